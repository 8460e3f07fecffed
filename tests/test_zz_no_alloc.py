"""plhip_group_by_agg_noalloc (include/plhip.h): the entry point that allocates nothing, for callers that cannot call
hipMalloc.  It has no counterpart in the reference; its bar is the existing entry point's: through the C ABI, its output
against plhip_group_by_agg_opts' on the same device-resident columns (keys, integer sums, counts, validity bit-exact;
f64 within 1e-12 relative), on two calls in a row, the second with a smaller input -- both share the GPU's static
scratch array, which keeps the first call's contents."""
import ctypes as C

import numpy as np
import pytest

from polars_amd import _ffi
from tests import util

pytestmark = pytest.mark.gpu


def _dev(a, torch):
    return torch.from_numpy(a).to("cuda:0")


def _col(t, code):
    return _ffi.ColumnStruct(t.data_ptr(), None, 0, t.numel(), code, _ffi.DEVICE)


def _existing(L, karr, varr, nv, aarr, na, hint):
    import polars_amd as pa

    handle = C.c_void_p()
    opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), int(hint))
    _ffi.check(L.plhip_group_by_agg_opts(karr, 1, varr, nv, aarr, na, 0, C.byref(opts), C.byref(handle)))
    try:
        out = []
        for i in range(L.plhip_result_num_columns(handle)):
            cs = _ffi.ColumnStruct()
            _ffi.check(L.plhip_result_column(handle, i, C.byref(cs)))
            c = pa.frame._host_column(f"c{i}", cs)
            m = c.valid_mask()
            out.append((c.to_numpy().copy(), None if m is None else m.copy()))
        return out
    finally:
        L.plhip_result_free(handle)


def _run_pair(n, groups, seed, nullable):
    import torch

    L = _ffi.lib()
    _ffi.check(L.plhip_init(0))
    rng = np.random.default_rng(seed)
    key = rng.integers(-(groups // 2), groups - groups // 2, n).astype(np.int32)
    v1 = rng.integers(-1000, 1000, n).astype(np.int32)
    v3 = np.round(rng.uniform(-100, 100, n), 6)
    tk, t1, t3 = _dev(key, torch), _dev(v1, torch), _dev(v3, torch)
    karr = (_ffi.ColumnStruct * 1)(_col(tk, _ffi.I32))
    c1, c3 = _col(t1, _ffi.I32), _col(t3, _ffi.F64)
    keep = []
    if nullable:  # 10 % null values in v3: MEAN / MIN of it become nullable columns
        mask = rng.random(n) >= 0.1
        mask[key == key[0]] = False  # one all-null group
        bm = _dev(np.packbits(mask, bitorder="little"), torch)
        keep.append(bm)
        c3.validity = bm.data_ptr()
    varr = (_ffi.ColumnStruct * 2)(c1, c3)
    aggs = [(0, _ffi.AGG_SUM), (1, _ffi.AGG_MEAN), (1, _ffi.AGG_MIN), (0, _ffi.AGG_MAX), (0, _ffi.AGG_COUNT)]
    aarr = (_ffi.AggStruct * len(aggs))(*[_ffi.AggStruct(c, k) for c, k in aggs])
    want = _existing(L, karr, varr, 2, aarr, len(aggs), groups)

    cap = groups + 8
    dts = [_ffi.I32, _ffi.I32, _ffi.F64, _ffi.F64, _ffi.I32, _ffi.U32]
    tdt = {_ffi.I32: torch.int32, _ffi.F64: torch.float64, _ffi.U32: torch.int32}
    bufs = [torch.full((cap,), -7, dtype=tdt[d], device="cuda:0") for d in dts]
    maps = [torch.full(((cap // 64 + 1) * 8,), 0xFF, dtype=torch.uint8, device="cuda:0") for _ in dts]
    oarr = (_ffi.ColumnStruct * len(dts))(*[
        _ffi.ColumnStruct(b.data_ptr(), m.data_ptr(), 0, cap, d, _ffi.DEVICE) for b, m, d in zip(bufs, maps, dts)])
    g = C.c_int64(-1)
    torch.cuda.synchronize()
    _ffi.check(L.plhip_group_by_agg_noalloc(karr, 1, varr, 2, aarr, len(aggs), 0, groups, oarr, len(dts), C.byref(g)))
    G = g.value
    assert G == len(want[0][0])
    got = []
    for j, (b, m, d) in enumerate(zip(bufs, maps, dts)):
        vals = b.cpu().numpy()[:G].view(_ffi.code_np_dtype(d)).copy()
        has_map = want[j][1] is not None
        mk = np.unpackbits(m.cpu().numpy(), bitorder="little")[:G].astype(bool) if has_map else None
        got.append((vals, mk))
    util.assert_frames_match(got, want, 1, what=f"noalloc n={n} groups={groups} nullable={nullable}")
    return G


def test_noalloc_matches_existing_entry_point_twice():
    # two calls in a row on the same GPU's scratch array, the second with a smaller input and fewer groups
    g1 = _run_pair(3_000_000, 1000, 11, nullable=True)
    g2 = _run_pair(70_001, 300, 12, nullable=False)
    assert g1 == 1000 and g2 == 300
    g3 = _run_pair(1_000_000, 97, 13, nullable=True)
    assert g3 == 97


def test_noalloc_rejects_what_it_cannot_hold():
    import torch

    L = _ffi.lib()
    _ffi.check(L.plhip_init(0))
    n = 1 << 16
    tk = torch.arange(n, dtype=torch.int32, device="cuda:0")
    tv = torch.ones(n, dtype=torch.float64, device="cuda:0")
    karr = (_ffi.ColumnStruct * 1)(_col(tk, _ffi.I32))
    varr = (_ffi.ColumnStruct * 1)(_col(tv, _ffi.F64))
    aarr = (_ffi.AggStruct * 1)(_ffi.AggStruct(0, _ffi.AGG_SUM))
    ok = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    ov = torch.zeros(n, dtype=torch.float64, device="cuda:0")
    oarr = (_ffi.ColumnStruct * 2)(_ffi.ColumnStruct(ok.data_ptr(), None, 0, n, _ffi.I32, _ffi.DEVICE),
                                   _ffi.ColumnStruct(ov.data_ptr(), None, 0, n, _ffi.F64, _ffi.DEVICE))
    g = C.c_int64(-1)
    # all-unique keys: far beyond one LDS table -> a clear error, nothing launched
    rc = L.plhip_group_by_agg_noalloc(karr, 1, varr, 1, aarr, 1, 0, n, oarr, 2, C.byref(g))
    assert rc == 6 and b"do not fit" in L.plhip_last_error()
    # no hint -> rejected (sampling would allocate)
    rc = L.plhip_group_by_agg_noalloc(karr, 1, varr, 1, aarr, 1, 0, 0, oarr, 2, C.byref(g))
    assert rc == 6 and b"n_groups_hint" in L.plhip_last_error()
    # a hint that promises too few groups: the tables overflow -> ComputeError, not a wrong frame
    tk2 = (torch.arange(n, dtype=torch.int32, device="cuda:0") % 3000).contiguous()
    karr2 = (_ffi.ColumnStruct * 1)(_col(tk2, _ffi.I32))
    rc = L.plhip_group_by_agg_noalloc(karr2, 1, varr, 1, aarr, 1, 0, 40, oarr, 2, C.byref(g))
    assert rc == 1
    # and the entry point still works afterwards
    tk3 = (torch.arange(n, dtype=torch.int32, device="cuda:0") % 40).contiguous()
    karr3 = (_ffi.ColumnStruct * 1)(_col(tk3, _ffi.I32))
    _ffi.check(L.plhip_group_by_agg_noalloc(karr3, 1, varr, 1, aarr, 1, 0, 40, oarr, 2, C.byref(g)))
    assert g.value == 40
    s = ov.cpu().numpy()[:40]
    assert np.all((s == n // 40) | (s == n // 40 + 1)) and s.sum() == n


def test_noalloc_allocates_nothing():
    """The entry point's one contract: no device allocation.  After a warm-up call, the device's free memory
    (hipMemGetInfo through torch) and the library's pooled result buffers must be the same before and after a run of calls."""
    import torch

    L = _ffi.lib()
    _ffi.check(L.plhip_init(0))
    n, groups = 500_000, 200
    rng = np.random.default_rng(3)
    tk = _dev(rng.integers(0, groups, n).astype(np.int32), torch)
    tv = _dev(rng.random(n), torch)
    karr = (_ffi.ColumnStruct * 1)(_col(tk, _ffi.I32))
    varr = (_ffi.ColumnStruct * 1)(_col(tv, _ffi.F64))
    aarr = (_ffi.AggStruct * 2)(_ffi.AggStruct(0, _ffi.AGG_SUM), _ffi.AggStruct(0, _ffi.AGG_COUNT))
    cap = groups + 8
    ok = torch.zeros(cap, dtype=torch.int32, device="cuda:0")
    ov = torch.zeros(cap, dtype=torch.float64, device="cuda:0")
    oc = torch.zeros(cap, dtype=torch.int32, device="cuda:0")
    oarr = (_ffi.ColumnStruct * 3)(_ffi.ColumnStruct(ok.data_ptr(), None, 0, cap, _ffi.I32, _ffi.DEVICE),
                                   _ffi.ColumnStruct(ov.data_ptr(), None, 0, cap, _ffi.F64, _ffi.DEVICE),
                                   _ffi.ColumnStruct(oc.data_ptr(), None, 0, cap, _ffi.U32, _ffi.DEVICE))
    g = C.c_int64(-1)
    _ffi.check(L.plhip_group_by_agg_noalloc(karr, 1, varr, 1, aarr, 2, 0, groups, oarr, 3, C.byref(g)))  # warm-up
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    pooled0 = L.plhip_pooled_buffers(0)
    for _ in range(20):
        _ffi.check(L.plhip_group_by_agg_noalloc(karr, 1, varr, 1, aarr, 2, 0, groups, oarr, 3, C.byref(g)))
        assert g.value == groups
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free1 == free0, f"device memory changed across no-alloc calls: {free0} -> {free1}"
    assert L.plhip_pooled_buffers(0) == pooled0
    assert int(oc.cpu().numpy()[:groups].view(np.uint32).sum()) == n
