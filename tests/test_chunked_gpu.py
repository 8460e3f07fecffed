"""Chunked columns at the boundary (plhip_group_by_agg_chunked): the reference rechunks the frame before the group-by
(`df.as_single_chunk_par()`, crates/polars-lazy/src/physical_plan/executors/group_by.rs:66); here the library lays the chunks
of every column end to end in HBM -- values by copies, validity by a bit-granular kernel -- and must give the result of the
single-chunk query.  Ragged chunk boundaries that differ between columns, empty chunks, chunks with and without a validity
bitmap, Arrow offsets inside chunks, host and device chunks mixed."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from polars_amd.frame import _host_column, group_by_agg_chunked_raw
from tests import util

pytestmark = pytest.mark.gpu


def split(name, values, mask, bounds, rng, device=False, drop_full_masks=True):
    """Chunks of `values` at `bounds`; every chunk sits at a random offset inside a larger buffer (an Arrow slice)."""
    import torch

    out = []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        pad = int(rng.integers(0, 13))
        v = np.concatenate([np.zeros(pad, values.dtype), values[lo:hi], np.zeros(3, values.dtype)])
        m = None
        if mask is not None and not (drop_full_masks and mask[lo:hi].all() and rng.random() < 0.5):
            m = np.concatenate([rng.random(pad) < 0.5, mask[lo:hi], rng.random(3) < 0.5])
        if device:
            out.append(Column(name, torch.from_numpy(v).to("cuda:0"), m, offset=pad, length=hi - lo))
        else:
            out.append(Column(name, v, m, offset=pad, length=hi - lo))
    return out


def bounds_for(rng, n, k):
    cuts = np.sort(rng.integers(0, n + 1, k - 1)).tolist()
    b = [0] + cuts + [n]
    if k > 3:
        b[2] = b[1]  # an empty chunk
    return sorted(b)


@pytest.mark.parametrize("device", [False, True], ids=["host_chunks", "device_chunks"])
@pytest.mark.parametrize("n", [1000, 300_000, 3_000_000])
def test_chunked_columns_match_the_single_chunk_query(oracle, device, n):
    rng = np.random.default_rng(n + (7 if device else 0))
    k1 = rng.integers(-40, 40, n).astype(np.int32)
    k1m = rng.random(n) >= 0.03
    k2 = rng.integers(0, 7, n).astype(np.int64)
    v0 = rng.integers(-1000, 1000, n).astype(np.int64)
    v1 = np.round(rng.uniform(0, 100, n), 6)
    v1m = rng.random(n) >= 0.1
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (1, "sum")]
    keys = [split("k1", k1, k1m, bounds_for(rng, n, 5), rng, device), split("k2", k2, None, bounds_for(rng, n, 3), rng, device)]
    vals = [split("v0", v0, None, bounds_for(rng, n, 1), rng, device), split("v1", v1, v1m, bounds_for(rng, n, 7), rng, device)]
    codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
    handle, cols = group_by_agg_chunked_raw(keys, vals, codes)
    try:
        got = [_host_column(str(i), cs) for i, cs in enumerate(cols)]
    finally:
        _ffi.lib().plhip_result_free(handle)
    got = [(c.to_numpy(), c.valid_mask()) for c in got]
    want, _ = oracle.group_by_agg([Column("k1", k1, k1m), Column("k2", k2)], [Column("v0", v0), Column("v1", v1, v1m)], aggs)
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what=f"chunked columns, n={n}, device={device}")


def test_chunked_single_key_dense_and_errors(oracle):
    """A chunked u32 code column with the dense flag; dtype mismatch between chunks and length mismatch between columns are errors."""
    rng = np.random.default_rng(5)
    n, ncat = 2_000_000, 30_000
    k = rng.integers(0, ncat, n).astype(np.uint32)
    km = rng.random(n) >= 0.02
    v = rng.integers(1, 6, n).astype(np.int32)
    keys = [split("k", k, km, bounds_for(rng, n, 4), rng, True)]
    vals = [split("v", v, None, bounds_for(rng, n, 6), rng, False)]
    handle, cols = group_by_agg_chunked_raw(keys, vals, [(0, _ffi.AGG_SUM), (0, _ffi.AGG_COUNT)], _ffi.FLAG_KEYS_DENSE, ncat)
    try:
        got = [_host_column(str(i), cs) for i, cs in enumerate(cols)]
    finally:
        _ffi.lib().plhip_result_free(handle)
    assert pa.last_strategy() == "dense"
    want, _ = oracle.group_by_agg([Column("k", k, km)], [Column("v", v)], [(0, "sum"), (0, "count")])
    util.assert_frames_match([(c.to_numpy(), c.valid_mask()) for c in got], util.oracle_to_np(want), 1, what="chunked dense codes")
    bad = [[Column("k", k[:10]), Column("k", k[10:20].astype(np.int64))]]
    with pytest.raises(_ffi.PlHipError):
        group_by_agg_chunked_raw(bad, [[Column("v", v[:20])]], [(0, _ffi.AGG_SUM)])
    with pytest.raises(_ffi.PlHipError):
        group_by_agg_chunked_raw([[Column("k", k[:10]), Column("k", k[10:20])]], [[Column("v", v[:19])]], [(0, _ffi.AGG_SUM)])


def test_pinned_host_columns(oracle):
    """Columns in page-locked memory from plhip_alloc_pinned (the host layer's Arrow allocator): same results, and the host ->
    HBM copy runs at DMA rate (printed, not asserted: the pageable path's rate depends on the box)."""
    import ctypes as C
    import time

    L = _ffi.lib()
    n = 20_000_000
    rng = np.random.default_rng(12)
    k = rng.integers(0, 1000, n).astype(np.int32)
    v = rng.integers(-9, 9, n).astype(np.int64)
    pk, pv = C.c_void_p(), C.c_void_p()
    _ffi.check(L.plhip_alloc_pinned(k.nbytes, C.byref(pk)))
    _ffi.check(L.plhip_alloc_pinned(v.nbytes, C.byref(pv)))
    try:
        kp = np.ctypeslib.as_array(C.cast(pk, C.POINTER(C.c_int32)), shape=(n,))
        vp = np.ctypeslib.as_array(C.cast(pv, C.POINTER(C.c_int64)), shape=(n,))
        kp[:], vp[:] = k, v
        times = {}
        for name, (kk, vv) in {"pageable": (k, v), "pinned": (kp, vp)}.items():
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                out = pa.DataFrame({"k": kk, "v": vv}).group_by("k").agg(pa.col("v").sum(), pa.count())
                best = min(best, time.perf_counter() - t0)
            times[name] = best
            want, _ = oracle.group_by_agg([Column("k", k)], [Column("v", v)], [(0, "sum"), (0, "count")])
            util.assert_frames_match(util.frame_to_np(out), util.oracle_to_np(want), 1, what=f"{name} host columns")
        print(f"host columns, {(k.nbytes + v.nbytes) / 1e6:.0f} MB: pageable {times['pageable'] * 1e3:.1f} ms, pinned {times['pinned'] * 1e3:.1f} ms "
              f"({(k.nbytes + v.nbytes) / times['pinned'] / 1e9:.1f} GB/s end to end)")
    finally:
        _ffi.check(L.plhip_free_pinned(pk))
        _ffi.check(L.plhip_free_pinned(pv))
