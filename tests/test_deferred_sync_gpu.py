"""A DEVICE-output frame of plhip_group_by_agg[_opts] comes back as soon as its row count is on the host; the tail of its last
kernel is waited for by the first reader of the buffers (plhip_result_column / pack / finalize), and a frame freed unread hands
its pooled buffers on together with the producing stream's token (include/plhip.h, "Deferred completion").  These tests chain
queries without reading them, reuse the pooled buffers from the same and from another thread, and compare what is finally read
with the oracle: a reader that got a buffer too early shows up as a stale or half-written frame."""
import ctypes as C
import threading

import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _reset_strategy():
    yield
    pa.set_strategy("auto")


class Query:
    """One group-by with device-resident inputs, callable with or without reading the frame."""

    def __init__(self, k, vals, aggs, flags=0, hint=0):
        import torch

        self.host_keys = [Column("k", k)]
        self.host_vals = [Column(f"v{i}", v) for i, v in enumerate(vals)]
        self.aggs = aggs
        self.keep = [torch.from_numpy(k).to("cuda:0")] + [torch.from_numpy(v).to("cuda:0") for v in vals]
        torch.cuda.synchronize()
        keys = [Column("k", self.keep[0])]
        dvals = [Column(f"v{i}", t) for i, t in enumerate(self.keep[1:])]
        codes = [(ci, _ffi.AGG_NAMES[a]) for ci, a in aggs]
        self.karr = (_ffi.ColumnStruct * 1)(*[c.struct() for c in keys])
        self.varr = (_ffi.ColumnStruct * len(dvals))(*[c.struct() for c in dvals])
        self.aarr = (_ffi.AggStruct * len(codes))(*[_ffi.AggStruct(c, a) for c, a in codes])
        self.opts = _ffi.OptionsStruct(C.sizeof(_ffi.OptionsStruct), int(hint))
        self.flags = flags | _ffi.FLAG_OUTPUT_DEVICE
        self.n_vals, self.n_aggs = len(dvals), len(codes)

    def call(self):
        h = C.c_void_p()
        _ffi.check(_ffi.lib().plhip_group_by_agg_opts(self.karr, 1, self.varr, self.n_vals, self.aarr, self.n_aggs, self.flags,
                                                      C.byref(self.opts), C.byref(h)))
        return h

    def unread(self):
        L = _ffi.lib()
        h = self.call()
        g = L.plhip_result_num_rows(h)
        L.plhip_result_free(h)
        return g

    def read(self):
        L = _ffi.lib()
        h = self.call()
        try:
            out = []
            for i in range(L.plhip_result_num_columns(h)):
                cs = _ffi.ColumnStruct()
                _ffi.check(L.plhip_result_column(h, i, C.byref(cs)))
                c = pa.frame._host_column("c", cs)
                out.append((c.to_numpy().copy(), None if c.valid_mask() is None else c.valid_mask().copy()))
            return out
        finally:
            L.plhip_result_free(h)

    def want(self, oracle):
        w, _ = oracle.group_by_agg(self.host_keys, self.host_vals, self.aggs)
        return util.oracle_to_np(w)


def dense_query(seed, n=3_000_000, g=100_000):
    """db-benchmark q3's shape: Categorical codes (declared dense), an integer and a float value column"""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, g, n).astype(np.uint32)
    v1 = rng.integers(1, 6, n).astype(np.int32)
    v3 = np.round(rng.random(n) * 100, 6)
    return Query(k, [v1, v3], [(0, "sum"), (1, "mean")], _ffi.FLAG_KEYS_DENSE, g)


def hashed_query(seed, n=2_000_000, g=50_000):
    rng = np.random.default_rng(seed)
    k = rng.integers(-g // 2, g // 2, n).astype(np.int64) * 1_000_003
    v = rng.integers(-100, 100, n).astype(np.int64)
    w = rng.random(n)
    return Query(k, [v, w], [(0, "sum"), (1, "min"), (1, "max"), (0, "count")])


def check(got, want, what):
    util.assert_frames_match(got, want, 1, rtol_f64=1e-12, rtol_f32=1e-6, what=what)


@pytest.mark.parametrize("make,strategies", [(dense_query, ("dense",)), (hashed_query, ("tilesort", "direct"))], ids=["dense", "hashed"])
def test_chained_unread_queries_then_a_read(oracle, make, strategies):
    """A and B have the same shapes, so B's output buffers are the ones A's unread frame has just handed back."""
    A, B = make(11), make(12)
    want_a, want_b = A.want(oracle), B.want(oracle)
    for _ in range(3):
        got = A.read()
    assert pa.last_strategy() in strategies, pa.last_strategy()
    check(got, want_a, "A, read")
    for rnd in range(20):
        ga = A.unread()
        assert ga == len(want_a[0][0])
        got = B.read()
        check(got, want_b, f"B after an unread A, round {rnd}")
        for _ in range(3):
            B.unread()
        got = A.read()
        check(got, want_a, f"A after unread Bs, round {rnd}")


def test_pooled_buffers_cross_threads_with_their_token(oracle):
    """Thread T frees frames unread in a tight loop; the main thread's queries of the same shape draw on the same pool (another
    stream: it must wait for T's stream before it lets its own kernels write the buffer) and read every frame."""
    A, B = dense_query(21, 2_000_000, 60_000), dense_query(22, 2_000_000, 60_000)
    want_b = B.want(oracle)
    B.read()
    stop, err = threading.Event(), []

    def spin():
        try:
            _ffi.lib().plhip_init(0)
            while not stop.is_set():
                A.unread()
        except Exception as e:  # noqa: BLE001
            err.append(e)

    t = threading.Thread(target=spin)
    t.start()
    try:
        for rnd in range(40):
            check(B.read(), want_b, f"B beside a thread freeing unread frames, round {rnd}")
    finally:
        stop.set()
        t.join()
    assert not err, err
    # the thread has ended (its stream is gone): frames it left in the pool are still usable
    for rnd in range(3):
        check(B.read(), want_b, f"B after the other thread has ended, round {rnd}")


def test_packing_an_unread_partial_frame_waits_for_it(oracle):
    """plhip_result_pack_by_owner on a frame nobody has read: the pack kernels run on the caller's stream, possibly another thread's."""
    L = _ffi.lib()
    Q = hashed_query(31, 1_000_000, 20_000)
    Q.flags |= _ffi.FLAG_PARTIAL
    h = Q.call()
    box = {}

    def pack():
        try:
            L.plhip_init(0)
            n_parts = 4
            rb = L.plhip_result_record_bytes(h)
            g = L.plhip_result_num_rows(h)
            import torch

            buf = torch.empty(max(1, rb * g), dtype=torch.uint8, device="cuda:0")
            counts = (C.c_int64 * n_parts)()
            _ffi.check(L.plhip_result_pack_by_owner(h, 1, 0x9E3779B97F4A7C15, n_parts, buf.data_ptr(), counts))
            box["counts"] = list(counts)
            box["g"] = g
        except Exception as e:  # noqa: BLE001
            box["err"] = e

    t = threading.Thread(target=pack)
    t.start()
    t.join()
    L.plhip_result_free(h)
    assert "err" not in box, box.get("err")
    assert sum(box["counts"]) == box["g"] > 0
