"""PLHIP_FLAG_EXACT_FOLD: float SUM / MEAN folded strictly in ascending row order, the association of the reference's
GroupsProxy::Idx path (take_agg_no_null_primitive_iter_unchecked, crates/polars-arrow/src/kernels/take_agg/mod.rs:17-37;
agg_sum / agg_mean, crates/polars-core/src/frame/group_by/aggregations/mod.rs:622-651, 701-742), with f32 columns folded in
f32 (aggregations/mod.rs:719-724).  The bar here is BIT equality with the oracle's Idx-path fold -- f64 and f32, sums and
means, -0.0 included -- for every group size regime of the sort strategy (a thread per short run, a wave per run of
>= 2048 rows), whatever strategy is forced, and identical bits from run to run."""
import ctypes as C

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


def run(keys, vals, aggs, flags=0, hint=0):
    codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
    handle, cols = pa.group_by_agg_raw(keys, vals, codes, flags, hint)
    try:
        out = []
        for cs in cols:
            c = pa.frame._host_column("c", cs)
            out.append((c.to_numpy().copy(), None if c.valid_mask() is None else c.valid_mask().copy()))
        return out
    finally:
        _ffi.lib().plhip_result_free(handle)


def assert_bits_equal(got, want, n_keys, what):
    """canonical key sort, then every column bit for bit (floats compared as their integer images: -0.0 != +0.0)."""
    util.assert_frames_match(got, want, n_keys, rtol_f64=0.0, rtol_f32=0.0, what=what)
    og, ow = util.canonical_order(got[:n_keys]), util.canonical_order(want[:n_keys])
    for j in range(n_keys, len(got)):
        gv, wv = got[j][0][og], want[j][0][ow]
        m = np.ones(len(gv), bool) if want[j][1] is None else want[j][1][ow]
        if gv.dtype.kind == "f":
            iv = np.uint32 if gv.dtype.itemsize == 4 else np.uint64
            a, b = gv[m].view(iv), wv[m].view(iv)
            nan = np.isnan(gv[m])
            bad = np.flatnonzero((a != b) & ~nan)
            assert len(bad) == 0, f"{what}: column {j}: {len(bad)} of {len(a)} groups differ in bits, first {gv[m][bad[:3]]!r} vs {wv[m][bad[:3]]!r}"


def mixed_values(rng, n, dtype):
    """magnitudes over 12 (f64) / 6 (f32) decades and both signs: every re-association changes low bits"""
    decades = 12 if dtype == np.float64 else 6
    v = (rng.standard_normal(n) * 10.0 ** rng.integers(-decades // 2, decades // 2, n)).astype(dtype)
    v[rng.integers(0, n, n // 50)] = dtype(-0.0)
    return v


SHAPES = [
    ("short runs", 200_000, 40_000, None),           # ~5 rows per group: a thread per run
    ("runs around the 2048-row switch", 600_000, 280, None),
    ("a wave per run", 1_000_000, 37, None),          # ~27 000 rows per group
    ("one heavy key + a tail", 1_500_000, 5_000, 0.4),  # a 600 000-row run beside short ones
]


@pytest.mark.parametrize("what,n,g,hot", SHAPES, ids=[s[0] for s in SHAPES])
@pytest.mark.parametrize("vdt", [np.float64, np.float32], ids=["f64", "f32"])
def test_exact_fold_is_bit_identical_to_the_idx_path(oracle, what, n, g, hot, vdt):
    import zlib

    rng = np.random.default_rng(zlib.crc32(f"{what}{vdt.__name__}".encode()))
    k = rng.integers(0, g, n).astype(np.int64)
    if hot:
        k[rng.random(n) < hot] = 7
    keys = [Column("k", k)]
    v = mixed_values(rng, n, vdt)
    w = mixed_values(rng, n, vdt)
    wm = rng.random(n) >= 0.1
    i64 = rng.integers(-2**62, 2**62, n).astype(np.int64)  # an i64 mean folds (double)x in f64: order matters up here
    vals = [Column("v", v), Column("w", w, wm), Column("i", i64)]
    aggs = [(0, "sum"), (0, "mean"), (1, "sum"), (1, "mean"), (2, "mean"), (2, "sum"), (0, "min"), (1, "max"), (0, "count")]
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    want = util.oracle_to_np(want)
    first = None
    for strategy in ("auto", "direct", "tilesort", "global"):  # the flag wins over a forced strategy
        pa.set_strategy(strategy)
        got = run(keys, vals, aggs, _ffi.FLAG_EXACT_FOLD)
        assert pa.last_strategy() == "sort", pa.last_strategy()
        assert_bits_equal(got, want, 1, f"exact fold, {what}, {vdt.__name__}, forced {strategy}")
        order = util.canonical_order(got[:1])
        bits = [c[0][order].tobytes() for c in got]
        if first is None:
            first = bits
        assert bits == first, "the exact fold must not change from run to run"


def test_exact_fold_one_row_groups_and_all_null_groups(oracle):
    # aggregations/mod.rs:631-633: a 1-row group's sum is `arr.get(first).unwrap_or(0)`: -0.0 stays -0.0 (0.0 + -0.0 would be +0.0);
    # an all-null group sums to +0.0 and its mean is null
    k = np.array([1, 2, 3, 3, 4, 4, 5], dtype=np.int32)
    v = np.array([-0.0, 5.5, -0.0, -0.0, 1.0, 2.0, 9.0], dtype=np.float64)
    m = np.array([True, True, True, True, False, False, False])
    keys, vals = [Column("k", k)], [Column("v", v, m), Column("u", v.astype(np.float32), m)]
    aggs = [(0, "sum"), (0, "mean"), (1, "sum"), (1, "mean")]
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    got = run(keys, vals, aggs, _ffi.FLAG_EXACT_FOLD)
    assert_bits_equal(got, util.oracle_to_np(want), 1, "1-row and all-null groups")
    order = util.canonical_order(got[:1])
    s = got[1][0][order]
    assert np.signbit(s[0]) and not np.signbit(s[2]) and s[3] == 0.0 and not np.signbit(s[3])  # -0.0 | 0.0 + -0.0 + -0.0 = 0.0 | all null


def test_exact_fold_multi_key_sorted_flag_and_stable_order(oracle):
    rng = np.random.default_rng(5)
    n = 300_000
    a, b = rng.integers(0, 30, n).astype(np.int32), rng.integers(-5, 5, n).astype(np.int64)
    v = mixed_values(rng, n, np.float64)
    keys, vals = [Column("a", a), Column("b", b)], [Column("v", v)]
    aggs = [(0, "sum"), (0, "mean")]
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    got = run(keys, vals, aggs, _ffi.FLAG_EXACT_FOLD)
    assert_bits_equal(got, util.oracle_to_np(want), 2, "two keys")
    got = run(keys, vals, aggs, _ffi.FLAG_EXACT_FOLD | _ffi.FLAG_MAINTAIN_ORDER)
    assert_bits_equal(got, util.oracle_to_np(want), 2, "two keys, stable order")
    first = {}
    for i, t in enumerate(zip(a.tolist(), b.tolist())):
        first.setdefault(t, i)
    rows = [first[t] for t in zip(got[0][0].tolist(), got[1][0].tolist())]
    assert rows == sorted(rows), "maintain_order: groups by first row"
    ks = np.sort(rng.integers(0, 100, n).astype(np.int64))
    want, _ = oracle.group_by_agg([Column("k", ks)], vals, aggs)
    got = run([Column("k", ks)], vals, aggs, _ffi.FLAG_EXACT_FOLD | _ffi.FLAG_KEYS_SORTED)
    assert pa.last_strategy() == "sorted_keys"
    assert_bits_equal(got, util.oracle_to_np(want), 1, "sorted-key fast path")


def test_exact_fold_without_float_sums_keeps_the_fast_paths(oracle):
    # integer sums, extrema and counts do not depend on the order: the flag changes nothing for them
    rng = np.random.default_rng(6)
    n = 400_000
    keys = [Column("k", rng.integers(0, 50, n).astype(np.int32))]
    vals = [Column("v", rng.integers(-9, 9, n).astype(np.int32)), Column("f", rng.random(n))]
    aggs = [(0, "sum"), (1, "min"), (1, "max"), (0, "count")]
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    got = run(keys, vals, aggs, _ffi.FLAG_EXACT_FOLD)
    assert pa.last_strategy() != "sort"
    assert_bits_equal(got, util.oracle_to_np(want), 1, "no float sum")


def test_exact_fold_rejects_partial():
    L = _ffi.lib()
    k, v = np.arange(8, dtype=np.int32), np.ones(8)
    with pytest.raises(_ffi.PlHipError):
        run([Column("k", k)], [Column("v", v)], [(0, "sum")], _ffi.FLAG_EXACT_FOLD | _ffi.FLAG_PARTIAL)
    assert b"EXACT_FOLD" in L.plhip_last_error()
