"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference's own
known answers.  Integers, keys, counts and validity bit-exact after a canonical key sort; f64 sum/mean
within 1e-12 relative (the north star's tolerance); f32 within 1e-5 (the reference folds f32 in f32)."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from tests import util

pytestmark = pytest.mark.gpu

KA = util.load_known_answers()
STRATEGIES = ["auto", "global", "lds", "partitioned", "direct", "tilesort", "sort", "tilesort2"]


@pytest.fixture(autouse=True)
def _reset_strategy():
    yield
    pa.set_strategy("auto")


def run_product(keys, values, aggs, maintain_order=False, strategy="auto"):
    pa.set_strategy(strategy)
    # value columns may share a name with a key column (reference tests do that): build a frame that keeps both
    cols = {f"k{j}": k for j, k in enumerate(keys)}
    cols.update({f"v{j}": v for j, v in enumerate(values)})
    df = pa.DataFrame({n: Column(n, c.values, c.valid_mask()) for n, c in cols.items()})
    exprs = []
    for i, (ci, kind) in enumerate(aggs):
        e = pa.count() if kind == "count" else getattr(pa.col(f"v{ci}"), kind)()
        exprs.append(e.alias(f"agg{i}"))
    return df.group_by([f"k{j}" for j in range(len(keys))], maintain_order=maintain_order).agg(exprs)


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("case", KA["group_by"], ids=lambda c: c["name"])
def test_reference_known_answers(case, strategy):
    keys = [util.make_column(k) for k in case["keys"]]
    values = [util.make_column(v) for v in case["values"]]
    out = run_product(keys, values, [tuple(a) for a in case["aggs"]], case["maintain_order"], strategy)
    cols = [out[c].to_list() for c in out.columns]
    dts = [_ffi.DTYPE_NAMES[out[c].dtype] for c in out.columns]
    util.check_known_answer(case, cols, dts)


@pytest.mark.parametrize("case", KA["hash"], ids=lambda c: c["name"])
def test_hash_known_answers(case):
    import ctypes as C

    keys = [util.make_column(k) for k in case["keys"]]
    out = np.empty(len(keys[0]), dtype=np.uint64)
    arr = (_ffi.ColumnStruct * len(keys))(*[k.struct() for k in keys])
    _ffi.check(_ffi.lib().plhip_vec_hash(arr, len(keys), case["null_hash"], out.ctypes.data, _ffi.HOST))
    assert out.tolist() == case["expected"]


# ---------------------------------------------------------------------------------------------------
# randomised parity against the oracle
# ---------------------------------------------------------------------------------------------------
INT_DT = [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64]
ALL_DT = INT_DT + [np.float32, np.float64]


def rand_col(rng, name, dt, n, card=None, null_frac=0.0, nan_frac=0.0, lo=None, hi=None):
    dt = np.dtype(dt)
    if dt.kind == "f":
        if card:
            pool = np.round(rng.uniform(-100, 100, card), 3).astype(dt)
            v = pool[rng.integers(0, card, n)]
        else:
            v = np.round(rng.uniform(0, 100, n), 6).astype(dt)
        if nan_frac:
            v[rng.random(n) < nan_frac] = np.nan
    else:
        info = np.iinfo(dt)
        a = info.min if lo is None else lo
        b = info.max if hi is None else hi
        if card:
            pool = rng.integers(a, b, card, dtype=dt, endpoint=True)
            v = pool[rng.integers(0, card, n)]
        else:
            v = rng.integers(a, b, n, dtype=dt, endpoint=True)
    mask = None
    if null_frac:
        mask = rng.random(n) >= null_frac
    return Column(name, v, mask)


def check_parity(oracle, keys, values, aggs, strategy="auto", maintain_order=False, what=""):
    want, _ = oracle.group_by_agg(keys, values, aggs, maintain_order=maintain_order)
    got = run_product(keys, values, aggs, maintain_order, strategy)
    ran = pa.last_strategy()
    if strategy in ("direct", "tilesort", "sort", "tilesort2") and ran != strategy:
        FELL_BACK.append((what, strategy, ran))
    g, w = util.frame_to_np(got), util.oracle_to_np(want)
    if maintain_order:  # row order itself is part of the contract
        for (gv, _), (wv, _) in zip(g[:len(keys)], w[:len(keys)]):
            assert np.array_equal(util._bits(gv), util._bits(wv)), f"{what}: stable group order differs"
    util.assert_frames_match(g, w, len(keys), what=what)
    return got


ALL_AGGS = ["sum", "mean", "min", "max", "count"]
FELL_BACK: list = []  # (case, forced strategy, strategy that ran): the LDS-only pipeline declined the query


def test_zz_fast_path_coverage():
    """The forced LDS-only strategies may decline (stable order, wide key tuples, more groups than fit);
    they must not decline the plain cases."""
    allowed = {"direct": ("multi-key", "stable", "card=None", "n=0"), "tilesort": ("multi-key", "stable", "card=None", "n=0"),
               "sort": ("n=0",), "tilesort2": ("multi-key", "stable", "card=None", "n=0")}
    bad = [w for w, s, r in FELL_BACK if not any(w.startswith(a) or a in w for a in allowed[s]) and not (s == "tilesort2" and r == "leafsort")]
    assert not bad, f"fast path declined: {sorted(bad)}"


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("kdt", ALL_DT, ids=lambda d: np.dtype(d).name)
def test_single_key_dtypes(oracle, kdt, strategy):
    rng = np.random.default_rng(hash(np.dtype(kdt).name) % 1000)
    n = 20000
    keys = [rand_col(rng, "k", kdt, n, card=300, null_frac=0.05)]
    values = [rand_col(rng, "a", np.int32, n, lo=-1000, hi=1000), rand_col(rng, "b", np.float64, n, null_frac=0.1)]
    aggs = [(0, a) for a in ALL_AGGS] + [(1, a) for a in ALL_AGGS]
    check_parity(oracle, keys, values, aggs, strategy, what=f"key {np.dtype(kdt).name}")


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("vdt", ALL_DT, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("nulls", [0.0, 0.3])
def test_value_dtypes(oracle, vdt, nulls, strategy):
    rng = np.random.default_rng(7)
    n = 30000
    keys = [rand_col(rng, "k", np.int32, n, card=50)]
    dt = np.dtype(vdt)
    if dt.kind == "f":
        v = rand_col(rng, "v", vdt, n, null_frac=nulls, nan_frac=0.02)
    elif dt.itemsize == 8:  # keep |sum| < 2^53 so that the f64 fold of the reference is order independent
        v = rand_col(rng, "v", vdt, n, null_frac=nulls, lo=0 if dt.kind == "u" else -(1 << 40), hi=1 << 40)
    else:
        v = rand_col(rng, "v", vdt, n, null_frac=nulls)
    check_parity(oracle, keys, [v], [(0, a) for a in ALL_AGGS], strategy, what=f"value {dt.name} nulls={nulls}")


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_wrapping_sums(oracle, strategy):
    # i32/u32/i64/u64 sums wrap in their own type (aggregations/mod.rs:622-651, release build)
    rng = np.random.default_rng(3)
    n = 5000
    keys = [rand_col(rng, "k", np.uint32, n, card=7)]
    values = [rand_col(rng, f"v{i}", dt, n) for i, dt in enumerate([np.int32, np.uint32, np.int64, np.uint64])]
    check_parity(oracle, keys, values, [(i, "sum") for i in range(4)], strategy, what="wrapping")


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_multi_key(oracle, strategy):
    rng = np.random.default_rng(11)
    n = 50000
    keys = [rand_col(rng, "id1", np.uint32, n, lo=1, hi=100), rand_col(rng, "id2", np.int64, n, lo=-50, hi=50, null_frac=0.05),
            rand_col(rng, "id3", np.int16, n, card=20), rand_col(rng, "f", np.float64, n, card=5, null_frac=0.1)]
    values = [rand_col(rng, "v1", np.int64, n, lo=-1000, hi=1000), rand_col(rng, "v3", np.float64, n)]
    aggs = [(0, "sum"), (0, "min"), (0, "max"), (1, "sum"), (1, "mean"), (0, "count")]
    check_parity(oracle, keys, values, aggs, strategy, what="multi-key")


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("mo", [False, True])
def test_maintain_order(oracle, strategy, mo):
    rng = np.random.default_rng(5)
    n = 40000
    keys = [rand_col(rng, "k", np.int64, n, card=1000, null_frac=0.01)]
    values = [rand_col(rng, "v", np.int32, n, lo=0, hi=100)]
    check_parity(oracle, keys, values, [(0, "sum"), (0, "count")], strategy, maintain_order=mo, what="stable")


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_edge_keys(oracle, strategy):
    # the table's empty sentinel as a real key, 0, extremes, -0.0 vs +0.0 (distinct groups: bit pattern)
    k = np.array([0xFFFFFFFFFFFFFFFF, 0, 1, 0xFFFFFFFFFFFFFFFF, 0x7FFFFFFFFFFFFFFF, 0, 0x8000000000000000] * 30, dtype=np.uint64)
    v = np.arange(len(k), dtype=np.int64)
    check_parity(oracle, [Column("k", k)], [Column("v", v)], [(0, "sum"), (0, "count"), (0, "min")], strategy, what="sentinel")
    f = np.array([0.0, -0.0, np.nan, 1.5, -0.0, np.nan, np.inf, -np.inf] * 10, dtype=np.float64)
    v = np.arange(len(f), dtype=np.int32)
    check_parity(oracle, [Column("f", f)], [Column("v", v)], [(0, "sum"), (0, "count")], strategy, what="float keys")
    ki = np.array([-(2 ** 63), 2 ** 63 - 1, -1, 0] * 25, dtype=np.int64)
    check_parity(oracle, [Column("k", ki, np.arange(100) % 7 != 0)], [Column("v", np.ones(100, np.float32))],
                 [(0, "sum"), (0, "mean")], strategy, what="i64 extremes + nulls")


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_float_extremum_rules(oracle, strategy):
    # NaN ignored; +-inf and all-NaN groups clamp to the fold's init; 1-row groups return their value;
    # all-null -> null (aggregations/mod.rs:459-620)
    inf, nan = np.inf, np.nan
    k = np.array([0, 0, 1, 1, 2, 3, 4, 4, 5, 5, 6, 7, 7, 8, 8, 8], dtype=np.int32)
    v = np.array([nan, nan, inf, inf, nan, inf, -inf, -inf, 1.0, nan, -inf, 2.0, 3.0, nan, 5.0, nan])
    m = np.ones(len(k), bool)
    m[[11, 12]] = False  # group 7 all null
    for vdt in (np.float64, np.float32):
        cols = [Column("v", v.astype(vdt)), Column("vn", v.astype(vdt), m)]
        check_parity(oracle, [Column("k", k)], cols, [(0, "min"), (0, "max"), (1, "min"), (1, "max"), (1, "mean"), (1, "sum")],
                     strategy, what=f"extremum {np.dtype(vdt).name}")


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 1000, 1025])
def test_tiny_and_ragged_sizes(oracle, strategy, n):
    rng = np.random.default_rng(n)
    keys = [rand_col(rng, "k", np.int32, n, lo=0, hi=9)]
    values = [rand_col(rng, "v", np.float64, n, null_frac=0.2 if n else 0.0)]
    check_parity(oracle, keys, values, [(0, a) for a in ALL_AGGS], strategy, what=f"n={n}")


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_arrow_offsets(oracle, strategy):
    # a sliced Arrow array: values/validity offset that is not byte aligned
    rng = np.random.default_rng(9)
    n, off = 5000, 13
    kbuf = rng.integers(0, 40, n + off + 5).astype(np.int32)
    vbuf = rng.integers(-100, 100, n + off + 5).astype(np.int64)
    vmask = rng.random(n + off + 5) > 0.25
    kmask = rng.random(n + off + 5) > 0.1
    keys = [Column("k", kbuf, kmask, offset=off, length=n)]
    values = [Column("v", vbuf, vmask, offset=off, length=n)]
    want, _ = oracle.group_by_agg(keys, values, [(0, a) for a in ALL_AGGS])
    pa.set_strategy(strategy)
    handle, cols = pa.group_by_agg_raw(keys, values, [(0, _ffi.AGG_NAMES[a]) for a in ALL_AGGS], 0)
    try:
        from polars_amd.frame import _host_column
        got = [_host_column(str(i), cs) for i, cs in enumerate(cols)]
        g = [(c.to_numpy(), c.valid_mask()) for c in got]
    finally:
        _ffi.lib().plhip_result_free(handle)
    util.assert_frames_match(g, util.oracle_to_np(want), 1, what="offsets")


@pytest.mark.parametrize("strategy", STRATEGIES)
@pytest.mark.parametrize("card", [3, 2000, 100000, None])
def test_cardinalities(oracle, strategy, card):
    # from a handful of hot groups to all-unique keys
    rng = np.random.default_rng(17)
    n = 300000
    if card is None:
        k = rng.permutation(n).astype(np.int64) * 7919
        keys = [Column("k", k)]
    else:
        keys = [rand_col(rng, "k", np.int64, n, card=card)]
    values = [rand_col(rng, "v1", np.int32, n, lo=1, hi=5), rand_col(rng, "v3", np.float64, n)]
    check_parity(oracle, keys, values, [(0, "sum"), (1, "mean"), (1, "sum"), (0, "count"), (1, "max")], strategy, what=f"card={card}")


@pytest.mark.parametrize("strategy", STRATEGIES)
def test_skewed_keys(oracle, strategy):
    rng = np.random.default_rng(23)
    n = 400000
    z = rng.zipf(1.3, n).astype(np.int64)  # heavy hitters + a long tail
    keys = [Column("k", z)]
    values = [rand_col(rng, "v", np.float64, n), rand_col(rng, "w", np.int64, n, lo=-5, hi=5)]
    check_parity(oracle, keys, values, [(0, "sum"), (0, "mean"), (1, "sum"), (1, "min"), (1, "max"), (0, "count")], strategy, what="zipf")


def test_device_resident_inputs(oracle):
    import torch

    rng = np.random.default_rng(29)
    n = 100000
    k = rng.integers(0, 500, n).astype(np.int32)
    v = rng.uniform(0, 100, n)
    m = rng.random(n) > 0.05
    keys_h, vals_h = [Column("k", k)], [Column("v", v, m)]
    want, _ = oracle.group_by_agg(keys_h, vals_h, [(0, "sum"), (0, "mean"), (0, "count")])
    dk = torch.from_numpy(k).cuda()
    dv = torch.from_numpy(v).cuda()
    dm = torch.from_numpy(pa.column.pack_validity(m)).cuda()
    df = pa.DataFrame([Column("k", dk), Column("v", dv, dm, packed=True)])
    got = df.group_by("k").agg(pa.col("v").sum().alias("s"), pa.col("v").mean().alias("m"), pa.count())
    util.assert_frames_match(util.frame_to_np(got), util.oracle_to_np(want), 1, what="device inputs")


def test_vec_hash_parity(oracle):
    rng = np.random.default_rng(31)
    n = 10000
    keys = [rand_col(rng, "a", np.int32, n, null_frac=0.1), rand_col(rng, "b", np.float64, n),
            rand_col(rng, "c", np.int8, n, null_frac=0.2), rand_col(rng, "d", np.uint64, n)]
    null_h = 988796329533502010
    want = oracle.vec_hash(keys, null_h)
    out = np.empty(n, dtype=np.uint64)
    arr = (_ffi.ColumnStruct * len(keys))(*[k.struct() for k in keys])
    _ffi.check(_ffi.lib().plhip_vec_hash(arr, len(keys), null_h, out.ctypes.data, _ffi.HOST))
    assert np.array_equal(out, want)


def test_vec_hash_boolean_columns(oracle):
    """plhip_vec_hash_seeded: Boolean keys hash to the caller's true / false / null constants and combine with
    _boost_hash_combine (VecHash for BooleanChunked, vector_hasher.rs:241-302; hashing/mod.rs:20-22): against the oracle, host and
    device bitmaps, with an Arrow offset that is not a multiple of 8."""
    import ctypes as C

    import torch

    rng = np.random.default_rng(33)
    n = 50_003
    seeds = _ffi.HashSeeds(0x0123456789ABCDEF, 0xF00DFACE12345678, 0x0BADC0DE87654321)
    b1 = Column("b1", rng.random(n) < 0.5)
    b2 = Column("b2", rng.random(n) < 0.3, rng.random(n) >= 0.2)
    k = rand_col(rng, "k", np.int64, n, null_frac=0.1)
    for keys in ([b1], [b1, k, b2], [k, b2, b1], [b2]):
        want = oracle.vec_hash_seeded(keys, seeds.null_hash, seeds.true_hash, seeds.false_hash)
        out = np.empty(n, dtype=np.uint64)
        arr = (_ffi.ColumnStruct * len(keys))(*[c.struct() for c in keys])
        _ffi.check(_ffi.lib().plhip_vec_hash_seeded(arr, len(keys), C.byref(seeds), out.ctypes.data, _ffi.HOST))
        assert np.array_equal(out, want), [c.name for c in keys]
    # a slice at an odd bit offset, bitmaps resident in HBM
    off, m = 13, 20_000
    structs, keep = [], []
    for c in (b2, b1):
        cs = c.struct()
        nb = (n + 7) // 8
        tv = torch.from_numpy(np.ctypeslib.as_array(C.cast(cs.values, C.POINTER(C.c_uint8)), shape=(nb,)).copy()).to("cuda:0")
        tm = None
        if cs.validity:
            tm = torch.from_numpy(np.ctypeslib.as_array(C.cast(cs.validity, C.POINTER(C.c_uint8)), shape=(nb,)).copy()).to("cuda:0")
        keep += [tv, tm]
        structs.append(_ffi.ColumnStruct(tv.data_ptr(), tm.data_ptr() if tm is not None else None, off, m, _ffi.BOOL, _ffi.DEVICE))
    torch.cuda.synchronize()
    want = oracle.vec_hash_seeded([b2, b1], seeds.null_hash, seeds.true_hash, seeds.false_hash)[off:off + m]  # (row-wise: a slice of the hashes)
    out = np.empty(m, dtype=np.uint64)
    _ffi.check(_ffi.lib().plhip_vec_hash_seeded((_ffi.ColumnStruct * 2)(*structs), 2, C.byref(seeds), out.ctypes.data, _ffi.HOST))
    assert np.array_equal(out, want)
    # the unseeded entry point refuses Boolean keys instead of hashing them with made-up constants
    out1 = np.empty(n, dtype=np.uint64)
    rc = _ffi.lib().plhip_vec_hash((_ffi.ColumnStruct * 1)(b1.struct()), 1, 7, out1.ctypes.data, _ffi.HOST)
    assert rc == 4 and b"plhip_vec_hash_seeded" in _ffi.lib().plhip_last_error()


def test_error_kinds():
    with pytest.raises(pa.ComputeError):
        pa.DataFrame({"a": np.zeros(3, np.int32)}).group_by().agg(pa.count())


# ---------------------------------------------------------------------------------------------------
# two-step API: plhip_group_tuples (GroupsProxy::Idx as CSR) + plhip_agg_groups (Series::agg_*)
# ---------------------------------------------------------------------------------------------------
def _canon(lists):
    return sorted(lists, key=lambda l: l[0])


@pytest.mark.parametrize("sorted_", [False, True])
@pytest.mark.parametrize("case", ["one_key", "nullable_key", "multi_key", "all_unique", "one_group"])
def test_group_tuples(oracle, case, sorted_):
    rng = np.random.default_rng(41)
    n = 30000
    if case == "one_key":
        keys = [rand_col(rng, "k", np.int32, n, card=700)]
    elif case == "nullable_key":
        keys = [rand_col(rng, "k", np.float64, n, card=50, null_frac=0.1)]
    elif case == "multi_key":
        keys = [rand_col(rng, "a", np.uint32, n, lo=1, hi=30), rand_col(rng, "b", np.int64, n, lo=-5, hi=5, null_frac=0.05),
                rand_col(rng, "c", np.int8, n, card=4)]
    elif case == "all_unique":
        keys = [Column("k", rng.permutation(n).astype(np.int64) * 7919)]
    else:
        keys = [Column("k", np.full(n, 7, dtype=np.uint16))]
    want = oracle.Groups(keys, sorted_)
    got = pa.group_tuples(keys, sorted_)
    assert got.n_groups == want.n_groups and got.n_rows == n
    first, off, idx = got.buffers()
    lists = [idx[off[g]:off[g + 1]].tolist() for g in range(got.n_groups)]
    for g, l in enumerate(lists):  # GroupsIdx invariants (proxy.rs:16-20)
        assert l == sorted(l) and first[g] == l[0]
    if sorted_:  # groups ordered by first row (hashing.rs:32-79): the order itself is part of the contract
        assert lists == want.lists()
    else:
        assert _canon(lists) == _canon(want.lists())


@pytest.mark.parametrize("vdt", [np.int16, np.int32, np.uint32, np.int64, np.float32, np.float64], ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("card", [40, 5000])
def test_agg_groups(oracle, vdt, card):
    # card 40: long groups -> the wave-per-group kernel; card 5000: the one-thread-per-group left fold
    rng = np.random.default_rng(43)
    n = 30000
    keys = [rand_col(rng, "k", np.int32, n, card=card)]
    dt = np.dtype(vdt)
    if dt.kind == "f":
        v = rand_col(rng, "v", vdt, n, null_frac=0.15, nan_frac=0.02)
    else:
        lo = None if dt.itemsize <= 2 else (0 if dt.kind == "u" else -(1 << 20))
        v = rand_col(rng, "v", vdt, n, null_frac=0.15, lo=lo, hi=(1 << 20) if dt.itemsize > 2 else None)
    want_g = oracle.Groups(keys, True)
    got_g = pa.group_tuples(keys, True)
    for kind in ALL_AGGS:
        wv, wm = oracle.agg(None if kind == "count" else v, kind, want_g)
        col = pa.agg_groups(None if kind == "count" else v, kind, got_g)
        gv, gm = col.to_numpy(), col.valid_mask()
        gm = np.ones(len(gv), bool) if gm is None else gm
        assert np.array_equal(gm, wm), f"{kind}: validity"
        rel = 1e-5 if dt == np.float32 else 1e-12
        assert gv.dtype == wv.dtype, f"{kind}: dtype {gv.dtype} vs {wv.dtype}"
        for i in np.nonzero(wm)[0]:
            assert util.values_equal(gv[i].item(), wv[i].item(), rel), f"{kind} group {i}: {gv[i]!r} vs {wv[i]!r}"


def test_agg_groups_left_fold_is_bit_exact(oracle):
    # short groups take the sequential kernel: same fold order as the reference's Idx path -> identical f64 bits
    rng = np.random.default_rng(47)
    n = 20000
    keys = [rand_col(rng, "k", np.int64, n, card=3000)]
    v = rand_col(rng, "v", np.float64, n)
    g_o, g_p = oracle.Groups(keys, True), pa.group_tuples(keys, True)
    for kind in ("sum", "mean"):
        wv, _ = oracle.agg(v, kind, g_o)
        gv = pa.agg_groups(v, kind, g_p).to_numpy()
        assert np.array_equal(gv.view(np.uint64), wv.view(np.uint64)), kind


@pytest.mark.parametrize("dt", [np.int32, np.uint64, np.float32, np.float64])
def test_agg_groups_first_last_var_std(oracle, dt):
    """first / last / var / std over CSR groups against the oracle's restatement of
    dispatch.rs:47-76 and take_agg/var.rs:11-40; short groups (thread per group: the reference's own update
    order) and long ones (wave per group, pairwise merge)."""
    rng = np.random.default_rng(97)
    for n, card in ((20000, 3000), (60000, 40)):
        keys = [rand_col(rng, "k", np.int64, n, card=card)]
        v = rand_col(rng, "v", dt, n, null_frac=0.15, lo=0 if np.dtype(dt).kind != "f" else None, hi=1000 if np.dtype(dt).kind != "f" else None)
        g_o, g_p = oracle.Groups(keys, True), pa.group_tuples(keys, True)
        for kind in ("first", "last", "var", "std", "median", "n_unique"):
            for ddof in ((0, 1, 2) if kind in ("var", "std") else (1,)):
                wv, wm = oracle.agg_ext(v, kind, g_o, ddof)
                got = pa.agg_groups(v, kind, g_p, ddof=ddof)
                gv, gm = got.to_numpy(), got.valid_mask()
                gm = np.ones(len(gv), bool) if gm is None else gm
                assert gv.dtype == wv.dtype and np.array_equal(gm, wm), (kind, ddof)
                if kind in ("first", "last", "n_unique"):
                    assert np.array_equal(gv[gm].view(np.uint8), wv[wm].view(np.uint8)), kind
                else:
                    np.testing.assert_allclose(gv[gm].astype(np.float64), wv[wm].astype(np.float64),
                                               rtol=1e-5 if gv.dtype == np.float32 else 1e-11, atol=0, err_msg=f"{kind} ddof={ddof}")


@pytest.mark.parametrize("vdt", [np.int32, np.float64], ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("card", [20, 40_000], ids=["wave_mode", "thread_mode"])
def test_csr_seams_with_a_heavy_key(oracle, vdt, card, monkeypatch):
    """One key on 30 % of the rows (a 180 000-row group beside short ones): the group is cut into chunks of 4 096 rows for
    sum / mean / min / max / var / std (k_agg_groups_long: several workgroups, fixed-order merge) and the values are
    sorted inside the groups with two device-wide sorts for median / quantile / n_unique (no segment is one workgroup's)."""
    rng = np.random.default_rng(211)
    n = 600_000
    k = rng.integers(0, card, n).astype(np.int32)
    k[rng.random(n) < 0.3] = 7
    keys = [Column("k", k)]
    dt = np.dtype(vdt)
    v = rand_col(rng, "v", vdt, n, null_frac=0.1, lo=-1000 if dt.kind == "i" else None, hi=1000 if dt.kind == "i" else None,
                 nan_frac=0.01 if dt.kind == "f" else 0.0)
    monkeypatch.delenv("PLHIP_FORCE_TWO_SORTS", raising=False)
    g_o, g_p = oracle.Groups(keys, True), pa.group_tuples(keys, True)
    assert g_p.n_groups == len(np.unique(k))
    for kind in ALL_AGGS:
        wv, wm = oracle.agg(None if kind == "count" else v, kind, g_o)
        col = pa.agg_groups(None if kind == "count" else v, kind, g_p)
        gv, gm = col.to_numpy(), col.valid_mask()
        gm = np.ones(len(gv), bool) if gm is None else gm
        assert np.array_equal(gm, wm), f"{kind}: validity"
        assert gv.dtype == wv.dtype, kind
        if gv.dtype.kind == "f":
            np.testing.assert_allclose(gv[gm], wv[wm], rtol=1e-11, atol=0, equal_nan=True, err_msg=kind)
        else:
            assert np.array_equal(gv[gm], wv[wm]), kind
    for kind in ("first", "last", "var", "std", "median", "n_unique"):
        wv, wm = oracle.agg_ext(v, kind, g_o, 1)
        got = pa.agg_groups(v, kind, g_p, ddof=1)
        gv, gm = got.to_numpy(), got.valid_mask()
        gm = np.ones(len(gv), bool) if gm is None else gm
        assert gv.dtype == wv.dtype and np.array_equal(gm, wm), kind
        if kind in ("first", "last", "n_unique"):
            assert np.array_equal(gv[gm].view(np.uint8), wv[wm].view(np.uint8)), kind
        else:
            np.testing.assert_allclose(gv[gm].astype(np.float64), wv[wm].astype(np.float64), rtol=1e-10, atol=0, err_msg=kind)
    # the two-sort route on uniform groups too (forced): the layout it produces must be the segmented sort's
    monkeypatch.setenv("PLHIP_FORCE_TWO_SORTS", "1")
    for kind in ("median", "n_unique"):
        wv, wm = oracle.agg_ext(v, kind, g_o, 1)
        got = pa.agg_groups(v, kind, g_p)
        gv, gm = got.to_numpy(), got.valid_mask()
        gm = np.ones(len(gv), bool) if gm is None else gm
        assert np.array_equal(gm, wm) and np.array_equal(gv[gm], wv[wm], equal_nan=gv.dtype.kind == "f"), kind
    from polars_amd.frame import agg_groups_quantile

    got = agg_groups_quantile(v, 0.3, g_p, "linear")
    wv, wm = oracle.agg_quantile(v, 0.3, "linear", g_o)
    gv, gm = got.to_numpy(), got.valid_mask()
    gm = np.ones(len(gv), bool) if gm is None else gm
    assert np.array_equal(gm, wm) and np.array_equal(gv[gm], wv[wm], equal_nan=True), "quantile 0.3 linear"


def test_eager_group_by_var_std_first_last():
    # test_group_by_var (crates/polars-core/src/frame/group_by/mod.rs:1134-1156) through the eager GroupBy mirror
    df = pa.DataFrame({"g": np.array([4, 4, 2], dtype=np.int32), "flt": np.array([1.0, 2.0, 3.0]), "int": np.array([1, 2, 3], dtype=np.int32)})
    out = df.group_by("g", maintain_order=True).select(["int"]).var(1)
    assert out.columns == ["g", "int_agg_var"] and out["g"].to_list() == [4, 2] and out["int_agg_var"].to_list() == [0.5, None]
    out = df.group_by("g", maintain_order=True).select(["int"]).std(1)
    assert abs(out["int_agg_std"].to_list()[0] - 2 ** -0.5) < 1e-6
    out = df.group_by("g", maintain_order=True).last()
    assert out.columns == ["g", "flt_last", "int_last"] and out["flt_last"].to_list() == [2.0, 3.0] and out["int_last"].to_list() == [2, 3]
    out = df.group_by("g", maintain_order=True).first()
    assert out["flt_first"].to_list() == [1.0, 3.0]


def test_group_by_shorthands_first_last_median():
    """test_group_by_shorthands (py-polars/tests/unit/operations/test_group_by.py:59-93): a = [1..5] by b = [a, a, b, b, b],
    c = [None, 1, None, 1, None] -> first (1, None), (3, None); last (2, 1), (5, None); median (1.5, 1.0), (4.0, 1.0)."""
    df = pa.DataFrame({"a": np.array([1, 2, 3, 4, 5], dtype=np.int64), "b": np.array([10, 10, 20, 20, 20], dtype=np.int32),
                       "c": [None, 1, None, 1, None]})
    gb = lambda: df.group_by("b", maintain_order=True)
    assert gb().first().rows() == [(10, 1, None), (20, 3, None)]
    assert gb().last().rows() == [(10, 2, 1), (20, 5, None)]
    assert gb().median().rows() == [(10, 1.5, 1.0), (20, 4.0, 1.0)]
    assert gb().n_unique().rows() == [(10, 2, 2), (20, 3, 2)]  # test_group_by.py:81: the nulls are one value


def test_agg_groups_quantile(oracle):
    """plhip_agg_groups_quantile against the reference's exact expectations (test_quantile, aggregate/mod.rs:789-944; the
    column as one group) and against the oracle's restatement of quantile_idx / generic_quantile on random groups with
    nulls: every interpolation, several quantiles, all value dtype classes; bit-exact (the arithmetic is a handful of
    f64 operations in a fixed order).  test_group_by_shorthand_quantile (test_group_by.py:96-102) through the mirror."""
    from tests.util import QUANTILE_KNOWN, quantile_known_column

    for data, table in QUANTILE_KNOWN:
        col = quantile_known_column(data)
        g = pa.group_tuples([Column("g", np.zeros(len(data), dtype=np.int32))], True)
        for interp, qs in table.items():
            for q, want in qs.items():
                got = pa.frame.agg_groups_quantile(col, q, g, interp)
                assert got.to_numpy().dtype == np.float64 and got.to_list() == [want], (data, interp, q, got.to_list())
    rng = np.random.default_rng(139)
    for dt in (np.int32, np.uint64, np.float32, np.float64, np.int8):
        for n, card in ((20000, 3000), (30000, 7)):
            keys = [rand_col(rng, "k", np.int64, n, card=card)]
            isf = np.dtype(dt).kind == "f"
            v = rand_col(rng, "v", dt, n, null_frac=0.2, lo=None if isf else 0, hi=None if isf else 100)
            g_o, g_p = oracle.Groups(keys, True), pa.group_tuples(keys, True)
            for interp in oracle.INTERPOLATIONS:
                for q in (0.0, 0.1, 0.5, 0.73, 1.0):
                    wv, wm = oracle.agg_quantile(v, q, interp, g_o)
                    got = pa.frame.agg_groups_quantile(v, q, g_p, interp)
                    gv, gm = got.to_numpy(), got.valid_mask()
                    gm = np.ones(len(gv), bool) if gm is None else gm
                    assert gv.dtype == wv.dtype and np.array_equal(gm, wm), (dt, interp, q)
                    assert np.array_equal(gv[gm].view(np.uint8), wv[wm].view(np.uint8)), (dt, interp, q)
            wv, wm = oracle.agg_quantile(v, 1.5, "linear", g_o)
            got = pa.frame.agg_groups_quantile(v, 1.5, g_p, "linear")
            assert got.to_numpy().dtype == wv.dtype and not got.valid_mask().any()
    df = pa.DataFrame({"a": np.array([1, 2, 3, 4, 5], dtype=np.int64), "b": np.array([10, 10, 20, 20, 20], dtype=np.int32),
                       "c": [None, 1, None, 1, None]})
    out = df.group_by("b", maintain_order=True).quantile(0.5)
    assert out.columns == ["b", "a_quantile_0.50", "c_quantile_0.50"] and out.rows() == [(10, 2.0, 1.0), (20, 4.0, 1.0)]
    with pytest.raises(pa.PlHipError):
        df.group_by("b").quantile(1.5)


def test_agg_groups_n_unique_edges(oracle):
    """agg_n_unique (dispatch.rs:82-107 over unique/mod.rs:136-171): test_group_by_agg_n_unique_floats (test_df.py:1726-1734:
    b = [1.0, 2.0 | 2.0] by a = [1, 1 | 3] -> [2, 1], f32 and f64); -0.0 == 0.0, every NaN counts; a value equal to the
    dtype's maximum beside nulls (they share the sort code and are told apart by the per-group marks); i64 values that
    differ only below f64 precision; an empty group (slice of nothing) stays null."""
    nan = float("nan")
    for dt in (np.float32, np.float64):
        g = pa.group_tuples([Column("a", np.array([1, 1, 3], dtype=np.int64))], True)
        got = pa.agg_groups(Column("b", np.array([1.0, 2.0, 2.0], dtype=dt)), "n_unique", g)
        assert got.to_numpy().dtype == np.uint32 and got.to_list() == [2, 1]
    cases = [
        Column("f", np.array([-0.0, 0.0, nan, nan, 1.0, 1.0, nan, 5.0], dtype=np.float64)),
        Column("f", np.array([-0.0, 0.0, nan, nan, 1.0, 1.0, nan, 5.0], dtype=np.float32), np.array([1, 1, 1, 0, 1, 1, 1, 0], bool)),
        Column("u", np.array([2**64 - 1, 2**64 - 1, 0, 7, 2**64 - 1, 7, 1, 1], dtype=np.uint64), np.array([1, 0, 1, 1, 1, 0, 1, 1], bool)),
        Column("i", np.array([2**63 - 1, 2**63 - 2, 2**53, 2**53 + 1, -2**63, 2**53 + 1, 0, 0], dtype=np.int64), np.array([1, 1, 1, 1, 1, 1, 0, 0], bool)),
        Column("b", np.array([-128, 127, 127, 0, 0, -128, 5, 5], dtype=np.int8)),
    ]
    keys = [Column("k", np.array([1, 1, 1, 1, 2, 2, 2, 2], dtype=np.int32))]
    g_o, g_p = oracle.Groups(keys, True), pa.group_tuples(keys, True)
    for v in cases:
        wv, wm = oracle.agg_ext(v, "n_unique", g_o)
        got = pa.agg_groups(v, "n_unique", g_p)
        assert got.to_numpy().tolist() == wv.tolist() and wm.all() and (got.valid_mask() is None or got.valid_mask().all()), v.name
    nan = float("nan")
    f = pa.DataFrame({"k": np.zeros(5, dtype=np.int32), "x": np.array([3.0, nan, 1.0, 2.0, nan])})
    assert f.group_by("k").median()["x_median"].to_list() == [3.0]  # NaN sorts last: [1, 2, 3, nan, nan] -> 3


def test_agg_groups_var_known_answer():
    # crates/polars-core/src/frame/group_by/mod.rs:1134-1156
    g = pa.group_tuples([Column("g", np.array([7, 7, 9], dtype=np.int32))], True)
    x = Column("int", np.array([1, 2, 3], dtype=np.int32))
    v = pa.agg_groups(x, "var", g, ddof=1)
    assert v.to_list() == [0.5, None]
    s = pa.agg_groups(x, "std", g, ddof=1)
    assert abs(s.to_list()[0] - 2 ** -0.5) < 1e-6 and s.to_list()[1] is None


def test_group_tuples_empty_and_errors():
    g = pa.group_tuples([Column("k", np.empty(0, np.int32))])
    assert g.n_groups == 0 and g.n_rows == 0
    with pytest.raises(pa.ShapeMismatch):
        pa.group_tuples([Column("a", np.zeros(3, np.int32)), Column("b", np.zeros(4, np.int32))])


def test_sharded_engine_two_shards_one_gpu(oracle):
    """The two local passes of polars_amd/distributed.py on the GPU (HipEngine.partial / key_hashes / merge)
    and its finalize step, with the all-to-all replaced by its definition (rows go to owner = hash % world):
    two row shards, regrouped per owner, must give the single-pass result."""
    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(53)
    n, world = 60000, 2
    dev = torch.device("cuda:0")
    k1 = rng.integers(-30, 30, n).astype(np.int32); k1m = rng.random(n) > 0.05
    k2 = rng.integers(0, 9, n).astype(np.int64)
    v0 = rng.integers(-1000, 1000, n).astype(np.int32)
    v1 = np.round(rng.uniform(0, 100, n), 6); v1m = rng.random(n) > 0.1
    v2 = rng.integers(0, 50, n).astype(np.uint32)
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (2, "mean"), (2, "sum")]
    eng = pd_.HipEngine(dev)
    shards = []
    for r in range(world):
        lo, hi = r * n // world, (r + 1) * n // world
        keys = [Column("k1", k1[lo:hi], k1m[lo:hi]), Column("k2", k2[lo:hi])]
        vals = [Column("v0", v0[lo:hi]), Column("v1", v1[lo:hi], v1m[lo:hi]), Column("v2", v2[lo:hi])]
        part = eng.partial(keys, vals, aggs)
        owner = ((eng.key_hashes(part[:2]) >> 33) & 0x7FFFFFFF) % world
        shards.append((part, owner))
    names = ["s", "m", "mn", "mx", "c", "m2", "s2"]
    outs = []
    for r in range(world):  # what rank r holds after the exchange
        cols = []
        for j in range(len(shards[0][0])):
            ts, ms = [], []
            for part, owner in shards:
                sel = owner == r
                ts.append(part[j].t[sel])
                ms.append((part[j].mask if part[j].mask is not None else torch.ones_like(sel))[sel])
            c0 = shards[0][0][j]
            has_mask = any(p[j].mask is not None for p, _ in shards)
            cols.append(pd_.Col(c0.name, torch.cat(ts).contiguous(), c0.code, torch.cat(ms) if has_mask else None))
        merged = eng.merge(cols[:2], cols[2:], pd_._merge_ops(aggs))
        outs.append(pd_.finalize(merged[:2], merged[2:], aggs, [_ffi.I32, _ffi.F64, _ffi.U32], names))
    got = []
    for j in range(len(outs[0])):
        v = np.concatenate([o[j].t.cpu().numpy().view(_ffi.code_np_dtype(o[j].code)) for o in outs])
        m = np.concatenate([(np.ones(o[j].t.numel(), bool) if o[j].mask is None else o[j].mask.cpu().numpy()) for o in outs])
        got.append((v, m))
    keys = [Column("k1", k1, k1m), Column("k2", k2)]
    vals = [Column("v0", v0), Column("v1", v1, v1m), Column("v2", v2)]
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what="two shards, one GPU")


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_slotted_exchange_one_gpu(oracle, world):
    """plhip_result_pack_slots / plhip_merge_slots with the equal-split all-to-all replaced by its definition
    (rank r receives slot r of every shard); one shard is empty.  Then a slot capacity that is too small: the
    headers must say so (and the clamped slots must still hold exactly slot_rows records' worth of space)."""
    import ctypes as C

    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(67)
    n = 60000
    K, V = _shard_inputs(rng, n)
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (2, "mean"), (2, "sum")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    bounds = [0] + [n * (r + 1) // (world - 1) for r in range(world - 1)] + [n]
    slot_rows = 400
    shard_cols, sends = [], []
    for r in range(world):
        keys, vals = _shard_columns(K, V, bounds[r], bounds[r + 1])
        shard_cols.append((keys, vals))
        send, counts = eng.pack(keys, vals, aggs, world, slot_rows=slot_rows)
        sends.append((send, counts))
    schema = pd_.partial_schema(*shard_cols[0], aggs)
    rec = sum(_ffi.code_np_dtype(c).itemsize + (1 if b else 0) for c, b in schema)
    stride = int(_ffi.lib().plhip_slot_bytes(rec, slot_rows))
    names = ["s", "m", "mn", "mx", "c", "m2", "s2"]
    outs = []
    for r in range(world):
        recv = torch.cat([send.view(world, stride)[r] for send, _ in sends]).contiguous()
        hdr = recv.view(world, stride)[:, :16].cpu().numpy().view(np.uint64)
        assert not hdr[:, 1].any()
        assert hdr[:, 0].tolist() == [counts[r] for _, counts in sends]
        rc = (C.c_int64 * world)(*[int(x) for x in hdr[:, 0]])
        keys, vals = shard_cols[r]
        merged = eng._merge_call(recv, schema, 2, aggs, 0, sum(rc), slots=(world, slot_rows, rc))
        outs.append(eng.finalize_frame(merged, keys, vals, aggs, names))
    got = []
    for j in range(len(outs[0])):
        v = np.concatenate([o[j].t.cpu().numpy().view(_ffi.code_np_dtype(o[j].code)) for o in outs])
        m = np.concatenate([(np.ones(o[j].t.numel(), bool) if o[j].mask is None else o[j].mask.cpu().numpy()) for o in outs])
        got.append((v, m))
    keys, vals = _shard_columns(K, V, 0, n)
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what=f"slotted exchange, world {world}")
    # overflow: 549 groups over `world` owners cannot fit slots of 16 rows
    keys, vals = shard_cols[0]
    send, counts = eng.pack(keys, vals, aggs, world, slot_rows=16)
    stride = int(_ffi.lib().plhip_slot_bytes(rec, 16))
    hdr = send.view(world, stride)[:, :16].cpu().numpy().view(np.uint64)
    assert max(counts) > 16 and hdr[:, 1].all() and hdr[:, 0].tolist() == counts


# ---------------------------------------------------------------------------------------------------
# sorted-key fast path and slice pushdown (a17)
# ---------------------------------------------------------------------------------------------------
def _sorted_frame(key_col, n):
    v = np.arange(n, dtype=np.int64)
    return pa.DataFrame({"a": key_col.set_sorted(), "v": Column("v", v)})


@pytest.mark.parametrize("vals,want", __import__("tests.test_oracle_golden", fromlist=["x"]).SORTED_KEY_CASES)
def test_sorted_key_known_answers(vals, want):
    """crates/polars/tests/it/core/group_by.rs:6-96 + sort_partition.rs:143-160: the [first, len] slices, read back
    here as (min row, count) per output row, in row order."""
    from tests.test_oracle_golden import sorted_case_column

    out = _sorted_frame(sorted_case_column(vals), len(vals)).group_by("a").agg(pa.col("v").min().alias("first"), pa.count().alias("len"))
    assert pa.last_strategy() == "sorted_keys"
    assert [[f, l] for f, l in zip(out["first"].to_list(), out["len"].to_list())] == want
    keys = out["a"].to_list()
    assert keys == [vals[f] for f, _ in want]


def test_sorted_key_value_semantics_and_trusted_flag(oracle):
    nan = float("nan")
    f = np.array([-0.0, 0.0, 1.5, 1.5, nan, nan], dtype=np.float64)
    out = _sorted_frame(Column("a", f), len(f)).group_by("a").agg(pa.col("v").min().alias("first"), pa.count().alias("len"))
    assert [[a, b] for a, b in zip(out["first"].to_list(), out["len"].to_list())] == oracle.sorted_key_groups(Column("a", f))
    # the flag is trusted: unsorted input yields one row per run, like the reference
    u = np.array([3, 3, 1, 3, 3, 3, 2], dtype=np.int32)
    out = _sorted_frame(Column("a", u), len(u)).group_by("a").agg(pa.count().alias("len"))
    assert out["a"].to_list() == [3, 1, 3, 2] and out["len"].to_list() == [2, 1, 3, 1]


@pytest.mark.parametrize("descending", [False, True])
@pytest.mark.parametrize("nulls", ["none", "first", "last"])
def test_sorted_key_parity_large(oracle, descending, nulls):
    """Really sorted keys: the runs are the groups, so the result must equal the stable hash group-by of the
    oracle, row for row (same order), incl. long runs (the per-workgroup fold) and a run of nulls."""
    rng = np.random.default_rng(71)
    n = 300_000
    k = np.sort(np.concatenate([rng.integers(-5000, 5000, n - 20000), np.full(20000, 77)]).astype(np.int64))
    if descending:
        k = k[::-1].copy()
    mask = None
    if nulls != "none":
        mask = np.ones(n, bool)
        if nulls == "first":
            mask[:1234] = False
        else:
            mask[-1234:] = False
    v0 = rng.integers(-100, 100, n).astype(np.int32)
    v1 = np.round(rng.uniform(0, 100, n), 6)
    v1m = rng.random(n) > 0.1
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count")]
    df = pa.DataFrame({"k": Column("k", k, mask).set_sorted(descending), "v0": Column("v0", v0), "v1": Column("v1", v1, v1m)})
    got = df.group_by("k").agg(pa.col("v0").sum().alias("a0"), pa.col("v1").mean().alias("a1"), pa.col("v0").min().alias("a2"),
                               pa.col("v1").max().alias("a3"), pa.count().alias("a4"))
    assert pa.last_strategy() == "sorted_keys"
    want, _ = oracle.group_by_agg([Column("k", k, mask)], [Column("v0", v0), Column("v1", v1, v1m)], aggs, maintain_order=True)
    g, w = util.frame_to_np(got), util.oracle_to_np(want)
    assert np.array_equal(g[0][0][g[0][1] if g[0][1] is not None else slice(None)], w[0][0][w[0][1] if w[0][1] is not None else slice(None)])
    util.assert_frames_match(g, w, 1, what=f"sorted keys desc={descending} nulls={nulls}")


@pytest.mark.parametrize("sl", [(0, 5), (3, 4), (-4, 2), (-100, 3), (7, 100), (50, 2), (0, 0)])
def test_slice_pushdown(oracle, sl):
    """GroupByExec's slice (group_by.rs:79-82) with slice_offsets semantics (utils/mod.rs:199-218) on the stable
    group order."""
    rng = np.random.default_rng(73)
    n = 5000
    k = rng.integers(0, 12, n).astype(np.int32); km = rng.random(n) > 0.05
    v = np.round(rng.uniform(0, 10, n), 3); vm = rng.random(n) > 0.2
    df = pa.DataFrame({"k": Column("k", k, km), "v": Column("v", v, vm)})
    got = df.lazy().group_by("k", maintain_order=True).agg(pa.col("v").sum().alias("s"), pa.col("v").max().alias("m"), pa.count().alias("c")).slice(*sl).collect()
    want, _ = oracle.group_by_agg([Column("k", k, km)], [Column("v", v, vm)], [(0, "sum"), (0, "max"), (0, "count")], maintain_order=True)
    w = util.oracle_to_np(want)
    g_all = len(w[0][0])
    off, ln = sl
    if off < 0:
        start, cnt = (g_all + off, min(ln, -off)) if -off <= g_all else (0, min(ln, g_all))
    else:
        start, cnt = (off, min(ln, g_all - off)) if off <= g_all else (g_all, 0)
    w = [(vv[start:start + cnt], None if mm is None else mm[start:start + cnt]) for vv, mm in w]
    g = util.frame_to_np(got)
    assert len(g[0][0]) == cnt
    if cnt:
        util.assert_frames_match(g, w, 1, what=f"slice {sl}")
        gm = g[0][1] if g[0][1] is not None else np.ones(cnt, bool)
        wm = w[0][1] if w[0][1] is not None else np.ones(cnt, bool)
        assert np.array_equal(gm, wm) and np.array_equal(g[0][0][gm], w[0][0][wm])  # same rows in the same order


def _shard_inputs(rng, n):
    k1 = rng.integers(-30, 30, n).astype(np.int32); k1m = rng.random(n) > 0.05
    k2 = rng.integers(0, 9, n).astype(np.int64)
    v0 = rng.integers(-1000, 1000, n).astype(np.int32)
    v1 = np.round(rng.uniform(0, 100, n), 6); v1m = rng.random(n) > 0.1
    v2 = rng.integers(0, 50, n).astype(np.uint32)
    return (k1, k1m, k2), (v0, v1, v1m, v2)


def _shard_columns(K, V, lo, hi):
    k1, k1m, k2 = K
    v0, v1, v1m, v2 = V
    return ([Column("k1", k1[lo:hi], k1m[lo:hi]), Column("k2", k2[lo:hi])],
            [Column("v0", v0[lo:hi]), Column("v1", v1[lo:hi], v1m[lo:hi]), Column("v2", v2[lo:hi])])


def test_pack_by_owner_matches_its_definition():
    """plhip_result_pack_by_owner: counts = rows per owner, every record sits in its owner's bucket
    (owner = ((vec_hash >> 33) & 0x7FFFFFFF) % world, recomputed here from plhip_vec_hash), and the records are
    exactly the frame's rows; plhip_records_to_frame gives the columns back."""
    import ctypes as C

    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(59)
    n, world = 50000, 5
    K, V = _shard_inputs(rng, n)
    keys, vals = _shard_columns(K, V, 0, n)
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    part = eng.partial(keys, vals, aggs)  # the frame as tensors (python path)
    owner = (((eng.key_hashes(part[:2]) >> 33) & 0x7FFFFFFF) % world).cpu().numpy()
    send, counts = eng.pack(keys, vals, aggs, world)
    g = part[0].t.numel()
    assert send.shape[0] == g and sum(counts) == g
    assert counts == np.bincount(owner, minlength=world).tolist()
    schema = pd_.partial_schema(keys, vals, aggs)
    # decode the records on the host
    rec = send.cpu().numpy()
    off, cols = 0, []
    for code, nullable in schema:
        w = _ffi.code_np_dtype(code).itemsize
        v = np.ascontiguousarray(rec[:, off:off + w]).view(_ffi.code_np_dtype(code)).reshape(-1)
        off += w
        m = None
        if nullable:
            m = rec[:, off].astype(bool); off += 1
        cols.append((v, m))
    assert off == rec.shape[1]

    def by_key(cols_vm):  # rows ordered by the (unique) key tuple; null slots zeroed
        out = []
        for v, m in cols_vm:
            out.append(v if m is None else np.where(m, v, np.zeros_like(v)))
            if m is not None:
                out.append(m)
        order = np.lexsort((out[2], out[0], out[1]))  # k1 mask, k1, k2
        return [c[order] for c in out]

    want = [(c.t.cpu().numpy().view(_ffi.code_np_dtype(c.code)), None if c.mask is None else c.mask.cpu().numpy()) for c in part]
    for gc, wc in zip(by_key(cols), by_key(want)):  # two runs of the partial pass: f64 sums differ in the last bits
        if gc.dtype.kind == "f":
            np.testing.assert_allclose(gc, wc, rtol=1e-12)
        else:
            assert np.array_equal(gc, wc)
    # every bucket holds only rows of its owner: recompute the owner of the packed rows' keys
    pk = [pd_.Col("k1", torch.from_numpy(cols[0][0].copy()).cuda(), _ffi.I32, torch.from_numpy(cols[0][1].copy()).cuda()),
          pd_.Col("k2", torch.from_numpy(cols[1][0].copy()).cuda(), _ffi.I64)]
    o2 = (((eng.key_hashes(pk) >> 33) & 0x7FFFFFFF) % world).cpu().numpy()
    assert np.array_equal(o2, np.repeat(np.arange(world), counts))
    # and back to columns
    frame = C.c_void_p()
    dts = (C.c_int32 * len(schema))(*[c for c, _ in schema])
    nul = (C.c_uint8 * len(schema))(*[int(b) for _, b in schema])
    L = _ffi.lib()
    _ffi.check(L.plhip_records_to_frame(send.data_ptr(), g, dts, nul, len(schema), C.byref(frame)))
    cs_list = []
    for i in range(len(schema)):
        cs = _ffi.ColumnStruct()
        _ffi.check(L.plhip_result_column(frame, i, C.byref(cs)))
        cs_list.append(cs)
    back = eng._frame(frame, cs_list, [f"c{i}" for i in range(len(schema))])
    got = [(c.t.cpu().numpy().view(_ffi.code_np_dtype(c.code)), None if c.mask is None else c.mask.cpu().numpy()) for c in back]
    for (gv, gm), (wv, wm) in zip(got, cols):
        assert (gm is None) == (wm is None)
        if gm is not None:
            assert np.array_equal(gm, wm)
            assert np.array_equal(gv[gm], wv[wm])
        else:
            assert np.array_equal(gv, wv)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_library_exchange_one_gpu(oracle, world):
    """HipEngine.pack / merge_records (the library-side bucketing and unpacking of the exchange) with the
    all-to-all replaced by its definition: rank r receives bucket r of every shard.  One shard is empty."""
    import torch

    from polars_amd import distributed as pd_

    rng = np.random.default_rng(61)
    n = 60000
    K, V = _shard_inputs(rng, n)
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (2, "mean"), (2, "sum")]
    eng = pd_.HipEngine(torch.device("cuda:0"))
    bounds = [0] + [n * (r + 1) // (world - 1) for r in range(world - 1)] + [n]  # the last shard is empty
    packed, shard_cols = [], []
    for r in range(world):
        keys, vals = _shard_columns(K, V, bounds[r], bounds[r + 1])
        shard_cols.append((keys, vals))
        packed.append(eng.pack(keys, vals, aggs, world))
    names = ["s", "m", "mn", "mx", "c", "m2", "s2"]
    outs = []
    for r in range(world):
        pieces = []
        for send, counts in packed:
            lo = sum(counts[:r])
            pieces.append(send[lo:lo + counts[r]])
        recv = torch.cat(pieces).contiguous()
        keys, vals = shard_cols[r]
        if r % 2 == 0:  # both finalize routes: torch on the regrouped columns, and the library's kernel
            merged = eng.merge_records(recv, keys, vals, aggs)
            outs.append(pd_.finalize(merged[:2], merged[2:], aggs, [_ffi.I32, _ffi.F64, _ffi.U32], names, [False, True, False]))
        else:
            outs.append(eng.finalize_frame(eng.merge_records(recv, keys, vals, aggs, raw=True), keys, vals, aggs, names))
    got = []
    for j in range(len(outs[0])):
        v = np.concatenate([o[j].t.cpu().numpy().view(_ffi.code_np_dtype(o[j].code)) for o in outs])
        m = np.concatenate([(np.ones(o[j].t.numel(), bool) if o[j].mask is None else o[j].mask.cpu().numpy()) for o in outs])
        got.append((v, m))
    keys, vals = _shard_columns(K, V, 0, n)
    want, _ = oracle.group_by_agg(keys, vals, aggs)
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what=f"library exchange, world {world}")


# ---------------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: whole-frame parity where the oracle finishes in seconds, size-independent
# properties (checksum of checksums, counts add up to N, distinct keys) beyond that
# ---------------------------------------------------------------------------------------------------
def _bench_frame(workload, n, hint=True):
    import torch

    import bench
    from polars_amd import h2oai

    keys_n, vals_n, aggs, wide = bench.WORKLOADS[workload]
    dev = torch.device("cuda:0")
    tensors = h2oai.generate(n, seed=108, device=dev, wide=wide, columns=keys_n + vals_n)
    cols = h2oai.to_columns(tensors, wide)
    keys, vals = [cols[k] for k in keys_n], [cols[v] for v in vals_n]
    h = 0
    if hint and not wide and all(k in ("id1", "id2", "id3") for k in keys_n):
        h = 1
        for k in keys_n:
            h *= 100 if k in ("id1", "id2") else max(n // 100, 1)
    return tensors, keys, vals, aggs, min(h, n)


def _run_raw(keys, vals, aggs, hint):
    from polars_amd.frame import _host_column

    codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
    handle, cols = pa.group_by_agg_raw(keys, vals, codes, 0, hint)
    try:
        got = [_host_column(str(i), cs) for i, cs in enumerate(cols)]
    finally:
        _ffi.lib().plhip_result_free(handle)
    return [(c.to_numpy(), c.valid_mask() if c.valid_mask() is not None else np.ones(len(c), bool)) for c in got]


@pytest.mark.parametrize("workload,expect", [("q1", "direct"), ("q3", "dense"), ("q5", "dense"), ("q3", "tilesort"), ("q5", "tilesort")])
@pytest.mark.parametrize("hint", [True, False])
def test_db_benchmark_full_size_vs_oracle(oracle, workload, expect, hint, monkeypatch):
    """q3 / q5 keys are 32-bit integers in a compact range: under AUTO one min / max pass makes them dense codes (strategy
    "dense", no flag from the caller); with that routing switched off the hashed tile sort takes them as in round 1."""
    n = 10_000_000
    if expect == "tilesort":
        monkeypatch.setenv("PLHIP_NO_RANGE_DENSE", "1")
    tensors, keys, vals, aggs, h = _bench_frame(workload, n, hint)
    got = _run_raw(keys, vals, aggs, h)
    assert pa.last_strategy() == expect
    host = {k: v.cpu().numpy() for k, v in tensors.items()}
    hk = [Column(c.name, host[c.name].view(np.uint32) if c.dtype == _ffi.U32 else host[c.name]) for c in keys]
    hv = [Column(c.name, host[c.name]) for c in vals]
    want, _ = oracle.group_by_agg(hk, hv, aggs, n_threads=8)
    util.assert_frames_match(got, util.oracle_to_np(want), len(keys), what=f"{workload} 1e7 rows")


@pytest.mark.parametrize("workload", ["q2", "q4", "q7", "q10"])
def test_db_benchmark_other_queries_full_size_vs_oracle(oracle, workload):
    """The other fused queries of the reference's own harness (py-polars/tests/benchmark/run_h2oai_benchmark.py:66-160) at its
    size, 1e7 rows: q2 sum(v1) by (id1, id2); q4 mean(v1), mean(v2), mean(v3) by id4; q7 max(v1), min(v2) by id3 (the harness
    subtracts the two result columns afterwards); q10 sum(v3), count by all six ids (about one row per group).  Whole frame
    against the oracle, under AUTO with the harness's knowledge (category counts of id1..id3 as the hint, dense codes for a
    single Categorical key)."""
    n = 10_000_000
    tensors, keys, vals, aggs, h = _bench_frame(workload, n, True)
    codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
    flags = _ffi.FLAG_KEYS_DENSE if (len(keys) == 1 and keys[0].n_categories is not None) else 0
    from polars_amd.frame import _host_column

    handle, cols = pa.group_by_agg_raw(keys, vals, codes, flags, h)
    try:
        got = [_host_column(str(i), cs) for i, cs in enumerate(cols)]
    finally:
        _ffi.lib().plhip_result_free(handle)
    got = [(c.to_numpy(), c.valid_mask() if c.valid_mask() is not None else np.ones(len(c), bool)) for c in got]
    host = {k: v.cpu().numpy() for k, v in tensors.items()}
    hk = [Column(c.name, host[c.name].view(np.uint32) if c.dtype == _ffi.U32 else host[c.name]) for c in keys]
    hv = [Column(c.name, host[c.name]) for c in vals]
    want, _ = oracle.group_by_agg(hk, hv, aggs, n_threads=8)
    util.assert_frames_match(got, util.oracle_to_np(want), len(keys), what=f"{workload} 1e7 rows via {pa.last_strategy()}")
    if workload == "q7":  # max / min have no straight-line dense kind: AUTO prefers the hashed tile sort (0.14 against 0.26 ms)
        assert pa.last_strategy() == "tilesort", pa.last_strategy()


@pytest.mark.parametrize("workload,dense,expect", [("q3", True, ("dense",)), ("q3", False, ("dense",)), ("q1", True, ("direct", "dense")), ("q5", False, ("dense",))])
def test_db_benchmark_na_variant_full_size_vs_oracle(oracle, workload, dense, expect):
    """db-benchmark's NA variant (groupby-datagen.R:30-44) at BASELINE config 2's size: 5 % of the distinct values of the
    key are nulled (so the null key is a heavy group), 5 % of every value column is NA.  Whole frame against the oracle."""
    import torch

    import bench
    from polars_amd import h2oai

    n = 10_000_000
    keys_n, vals_n, aggs, wide = bench.WORKLOADS[workload]
    tensors = h2oai.generate(n, seed=108, device=torch.device("cuda:0"), wide=wide, columns=keys_n + vals_n, na=True)
    cols = h2oai.to_columns(tensors, wide)
    keys, vals = [cols[k] for k in keys_n], [cols[v] for v in vals_n]
    assert all(c.validity is not None for c in keys + vals), "the NA variant carries validity on every column"
    hint = keys[0].n_categories or 0
    codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
    from polars_amd.frame import _host_column
    handle, out = pa.group_by_agg_raw(keys, vals, codes, _ffi.FLAG_KEYS_DENSE if dense else 0, hint)
    try:
        got = [_host_column(str(i), cs) for i, cs in enumerate(out)]
    finally:
        _ffi.lib().plhip_result_free(handle)
    assert pa.last_strategy() in expect
    got = [(c.to_numpy(), c.valid_mask()) for c in got]
    host = {k: v.cpu().numpy() for k, v in tensors.items()}
    mk = lambda c: Column(c.name, host[c.name].view(np.uint32) if c.dtype == _ffi.U32 else host[c.name], host[c.name + "__valid"])
    want, _ = oracle.group_by_agg([mk(c) for c in keys], [mk(c) for c in vals], aggs, n_threads=8)
    w = util.oracle_to_np(want)
    assert w[0][1] is not None and not w[0][1].all(), "the null key must be a group of its own"
    util.assert_frames_match(got, w, len(keys), what=f"{workload} NA variant, 1e7 rows")


def test_wrong_hints_cost_time_not_correctness(oracle):
    n = 2_000_000
    tensors, keys, vals, aggs, _ = _bench_frame("q3", n)
    host = {k: v.cpu().numpy() for k, v in tensors.items()}
    want, _ = oracle.group_by_agg([Column("id3", host["id3"].view(np.uint32))], [Column(c.name, host[c.name]) for c in vals], aggs, n_threads=8)
    for h in (7, 500, n):  # far too low (tables overflow -> the query is redone), low, far too high
        got = _run_raw(keys, vals, aggs, h)
        util.assert_frames_match(got, util.oracle_to_np(want), 1, what=f"hint {h} via {pa.last_strategy()}")


@pytest.mark.parametrize("strategy", ["tilesort2", "sort"])
def test_high_cardinality_properties(strategy):
    """BASELINE config 3 shape at 2e7 rows (three i64 keys, nearly all-unique tuples): properties that do not
    need a second implementation at this size, for both strategies of that regime (with about one row per group
    AUTO sorts; from two rows per group on it takes the two partition levels)."""
    import torch

    n = 20_000_000
    tensors, keys, vals, aggs, _ = _bench_frame("c3", n)
    pa.set_strategy(strategy)
    got = _run_raw(keys, vals, aggs, 0)
    # (forced two levels: with about one row per group their leaves go to the leaf sort kernel, reported as "leafsort")
    assert pa.last_strategy() == (strategy if strategy == "sort" else "leafsort")
    if strategy == "sort":  # nearly all-unique keys: AUTO takes the two levels with the leaf sort
        pa.set_strategy("auto")
        got = _run_raw(keys, vals, aggs, 0)
        assert pa.last_strategy() == "leafsort"
    k1, k2, k3 = (got[i][0] for i in range(3))
    s1, mn, mx, s3, mean3, cnt = (got[3 + i][0] for i in range(6))
    assert int(cnt.astype(np.int64).sum()) == n                                        # counts add up to N
    assert int(s1.sum()) == int(tensors["v1"].sum().item())                            # checksum of integer sums: exact
    tot3 = float(tensors["v3"].sum().item())
    assert abs(float(s3.sum()) - tot3) <= 1e-9 * abs(tot3)                             # f64 checksum of sums
    packed = (tensors["id1"] * 101 + tensors["id2"]) * (n // 100 + 1) + tensors["id3"]  # injective for this data
    assert len(k1) == int(torch.unique(packed).numel())                                # one row per distinct key tuple
    gp = (k1.astype(np.int64) * 101 + k2) * (n // 100 + 1) + k3
    assert len(np.unique(gp)) == len(gp)                                               # and no tuple twice
    assert (mn <= mx).all() and np.allclose(mean3 * cnt, s3, rtol=1e-12)
    one = cnt == 1                                                                     # single-row groups: sum = min = max
    assert (s1[one] == mn[one]).all() and (mn[one] == mx[one]).all()


def test_config3_full_size_properties():
    """BASELINE configs[2] at its stated size: 1e8 rows, keys (id1, id2, id3) as i64, i64 sum/min/max + f64 sum/mean +
    count, under AUTO.  The oracle needs minutes here, so the frame is checked through size-independent properties."""
    import torch

    n = 100_000_000
    tensors, keys, vals, aggs, _ = _bench_frame("c3", n)
    got = _run_raw(keys, vals, aggs, 0)
    assert pa.last_strategy() in ("sort", "tilesort2", "leafsort")
    k1, k2, k3 = (got[i][0] for i in range(3))
    s1, mn, mx, s3, mean3, cnt = (got[3 + i][0] for i in range(6))
    assert int(cnt.astype(np.int64).sum()) == n                                        # counts add up to N
    assert int(s1.sum()) == int(tensors["v1"].sum().item())                            # checksum of integer sums: exact
    tot3 = float(tensors["v3"].sum().item())
    assert abs(float(s3.sum()) - tot3) <= 1e-9 * abs(tot3)                             # f64 checksum of sums
    packed = (tensors["id1"] * 101 + tensors["id2"]) * (n // 100 + 1) + tensors["id3"]  # injective for this data
    assert len(k1) == int(torch.unique(packed).numel())                                # one row per distinct key tuple
    del packed
    gp = (k1.astype(np.int64) * 101 + k2) * (n // 100 + 1) + k3
    assert len(np.unique(gp)) == len(gp)                                               # and no tuple twice
    assert (mn <= mx).all() and np.allclose(mean3 * cnt, s3, rtol=1e-12)
    one = cnt == 1                                                                     # single-row groups: sum = min = max
    assert (s1[one] == mn[one]).all() and (mn[one] == mx[one]).all()
    assert int(mn.min()) == int(tensors["v1"].min().item()) and int(mx.max()) == int(tensors["v1"].max().item())


def test_config5_full_size_properties():
    """BASELINE configs[4] at its stated size: 1e9 rows, 1e8 ids, half Zipf(1.1) heavy hitters and half a uniform tail
    (bench.skewed_keys, seed 108), f64 sum/mean + i64 sum/min/max + count under AUTO; size-independent properties."""
    import torch

    import bench

    n, ids = 1_000_000_000, 100_000_000
    dev = torch.device("cuda:0")
    t = bench.skewed_keys(n, ids, 108, dev)
    keys, vals = [Column("k", t["k"])], [Column("v3", t["v3"]), Column("v1", t["v1"])]
    aggs = bench.WORKLOADS["c5"][2]
    got = _run_raw(keys, vals, aggs, 0)
    k = got[0][0]
    assert len(k) > ids // 2, f"{len(k)} groups via {pa.last_strategy()}"
    s3, mean3, s1, mn, mx, cnt = (got[1 + i][0] for i in range(6))
    assert int(cnt.astype(np.int64).sum()) == n
    assert int(s1.sum()) == int(t["v1"].sum().item())
    tot3 = float(t["v3"].sum().item())
    assert abs(float(s3.sum()) - tot3) <= 1e-9 * abs(tot3)
    uniq, ucnt = torch.unique(t["k"], return_counts=True)
    assert len(k) == int(uniq.numel()) and len(np.unique(k)) == len(k)
    order = np.argsort(k)
    assert np.array_equal(k[order], uniq.cpu().numpy())                                # the keys themselves
    assert np.array_equal(cnt[order].astype(np.int64), ucnt.cpu().numpy())             # and every group's length, exactly
    del uniq, ucnt
    assert (mn <= mx).all() and (mn >= 1).all() and (mx <= 5).all()
    assert np.allclose(mean3 * cnt, s3, rtol=1e-12)
    top = int(np.argmax(cnt))                                                          # the heaviest hitter: ~3.3e7 rows
    hot = t["k"] == int(k[top])
    assert int(cnt[top]) == int(hot.sum().item()) and int(s1[top]) == int(t["v1"][hot].sum().item())
    ref = float(t["v3"][hot].sum().item())
    assert abs(float(s3[top]) - ref) <= 1e-12 * abs(ref)


@pytest.mark.parametrize("hot", [False, True], ids=["near_unique", "with_heavy_hitters"])
def test_leaf_sort_vs_oracle(oracle, hot):
    """The leaf kernel of the two-level pipeline for about one row per group (k_leaf_sort: a wave sorts a leaf and folds
    its runs): three i64 keys, nearly all tuples distinct, a nullable f64 column; with heavy hitters some leaves exceed
    the 512 rows a wave sorts and are left to k_fold_runs (which skips the leaves already folded)."""
    rng = np.random.default_rng(91 + hot)
    n = 3_000_000
    k1 = rng.integers(1, 101, n).astype(np.int64)
    k2 = rng.integers(-50, 50, n).astype(np.int64)
    k3 = rng.integers(1, 30_001, n).astype(np.int64)
    if hot:  # two tuples own 2 % and 0.5 % of the rows
        h = rng.random(n)
        k1[h < 0.02], k2[h < 0.02], k3[h < 0.02] = 7, -3, 11
        sel = (h >= 0.02) & (h < 0.025)
        k1[sel], k2[sel], k3[sel] = 99, 49, 29_999
    keys = [Column("a", k1), Column("b", k2, rng.random(n) >= 0.01), Column("c", k3)]
    values = [rand_col(rng, "v", np.int64, n, lo=-10**9, hi=10**9), rand_col(rng, "w", np.float64, n, null_frac=0.1)]
    aggs = [(0, "sum"), (0, "min"), (0, "max"), (1, "sum"), (1, "mean"), (0, "count")]
    check_parity(oracle, keys, values, aggs, "tilesort2", what=f"leaf sort hot={hot}")
    assert pa.last_strategy() == "leafsort"
    got = check_parity(oracle, keys, values, aggs, "auto", what=f"leaf sort under AUTO hot={hot}")
    assert pa.last_strategy() == "leafsort" and got.height > n // 2


def test_two_levels_vs_oracle_mid_size(oracle):
    """TILESORT2 against the oracle where the oracle still finishes in seconds: 6e6 rows, ~1.9e6 groups (beyond one
    partition level), a nullable i64 key, nullable f64 values; forced, and as AUTO's own choice."""
    rng = np.random.default_rng(83)
    n = 6_000_000
    keys = [rand_col(rng, "k", np.int64, n, card=2_000_000, null_frac=0.001)]
    values = [rand_col(rng, "a", np.int32, n, lo=-1000, hi=1000), rand_col(rng, "b", np.float64, n, null_frac=0.1)]
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count")]
    for strategy in ("tilesort2", "auto"):
        check_parity(oracle, keys, values, aggs, strategy, what=f"two levels {strategy}")
        assert pa.last_strategy() == "tilesort2"


def test_stable_order_many_groups_takes_the_sort(oracle):
    """maintain_order beyond the LDS range: the sort strategy, its runs reordered by first row (hashing.rs:32-79)."""
    rng = np.random.default_rng(103)
    n = 2_000_000
    keys = [rand_col(rng, "k", np.int64, n, card=400_000, null_frac=0.001)]
    values = [rand_col(rng, "a", np.int32, n, lo=-1000, hi=1000), rand_col(rng, "b", np.float64, n, null_frac=0.1)]
    check_parity(oracle, keys, values, [(0, "sum"), (1, "mean"), (1, "min"), (0, "count")], "auto", maintain_order=True, what="stable, many groups")
    assert pa.last_strategy() == "sort"
    check_parity(oracle, keys, values, [(0, "sum"), (0, "count")], "sort", maintain_order=True, what="stable, forced sort")
    assert pa.last_strategy() == "sort"


def test_skewed_keys_estimate_and_auto(oracle):
    """BASELINE config 5's recipe at a size the oracle handles: half of the rows Zipf-like(1.1) heavy hitters (the top
    key ~3 % of all rows), half a uniform tail over 2e6 ids.  Heavy hitters eat the cardinality sample; the estimator
    must still size the pipeline for the ~2e6 groups (once-seen keys / GEE / second sample), AUTO must take the two
    partition levels and not fall through to the sort, and the result is the oracle's."""
    rng = np.random.default_rng(101)
    n, ids = 8_000_000, 2_000_000
    z = np.floor(np.maximum(rng.random(n), 1e-300) ** -10.0)
    k = np.where(rng.random(n) < 0.5, np.mod(np.minimum(z, 1e15) - 1, ids), np.floor(rng.random(n) * ids)).astype(np.int64)
    keys = [Column("k", k)]
    values = [Column("a", rng.integers(-1000, 1000, n).astype(np.int64)), Column("b", np.round(rng.uniform(0, 100, n), 6))]
    check_parity(oracle, keys, values, [(0, "sum"), (1, "mean"), (0, "count")], "auto", what="skewed keys")
    assert pa.last_strategy() == "tilesort2"


def test_two_levels_skew_falls_back(oracle):
    """A heavy key plus many light ones: if a leaf table overflows the query is redone by the sort strategy;
    either way the result is the oracle's."""
    rng = np.random.default_rng(89)
    n = 2_000_000
    k = rng.integers(0, 1_000_000, n).astype(np.int64)
    k[rng.random(n) < 0.3] = 7
    keys = [Column("k", k)]
    values = [rand_col(rng, "a", np.int64, n, lo=-1000, hi=1000)]
    check_parity(oracle, keys, values, [(0, "sum"), (0, "count"), (0, "max")], "tilesort2", what="two levels skew")


@pytest.mark.parametrize("key_dtype", [np.int32, np.int64])
def test_tile_sort_hot_partitions_are_split(oracle, monkeypatch, key_dtype):
    """Single-level tile sort under skew: a partition that holds a heavy hitter -- or the null keys, db-benchmark's NA
    variant -- is folded by several workgroups that plan for themselves from the per-partition long-run counts and whose
    dumps the last finisher merges (fast.hpp, FoldRunsArgs::self_hot).  A lowered chunk size splits several partitions at
    oracle size; every accumulator kind, nullable values; then the same query with the split switched off."""
    monkeypatch.setenv("PLHIP_HOT_CHUNK", "1024")
    rng = np.random.default_rng(149)
    n = 600_000
    k = rng.integers(0, 20_000, n).astype(key_dtype)
    u = rng.random(n)
    k[u < 0.25] = 7
    k[(u >= 0.25) & (u < 0.35)] = 12345
    k[(u >= 0.35) & (u < 0.38)] = -3
    km = rng.random(n) > 0.06
    keys = [Column("k", k, km)]
    values = [Column("a", rng.integers(-1000, 1000, n).astype(np.int64)), rand_col(rng, "b", np.float64, n, null_frac=0.1)]
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (1, "sum")]
    check_parity(oracle, keys, values, aggs, "tilesort", what="hot partitions split")
    assert pa.last_strategy() == "tilesort"
    check_parity(oracle, [Column("k", k)], values, aggs, "auto", what="hot partitions split, no null keys")
    monkeypatch.delenv("PLHIP_HOT_CHUNK")
    check_parity(oracle, keys, values, aggs, "tilesort", what="hot partitions, default chunk size")
    monkeypatch.setenv("PLHIP_NOHOT", "1")
    check_parity(oracle, keys, values, aggs, "tilesort", what="hot partitions, one workgroup each")


def test_sort_strategy_long_runs_are_split(oracle):
    """The sort strategy under skew: a few long runs (a heavy hitter, the null keys) are cut into chunks that the
    workgroups share, combined with global atomics, the last chunk writes the row (sort_path.hip, k_fold_sorted_long);
    every accumulator kind, nullable key and values, unordered and in stable order; then more than 64 long runs (one
    workgroup per run)."""
    rng = np.random.default_rng(151)
    n = 700_000
    k = rng.integers(0, 50_000, n).astype(np.int64)
    u = rng.random(n)
    k[u < 0.3] = 7
    k[(u >= 0.3) & (u < 0.4)] = -12345
    keys = [Column("k", k, rng.random(n) > 0.05)]
    values = [Column("a", rng.integers(-1000, 1000, n).astype(np.int64)), rand_col(rng, "b", np.float64, n, null_frac=0.1)]
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (1, "sum"), (1, "min")]
    check_parity(oracle, keys, values, aggs, "sort", what="long runs split")
    assert pa.last_strategy() == "sort"
    check_parity(oracle, keys, values, aggs, "sort", maintain_order=True, what="long runs split, stable order")
    keys = [Column("k", rng.integers(0, 100, n).astype(np.int32))]  # 100 runs of ~7 000 rows
    check_parity(oracle, keys, values, aggs, "sort", what="many long runs")


def test_two_levels_hot_leaves_are_split(oracle, monkeypatch):
    """Heavy hitters in the two-level pipeline: leaves with many rows are folded by several workgroups whose tables the
    last one merges (fast.hpp, HotState).  A lowered chunk size makes a dozen leaves split at a size the oracle
    handles; nullable key (the null group is a heavy hitter too), nullable values, every accumulator kind."""
    monkeypatch.setenv("PLHIP_T2_HOT_CHUNK", "4096")
    rng = np.random.default_rng(131)
    n, ids = 4_000_000, 1_500_000
    z = np.floor(np.maximum(rng.random(n), 1e-300) ** -10.0)
    k = np.where(rng.random(n) < 0.5, np.mod(np.minimum(z, 1e15) - 1, ids), np.floor(rng.random(n) * ids)).astype(np.int64)
    km = rng.random(n) > 0.02
    keys = [Column("k", k, km)]
    values = [Column("a", rng.integers(-1000, 1000, n).astype(np.int64)), rand_col(rng, "b", np.float64, n, null_frac=0.1)]
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (1, "sum")]
    check_parity(oracle, keys, values, aggs, "tilesort2", what="hot leaves split")
    assert pa.last_strategy() == "tilesort2"
    monkeypatch.setenv("PLHIP_T2_NOHOT", "1")
    check_parity(oracle, keys, values, aggs, "tilesort2", what="hot leaves, one workgroup each")


# ---------------------------------------------------------------------------------------------------
# the reference's FfiPlugin boundary (include/plhip_plugin.h), driven through the Arrow C Data Interface
# ---------------------------------------------------------------------------------------------------
import ctypes as C  # noqa: E402


class CArrowSchema(C.Structure):
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64),
                ("n_children", C.c_int64), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p),
                ("private_data", C.c_void_p)]


class CArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64),
                ("n_children", C.c_int64), ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p),
                ("release", C.c_void_p), ("private_data", C.c_void_p)]


class CSeriesExport(C.Structure):
    pass


_SERIES_RELEASE = C.CFUNCTYPE(None, C.POINTER(CSeriesExport))
CSeriesExport._fields_ = [("field", C.POINTER(CArrowSchema)), ("arrays", C.POINTER(C.POINTER(CArrowArray))), ("len", C.c_size_t),
                          ("release", _SERIES_RELEASE), ("private_data", C.c_void_p)]


def _export_series(name, chunks, keep):
    """export_series (crates/polars-ffi/src/lib.rs:50-68) with pyarrow as the Arrow implementation."""
    import pyarrow as pa_

    schema = CArrowSchema()
    pa_.field(name, chunks[0].type)._export_to_c(C.addressof(schema))
    arrs = [CArrowArray() for _ in chunks]
    for a, ch in zip(arrs, chunks):
        ch._export_to_c(C.addressof(a))
    ptrs = (C.POINTER(CArrowArray) * len(arrs))(*[C.pointer(a) for a in arrs])
    released = []
    rel = _SERIES_RELEASE(lambda p: released.append(1))
    keep += [schema, arrs, ptrs, rel]
    e = CSeriesExport(C.pointer(schema), C.cast(ptrs, C.POINTER(C.POINTER(CArrowArray))), len(arrs), rel, None)
    return e, arrs, released


@pytest.mark.parametrize("chunked", [False, True])
def test_ffi_plugin_boundary(oracle, chunked):
    import pyarrow as pa_

    L = _ffi.lib()
    L.plhip_plugin_group_by.argtypes = [C.POINTER(CSeriesExport), C.c_size_t]
    L.plhip_plugin_group_by.restype = CSeriesExport
    rng = np.random.default_rng(61)
    n = 50000
    k = rng.integers(-40, 40, n).astype(np.int32); km = rng.random(n) > 0.05
    v0 = rng.integers(-100, 100, n).astype(np.int64)
    v1 = np.round(rng.uniform(0, 100, n), 6); v1m = rng.random(n) > 0.1
    split = [0, n] if not chunked else [0, 777, 20001, n]

    def chunks(vals, mask=None):
        out = []
        for a, b in zip(split[:-1], split[1:]):
            out.append(pa_.array(vals[a:b], mask=None if mask is None else ~mask[a:b]))
        return out

    keep = []
    spec, spec_arrs, spec_rel = _export_series("plhip:k=1;0:sum,1:mean,1:max,0:count", [pa_.array([0], type=pa_.uint32())], keep)
    ek, k_arrs, k_rel = _export_series("key", chunks(k, km), keep)
    e0, a0, r0 = _export_series("v0", chunks(v0), keep)
    e1, a1, r1 = _export_series("v1", chunks(v1, v1m), keep)
    inputs = (CSeriesExport * 4)(spec, ek, e0, e1)
    out = L.plhip_plugin_group_by(inputs, 4)
    # the plugin owns its inputs (plugin.rs:42-48): every array and every series export was released
    for arrs in (spec_arrs, k_arrs, a0, a1):
        assert all(not a.release for a in arrs)
    assert all(len(r) == 1 for r in (spec_rel, k_rel, r0, r1))
    assert out.len == 1
    st = pa_.Array._import_from_c(C.addressof(out.arrays[0].contents), C.addressof(out.field.contents))
    out.release(C.byref(out))  # frees the boxes; the array contents now belong to pyarrow (polars-ffi lib.rs:36-48)
    assert isinstance(st, pa_.StructArray)
    assert [f.name for f in st.type] == ["key", "v0_sum", "v1_mean", "v1_max", "count"]
    assert [str(f.type) for f in st.type] == ["int32", "int64", "double", "double", "uint32"]
    got = []
    for i in range(5):
        col = st.field(i)
        valid = ~np.asarray(col.is_null())
        got.append((np.asarray(col.fill_null(0)), valid))
    want, _ = oracle.group_by_agg([Column("key", k, km)], [Column("v0", v0), Column("v1", v1, v1m)],
                                  [(0, "sum"), (1, "mean"), (1, "max"), (0, "count")])
    util.assert_frames_match(got, util.oracle_to_np(want), 1, what="FfiPlugin boundary")


def test_ffi_plugin_field_and_errors():
    import pyarrow as pa_

    L = _ffi.lib()
    L.plhip_plugin_group_by_field.argtypes = [C.POINTER(CArrowSchema), C.c_size_t]
    L.plhip_plugin_group_by_field.restype = CArrowSchema
    fields = (CArrowSchema * 3)()
    for i, f in enumerate([pa_.field("plhip:k=1;0:mean,0:sum,0:count", pa_.uint32()), pa_.field("id", pa_.uint32()), pa_.field("x", pa_.int16())]):
        f._export_to_c(C.addressof(fields[i]))
    out = L.plhip_plugin_group_by_field(fields, 3)
    t = pa_.Field._import_from_c(C.addressof(out))
    assert t.name == "group_by" and [(f.name, str(f.type)) for f in t.type] == [("id", "uint32"), ("x_mean", "double"), ("x_sum", "int64"),
                                                                                ("count", "uint32")]
    # no error channel at this ABI version: a bad query comes back as a valid, empty struct + plhip_last_error()
    L.plhip_plugin_group_by.argtypes = [C.POINTER(CSeriesExport), C.c_size_t]
    L.plhip_plugin_group_by.restype = CSeriesExport
    keep = []
    bad, _, _ = _export_series("not a query", [pa_.array([0], type=pa_.uint32())], keep)
    col, _, _ = _export_series("a", [pa_.array([1, 2, 3], type=pa_.int32())], keep)
    out = L.plhip_plugin_group_by((CSeriesExport * 2)(bad, col), 2)
    st = pa_.Array._import_from_c(C.addressof(out.arrays[0].contents), C.addressof(out.field.contents))
    out.release(C.byref(out))
    assert len(st) == 0 and b"plhip:k=" in L.plhip_last_error()


@pytest.mark.parametrize("pattern", ["sorted", "constant", "two_values", "stride_256", "blocks", "reverse"])
def test_adversarial_key_patterns_auto(oracle, pattern):
    """Inputs that defeat sampling or hashing assumptions (sorted keys make a block sample look low-cardinality,
    strided keys stress the multiplicative hash): AUTO must stay correct through whatever fallbacks it takes."""
    n = 1_500_000
    rng = np.random.default_rng(67)
    if pattern == "sorted":
        k = np.sort(rng.integers(0, 200_000, n)).astype(np.int64)
    elif pattern == "reverse":
        k = np.sort(rng.integers(0, 3_000, n))[::-1].astype(np.int32).copy()
    elif pattern == "constant":
        k = np.full(n, -5, dtype=np.int32)
    elif pattern == "two_values":
        k = (np.arange(n) % 2).astype(np.uint32) * 0xFFFFFFFF  # 0 and the u32 table sentinel, alternating
    elif pattern == "stride_256":
        k = (rng.integers(0, 50_000, n) * 256).astype(np.int64)
    else:
        k = (np.arange(n) // 1000).astype(np.int32)  # 1500 groups, each one contiguous block
    v = rng.integers(-50, 50, n).astype(np.int32)
    w = np.round(rng.uniform(0, 10, n), 3)
    keys, vals = [Column("k", k)], [Column("v", v), Column("w", w)]
    aggs = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count")]
    want, _ = oracle.group_by_agg(keys, vals, aggs, n_threads=8)
    got = run_product(keys, vals, aggs, False, "auto")
    util.assert_frames_match(util.frame_to_np(got), util.oracle_to_np(want), 1, what=f"{pattern} via {pa.last_strategy()}")
