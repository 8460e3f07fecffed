"""The dense-code path (PLHIP_FLAG_KEYS_DENSE, strategy "dense"): a Categorical key's physical codes index the tables
directly, the counterpart of group_tuples_perfect (crates/polars-core/src/frame/group_by/perfect.rs:24-185, 190-227).
Through the C ABI against the oracle and the reference's own known answers: keys, counts, integer aggregates and
validity bit-exact after the canonical key sort, f64 within 1e-12 relative, f32 within 1e-5."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from tests import util

pytestmark = pytest.mark.gpu

KA = util.load_known_answers()
ALL_AGGS = ["sum", "mean", "min", "max", "count"]


@pytest.fixture(autouse=True)
def _reset_strategy():
    yield
    pa.set_strategy("auto")


def run_dense(key: Column, values, aggs, n_categories, strategy="dense"):
    pa.set_strategy(strategy)
    k = Column("k", key.values, key.valid_mask()).set_categories(n_categories)
    cols = {"k": k}
    cols.update({f"v{j}": Column(f"v{j}", v.values, v.valid_mask()) for j, v in enumerate(values)})
    df = pa.DataFrame(cols)
    exprs = []
    for i, (ci, kind) in enumerate(aggs):
        e = pa.count() if kind == "count" else getattr(pa.col(f"v{ci}"), kind)()
        exprs.append(e.alias(f"agg{i}"))
    return df.group_by("k").agg(exprs)


def check(oracle, key, values, aggs, n_categories, expect="dense", what="", strategy="dense"):
    want, _ = oracle.group_by_agg([key], values, aggs)
    got = run_dense(key, values, aggs, n_categories, strategy)
    assert pa.last_strategy() == expect if isinstance(expect, str) else pa.last_strategy() in expect, \
        f"{what}: ran {pa.last_strategy()!r}, expected {expect!r}"
    util.assert_frames_match(util.frame_to_np(got), util.oracle_to_np(want), 1, what=what)
    return got


def codes(rng, n, card, null_frac=0.0):
    v = rng.integers(0, card, n).astype(np.uint32)
    mask = (rng.random(n) >= null_frac) if null_frac else None
    return Column("k", v, mask)


def dense_cases():
    out = []
    for case in KA["group_by"]:
        ks = case["keys"]
        if len(ks) == 1 and ks[0]["dtype"] == "u32" and case["compare"] in ("sorted_keys", "n_groups"):
            out.append(case)
    return out


@pytest.mark.parametrize("case", dense_cases(), ids=lambda c: c["name"])
def test_reference_known_answers_dense(case):
    key = util.make_column(case["keys"][0])
    values = [util.make_column(v) for v in case["values"]]
    data = [x for x in util.expand(case["keys"][0]["data"]) if x is not None]
    ncat = case.get("n_categories", (max(data) + 1) if data else 1)
    out = run_dense(key, values, [tuple(a) for a in case["aggs"]], ncat)
    assert pa.last_strategy() == "dense"
    cols = [out[c].to_list() for c in out.columns]
    dts = [_ffi.DTYPE_NAMES[out[c].dtype] for c in out.columns]
    util.check_known_answer(case, cols, dts)


def test_perfect_hash_fixture_takes_the_dense_path_under_auto():
    # test_perfect_hash_table_null_values_8663 (py-polars/tests/unit/operations/test_group_by.py:636-832)
    case = next(c for c in KA["group_by"] if c["name"].endswith("8663"))
    key = util.make_column(case["keys"][0])
    out = run_dense(key, [], [(0, "count")], case["n_categories"], strategy="auto")
    # (101 rows: under AUTO tiny inputs stay with the general path; the forced run above covers the dense kernels)
    util.check_known_answer(case, [out[c].to_list() for c in out.columns])


@pytest.mark.parametrize("vdt", [np.int8, np.int16, np.int32, np.int64, np.uint8, np.uint16, np.uint32, np.uint64, np.float32, np.float64],
                         ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("nulls", [False, True], ids=["nonull", "nulls"])
def test_value_dtypes_all_aggs(oracle, vdt, nulls):
    from tests.test_parity_gpu import rand_col

    rng = np.random.default_rng(7 + np.dtype(vdt).itemsize)
    n = 50_000
    key = codes(rng, n, 700, 0.05 if nulls else 0.0)
    lo, hi = (None, None) if np.dtype(vdt).itemsize < 8 else (-(1 << 40) if np.dtype(vdt).kind == "i" else 0, 1 << 40)
    v = rand_col(rng, "v", vdt, n, null_frac=0.1 if nulls else 0.0, lo=lo, hi=hi, nan_frac=0.02 if np.dtype(vdt).kind == "f" else 0.0)
    check(oracle, key, [v], [(0, a) for a in ALL_AGGS], 700, what=f"dense values {np.dtype(vdt).name} nulls={nulls}")


@pytest.mark.parametrize("workload", ["q1", "q3", "q3_nulls", "two_cols_all_aggs"])
def test_db_benchmark_shapes(oracle, workload):
    rng = np.random.default_rng(108)
    n = 2_000_000 if workload != "q3" else 10_000_000
    ncat = 100 if workload == "q1" else n // 100
    key = codes(rng, n, ncat, 0.05 if workload == "q3_nulls" else 0.0)
    v1 = Column("v1", rng.integers(1, 6, n).astype(np.int32))
    v3m = (rng.random(n) >= 0.05) if workload == "q3_nulls" else None
    v3 = Column("v3", np.round(rng.uniform(0, 100, n), 6), v3m)
    if workload == "q1":
        check(oracle, key, [v1], [(0, "sum")], ncat, what="dense q1")
    elif workload == "two_cols_all_aggs":
        check(oracle, key, [v1, v3], [(0, a) for a in ALL_AGGS] + [(1, a) for a in ALL_AGGS], ncat, what="dense all aggs")
    else:
        got = check(oracle, key, [v1, v3], [(0, "sum"), (1, "mean")], ncat, what=f"dense {workload}", strategy="auto")  # AUTO takes it
        if workload == "q3":
            assert got.height == len(np.unique(key.to_numpy()))


def test_mixed_sign_f64_cancellation(oracle):
    # values in +-1e6 with groups that nearly cancel: the partial sums are order dependent, the bar stays 1e-12 of the
    # group's magnitude (sum |x|), which is what a re-associated IEEE sum can promise
    rng = np.random.default_rng(5)
    n, ncat = 1_000_000, 5_000
    key = codes(rng, n, ncat)
    x = rng.uniform(-1e6, 1e6, n)
    kk = key.to_numpy()
    order = np.argsort(kk, kind="stable")
    # make every second group cancel almost exactly: pair rows inside the group as (x, -x)
    xs = x[order]
    starts = np.flatnonzero(np.r_[True, kk[order][1:] != kk[order][:-1]])
    for gi, (a, b) in enumerate(zip(starts, np.r_[starts[1:], n])):
        if gi % 2 == 0 and b - a >= 2:
            h = (b - a) // 2
            xs[a + h:a + 2 * h] = -xs[a:a + h]
    x[order] = xs
    v = Column("v", x)
    want, _ = oracle.group_by_agg([key], [v], [(0, "sum"), (0, "mean")])
    got = run_dense(key, [v], [(0, "sum"), (0, "mean")], ncat)
    assert pa.last_strategy() == "dense"
    g, w = util.frame_to_np(got), util.oracle_to_np(want)
    og, ow = util.canonical_order(g[:1]), util.canonical_order(w[:1])
    assert np.array_equal(g[0][0][og], w[0][0][ow])
    mag = np.zeros(ncat)
    np.add.at(mag, kk, np.abs(x))
    cnt = np.bincount(kk, minlength=ncat)
    keys_sorted = g[0][0][og]
    for j, scale in ((1, mag[keys_sorted]), (2, mag[keys_sorted] / np.maximum(cnt[keys_sorted], 1))):
        err = np.abs(g[j][0][og] - w[j][0][ow])
        assert np.all(err <= 1e-12 * scale + 1e-300), f"column {j}: max err/scale {np.max(err / scale):.3e}"


def test_codes_that_never_occur_and_single_group(oracle):
    rng = np.random.default_rng(3)
    n = 100_000
    key = Column("k", (rng.integers(0, 50, n) * 7).astype(np.uint32))  # only multiples of 7 below 350 occur
    v = Column("v", rng.integers(-5, 5, n).astype(np.int64))
    got = check(oracle, key, [v], [(0, "sum"), (0, "count")], 400, what="sparse codes")
    assert got.height == 50
    one = Column("k", np.zeros(n, dtype=np.uint32))
    check(oracle, one, [v], [(0, "sum"), (0, "mean")], 1, what="one category")
    allnull = Column("k", np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=bool))
    got = check(oracle, allnull, [v], [(0, "sum")], 3, what="all-null key")
    assert got.height == 1 and got["k"].to_list() == [None]


def test_broken_promise_costs_time_not_correctness(oracle):
    rng = np.random.default_rng(4)
    n = 200_000
    key = codes(rng, n, 1000)
    v = Column("v", rng.uniform(0, 1, n))
    # the caller promises 900 categories, codes go up to 999: the dense pass notices and the hashed path redoes the query
    check(oracle, key, [v], [(0, "sum"), (0, "count")], 900, expect=("direct", "tilesort", "lds", "global", "partitioned"), what="broken promise")
    # code == n_categories exactly (the null slot's index) is a broken promise too
    k2 = Column("k", np.r_[key.to_numpy()[:-1], np.uint32(1000)].astype(np.uint32))
    check(oracle, k2, [v], [(0, "sum")], 1000, expect=("direct", "tilesort", "lds", "global", "partitioned"), what="code == n_categories")


@pytest.mark.parametrize("n", [1, 5, 63, 64, 1000, 8191, 8192, 8193, 70_001])
def test_ragged_sizes_forced(oracle, n):
    rng = np.random.default_rng(n)
    key = codes(rng, n, 37, 0.1)
    v = Column("v", rng.integers(0, 100, n).astype(np.int32), rng.random(n) >= 0.2)
    check(oracle, key, [v], [(0, "sum"), (0, "mean"), (0, "min"), (0, "count")], 37, what=f"dense n={n}")


@pytest.mark.parametrize("ncat", [2_000, 19_999, 20_000, 20_001, 40_000, 64_000])
def test_range_boundaries(oracle, ncat):
    # 20 000 groups x 8 bytes = one CU's LDS: the category counts around the range cuts, up to eight ranges
    rng = np.random.default_rng(ncat)
    n = 1_500_000
    key = codes(rng, n, ncat, 0.01)
    v1 = Column("v1", rng.integers(-3, 9, n).astype(np.int32))
    v3 = Column("v3", np.round(rng.uniform(0, 100, n), 6))  # (one sign: a relative bar on near-cancelling sums is test_mixed_sign_f64_cancellation's subject)
    check(oracle, key, [v1, v3], [(0, "sum"), (1, "mean"), (1, "max")], ncat, what=f"dense ncat={ncat}")


def test_too_many_categories_declines(oracle):
    # beyond eight code ranges: under AUTO the hashed pipelines take a query of this size (the partition mode starts at 2^22 rows and
    # 1.2e6 codes); the forced strategy takes the partition mode; a nullable key has no partition mode and declines even when forced
    rng = np.random.default_rng(9)
    n = 400_000
    key = codes(rng, n, 300_000)
    v = Column("v", rng.uniform(0, 1, n))
    hashed = ("tilesort", "tilesort2", "sort", "partitioned", "global")
    check(oracle, key, [v], [(0, "sum"), (0, "mean")], 300_000, expect=hashed, what="beyond eight ranges, auto", strategy="auto")
    check(oracle, key, [v], [(0, "sum"), (0, "mean")], 300_000, expect="dense_part", what="beyond eight ranges, forced")
    check(oracle, codes(rng, n, 300_000, 0.02), [v], [(0, "sum"), (0, "mean")], 300_000, expect=hashed, what="beyond eight ranges, nullable key")


def test_arrow_offsets_and_partial_flag(oracle):
    rng = np.random.default_rng(12)
    n = 300_000
    kbuf = rng.integers(0, 500, n + 8).astype(np.uint32)
    vbuf = rng.integers(-9, 9, n + 8).astype(np.int32)
    import torch

    pa.set_strategy("dense")  # (500 categories in one range: AUTO would leave this query to the hashed DIRECT strategy)
    for device in (False, True):
        for off in (0, 4, 3):  # device-resident columns: offset 4 keeps the 16-byte quads aligned, offset 3 does not -> declined
            kb = torch.from_numpy(kbuf.view(np.int32)).to("cuda:0") if device else kbuf
            vb = torch.from_numpy(vbuf).to("cuda:0") if device else vbuf
            key = Column("k", kb, offset=off, length=n, dtype=_ffi.U32).set_categories(500)
            v = Column("v", vb, offset=off, length=n)
            want, _ = oracle.group_by_agg([Column("k", kbuf[off:off + n])], [Column("v", vbuf[off:off + n])], [(0, "sum"), (0, "count")])
            got = pa.DataFrame([key, v]).group_by("k").agg(pa.col("v").sum().alias("s"), pa.count())
            util.assert_frames_match(util.frame_to_np(got), util.oracle_to_np(want), 1, what=f"offset {off} device={device}")
            assert (pa.last_strategy() == "dense") == (not (device and off == 3)), (off, device, pa.last_strategy())
    # mergeable partials (PLHIP_FLAG_PARTIAL): MEAN becomes (sum, valid count), sums stay 64-bit
    key = Column("k", kbuf[:n]).set_categories(500)
    v = Column("v", vbuf[:n], rng.random(n) >= 0.3)
    aggs = [(0, _ffi.AGG_SUM), (0, _ffi.AGG_MEAN), (0, _ffi.AGG_COUNT)]
    frames = []
    for flag in (_ffi.FLAG_KEYS_DENSE, 0):
        handle, cols = pa.group_by_agg_raw([key], [v], aggs, _ffi.FLAG_PARTIAL | flag, 500)
        try:
            assert pa.last_strategy() == ("dense" if flag else "direct")
            fr = [pa.frame._host_column(f"c{i}", cs) for i, cs in enumerate(cols)]
            frames.append([(c.to_numpy().copy(), None if c.valid_mask() is None else c.valid_mask().copy()) for c in fr])
        finally:
            _ffi.lib().plhip_result_free(handle)
    util.assert_frames_match(frames[0], frames[1], 1, what="partial frames: dense vs direct")


# ---- partition mode: more groups than DENSE_MAX_RANGES code ranges hold (the counterpart stays group_tuples_perfect, perfect.rs:24-185:
# codes index the tables; here the rows are first scattered by code range, api.hip run_dense) ----------------------------------------
PART_N = 4_400_000  # (under AUTO the mode starts at 2^22 rows and 1.2e6 codes; forced strategy "dense" takes it at any size)


def _part_case(rng, ncat, skew=None, never=()):
    k = rng.integers(0, ncat, PART_N).astype(np.uint32)
    if skew is not None:  # one code range holds most rows
        hot = rng.random(PART_N) < 0.6
        k[hot] = rng.integers(skew, skew + 50, int(hot.sum())).astype(np.uint32)
    for c in never:
        k[k == c] = (c + 1) % ncat
    return Column("k", k)


@pytest.mark.parametrize("ncat", [250_000, 1_300_000])
def test_partition_mode_q3_shape(oracle, ncat):
    rng = np.random.default_rng(ncat)
    key = _part_case(rng, ncat, never=(0, 17, ncat - 1))
    v1 = Column("v1", rng.integers(1, 6, PART_N).astype(np.int32))
    v3 = Column("v3", np.round(rng.uniform(0, 100, PART_N), 6))
    check(oracle, key, [v1, v3], [(0, "sum"), (1, "mean")], ncat, expect="dense_part", what=f"partition mode q3, {ncat} codes",
          strategy="auto" if ncat > 1_200_000 else "dense")


def test_partition_mode_all_aggregations_nullable_values_and_skew(oracle):
    rng = np.random.default_rng(99)
    ncat = 400_000
    key = _part_case(rng, ncat, skew=123_456)
    v1 = Column("v1", rng.integers(-1000, 1000, PART_N).astype(np.int32), rng.random(PART_N) >= 0.07)
    v3 = Column("v3", np.round(rng.uniform(0, 100, PART_N), 6), rng.random(PART_N) >= 0.1)  # (one sign: the f64 bound is relative to sum |x|)
    v8 = Column("v8", rng.integers(-2**40, 2**40, PART_N).astype(np.int64))
    aggs = [(0, a) for a in ALL_AGGS] + [(1, a) for a in ALL_AGGS] + [(2, "sum"), (2, "min"), (2, "max")]
    check(oracle, key, [v1, v3, v8], aggs, ncat, expect="dense_part", what="partition mode, every aggregation", strategy="dense")


def test_partition_mode_plain_i32_key_with_a_compact_range(oracle):
    # db-benchmark's id6 at scale: no flag from the caller, AUTO measures the key's range and takes the partition mode
    rng = np.random.default_rng(5)
    lo, span = -70_000, 1_500_000
    k = Column("k", rng.integers(lo, lo + span, PART_N).astype(np.int32))
    v = Column("v", rng.integers(0, 100, PART_N).astype(np.int32))
    want, _ = oracle.group_by_agg([k], [v], [(0, "sum"), (0, "count")])
    pa.set_strategy("auto")
    df = pa.DataFrame({"k": k, "v": v})
    got = df.group_by("k").agg(pa.col("v").sum().alias("s"), pa.count().alias("c"))
    assert pa.last_strategy() == "dense_part", pa.last_strategy()
    util.assert_frames_match(util.frame_to_np(got), util.oracle_to_np(want), 1, what="partition mode, plain i32 key")


def test_partition_mode_code_beyond_the_hint_falls_back(oracle):
    rng = np.random.default_rng(6)
    ncat = 1_250_000
    key = _part_case(rng, ncat)
    key.values[12345] = ncat + 7  # the caller's promise is broken: the hashed path redoes the query
    v = Column("v", rng.integers(0, 9, PART_N).astype(np.int32))
    check(oracle, key, [v], [(0, "sum"), (0, "count")], ncat, expect=("tilesort", "tilesort2", "sort"), what="partition mode, bad code", strategy="auto")


def test_config4_single_gpu_point_full_size_properties():
    """BASELINE configs[3]'s N = 1 point at its stated size -- q3 over 1e9 rows / 1e7 ids on ONE GPU, the regime config 4's
    shards run in -- takes the partition mode (range scatter + one-range dense fold).  No oracle finishes this in seconds:
    size-independent properties (counts add up to N and match torch's per-id counts exactly, the exact integer checksum, the
    f64 checksum, mean * count == sum per group, every id at most once)."""
    import torch

    from polars_amd import h2oai
    from polars_amd.frame import _host_column

    n = 1_000_000_000
    dev = torch.device("cuda:0")
    t = h2oai.generate(n, seed=108, device=dev, columns=["id3", "v1", "v3"])
    cols = h2oai.to_columns(t)
    keys, vals = [cols["id3"]], [cols["v1"], cols["v3"]]
    ncat = keys[0].n_categories
    assert ncat == n // 100
    codes = [(0, _ffi.AGG_NAMES["sum"]), (1, _ffi.AGG_NAMES["mean"]), (1, _ffi.AGG_NAMES["sum"]), (0, _ffi.AGG_NAMES["count"])]
    handle, rcols = pa.group_by_agg_raw(keys, vals, codes, _ffi.FLAG_KEYS_DENSE, ncat)
    try:
        assert pa.last_strategy() == "dense_part", pa.last_strategy()
        got = [_host_column(str(i), cs).to_numpy().copy() for i, cs in enumerate(rcols)]
    finally:
        _ffi.lib().plhip_result_free(handle)
    k, s1, mean3, s3, cnt = got
    assert len(np.unique(k)) == len(k)
    ids = t["id3"].view(torch.int32) if t["id3"].dtype != torch.int32 else t["id3"]
    per_id = torch.bincount(ids, minlength=ncat)
    present = (per_id > 0).sum().item()
    assert len(k) == present, (len(k), present)
    order = np.argsort(k)
    assert np.array_equal(cnt[order].astype(np.int64), per_id[per_id > 0].cpu().numpy())  # every group's length, exactly
    assert int(cnt.astype(np.int64).sum()) == n
    assert int(s1.astype(np.int64).sum()) == int(t["v1"].sum().item())  # (about 100 rows of 1..5 per id: no group's i32 sum wraps)
    tot3 = float(t["v3"].sum().item())
    assert abs(float(s3.sum()) - tot3) <= 1e-9 * abs(tot3)
    assert np.allclose(mean3 * cnt, s3, rtol=1e-12)
    top = int(np.argmax(cnt))
    sel = ids == int(k[top])
    assert int(s1[top]) == int(t["v1"][sel].sum().item())
    ref = float(t["v3"][sel].sum().item())
    assert abs(float(s3[top]) - ref) <= 1e-12 * abs(ref)
