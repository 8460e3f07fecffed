"""Pins the CPU oracle against the reference's own known-answer tests (tests/golden/reference_known_answers.json)."""
import numpy as np
import pytest

from polars_amd import _ffi
from tests import util

KA = util.load_known_answers()


@pytest.mark.parametrize("case", KA["hash"], ids=lambda c: c["name"])
def test_hash_known_answers(oracle, case):
    keys = [util.make_column(k) for k in case["keys"]]
    got = oracle.vec_hash(keys, case["null_hash"])
    assert got.tolist() == case["expected"]


def test_folded_multiply_matches_python_bigint(oracle):
    M = 6364136223846793005
    rng = np.random.default_rng(1)
    for v in [0, 1, 1234, 5678, 2**63, 2**64 - 1] + rng.integers(0, 2**63, 50).tolist():
        r = v * M
        assert oracle.folded_multiply(v, M) == ((r & (2**64 - 1)) ^ (r >> 64))


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("partitioned", [0, 4])
@pytest.mark.parametrize("case", KA["group_by"], ids=lambda c: c["name"])
def test_group_by_known_answers(oracle, case, threads, partitioned):
    keys = [util.make_column(k) for k in case["keys"]]
    values = [util.make_column(v) for v in case["values"]]
    aggs = [(ci, kind) for ci, kind in case["aggs"]]
    out, _ = oracle.group_by_agg(keys, values, aggs, maintain_order=case["maintain_order"], n_threads=threads,
                                 partitioned=partitioned)
    cols = [util.to_py(v, m) for _, v, m in out]
    dts = [_ffi.DTYPE_NAMES[_ffi.np_dtype_code(v.dtype)] for _, v, _ in out]
    util.check_known_answer(case, cols, dts)


# crates/polars/tests/it/core/group_by.rs:6-96 (test_sorted_group_by) and
# crates/polars-arrow/src/kernels/sort_partition.rs:143-160 (test_partition_points): [first, len] slices
SORTED_KEY_CASES = [
    ([1, 1, 1, 6, 6, None], [[0, 3], [3, 2], [5, 1]]),
    ([None, None, 1, 1, 1, 6, 6], [[0, 2], [2, 3], [5, 2]]),
    ([None, None, 3, 3, 1, 1, 1, -1], [[0, 2], [2, 2], [4, 3], [7, 1]]),
    ([15, 15, 15, 15, 14, 13, 11, 11, 3, 3, 1, 1, 1, -1, None, None, None],
     [[0, 4], [4, 1], [5, 1], [6, 2], [8, 2], [10, 3], [13, 1], [14, 3]]),
    ([1, 3, 3, 3, 3, 5, 5, 5, 9, 9, 10], [[0, 1], [1, 4], [5, 3], [8, 2], [10, 1]]),
    ([None] * 5 + [1, 3, 3, 3, 3, 5, 5, 5, 9, 9, 10], [[0, 5], [5, 1], [6, 4], [10, 3], [13, 2], [15, 1]]),
    ([1, 3, 3, 3, 3, 5, 5, 5, 9, 9, 10] + [None] * 5, [[0, 1], [1, 4], [5, 3], [8, 2], [10, 1], [11, 5]]),
]


def sorted_case_column(vals):
    from polars_amd.column import Column

    mask = np.array([v is not None for v in vals])
    data = np.array([0 if v is None else v for v in vals], dtype=np.int32)
    return Column("a", data, None if mask.all() else mask)


@pytest.mark.parametrize("vals,want", SORTED_KEY_CASES)
def test_sorted_key_groups_known_answers(oracle, vals, want):
    assert oracle.sorted_key_groups(sorted_case_column(vals)) == want


def test_sorted_key_groups_compare_values_not_bits(oracle):
    from polars_amd.column import Column

    nan = float("nan")
    got = oracle.sorted_key_groups(Column("f", np.array([-0.0, 0.0, 1.5, 1.5, nan, nan], dtype=np.float64)))
    assert got == [[0, 2], [2, 2], [4, 1], [5, 1]]


def test_agg_ext_known_answers(oracle):
    """test_group_by_var (crates/polars-core/src/frame/group_by/mod.rs:1134-1156): int [1, 2 | 3] by [foo, foo | bar]
    -> var(1) = 0.5 for foo, std(1) = 1/sqrt(2); the one-row group has count <= ddof -> null."""
    from polars_amd.column import Column

    g = oracle.Groups([Column("g", np.array([7, 7, 9], dtype=np.int32))], True)
    v, m = oracle.agg_ext(Column("int", np.array([1, 2, 3], dtype=np.int32)), "var", g, 1)
    assert v.dtype == np.float64 and m.tolist() == [True, False] and v[0] == 0.5
    v, m = oracle.agg_ext(Column("int", np.array([1, 2, 3], dtype=np.int32)), "std", g, 1)
    assert abs(v[0] - 2 ** -0.5) < 1e-15 and m.tolist() == [True, False]
    v, m = oracle.agg_ext(Column("x", np.array([5, 6, 8], dtype=np.int64), np.array([False, True, True])), "first", g)
    assert m.tolist() == [False, True] and v[1] == 8
    v, m = oracle.agg_ext(Column("x", np.array([5, 6, 8], dtype=np.int64), np.array([False, True, True])), "last", g)
    assert m.tolist() == [True, True] and v.tolist() == [6, 8]
    # test_group_by_agg_n_unique_floats (py-polars/tests/unit/dataframe/test_df.py:1726-1734): [1.0, 2.0 | 2.0] -> [2, 1]
    for dt in (np.float32, np.float64):
        v, m = oracle.agg_ext(Column("b", np.array([1.0, 2.0, 2.0], dtype=dt)), "n_unique", g)
        assert v.dtype == np.uint32 and v.tolist() == [2, 1] and m.all()
    # test_group_by_shorthands (test_group_by.py:59-93): a = [1..5], c = [None, 1, None, 1, None] by b = [a, a, b, b, b]
    g5 = oracle.Groups([Column("b", np.array([10, 10, 20, 20, 20], dtype=np.int32))], True)
    v, _ = oracle.agg_ext(Column("a", np.array([1, 2, 3, 4, 5], dtype=np.int64)), "n_unique", g5)
    assert v.tolist() == [2, 3]
    v, _ = oracle.agg_ext(Column("c", np.array([0, 1, 0, 1, 0], dtype=np.int64), np.array([False, True, False, True, False])), "n_unique", g5)
    assert v.tolist() == [2, 2]


def test_agg_quantile_known_answers(oracle):
    """test_quantile, test_quantile_all_null, test_quantile_single_value, test_quantile_min_max (crates/polars-core/src/
    chunked_array/ops/aggregate/mod.rs:696-944) with the column as one group, test_median (:639-660) and
    test_group_by_shorthand_quantile (py-polars/tests/unit/operations/test_group_by.py:96-102)."""
    from polars_amd.column import Column
    from tests.util import QUANTILE_KNOWN, quantile_known_column

    for data, table in QUANTILE_KNOWN:
        col = quantile_known_column(data)
        g = oracle.Groups([Column("g", np.zeros(len(data), dtype=np.int32))], True)
        for interp, qs in table.items():
            for q, want in qs.items():
                v, m = oracle.agg_quantile(col, q, interp, g)
                assert m.tolist() == [True] and v.dtype == np.float64 and v[0] == want, (data, interp, q, v[0])
        v, m = oracle.agg_ext(col, "median", g)
        assert v[0] == (3.0 if len(data) == 7 else 4.0)
    g3 = oracle.Groups([Column("g", np.zeros(3, dtype=np.int32))], True)
    g4 = oracle.Groups([Column("g", np.zeros(4, dtype=np.int32))], True)
    g1 = oracle.Groups([Column("g", np.zeros(1, dtype=np.int32))], True)
    for dt in (np.float32, np.int32, np.float64, np.int64):
        for interp in oracle.INTERPOLATIONS:
            v, m = oracle.agg_quantile(Column("x", np.zeros(3, dtype=dt), np.zeros(3, bool)), 0.9, interp, g3)
            assert m.tolist() == [False]
            v, m = oracle.agg_quantile(Column("x", np.ones(1, dtype=dt)), 0.5, interp, g1)
            assert m.tolist() == [True] and v[0] == 1.0
            c = Column("x", np.array([0, 1, 5, 1], dtype=dt), np.array([False, True, True, True]))
            assert oracle.agg_quantile(c, 0.0, interp, g4)[0][0] == 1.0 and oracle.agg_quantile(c, 1.0, interp, g4)[0][0] == 5.0
    g5 = oracle.Groups([Column("b", np.array([10, 10, 20, 20, 20], dtype=np.int32))], True)
    v, m = oracle.agg_quantile(Column("a", np.array([1, 2, 3, 4, 5], dtype=np.int64)), 0.5, "nearest", g5)
    assert v.tolist() == [2.0, 4.0]
    v, m = oracle.agg_quantile(Column("c", np.array([0, 1, 0, 1, 0], dtype=np.int64), np.array([False, True, False, True, False])), 0.5, "nearest", g5)
    assert v.tolist() == [1.0, 1.0] and m.all()
    v, m = oracle.agg_quantile(Column("a", np.array([1, 2, 3, 4, 5], dtype=np.int64)), 1.5, "nearest", g5)
    assert v.dtype == np.int64 and not m.any()  # aggregations/mod.rs:346-349


def test_boolean_vec_hash_combine_formula(oracle):
    """VecHash for BooleanChunked (vector_hasher.rs:241-302) + _boost_hash_combine (hashing/mod.rs:20-22), restated in plain Python:
    the first column hashes to true_h / false_h / null_h, a later Boolean column combines as l ^ (h + 0x9e3779b9 + (l << 6) + (h >> 2));
    integer columns keep folded_multiply.  The three constants are seed dependent in the reference (random_state.hash_one), hence
    arguments here: what is pinned is the combine arithmetic."""
    import numpy as np

    from polars_amd.column import Column

    M64 = (1 << 64) - 1
    MULT = 6364136223846793005
    fm = lambda s, by: ((s * by) & M64) ^ ((s * by) >> 64)
    boost = lambda l, r: l ^ ((r + ((0x9E3779B9 + ((l << 6) & M64) + (r >> 2)) & M64)) & M64)
    null_h, true_h, false_h = 0x0123456789ABCDEF, 0xF00DFACE12345678, 0x0BADC0DE87654321
    rng = np.random.default_rng(1)
    n = 301
    b1, b2 = rng.random(n) < 0.5, rng.random(n) < 0.3
    m2 = rng.random(n) >= 0.2
    k = rng.integers(-5, 5, n).astype(np.int32)
    cols = [Column("b1", b1), Column("k", k), Column("b2", b2, m2)]
    got = oracle.vec_hash_seeded(cols, null_h, true_h, false_h)
    for i in range(n):
        h = true_h if b1[i] else false_h
        h = fm((int(k[i]) & 0xFFFFFFFF) ^ h, MULT)
        l = null_h if not m2[i] else (true_h if b2[i] else false_h)
        h = boost(l, h)
        assert int(got[i]) == h, (i, hex(int(got[i])), hex(h))
    # without Boolean columns it is plhip_vec_hash / orc_vec_hash
    assert np.array_equal(oracle.vec_hash_seeded([Column("k", k)], null_h, 1, 2), oracle.vec_hash([Column("k", k)], null_h))
