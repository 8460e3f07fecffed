"""Boolean columns at the ABI (PLHIP_BOOL: the values buffer is a bitmap).  The reference's casts: a Boolean key groups as
UInt8 and stays Boolean (into_groups.rs:215-230); sum -> u32 (aggregations/boolean.rs:107-109: cast to IdxSize), mean -> f64
(dispatch.rs:164), min / max -> Boolean (boolean.rs:20-106).  Checked against the oracle run on the cast columns, and against the
reference's own `bool_sum: UInt32 == 1` (py-polars/tests/unit/streaming/test_streaming_group_by.py:35-105)."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from tests import util

pytestmark = pytest.mark.gpu


def test_reference_bool_sum_is_u32():
    df = pa.DataFrame({"person_id": np.array([1, 1], np.int64), "bool": np.array([True, False])})
    out = df.group_by("person_id").agg(pa.col("bool").sum().alias("bool_sum"), pa.col("bool").max().alias("bool_max"), pa.col("bool").mean().alias("m"))
    assert out["bool_sum"].dtype == _ffi.U32 and out["bool_sum"].to_list() == [1]
    assert out["bool_max"].dtype == _ffi.BOOL and out["bool_max"].to_list() == [True]
    assert out["m"].dtype == _ffi.F64 and out["m"].to_list() == [0.5]


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 300_000])
@pytest.mark.parametrize("offset", [0, 3])
def test_bool_key_and_values_vs_oracle(oracle, n, offset):
    rng = np.random.default_rng(n + offset)
    tot = n + offset + 5
    kb, kbm = rng.random(tot) < 0.4, rng.random(tot) >= 0.1
    k2 = rng.integers(0, 5, tot).astype(np.int64)
    vb, vbm = rng.random(tot) < 0.7, rng.random(tot) >= 0.2
    w = rng.integers(-50, 50, tot).astype(np.int32)
    sl = slice(offset, offset + n)
    keys = [Column("kb", kb, kbm, offset=offset, length=n), Column("k2", k2, offset=offset, length=n)]
    vals = [Column("vb", vb, vbm, offset=offset, length=n), Column("w", w, offset=offset, length=n)]
    aggs = [(0, "sum"), (0, "mean"), (0, "min"), (0, "max"), (0, "count"), (1, "sum")]
    df = pa.DataFrame({"kb": keys[0], "k2": keys[1], "vb": vals[0], "w": vals[1]})
    out = df.group_by("kb", "k2").agg(pa.col("vb").sum().alias("s"), pa.col("vb").mean().alias("m"), pa.col("vb").min().alias("lo"),
                                      pa.col("vb").max().alias("hi"), pa.col("vb").count().alias("c"), pa.col("w").sum().alias("ws"))
    assert [out[c].dtype for c in out.columns] == [_ffi.BOOL, _ffi.I64, _ffi.U32, _ffi.F64, _ffi.BOOL, _ffi.BOOL, _ffi.U32, _ffi.I32]
    ok = [Column("kb", kb[sl].astype(np.uint8), kbm[sl]), Column("k2", k2[sl])]
    ov = [Column("vb", vb[sl].astype(np.uint32), vbm[sl]), Column("w", w[sl])]
    want, _ = oracle.group_by_agg(ok, ov, aggs)
    got = []
    for name in out.columns:
        c = out[name]
        v = c.to_numpy()
        got.append((v.astype(np.uint8 if name == "kb" else np.uint32) if c.dtype == _ffi.BOOL else v, c.valid_mask()))
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what=f"bool key + values, n={n}, offset={offset}")


def test_bool_device_column_and_single_key(oracle):
    import torch

    rng = np.random.default_rng(8)
    n = 2_000_000
    kb = rng.random(n) < 0.5
    v = np.round(rng.uniform(0, 10, n), 6)
    bits = np.concatenate([np.packbits(kb, bitorder="little"), np.zeros(8, np.uint8)])
    dev = torch.device("cuda:0")
    kcol = Column("kb", torch.from_numpy(bits).to(dev), dtype=_ffi.BOOL, length=n)
    from polars_amd.frame import _host_column, group_by_agg_raw

    handle, cols = group_by_agg_raw([kcol], [Column("v", torch.from_numpy(v).to(dev))], [(0, _ffi.AGG_SUM), (-1, _ffi.AGG_COUNT)])
    try:
        got = [_host_column(str(i), cs) for i, cs in enumerate(cols)]
    finally:
        _ffi.lib().plhip_result_free(handle)
    want, _ = oracle.group_by_agg([Column("kb", kb.astype(np.uint8))], [Column("v", v)], [(0, "sum"), (0, "count")])
    g = [(got[0].to_numpy().astype(np.uint8), got[0].valid_mask()), (got[1].to_numpy(), got[1].valid_mask()), (got[2].to_numpy(), got[2].valid_mask())]
    util.assert_frames_match(g, util.oracle_to_np(want), 1, what="device-resident bool key")


def _canon(lists):
    return sorted(tuple(l) for l in lists)


@pytest.mark.parametrize("offset", [0, 5])
def test_bool_in_the_csr_seams(oracle, offset):
    """Boolean keys in plhip_group_tuples (grouped as their UInt8 cast, into_groups.rs:215-230) and Boolean value columns in
    plhip_agg_groups: sum over the IdxSize cast (boolean.rs:107-109), mean over the Float64 cast (dispatch.rs:164), min / max
    (boolean.rs:20-106) and first / last stay Boolean.  Against the oracle's CSR restatement on the cast columns."""
    rng = np.random.default_rng(17 + offset)
    n = 40_000
    tot = n + offset + 3
    kb, kbm = rng.random(tot) < 0.45, rng.random(tot) >= 0.08
    k2 = rng.integers(0, 300, tot).astype(np.int32)
    vb, vbm = rng.random(tot) < 0.6, rng.random(tot) >= 0.25
    sl = slice(offset, offset + n)
    keys_b = [Column("kb", kb, kbm, offset=offset, length=n), Column("k2", k2, offset=offset, length=n)]
    keys_u = [Column("kb", kb[sl].astype(np.uint8), kbm[sl]), Column("k2", k2[sl])]
    for sorted_ in (True, False):
        want, got = oracle.Groups(keys_u, sorted_), pa.group_tuples(keys_b, sorted_)
        assert got.n_groups == want.n_groups and got.n_rows == n
        first, off, idx = got.buffers()
        lists = [idx[off[g]:off[g + 1]].tolist() for g in range(got.n_groups)]
        assert all(l == sorted(l) and first[g] == l[0] for g, l in enumerate(lists))
        if sorted_:
            assert lists == want.lists()
        else:
            assert _canon(lists) == _canon(want.lists())
    g_o, g_p = oracle.Groups(keys_u, True), pa.group_tuples(keys_b, True)
    v_b = Column("vb", vb, vbm, offset=offset, length=n)
    v_u32, v_u8 = Column("vb", vb[sl].astype(np.uint32), vbm[sl]), Column("vb", vb[sl].astype(np.uint8), vbm[sl])
    for kind in ("sum", "mean", "min", "max", "first", "last"):
        src = v_u32 if kind in ("sum", "mean") else v_u8
        wv, wm = oracle.agg(src, kind, g_o) if kind in ("sum", "mean", "min", "max") else oracle.agg_ext(src, kind, g_o, 1)
        col = pa.agg_groups(v_b, kind, g_p)
        gv, gm = col.to_numpy(), col.valid_mask()
        gm = np.ones(len(gv), bool) if gm is None else gm
        assert np.array_equal(gm, wm), f"{kind}: validity"
        if kind in ("sum", "mean"):
            assert col.dtype == (_ffi.U32 if kind == "sum" else _ffi.F64) and gv.dtype == wv.dtype
            assert np.array_equal(gv[gm], wv[wm]), kind  # 0 / 1 terms: the sums are exact integers, the means their exact quotients
        else:
            assert col.dtype == _ffi.BOOL, kind
            assert np.array_equal(gv[gm].astype(np.uint8), wv[wm]), kind
    # a single Boolean key through the mirror: the reference's three possible groups (true, false, null)
    out = pa.DataFrame({"k": Column("k", kb[sl], kbm[sl])}).group_by("k").groups()
    assert out.n_groups == 3
    for kind in ("var", "median", "n_unique"):
        with pytest.raises(_ffi.PlHipError):
            pa.agg_groups(v_b, kind, g_p)


def test_reference_bool_min_max_known_answers():
    """py-polars/tests/unit/datatypes/test_bool.py:33-43 (test_bool_min_max), the column as ONE group over the CSR seam: nulls are
    skipped, min is the AND and max the OR of the valid values; and :29-30 (test_bool_sum_empty): an empty Boolean column sums to 0."""
    cases = [([None, True], True, True), ([None, True, False], False, True), ([False, True], False, True),
             ([True, True], True, True), ([False, False], False, False)]
    for data, want_min, want_max in cases:
        vals = np.array([bool(x) for x in data])
        mask = np.array([x is not None for x in data])
        col = Column("b", vals, None if mask.all() else mask)
        g = pa.group_tuples([Column("g", np.zeros(len(data), dtype=np.int32))], True)
        lo, hi = pa.agg_groups(col, "min", g), pa.agg_groups(col, "max", g)
        assert lo.dtype == _ffi.BOOL and hi.dtype == _ffi.BOOL
        assert lo.to_list() == [want_min] and hi.to_list() == [want_max], (data, lo.to_list(), hi.to_list())
        s = pa.agg_groups(col, "sum", g)
        assert s.dtype == _ffi.U32 and s.to_list() == [sum(1 for x in data if x)]
    # one group whose every value is null: min / max are null, the sum is 0 (aggregations/boolean.rs:20-106; agg_sum never null)
    col = Column("b", np.array([True, False]), np.array([False, False]))
    g = pa.group_tuples([Column("g", np.zeros(2, dtype=np.int32))], True)
    assert pa.agg_groups(col, "min", g).to_list() == [None] and pa.agg_groups(col, "max", g).to_list() == [None]
    assert pa.agg_groups(col, "sum", g).to_list() == [0]
    out = pa.DataFrame({"k": np.zeros(0, np.int64), "b": np.zeros(0, bool)}).group_by("k").agg(pa.col("b").sum().alias("s"))
    assert out["s"].dtype == _ffi.U32 and out["s"].to_list() == []
