"""Hash join on one integer key (plhip_hash_join) against the oracle: the same pairs in the same order (the
reference's order is deterministic: probe rows in order, matches in ascending build-row order), every integer dtype, nulls
that join nulls, duplicate-heavy and unique keys, empty sides, device-resident keys; and the reference's own known answers."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from polars_amd.join import join_indices
from tests.test_oracle_join import NP, check_case, gold_cases

pytestmark = pytest.mark.gpu


def test_reference_known_answers_on_the_gpu():
    for case in gold_cases():
        for dt in case["dtypes"]:
            check_case(case, dt, join_indices)


@pytest.mark.parametrize("how", ["inner", "left", "semi", "anti"])
@pytest.mark.parametrize("dtype", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64])
def test_join_matches_the_oracle_pair_for_pair(oracle, how, dtype):
    rng = np.random.default_rng(np.dtype(dtype).itemsize * 7 + (how == "left"))
    info = np.iinfo(dtype)
    for nl, nr, span, nulls in ((0, 9, 5, 0.0), (9, 0, 5, 0.0), (1000, 1000, 50, 0.1), (50_000, 20_000, 30_000, 0.02), (20_000, 50_000, 100, 0.0)):
        lo = max(int(info.min), -span)
        hi = min(int(info.max), span)
        lk = rng.integers(lo, hi + 1, nl).astype(dtype)
        rk = rng.integers(lo, hi + 1, nr).astype(dtype)
        if nl > 10:
            lk[:3] = [info.min, info.max, 0]  # the extremes are ordinary keys (u64 max is not a sentinel)
            rk[-3:] = [info.max, info.min, 0]
        lm = rng.random(nl) >= nulls if nulls else None
        rm = rng.random(nr) >= nulls if nulls else None
        left, right = Column("k", lk, lm), Column("k", rk, rm)
        gl, gr = join_indices(left, right, how)
        wl, wr = oracle.hash_join(left, right, how)
        assert len(gl) == len(wl), (how, dtype, nl, nr)
        assert np.array_equal(gl, wl) and (how in ("semi", "anti") or np.array_equal(gr, wr)), (how, dtype, nl, nr)


def test_join_large_and_device_resident(oracle):
    import torch

    rng = np.random.default_rng(9)
    nl, nr = 3_000_000, 1_000_000
    lk = rng.integers(0, 1_500_000, nl).astype(np.int64)
    rk = rng.permutation(2_000_000)[:nr].astype(np.int64)  # unique build keys: at most one match per probe row
    dev = torch.device("cuda:0")
    left, right = Column("k", torch.from_numpy(lk).to(dev)), Column("k", torch.from_numpy(rk).to(dev))
    for how in ("inner", "left", "semi", "anti"):
        gl, gr = join_indices(left, right, how)
        wl, wr = oracle.hash_join(Column("k", lk), Column("k", rk), how)
        assert np.array_equal(gl, wl) and (how in ("semi", "anti") or np.array_equal(gr, wr)), how
    assert _ffi.lib().plhip_last_strategy() == b"hash_join"


def test_join_errors():
    with pytest.raises(_ffi.PlHipError):
        join_indices(Column("k", np.zeros(3, np.int32)), Column("k", np.zeros(3, np.int64)))
    with pytest.raises(_ffi.PlHipError):
        join_indices(Column("k", np.zeros(3, np.float64)), Column("k", np.zeros(3, np.float64)))


def test_frame_join():
    """DataFrame.join end to end (payload gather on the host), the reference's docstring frame with integer keys."""
    df = pa.DataFrame({"foo": np.array([1, 2, 3], np.int64), "bar": np.array([6.0, 7.0, 8.0]), "ham": np.array([10, 20, 30], np.int32)})
    other = pa.DataFrame({"apple": ["x", "y", "z"], "ham": np.array([10, 20, 40], np.int32)})
    out = df.join(other, on="ham")
    assert out.to_dict() == {"foo": [1, 2], "bar": [6.0, 7.0], "ham": [10, 20], "apple": ["x", "y"]}
    out = df.join(other, on="ham", how="left")
    assert out.to_dict() == {"foo": [1, 2, 3], "bar": [6.0, 7.0, 8.0], "ham": [10, 20, 30], "apple": ["x", "y", None]}
    assert df.join(other, on="ham", how="semi").to_dict() == {"foo": [1, 2], "bar": [6.0, 7.0], "ham": [10, 20]}
    assert df.join(other, on="ham", how="anti").to_dict() == {"foo": [3], "bar": [8.0], "ham": [30]}


@pytest.mark.parametrize("how", ["inner", "left", "semi", "anti"])
def test_multi_key_join_matches_the_oracle(oracle, how):
    """hash_join/multiple_keys.rs: (i64, f64 by bit pattern, nullable i32) key tuples, pair for pair against the oracle."""
    rng = np.random.default_rng(21 + len(how))
    for nl, nr in ((0, 7), (7, 0), (500, 500), (40_000, 15_000), (15_000, 40_000)):
        def side(n):
            a = rng.integers(0, 30, n).astype(np.int64)
            b = rng.integers(0, 4, n).astype(np.float64) * 0.5
            if n > 4:
                b[:2] = [-0.0, 0.0]  # different bit patterns: different keys
            c = rng.integers(0, 3, n).astype(np.int32)
            cm = rng.random(n) >= 0.1
            return [Column("a", a), Column("b", b), Column("c", c, cm)]
        left, right = side(nl), side(nr)
        gl, gr = join_indices(left, right, how)
        wl, wr = oracle.hash_join_multi(left, right, how)
        assert len(gl) == len(wl), (how, nl, nr)
        assert np.array_equal(gl, wl) and (how in ("semi", "anti") or np.array_equal(gr, wr)), (how, nl, nr)


def test_multi_key_frame_join_reference_shape():
    """crates/polars/tests/it/core/joins.rs:167-185 (get_dfs) joined on two keys, a = foo and b = bar: rows (1, "a") x 1 and
    (1, "c") x 2 x 2 -> 5 rows; strings go in as integer ids here."""
    ids = {"a": 0, "b": 1, "c": 2}
    df_a = pa.DataFrame({"a": np.array([1, 2, 1, 1], np.int32), "b": np.array([ids[x] for x in ["a", "b", "c", "c"]], np.int32), "c": np.array([0, 1, 2, 3], np.int64)})
    df_b = pa.DataFrame({"foo": np.array([1, 1, 1], np.int32), "bar": np.array([ids[x] for x in ["a", "c", "c"]], np.int32), "ham": np.array([10, 11, 12], np.int64)})
    out = df_a.join(df_b, left_on=["a", "b"], right_on=["foo", "bar"])
    assert out.height == 5 and sorted(out["ham"].to_list()) == [10, 11, 11, 12, 12]


@pytest.mark.parametrize("multi", [False, True], ids=["one_key", "key_tuples"])
def test_outer_join(oracle, multi):
    """hash_join_tuples_outer (single_keys_outer.rs:62-): the probe rows' pairs in the reference's order, then the unmatched build
    rows -- whose order the reference leaves to its hash map, so that tail is compared as a set."""
    NONE = 0xFFFFFFFF
    rng = np.random.default_rng(31 + multi)
    for nl, nr in ((0, 6), (6, 0), (300, 300), (30_000, 12_000), (12_000, 30_000)):
        def side(n, lo, hi):
            a = rng.integers(lo, hi, n).astype(np.int32)
            am = rng.random(n) >= 0.05
            cols = [Column("a", a, am)]
            if multi:
                cols.append(Column("b", rng.integers(0, 3, n).astype(np.int64)))
            return cols
        left, right = side(nl, 0, 4000), side(nr, 2000, 6000)
        gl, gr = join_indices(left if multi else left[0], right if multi else right[0], "outer")
        wl, wr = (oracle.hash_join_multi(left, right, "outer") if multi else oracle.hash_join(left[0], right[0], "outer"))
        assert len(gl) == len(wl), (nl, nr)
        probe = 1 if not (nl > nr) else 0  # the longer relation probes (the right one on a tie)
        gp, wp = list(zip(gl.tolist(), gr.tolist())), list(zip(wl.tolist(), wr.tolist()))
        assert [p for p in gp if p[probe] != NONE] == [p for p in wp if p[probe] != NONE], (nl, nr)
        gt, wt = [p for p in gp if p[probe] == NONE], [p for p in wp if p[probe] == NONE]
        assert gp[len(gp) - len(gt):] == gt and sorted(gt) == sorted(wt), (nl, nr)  # the tail sits behind the probe rows' pairs


def test_outer_join_reference_answer_and_frame():
    """test_join_sorted_fast_paths_null's outer answer (py-polars/tests/unit/operations/test_join.py:593-596) through the frame API."""
    df1 = pa.DataFrame({"x": np.array([0, 0, 1], np.int64)})
    df2 = pa.DataFrame({"x": (np.array([0, 0], np.int64), np.array([True, False])), "y": np.array([0, 1], np.int64)})
    out = df1.join(df2, on="x", how="outer")
    assert out.to_dict() == {"x": [0, 0, 1, None], "y": [0, 0, None, 1]}


def test_string_key_join_reference_docstring():
    """DataFrame.join's own examples (py-polars/polars/dataframe/frame.py:5913-5979): a string key on both sides, every join kind.
    Both key columns are encoded under one dictionary on the device, the pairs come from plhip_hash_join on the codes.  (The outer
    example's two unmatched rows are compared as a set of rows: the reference prints them in its hash map's order.)"""
    import json
    import os

    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_join_known_answers.json")))["docstring_frames"]
    mk = lambda d: pa.DataFrame({k: (np.array(v, np.int64) if isinstance(v[0], int) else np.array(v) if isinstance(v[0], float) else v) for k, v in d.items()})
    df, other = mk(gold["df"]), mk(gold["other"])
    for how, want in gold["results"].items():
        out = df.join(other, on="ham", how=how).to_dict()
        rows = lambda d: sorted(zip(*[[repr(x) for x in d[c]] for c in want]))
        assert list(out) == list(want), how
        if how == "outer":
            assert rows(out) == rows(want), how
        else:
            assert out == want, how


def test_take_gathers_payload_on_the_device():
    """plhip_take: the payload half of a join (`take_unchecked` / `_take_opt_chunked_unchecked` behind DataFrame::_finish_join,
    crates/polars-ops/src/frame/join/hash_join/mod.rs:105-201): out[i] = col[idx[i]], null where idx[i] is PLHIP_JOIN_NONE or the
    source row is null -- against numpy's take, every width, host and device-resident indices."""
    import ctypes as C

    import torch

    from polars_amd.join import NONE

    L = _ffi.lib()
    rng = np.random.default_rng(8)
    n, m = 100_003, 250_001
    cols = [Column("a", rng.integers(-100, 100, n).astype(np.int8)), Column("b", rng.integers(0, 60000, n).astype(np.uint16), rng.random(n) >= 0.1),
            Column("c", rng.standard_normal(n).astype(np.float32)), Column("d", rng.integers(-2**60, 2**60, n).astype(np.int64), rng.random(n) >= 0.3),
            Column("e", rng.standard_normal(n))]
    idx = rng.integers(0, n, m).astype(np.uint32)
    idx[rng.random(m) < 0.05] = NONE
    tidx = torch.from_numpy(idx.view(np.int32)).to("cuda:0")
    torch.cuda.synchronize()
    arr = (_ffi.ColumnStruct * len(cols))(*[c.struct() for c in cols])
    for ptr, loc in ((idx.ctypes.data, _ffi.HOST), (tidx.data_ptr(), _ffi.DEVICE)):
        h = C.c_void_p()
        _ffi.check(L.plhip_take(arr, len(cols), ptr, m, loc, _ffi.TAKE_NULLABLE_IDX, C.byref(h)))
        try:
            assert L.plhip_result_num_rows(h) == m and L.plhip_result_num_columns(h) == len(cols)
            for j, c in enumerate(cols):
                cs = _ffi.ColumnStruct()
                _ffi.check(L.plhip_result_column(h, j, C.byref(cs)))
                got = pa.frame._host_column("x", cs)
                miss = idx == NONE
                safe = np.where(miss, 0, idx).astype(np.int64)
                want_valid = ~miss if c.valid_mask() is None else (c.valid_mask()[safe] & ~miss)
                assert np.array_equal(got.valid_mask(), want_valid), c.name
                gv, wv = got.to_numpy()[want_valid], c.to_numpy()[safe][want_valid]
                assert np.array_equal(gv.view(np.uint8), wv.view(np.uint8)), c.name
        finally:
            L.plhip_result_free(h)
    # without the flag and without nulls in the source no bitmap is made
    h = C.c_void_p()
    ok_idx = np.where(idx == NONE, 0, idx).astype(np.uint32)
    _ffi.check(L.plhip_take((_ffi.ColumnStruct * 1)(cols[0].struct()), 1, ok_idx.ctypes.data, m, _ffi.HOST, 0, C.byref(h)))
    cs = _ffi.ColumnStruct()
    _ffi.check(L.plhip_result_column(h, 0, C.byref(cs)))
    assert not cs.validity and np.array_equal(pa.frame._host_column("x", cs).to_numpy(), cols[0].to_numpy()[ok_idx])
    L.plhip_result_free(h)
