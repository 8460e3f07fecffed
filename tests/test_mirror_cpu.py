"""Host logic of the Python mirror that needs no GPU: the slice rules applied to a result frame (slice_offsets,
crates/polars-core/src/utils/mod.rs: negative offsets count from the end, out-of-range parts are dropped), expression objects, the
rank table behind string min / max (byte order = Rust's `str` order), and CSV / Arrow ingest up to the chunked columns."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
from polars_amd.frame import _slice_rows
from polars_amd.strings import DecodedColumn


def _ref_slice(xs, offset, length):
    n = len(xs)
    if offset < 0:
        back = min(-offset, n)
        start, ln = n - back, min(length, back)
    else:
        start = min(offset, n)
        ln = min(length, n - start)
    return xs[start:start + ln]


@pytest.mark.parametrize("offset,length", [(0, 3), (2, 100), (7, 2), (10, 1), (-3, 2), (-3, 10), (-100, 4), (0, 0), (-1, 1)])
def test_slice_rules_on_result_columns(offset, length):
    vals = list(range(10))
    mask = np.array([i % 3 != 0 for i in range(10)])
    c = _slice_rows(Column("x", np.array(vals, np.int64), mask), offset, length)
    want = _ref_slice([v if m else None for v, m in zip(vals, mask)], offset, length)
    assert c.to_list() == want and c.name == "x"
    s = _slice_rows(DecodedColumn("s", [None if v % 4 == 0 else f"v{v}" for v in vals]), offset, length)
    assert s.to_list() == _ref_slice([None if v % 4 == 0 else f"v{v}" for v in vals], offset, length)
    # polars: df.slice(-3, 2) on ten rows is rows 7, 8; df.slice(2, 100) is rows 2 .. 9
    assert _ref_slice(vals, -3, 2) == [7, 8] and _ref_slice(vals, 2, 100) == vals[2:]


def test_expressions_carry_their_parameters_through_alias():
    e = pa.col("v").std(0).alias("s0")
    assert (e.column, e.agg, e.ddof, e.output_name) == ("v", "std", 0, "s0")
    q = pa.col("v").alias("q").quantile(0.3, "linear")
    assert (q.agg, q.q, q.interpolation, q.output_name) == ("quantile", 0.3, "linear", "q")
    assert pa.col("v").median().output_name == "v" and pa.count().output_name == "count"
    with pytest.raises(_ffi.ComputeError):
        pa.col("v").sum().median()


def test_byte_order_is_the_rank_order():
    # what strings.ranks_of sorts by: UTF-8 bytes, i.e. code-point order -- 'Z' < 'a' < 'é' (2 bytes, 0xC3..) < 'e\\u0301' is FALSE:
    # 'e' + combining accent starts with 0x65 and sorts before 'é'
    strs = ["a", "Z", "é", "é", "", "ab", "b"]
    order = sorted(range(len(strs)), key=lambda i: strs[i].encode())
    assert [strs[i] for i in order] == ["", "Z", "a", "ab", "b", "é", "é"]
    assert [strs[i] for i in order] == sorted(strs)  # Python's str order is code-point order: the same thing


def test_csv_and_arrow_ingest_up_to_the_chunks(tmp_path):
    import pyarrow as pyar

    path = tmp_path / "t.csv"
    path.write_text("id1,id4,v1,v3\nid001,3,1,0.5\n,4,,1.5\nid002,,5,\nid001,3,2,2.5\n")
    df = pa.read_csv(path, dtypes={"id4": np.int32, "v1": np.int32, "v3": np.float64}, categorical=["id1"])
    assert df.columns == ["id1", "id4", "v1", "v3"] and df.height == 4
    assert df["id1"].dictionary == ["id001", "id002"] and df["id1"].dtype == _ffi.U32 and df["id1"].n_categories == 2
    assert df["id1"].valid_mask().tolist() == [True, False, True, True]
    assert df["id4"].dtype == _ffi.I32 and df["id4"].to_list() == [3, 4, None, 3]
    assert df["v3"].to_list() == [0.5, 1.5, None, 2.5]
    t = pyar.table({"a": pyar.chunked_array([pyar.array([1, 2, None], pyar.int64()), pyar.array([4], pyar.int64())]),
                    "b": pyar.chunked_array([pyar.array([True, False, True]), pyar.array([None], pyar.bool_())])})
    fr = pa.from_arrow(t)
    assert fr["a"].n_chunks == 2 and fr["a"].to_list() == [1, 2, None, 4]
    assert fr["b"].dtype == _ffi.BOOL and fr["b"].to_list() == [True, False, True, None]
    one = fr["a"].rechunk()
    assert isinstance(one, Column) and one.to_list() == [1, 2, None, 4]
