"""String VALUE columns under a group-by (Utf8Chunked::agg_min / agg_max, aggregations/utf8.rs:21-142; agg_first /
agg_last / agg_n_unique, aggregations/dispatch.rs:48-105,182-212): the strings are encoded on the device, min / max run on their
byte-order ranks, first / last / n_unique on their codes.  Checked against the oracle's restatement (oracle.agg_utf8, pinned in
tests/test_oracle_utf8.py) and against the answers the reference's own tests hold (tests/golden/reference_utf8_value_known_answers.json)."""
import json
import os

import numpy as np
import pytest

import polars_amd as pa
from polars_amd.column import Column
from polars_amd.strings import Utf8Column
from tests.test_utf8_gpu import make_strings

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind", ["db-benchmark", "ragged", "long"])
@pytest.mark.parametrize("n,n_distinct,n_groups,null_frac", [(1, 1, 1, 0.0), (2000, 7, 50, 0.3), (150_000, 20_000, 900, 0.05), (60_000, 60_000, 7, 0.0)])
def test_string_value_aggregates_match_the_oracle(oracle, kind, n, n_distinct, n_groups, null_frac):
    rng = np.random.default_rng(n + n_groups)
    vals = make_strings(rng, n, n_distinct, null_frac, kind)
    key = rng.integers(0, n_groups, n).astype(np.int32)
    g_o = oracle.Groups([Column("k", key)], True)
    want = {k: oracle.agg_utf8(vals, k, g_o) for k in ("min", "max", "first", "last", "n_unique")}
    # the fused query: min / max of the string column beside a count, groups in first-occurrence order
    df = pa.DataFrame({"k": key, "s": vals})
    out = df.group_by("k", maintain_order=True).agg([pa.col("s").min().alias("lo"), pa.col("s").max().alias("hi"), pa.count()]).to_dict()
    assert out["k"] == key[g_o.first].tolist()
    assert out["lo"] == want["min"] and out["hi"] == want["max"]
    assert out["count"] == np.diff(g_o.offsets).tolist()
    # the CSR seams: one column over prebuilt groups
    s = df.get_column("s")
    g = pa.group_tuples([df.get_column("k")], True)
    try:
        for k in ("first", "last", "min", "max"):
            assert pa.agg_groups(s, k, g).to_list() == want[k], k
        assert pa.agg_groups(s, "n_unique", g).to_list() == want["n_unique"]
    finally:
        g.close()


def test_all_null_and_single_row_groups(oracle):
    vals = [None, None, "b", None, "a", "", None, "ab"]
    key = np.array([0, 0, 1, 2, 1, 3, 3, 4], np.int64)
    df = pa.DataFrame({"k": key, "s": vals})
    out = df.group_by("k", maintain_order=True).agg([pa.col("s").min().alias("lo"), pa.col("s").max().alias("hi"),
                                                     pa.col("s").sum().alias("sum"), pa.col("s").mean().alias("mean")]).to_dict()
    assert out["lo"] == [None, "a", None, "", "ab"] and out["hi"] == [None, "b", None, "", "ab"]
    assert out["sum"] == [None] * 5 and out["mean"] == [None] * 5
    g_o = oracle.Groups([Column("k", key)], True)
    assert out["lo"] == oracle.agg_utf8(vals, "min", g_o) and out["hi"] == oracle.agg_utf8(vals, "max", g_o)
    # an empty frame
    empty = pa.DataFrame({"k": np.empty(0, np.int32), "s": Utf8Column.from_list("s", [])})
    assert empty.group_by("k").agg(pa.col("s").min()).to_dict() == {"k": [], "s": []}


def test_reference_answers_string_key_shorthands_and_string_values():
    gold = json.load(open(os.path.join(HERE, "golden", "reference_utf8_value_known_answers.json")))
    sh = gold["shorthands"]
    df = pa.DataFrame(sh["frame"])
    for method, rows in sh["expected"].items():
        gb = df.group_by(sh["by"], maintain_order=True)
        got = gb.agg(pa.count()) if method == "count" else getattr(gb, method)()  # (py-polars' GroupBy.count() is agg(pl.count()))
        assert [list(r) for r in got.rows()] == rows, method
    ty = gold["types"]
    df = pa.DataFrame(ty["frame"])
    for by in ty["by"]:
        lazy = df.lazy().group_by(by).agg([pa.col("person_name").mean().alias("str_mean"), pa.col("person_name").sum().alias("str_sum")]).collect()
        assert lazy.get_column("str_mean").to_list() == ty["expected"]["str_mean"]
        assert lazy.get_column("str_sum").to_list() == ty["expected"]["str_sum"]
        first = df.group_by(by).select(["person_name"]).first()
        last = df.group_by(by).select(["person_name"]).last()
        assert first.get_column("person_name_first").to_list() == ty["expected"]["str_first"]
        assert last.get_column("person_name_last").to_list() == ty["expected"]["str_last"]
