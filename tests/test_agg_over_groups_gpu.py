"""agg() with aggregations that have no fused accumulator -- first / last / median / n_unique / std / var / quantile -- takes the
reference's general route: the GroupsProxy once, every expression over it (group_by_helper, executors/group_by.rs:57-95;
AggregationExpr::evaluate_on_groups, aggregation.rs:44-258).  Here: plhip_group_tuples + plhip_agg_groups behind
`df.group_by(..).agg([..])`, checked against the oracle's CSR restatements, on db-benchmark's q6 shape among others."""
import numpy as np
import pytest

import polars_amd as pa
from polars_amd.column import Column
from tests import util

pytestmark = pytest.mark.gpu


def _close(a, b, rtol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return bool(np.all((a == b) | (np.abs(a - b) <= rtol * np.maximum(np.abs(a), np.abs(b)))))


@pytest.mark.parametrize("on_device", [False, True])
def test_q6_shape_median_and_std(oracle, on_device):
    """run_h2oai_benchmark.py q6: median(v3), std(v3) by id4, id5."""
    import torch

    rng = np.random.default_rng(6)
    n = 120_000
    id4, id5 = rng.integers(1, 31, n).astype(np.int32), rng.integers(1, 31, n).astype(np.int32)
    v3 = np.round(rng.uniform(0, 100, n), 6)
    put = (lambda a: torch.from_numpy(a).to("cuda:0")) if on_device else (lambda a: a)
    df = pa.DataFrame({"id4": Column("id4", put(id4)), "id5": Column("id5", put(id5)), "v3": Column("v3", put(v3))})
    out = df.lazy().group_by(["id4", "id5"], maintain_order=True).agg(
        [pa.col("v3").median().alias("v3_median"), pa.col("v3").std().alias("v3_std")]).collect()
    assert out.columns == ["id4", "id5", "v3_median", "v3_std"]
    g = oracle.Groups([Column("id4", id4), Column("id5", id5)], True)
    assert out.get_column("id4").to_numpy().tolist() == id4[g.first].tolist()
    assert out.get_column("id5").to_numpy().tolist() == id5[g.first].tolist()
    for name, kind in (("v3_median", "median"), ("v3_std", "std")):
        want, valid = oracle.agg_ext(Column("v3", v3), kind, g, 1)
        m = out.get_column(name).valid_mask()
        assert valid.all() and (m is None or m.all())
        assert _close(out.get_column(name).to_numpy(), want, 1e-11), kind


def test_mixed_expressions_over_one_set_of_groups(oracle):
    rng = np.random.default_rng(11)
    n = 50_000
    k = rng.integers(-40, 40, n).astype(np.int64)
    x = rng.integers(-1000, 1000, n).astype(np.int32)
    xm = rng.random(n) > 0.1
    y = rng.normal(size=n).astype(np.float32)
    df = pa.DataFrame({"k": k, "x": (x, xm), "y": y})
    exprs = [pa.col("x").sum().alias("x_sum"), pa.col("x").first().alias("x_first"), pa.col("x").last().alias("x_last"),
             pa.col("x").n_unique().alias("x_nu"), pa.col("y").var(0).alias("y_var0"), pa.col("y").quantile(0.3, "linear").alias("y_q"),
             pa.col("x").median().alias("x_med"), pa.count(), pa.col("y").max().alias("y_max")]
    out = df.group_by("k", maintain_order=True).agg(exprs)
    g = oracle.Groups([Column("k", k)], True)
    xc, yc = Column("x", x, xm), Column("y", y)
    assert out.get_column("k").to_numpy().tolist() == k[g.first].tolist()

    def check(name, want, valid, rtol=0.0):
        c = out.get_column(name)
        m = c.valid_mask()
        got_valid = np.ones(len(c), bool) if m is None else m
        assert got_valid.tolist() == np.asarray(valid, bool).tolist(), name
        a, b = c.to_numpy()[got_valid], np.asarray(want)[got_valid]
        assert _close(a, b, rtol), name

    check("x_sum", *oracle.agg(xc, "sum", g))
    check("x_first", *oracle.agg_ext(xc, "first", g))
    check("x_last", *oracle.agg_ext(xc, "last", g))
    check("x_nu", *oracle.agg_ext(xc, "n_unique", g))
    check("y_var0", *oracle.agg_ext(yc, "var", g, 0), rtol=1e-5)
    check("y_q", *oracle.agg_quantile(yc, 0.3, "linear", g), rtol=1e-6)
    check("x_med", *oracle.agg_ext(xc, "median", g), rtol=1e-12)
    check("y_max", *oracle.agg(yc, "max", g))
    assert out.get_column("count").to_numpy().tolist() == np.diff(g.offsets).tolist()
    # a slice behind the aggregation: the reference's rules, negative offsets from the end
    G = g.n_groups
    head = df.lazy().group_by("k", maintain_order=True).agg(exprs).slice(3, 5).collect()
    tail = df.lazy().group_by("k", maintain_order=True).agg(exprs).tail(4).collect()
    assert head.get_column("k").to_numpy().tolist() == k[g.first][3:8].tolist()
    assert tail.get_column("k").to_numpy().tolist() == k[g.first][G - 4:].tolist()
    assert head.get_column("x_med").to_list() == out.get_column("x_med").to_list()[3:8]
    # duplicate output names are an error, as in the reference
    with pytest.raises(pa._ffi.PlHipError):
        df.group_by("k").agg([pa.col("x").median(), pa.col("x").first()])


def test_string_key_and_string_value_over_groups(oracle):
    vals = ["pear", None, "apple", "fig", "apple", None, "kiwi", "fig"]
    key = ["x", "y", "x", "y", "z", "z", "x", "y"]
    v = np.arange(8, dtype=np.float64)
    df = pa.DataFrame({"k": key, "s": vals, "v": v})
    out = df.group_by("k", maintain_order=True).agg([pa.col("s").first().alias("s_first"), pa.col("s").last().alias("s_last"),
                                                     pa.col("s").min().alias("s_min"), pa.col("s").n_unique().alias("s_nu"),
                                                     pa.col("v").median().alias("v_med")]).to_dict()
    assert out == {"k": ["x", "y", "z"], "s_first": ["pear", None, "apple"], "s_last": ["kiwi", "fig", None],
                   "s_min": ["apple", "fig", "apple"], "s_nu": [3, 2, 2], "v_med": [2.0, 3.0, 4.5]}
