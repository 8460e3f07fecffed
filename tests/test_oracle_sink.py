"""Path B of the CPU baseline (oracle.sink_group_by: the streaming engine's PrimitiveGroupbySink restated,
crates/polars-pipe/src/executors/sinks/group_by/primitive/mod.rs:359-454) against path A (index lists + gather-fold):
two restatements of two reference code paths must agree on every group."""
import numpy as np
import pytest

from oracle import oracle as orc
from polars_amd.column import Column


@pytest.mark.parametrize("threads", [1, 3, 8])
@pytest.mark.parametrize("kdt", [np.int32, np.int64, np.uint32])
def test_sink_matches_index_list_path(threads, kdt):
    rng = np.random.default_rng(17)
    n = 120_000
    k = rng.integers(1, 3001, n).astype(kdt)
    km = rng.random(n) > 0.02
    v1 = rng.integers(-5, 6, n).astype(np.int32)
    v3 = np.round(rng.uniform(0, 100, n), 6)
    v3m = rng.random(n) > 0.05
    key, vals = Column("k", k, km), [Column("v1", v1), Column("v3", v3, v3m)]
    aggs = [(0, "sum"), (1, "mean"), (1, "min"), (1, "max"), (0, "count"), (1, "sum"), (0, "min")]
    ok, okv, ov, ovv = orc.sink_group_by(key, vals, aggs, n_threads=threads, morsel=7_000)
    want, _ = orc.group_by_agg([key], vals, aggs, n_threads=threads)
    width = np.dtype(kdt).itemsize * 8
    gk = np.where(okv, ok & ((1 << width) - 1), 0).astype(np.uint64)
    wk = np.where(want[0][2], want[0][1].astype(np.int64).astype(np.uint64) & np.uint64((1 << width) - 1), 0).astype(np.uint64)
    o, wo = np.lexsort((gk, okv)), np.lexsort((wk, want[0][2]))
    assert len(gk) == len(wk) and np.array_equal(gk[o], wk[wo]) and np.array_equal(okv[o], want[0][2][wo])
    for j in range(len(aggs)):
        gm, wm = ovv[j][o], want[1 + j][2][wo]
        assert np.array_equal(gm, wm), aggs[j]
        np.testing.assert_allclose(ov[j][o][gm], want[1 + j][1][wo][wm].astype(np.float64), rtol=1e-12, err_msg=str(aggs[j]))
    assert orc.sink_group_by(key, vals, aggs, n_threads=threads, outputs=False) == len(wk)
