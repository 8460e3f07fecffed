"""Several ranks of the sharded group-by on ONE GPU (gloo, records staged through host memory): the real
libplhip.so passes and exchange kernels on every rank, checked against the oracle on the whole input.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 scripts/rehearse_ranks.py
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polars_amd import _ffi  # noqa: E402
from polars_amd.distributed import group_by_agg_sharded  # noqa: E402
from tests import util  # noqa: E402
from tests.test_distributed_cpu import AGGS, columns, make_data  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda:0")
_ffi.check(_ffi.lib().plhip_init(0))
ok = True
for n, seed, empty_last in ((200_000, 5, False), (50_000, 6, True), (3_000_000, 7, False)):
    data = make_data(n, seed)
    lo, hi = rank * n // world, (rank + 1) * n // world
    if empty_last and rank == world - 1:
        lo = hi
    keys, vals = columns(data, lo, hi)
    res = group_by_agg_sharded(keys, vals, AGGS, device=dev)
    mine = []
    for c in res:
        mine.append(c.t.cpu().numpy().view(_ffi.code_np_dtype(c.code)))
        mine.append(np.ones(c.t.numel(), bool) if c.mask is None else c.mask.cpu().numpy())
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        from oracle import oracle as orc

        ncol = len(AGGS) + 2
        got = [(np.concatenate([g[2 * j] for g in gathered]), np.concatenate([g[2 * j + 1] for g in gathered])) for j in range(ncol)]
        top = n if not empty_last else (world - 1) * n // world
        k, v = columns(data, 0, top)
        want, _ = orc.group_by_agg(k, v, AGGS)
        util.assert_frames_match(got, util.oracle_to_np(want), 2, what=f"{world} ranks on one GPU, n={n}")
        print(f"ok: n={n} world={world} groups={len(got[0][0])} rows per rank={[len(g[0]) for g in gathered]}", flush=True)
# the bench's shape: one i32 key with many groups, sum(i32) + mean(f64), no nulls
from polars_amd.column import Column  # noqa: E402

n, g = 4_000_000, 50_000
rng = np.random.default_rng(11)
k = rng.integers(1, g + 1, n).astype(np.int32)
v1 = rng.integers(1, 6, n).astype(np.int32)
v3 = np.round(rng.uniform(0, 100, n), 6)
lo, hi = rank * n // world, (rank + 1) * n // world
aggs = [(0, "sum"), (1, "mean")]
for hint, what in ((g, "slots"), (100, "hint too small: slots overflow, all ranks fall back"), (0, "no hint: counts protocol")):
    res = group_by_agg_sharded([Column("id3", k[lo:hi])], [Column("v1", v1[lo:hi]), Column("v3", v3[lo:hi])], aggs, device=dev, n_groups_hint=hint)
    mine = []
    for c in res:
        mine.append(c.t.cpu().numpy().view(_ffi.code_np_dtype(c.code)))
        mine.append(np.ones(c.t.numel(), bool) if c.mask is None else c.mask.cpu().numpy())
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        from oracle import oracle as orc

        got = [(np.concatenate([x[2 * j] for x in gathered]), np.concatenate([x[2 * j + 1] for x in gathered])) for j in range(3)]
        want, _ = orc.group_by_agg([Column("id3", k)], [Column("v1", v1), Column("v3", v3)], aggs)
        util.assert_frames_match(got, util.oracle_to_np(want), 1, what=f"q3 shape, {world} ranks, {what}")
        print(f"ok: q3 shape ({what}) n={n} world={world} groups={len(got[0][0])} rows per rank={[len(x[0]) for x in gathered]}", flush=True)
dist.destroy_process_group()
