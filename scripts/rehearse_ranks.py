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
# dense codes (a Categorical key with a global rev-map): owners by code range, raw partial arrays on the wire.  Cases: the plain
# one with a null key and nullable values; a last rank without rows; a rank whose buffers are misaligned (it flags its slot
# headers and ALL ranks fall back to the hashed protocol together); a rank with a code beyond the hint (same fall-back).
from polars_amd.distributed import HipEngine  # noqa: E402

eng = HipEngine(dev)
eng.dense_keys = True
ncat = 30_001
n = 2_000_000
rng = np.random.default_rng(23)
kc = rng.integers(0, ncat, n).astype(np.uint32)
kc[kc == 17] = 18  # a code that never occurs
km = rng.random(n) >= 0.01
w1 = rng.integers(-5, 6, n).astype(np.int32)
w3 = np.round(rng.uniform(0, 100, n), 6)
w3m = rng.random(n) >= 0.05
daggs = [(0, "sum"), (1, "mean"), (1, "min"), (1, "max"), (0, "count"), (1, "sum"), (0, "mean")]
for what in ("dense", "dense, last rank empty", "dense, rank 0 misaligned", "dense, a code beyond the hint on the last rank"):
    lo, hi = rank * n // world, (rank + 1) * n // world
    if "empty" in what and rank == world - 1:
        lo = hi
    if "misaligned" in what and rank == 0:
        lo += 1
    kk = kc[lo:hi].copy()
    if "beyond" in what and rank == world - 1:
        kk[np.flatnonzero(km[lo:hi])[:3]] = ncat + 3  # (valid rows: a null row's code is never looked at)
    keys = [Column("id3", kk, km[lo:hi], dtype=_ffi.U32).set_categories(ncat)]
    vals = [Column("v1", w1[lo:hi]), Column("v3", w3[lo:hi], w3m[lo:hi])]
    if "misaligned" in what and rank == 0:  # device-resident buffers one element into their allocations (host columns are staged)
        keys = [Column("id3", torch.from_numpy(kc[lo - 1:hi].view(np.int32)).to(dev)[1:], km[lo:hi], dtype=_ffi.U32).set_categories(ncat)]
    res = group_by_agg_sharded(keys, vals, daggs, device=dev, engine=eng, n_groups_hint=ncat)
    want_proto = "dense" if what in ("dense", "dense, last rank empty") else ("slots", "exact")
    assert eng.last_protocol.startswith(want_proto), (what, eng.last_protocol)
    mine = []
    for c in res:
        mine.append(c.t.cpu().numpy().view(_ffi.code_np_dtype(c.code)))
        mine.append(np.ones(c.t.numel(), bool) if c.mask is None else c.mask.cpu().numpy())
    mine.append(np.concatenate([kk, np.zeros(0, np.uint32)]))
    mine.append((lo, hi))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        from oracle import oracle as orc

        ncol = len(daggs) + 1
        got = [(np.concatenate([x[2 * j] for x in gathered]), np.concatenate([x[2 * j + 1] for x in gathered])) for j in range(ncol)]
        rows = np.concatenate([np.arange(x[-1][0], x[-1][1]) for x in gathered])
        allk = np.concatenate([x[-2] for x in gathered])
        want, _ = orc.group_by_agg([Column("id3", allk, km[rows])], [Column("v1", w1[rows]), Column("v3", w3[rows], w3m[rows])], daggs)
        util.assert_frames_match(got, util.oracle_to_np(want), 1, what=f"{what}, {world} ranks")
        print(f"ok: {what} n={len(rows)} world={world} groups={len(got[0][0])} rows per rank={[len(x[0]) for x in gathered]} "
              f"protocol={eng.last_protocol[:12]} host syncs={eng.last_host_syncs}", flush=True)
# more codes than eight LDS ranges hold (config 4's regime): the local pass range-partitions its rows (dense_part) and the exchange
# stays the dense one; a last rank without rows sends the accumulators' initial values
ncat2, n2 = 400_003, 3_000_000
rng = np.random.default_rng(29)
kd = rng.integers(0, ncat2, n2).astype(np.uint32)
u1 = rng.integers(-5, 6, n2).astype(np.int32)
u3 = np.round(rng.uniform(0, 100, n2), 6)
u3m = rng.random(n2) >= 0.05
for what in ("dense beyond eight ranges", "dense beyond eight ranges, last rank empty"):
    lo, hi = rank * n2 // world, (rank + 1) * n2 // world
    if "empty" in what and rank == world - 1:
        lo = hi
    keys = [Column("id3", kd[lo:hi], dtype=_ffi.U32).set_categories(ncat2)]
    vals = [Column("v1", u1[lo:hi]), Column("v3", u3[lo:hi], u3m[lo:hi])]
    res = group_by_agg_sharded(keys, vals, daggs, device=dev, engine=eng, n_groups_hint=ncat2)
    assert eng.last_protocol.startswith("dense"), (what, eng.last_protocol)
    assert lo == hi or eng.last_partial_strategy == "dense_part", (what, eng.last_partial_strategy)
    mine = []
    for c in res:
        mine.append(c.t.cpu().numpy().view(_ffi.code_np_dtype(c.code)))
        mine.append(np.ones(c.t.numel(), bool) if c.mask is None else c.mask.cpu().numpy())
    mine.append((lo, hi))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        from oracle import oracle as orc

        ncol = len(daggs) + 1
        got = [(np.concatenate([x[2 * j] for x in gathered]), np.concatenate([x[2 * j + 1] for x in gathered])) for j in range(ncol)]
        rows = np.concatenate([np.arange(x[-1][0], x[-1][1]) for x in gathered])
        want, _ = orc.group_by_agg([Column("id3", kd[rows])], [Column("v1", u1[rows]), Column("v3", u3[rows], u3m[rows])], daggs)
        util.assert_frames_match(got, util.oracle_to_np(want), 1, what=f"{what}, {world} ranks")
        print(f"ok: {what} n={len(rows)} world={world} groups={len(got[0][0])} rows per rank={[len(x[0]) for x in gathered]} "
              f"protocol={eng.last_protocol[:12]} host syncs={eng.last_host_syncs}", flush=True)
dist.destroy_process_group()
