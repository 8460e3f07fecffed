"""The N > 1 path on CPU: the sharded group-by protocol of polars_amd/distributed.py (owner hashing, counts +
ONE all_to_all_single of packed partial records, regroup, finalize) over gloo with world_size 2 and 3.  The two
local group-by passes are played by the oracle here (test stand-in; on the GPU box they are libplhip.so)."""
import os
import socket
import sys
import tempfile
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

AGGS = [(0, "sum"), (1, "mean"), (0, "min"), (1, "max"), (0, "count"), (2, "mean"), (2, "sum")]


def make_data(n, seed):
    rng = np.random.default_rng(seed)
    k1 = rng.integers(-20, 20, n).astype(np.int32)
    k2 = rng.integers(0, 7, n).astype(np.int64)
    k1m = rng.random(n) > 0.05
    v0 = rng.integers(-1000, 1000, n).astype(np.int32)
    v1 = np.round(rng.uniform(0, 100, n), 6)
    v1m = rng.random(n) > 0.1
    v2 = rng.integers(0, 50, n).astype(np.uint32)
    return (k1, k1m, k2), (v0, v1, v1m, v2)


def columns(data, lo, hi):
    from polars_amd.column import Column

    (k1, k1m, k2), (v0, v1, v1m, v2) = data
    keys = [Column("k1", k1[lo:hi], k1m[lo:hi]), Column("k2", k2[lo:hi])]
    vals = [Column("v0", v0[lo:hi]), Column("v1", v1[lo:hi], v1m[lo:hi]), Column("v2", v2[lo:hi])]
    return keys, vals


class OracleEngine:
    """Stand-in for HipEngine: same partial-frame layout, computed by the CPU oracle."""

    def _col(self, name, arr, valid=None, nullable=None):
        import torch

        from polars_amd import _ffi
        from polars_amd.distributed import Col

        code = _ffi.np_dtype_code(arr.dtype)
        view = {np.dtype(np.uint32): np.int32, np.dtype(np.uint64): np.int64, np.dtype(np.uint16): np.int16}.get(arr.dtype)
        t = torch.from_numpy(np.ascontiguousarray(arr.view(view) if view else arr))
        # a mask goes with the column exactly when the source is nullable (schema property, same on all ranks)
        if nullable is None:
            nullable = valid is not None and not valid.all()
        mask = torch.from_numpy((valid if valid is not None else np.ones(len(arr), bool)).copy()) if nullable else None
        return Col(name, t, code, mask)

    def _np_column(self, c):
        from polars_amd import _ffi
        from polars_amd.column import Column

        a = c.t.numpy().view(_ffi.code_np_dtype(c.code))
        return Column(c.name, a, None if c.mask is None else c.mask.numpy())

    def partial(self, keys, vals, aggs):
        from oracle import oracle as orc
        from polars_amd.column import Column

        g = orc.Groups(keys, False)
        out = []
        for kc in keys:
            v, m = orc.take_keys(kc, g)
            out.append(self._col(kc.name, v, m, nullable=kc.validity is not None))
        for i, (ci, kind) in enumerate(aggs):
            src = vals[ci]
            a = src.to_numpy()
            wide = a.astype(np.float64) if a.dtype.kind == "f" else a.astype(np.uint64 if a.dtype.kind == "u" else np.int64)
            m = src.valid_mask()
            if kind == "sum":
                v, _ = orc.agg(Column("w", wide, m), "sum", g)
                out.append(self._col(f"a{i}_sum", v))
            elif kind == "mean":
                v, _ = orc.agg(Column("w", wide, m), "sum", g)
                c, _ = orc.agg(Column("one", np.ones(len(a), np.uint64), m), "sum", g)
                out += [self._col(f"a{i}_sum", v), self._col(f"a{i}_cnt", c)]
            elif kind == "count":
                v, _ = orc.agg(None, "count", g)
                out.append(self._col(f"a{i}_count", v.astype(np.uint64)))
            else:
                v, ok = orc.agg(src, kind, g)
                out.append(self._col(f"a{i}_{kind}", v, ok, nullable=src.validity is not None))
        return out

    def key_hashes(self, keys):
        import torch

        from oracle import oracle as orc
        from polars_amd.distributed import NULL_HASH

        if keys[0].t.numel() == 0:
            return torch.empty(0, dtype=torch.int64)
        return torch.from_numpy(orc.vec_hash([self._np_column(k) for k in keys], NULL_HASH).view(np.int64).copy())

    def merge(self, keys, parts, ops):
        from oracle import oracle as orc

        kc, pc = [self._np_column(k) for k in keys], [self._np_column(p) for p in parts]
        out, _ = orc.group_by_agg(kc, pc, [(i, op) for i, op in enumerate(ops)])
        names = [k.name for k in keys] + [p.name for p in parts]
        return [self._col(nm, v, m, nullable=c.mask is not None) for nm, (_, v, m), c in zip(names, out, list(keys) + list(parts))]


def _worker(rank, world, port, n, seed, outdir):
    import torch.distributed as dist

    from polars_amd import _ffi
    from polars_amd.distributed import group_by_agg_sharded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = make_data(n, seed)
        lo, hi = rank * n // world, (rank + 1) * n // world
        if rank == world - 1 and world == 3:
            lo = hi  # one rank holds no rows at all: empty sends / receives must work
        keys, vals = columns(data, lo, hi)
        res = group_by_agg_sharded(keys, vals, AGGS, engine=OracleEngine())
        arrays = {}
        for j, c in enumerate(res):
            arrays[f"v{j}"] = c.t.numpy().view(_ffi.code_np_dtype(c.code))
            arrays[f"m{j}"] = np.ones(c.t.numel(), bool) if c.mask is None else c.mask.numpy()
        np.savez(os.path.join(outdir, f"rank{rank}.npz"), **arrays)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_group_by_matches_single_process(world):
    import torch.multiprocessing as mp

    from oracle import oracle as orc
    from tests import util

    n, seed = 20000, 5
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), n, seed, d), nprocs=world, join=True)
        parts = [np.load(os.path.join(d, f"rank{r}.npz")) for r in range(world)]
    ncol = len(AGGS) + 2
    got = [(np.concatenate([p[f"v{j}"] for p in parts]), np.concatenate([p[f"m{j}"] for p in parts])) for j in range(ncol)]
    data = make_data(n, seed)
    hi = n if world != 3 else 2 * n // 3  # the last of three ranks contributed no rows
    keys, vals = columns(data, 0, hi)
    want, _ = orc.group_by_agg(keys, vals, AGGS)
    # the shares are disjoint in the keys: no key tuple may come back from two ranks
    kt = set()
    for i in range(len(got[0][0])):
        t = tuple((got[j][0][i].item() if got[j][1][i] else None) for j in range(2))
        assert t not in kt, f"key {t} returned by two ranks"
        kt.add(t)
    util.assert_frames_match(got, util.oracle_to_np(want), 2, what=f"sharded world={world}")
