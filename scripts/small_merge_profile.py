"""Development aid: kernel / wall time of the merge pass shape (few rows, every row its own group)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
for n, g in ((100_000, 100_000), (100_000, 12_500), (100_000, 50_000), (400_000, 100_000)):
    k = torch.from_numpy(rng.integers(1, g + 1, n).astype(np.int32)).to(dev)
    s = torch.from_numpy(rng.integers(1, 1000, n).astype(np.int64)).to(dev)
    f = torch.from_numpy(rng.uniform(0, 100, n)).to(dev)
    c = torch.from_numpy(rng.integers(1, 100, n).astype(np.uint64).view(np.int64)).to(dev)
    keys = [Column("k", k)]
    vals = [Column("s", s), Column("f", f), Column("c", c, dtype=_ffi.U64)]
    for strategy in ("auto", "tilesort", "sort", "lds", "global"):
        pa.set_strategy(strategy)
        best = wall = 1e9
        for it in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h, _ = pa.group_by_agg_raw(keys, vals, [(0, 0), (1, 0), (2, 0)], flags=2, n_groups_hint=min(n, g))
            dt = (time.perf_counter() - t0) * 1e3
            tot, dom, name = pa.last_kernel_ms()
            if it >= 2:
                best = min(best, tot); wall = min(wall, dt)
            _ffi.lib().plhip_result_free(h)
        print(f"n={n} g={g} {strategy}->{pa.last_strategy():10s} kernels {best:.3f} wall {wall:.3f} ms", flush=True)
pa.set_strategy("auto")
