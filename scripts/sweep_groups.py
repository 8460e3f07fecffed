"""Development aid: q3-shaped query at a fixed row count over a range of group counts, per strategy.
Prints the library's own kernel time (HIP events) so the AUTO thresholds in run_fast can be set from data.
usage: python scripts/sweep_groups.py [rows] [hint 0/1]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polars_amd as pa  # noqa: E402
from polars_amd import _ffi  # noqa: E402
from polars_amd.column import Column  # noqa: E402

FLAG_DEV = 2  # PLHIP_FLAG_OUTPUT_DEVICE

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
use_hint = len(sys.argv) > 2 and sys.argv[2] == "1"
dev = torch.device("cuda:0")
rng = np.random.default_rng(7)
v1 = torch.from_numpy(rng.integers(1, 6, n).astype(np.int32)).to(dev)
v3 = torch.from_numpy(np.round(rng.uniform(0, 100, n), 6)).to(dev)
for g in (200, 500, 1000, 1500, 2000, 2500, 3000, 4000, 6000, 8000, 12000, 16000, 30000, 100000):
    id3 = torch.from_numpy(rng.integers(1, g + 1, n).astype(np.int32)).to(dev)
    keys = [Column("id3", id3)]
    vals = [Column("v1", v1), Column("v3", v3)]
    line = [f"G={g:7d}"]
    for strategy in ("direct", "tilesort", "auto"):
        pa.set_strategy(strategy)
        best = wall = 1e9
        for it in range(8):
            t0 = time.perf_counter()
            h, _ = pa.group_by_agg_raw(keys, vals, [(0, 0), (1, 1)], flags=FLAG_DEV, n_groups_hint=g if use_hint else 0)
            dt = (time.perf_counter() - t0) * 1e3
            bd = pa.last_kernel_breakdown()
            if it >= 2:
                best = min(best, sum(bd))
                wall = min(wall, dt)
            _ffi.lib().plhip_result_free(h)
        line.append(f"{strategy}->{pa.last_strategy():9s} {best:.3f}/{wall:.3f} ms")
    print("  ".join(line), flush=True)
pa.set_strategy("auto")
