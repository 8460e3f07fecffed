"""Development aid: plhip_group_tuples + agg_groups over q3's key shape with and without a heavy key."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polars_amd as pa
from polars_amd.column import Column
n = 10_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(108)
for hot in (0.0, 0.3):
    k = torch.randint(1, n // 100 + 1, (n,), device=dev, dtype=torch.int32, generator=g)
    k = torch.where(torch.rand(n, device=dev, generator=g) < hot, torch.full_like(k, 7), k)
    v = torch.rand(n, device=dev, dtype=torch.float64, generator=g)
    keys = [Column("k", k)]
    for sorted_ in (False, True):
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gr = pa.group_tuples(keys, sorted_)
            dt = (time.perf_counter() - t0) * 1e3
            best = min(best, dt)
            if it < 2: gr.close()
        tb = {}
        for kind in ("sum", "median", "n_unique", "std"):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            pa.agg_groups(Column("v", v), kind, gr)
            tb[kind] = round((time.perf_counter() - t0) * 1e3, 2)
        gr.close()
        print(f"hot {hot} sorted {sorted_}: group_tuples {best:.2f} ms; agg ms {tb}", flush=True)
