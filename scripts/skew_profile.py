"""Development aid: how the strategies behave under heavy-hitter skew.  Keys ~ Zipf(s = 1.1) by inverse CDF
(X = floor(U^(-1/(s-1))), P(X >= k) = k^-(s-1): the top key holds ~6.7 % of the rows), folded onto `ids` distinct ids,
mixed half and half with a uniform tail so that (nearly) all ids appear -- BASELINE config 5's recipe at a smaller size.
usage: python scripts/skew_profile.py [rows] [ids]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
ids = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(108)
u = torch.rand(n, device=dev, dtype=torch.float64, generator=g).clamp_min(1e-300)
z = torch.floor(u.pow(-10.0)).clamp_max(float(ids) * 1e6)
k = torch.where(torch.rand(n, device=dev, generator=g) < float(os.environ.get("SKEW_FRAC", "0.5")), (z - 1) % ids, torch.floor(torch.rand(n, device=dev, dtype=torch.float64, generator=g) * ids)).to(torch.int64)
del u, z
v3 = torch.rand(n, device=dev, dtype=torch.float64, generator=g) * 100
v1 = torch.randint(1, 6, (n,), device=dev, dtype=torch.int64, generator=g)
top = torch.bincount((k % 1024).to(torch.int64)).max().item()
keys, vals = [Column("k", k)], [Column("v3", v3), Column("v1", v1)]
print(f"rows {n} ids {ids} distinct {int(torch.unique(k).numel())} rows of key 0: {int((k == 0).sum())}", flush=True)
for strategy in os.environ.get("STRATS", "auto,tilesort2,sort").split(","):
    pa.set_strategy(strategy)
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        h, _ = pa.group_by_agg_raw(keys, vals, [(0, 0), (1, 0)], flags=2)
        dt = (time.perf_counter() - t0) * 1e3
        G = _ffi.lib().plhip_result_num_rows(h)
        _ffi.lib().plhip_result_free(h)
        best = min(best, dt)
    print(f"{strategy:10s} -> {pa.last_strategy():10s} {best:9.3f} ms  groups {G}  breakdown {pa.last_kernel_breakdown()}", flush=True)
pa.set_strategy("auto")
