"""Development aid: one heavy key (or null keys) at the headline size -- q3's shape, 1e7 rows, 1e5 ids.
usage: python scripts/skew_small.py [rows] [hot fraction] [null fraction]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column, pack_validity

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
hot = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
nul = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(108)
ids = max(n // 100, 1)
k = torch.randint(1, ids + 1, (n,), device=dev, dtype=torch.int32, generator=g)
k = torch.where(torch.rand(n, device=dev, generator=g) < hot, torch.full_like(k, 7), k)
v1 = torch.randint(1, 6, (n,), device=dev, dtype=torch.int32, generator=g)
v3 = torch.rand(n, device=dev, dtype=torch.float64, generator=g) * 100
kc = Column("id3", k)
if nul > 0:
    m = (torch.rand(n, device=dev, generator=g) >= nul).cpu().numpy()
    kc = Column("id3", k, torch.from_numpy(pack_validity(m)).to(dev), packed=True)
keys, vals = [kc], [Column("v1", v1), Column("v3", v3)]
for strategy in os.environ.get("STRATS", "auto,tilesort,sort").split(","):
    pa.set_strategy(strategy)
    for hint in (ids, 0):
        best = 1e9
        for it in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            h, _ = pa.group_by_agg_raw(keys, vals, [(0, 0), (1, 1)], 2, hint)
            dt = (time.perf_counter() - t0) * 1e3
            G = _ffi.lib().plhip_result_num_rows(h)
            _ffi.lib().plhip_result_free(h)
            best = min(best, dt)
        print(f"hot {hot} null {nul} {strategy:9s} hint {hint:7d} -> {pa.last_strategy():10s} {best:8.3f} ms groups {G} breakdown {tuple(round(x, 3) for x in pa.last_kernel_breakdown())}", flush=True)
pa.set_strategy("auto")
