"""Development aid: plhip_group_tuples at q3's key shape, 1e7 rows, wall time per call (run under rocprofv3 --kernel-trace --stats
for the per-kernel split)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import polars_amd as pa
from polars_amd.column import Column
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(108)
k = torch.randint(1, n // 100 + 1, (n,), device=dev, dtype=torch.int32, generator=g)
keys = [Column("k", k)]
from polars_amd import _ffi
for loc, what in ((_ffi.DEVICE, "CSR stays in HBM"), (_ffi.HOST, "CSR copied to pageable host memory")):
    for sorted_ in (False, True):
        ts = []
        for it in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            gr = pa.group_tuples(keys, sorted_, loc)
            ts.append((time.perf_counter() - t0) * 1e3)
            gr.close()
        print(f"n {n} sorted {sorted_} ({what}): group_tuples ms {[round(t, 2) for t in ts]}", flush=True)
