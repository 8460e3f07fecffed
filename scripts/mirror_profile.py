"""Where the Python mirror's per-query time goes (q3 of the harness through `x.group_by(..).agg(..).collect()`, table in HBM)."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polars_amd as pa  # noqa: E402
from polars_amd import h2oai  # noqa: E402

t = h2oai.generate(10_000_000, seed=108, device=torch.device("cuda:0"), na=True, columns=["id3", "v1", "v3"])
x = pa.DataFrame(h2oai.to_columns(t)).lazy()
q = lambda: x.group_by("id3").agg([pa.col("v1").sum().alias("v1_sum"), pa.col("v3").mean().alias("v3_mean")]).collect()
for _ in range(20):
    q()
t0 = time.perf_counter()
for _ in range(200):
    q()
print(f"{(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per query", flush=True)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    q()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
