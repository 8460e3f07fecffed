"""Dense run-time kind against the hashed pipelines for a few query shapes (q3's frame, 1e7 rows, 1e5 codes):
    python scripts/dense_rt_probe.py        (run twice: with and without PLHIP_DENSE_ALLOW_RT=1)"""
import ctypes as C
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polars_amd as pa  # noqa: E402
from polars_amd import _ffi, h2oai  # noqa: E402

n = 10_000_000
dev = torch.device("cuda:0")
t = h2oai.generate(n, seed=108, device=dev, columns=["id3", "v1", "v2", "v3"])
cols = h2oai.to_columns(t)
L = _ffi.lib()
shapes = {
    "sum(v3)": (["v3"], [(0, "sum")]),
    "sum(v3),count": (["v3"], [(0, "sum"), (0, "count")]),
    "mean(v3)": (["v3"], [(0, "mean")]),
    "q3+max(v1)": (["v1", "v3"], [(0, "sum"), (1, "mean"), (0, "max")]),
    "mean(v1)": (["v1"], [(0, "mean")]),
    "max(v1),min(v2)": (["v1", "v2"], [(0, "max"), (1, "min")]),
    "min(v3),max(v3)": (["v3"], [(0, "min"), (0, "max")]),
    "count": ([], [(-1, "count")]),
}
for name, (vn, aggs) in shapes.items():
    keys, vals = [cols["id3"]], [cols[v] for v in vn]
    codes = [(ci, _ffi.AGG_NAMES[k]) for ci, k in aggs]
    out = []
    for dense in (True, False):
        flags = _ffi.FLAG_OUTPUT_DEVICE | (_ffi.FLAG_KEYS_DENSE if dense else 0)
        if not dense:
            os.environ["PLHIP_NO_RANGE_DENSE"] = "1"
        else:
            os.environ.pop("PLHIP_NO_RANGE_DENSE", None)
        for _ in range(5):
            h, _c = pa.group_by_agg_raw(keys, vals, codes, flags, n // 100)
            L.plhip_result_free(h)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            h, _c = pa.group_by_agg_raw(keys, vals, codes, flags, n // 100)
            L.plhip_result_free(h)
        torch.cuda.synchronize()
        out.append((pa.last_strategy(), round((time.perf_counter() - t0) * 10, 4)))
    print(f"{name:20s} dense-flag: {out[0]}   no flag: {out[1]}", flush=True)
