"""q6 of the reference's harness (run_h2oai_benchmark.py: median(v3), std(v3) by id4, id5) through the CSR seams, 1e7 rows:
plhip_group_tuples (device-resident CSR) + plhip_agg_groups(median) + plhip_agg_groups(std), checked against the oracle."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polars_amd as pa  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (checker only)
from polars_amd import _ffi, h2oai  # noqa: E402
from polars_amd.column import Column  # noqa: E402

n = 10_000_000
dev = torch.device("cuda:0")
t = h2oai.generate(n, seed=108, device=dev, columns=["id4", "id5", "v3"])
cols = h2oai.to_columns(t)
keys, v3 = [cols["id4"], cols["id5"]], cols["v3"]
L = _ffi.lib()


def once(check=False):
    g = pa.group_tuples(keys, False, _ffi.DEVICE)
    out = []
    for kind in ("median", "std"):
        h = C.c_void_p()
        _ffi.check(L.plhip_agg_groups(C.byref(v3.struct()), _ffi.AGG_GROUPS_ONLY[kind], g._h, (1 << 8) | _ffi.FLAG_OUTPUT_DEVICE, C.byref(h)))
        if check:
            cs = _ffi.ColumnStruct()
            _ffi.check(L.plhip_result_column(h, 0, C.byref(cs)))
            out.append(pa.frame._host_column(kind, cs).to_numpy().copy())
        L.plhip_result_free(h)
    ng = g.n_groups
    g.close()
    return ng, out


for _ in range(3):
    once()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 20
for _ in range(K):
    ng, _ = once()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / K * 1e3
ng, got = once(check=True)
hk = [Column("id4", t["id4"].cpu().numpy()), Column("id5", t["id5"].cpu().numpy())]
hv = Column("v3", t["v3"].cpu().numpy())
g_o = orc.Groups(hk, True)
g_p = pa.group_tuples(hk, True)
ok = True
for kind in ("median", "std"):
    wv, wm = orc.agg_ext(hv, kind, g_o, 1)
    gv = pa.agg_groups(hv, kind, g_p).to_numpy()
    ok &= bool(np.allclose(gv, wv, rtol=1e-11, atol=0))
print(f"q6: {ng} groups, {ms:.3f} ms per query (group_tuples + median + std, device-resident), {n / ms * 1e3:.3g} rows/s, parity {ok}", flush=True)
