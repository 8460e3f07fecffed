"""Times plhip_hash_join, device-resident i32 keys: python scripts/join_profile.py [probe rows] [build rows]
(db-benchmark's join task shape: a 1e7-row table against a 1e6-row table of unique ids)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_amd import _ffi  # noqa: E402
from polars_amd.column import Column  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
m = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
L = _ffi.lib()
_ffi.check(L.plhip_init(0))
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(3)
lk = torch.randint(0, int(m * 1.1), (n,), generator=gen, device=dev, dtype=torch.int32)
rk = torch.randperm(int(m * 1.1), generator=gen, device=dev)[:m].to(torch.int32).contiguous()
torch.cuda.synchronize()
left, right = Column("k", lk), Column("k", rk)
ls, rs = left.struct(), right.struct()
for how, code in (("inner", 0), ("left", 1)):
    best, rows = 1e9, 0
    for it in range(6):
        h = C.c_void_p()
        t0 = time.perf_counter()
        _ffi.check(L.plhip_hash_join(C.byref(ls), C.byref(rs), code, _ffi.DEVICE, C.byref(h)))
        dt = time.perf_counter() - t0
        rows = L.plhip_join_num_rows(h)
        L.plhip_join_free(h)
        if it:
            best = min(best, dt)
    print(f"{how} join: {n} x {m} rows -> {rows} pairs: {best * 1e3:.3f} ms = {(n + m) / best / 1e9:.2f} Grows/s", flush=True)

# the CPU port beside it (oracle: the reference's single-threaded build + probe restated in C; checker, timed once here)
from oracle import oracle as orc  # noqa: E402

hl, hr = Column("k", lk.cpu().numpy()), Column("k", rk.cpu().numpy())
orc.lib()
t0 = time.perf_counter()
li, ri = orc.hash_join(hl, hr, "inner")
dt = time.perf_counter() - t0
print(f"oracle (CPU port, 1 thread) inner join: {len(li)} pairs in {dt * 1e3:.0f} ms = {(n + m) / dt / 1e9:.3f} Grows/s", flush=True)
# algorithmic bytes: both key columns read once, two u32 index arrays written once
ab = (n + m) * 4 + 2 * 4 * rows
print(f"algorithmic bytes {ab / 1e6:.0f} MB -> {ab / best / 1e9:.0f} GB/s = {ab / best / 8e12 * 100:.1f} % of 8 TB/s (left join timing)", flush=True)
