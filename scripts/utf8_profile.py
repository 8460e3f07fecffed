"""Times plhip_utf8_encode on db-benchmark's id3 as strings ("id" + 10 digits): python scripts/utf8_profile.py [rows] [distinct].
Device-resident input (offsets + data in HBM), then the group-by on the codes (q3: sum(v1), mean(v3))."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polars_amd import _ffi  # noqa: E402
from polars_amd.column import Column  # noqa: E402
from polars_amd.frame import group_by_agg_raw  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
g = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000
L = _ffi.lib()
_ffi.check(L.plhip_init(0))
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev)
gen.manual_seed(1)
ids = torch.randint(1, g + 1, (n,), generator=gen, device=dev, dtype=torch.int64)
# "id%010d" built on the device: 12 bytes per row
digits = torch.stack([(ids // 10**p) % 10 for p in range(9, -1, -1)], dim=1).to(torch.uint8) + 48
data = torch.cat([torch.full((n, 1), ord("i"), dtype=torch.uint8, device=dev), torch.full((n, 1), ord("d"), dtype=torch.uint8, device=dev), digits], dim=1).contiguous().view(-1)
offsets = torch.arange(0, 12 * (n + 1), 12, dtype=torch.int64, device=dev)
v1 = torch.randint(1, 6, (n,), generator=gen, device=dev, dtype=torch.int32)
v3 = torch.rand((n,), generator=gen, device=dev, dtype=torch.float64) * 100
torch.cuda.synchronize()
cs = _ffi.Utf8ColumnStruct(offsets.data_ptr(), data.data_ptr(), None, 0, n, _ffi.DEVICE, 0)
best = 1e9
for it in range(6):
    handle, codes = C.c_void_p(), _ffi.ColumnStruct()
    t0 = time.perf_counter()
    _ffi.check(L.plhip_utf8_encode(C.byref(cs), C.byref(handle), C.byref(codes)))
    dt = time.perf_counter() - t0
    if it:
        best = min(best, dt)
    ncat = L.plhip_dictionary_len(handle)
    if it < 5:
        L.plhip_dictionary_free(handle)
print(f"encode: {n} rows, {ncat} distinct strings, {data.numel() / 1e6:.0f} MB of bytes + {offsets.numel() * 8 / 1e6:.0f} MB of offsets: "
      f"{best * 1e3:.3f} ms = {n / best / 1e9:.2f} Grows/s, {(data.numel() + offsets.numel() * 8) / best / 1e9:.0f} GB/s of input", flush=True)


class _Codes(Column):
    def __init__(self, cs, ncat):
        self.name, self.sorted_flag, self.n_categories = "id3", None, ncat
        self.location, self.dtype, self._ptr, self._vptr = _ffi.DEVICE, _ffi.U32, cs.values, cs.validity
        self.values, self.validity, self.offset, self.length = None, None, 0, int(cs.length)


kc = _Codes(codes, ncat)
vals = [Column("v1", v1), Column("v3", v3)]
aggs = [(0, _ffi.AGG_SUM), (1, _ffi.AGG_MEAN)]
best = 1e9
for it in range(20):
    t0 = time.perf_counter()
    h, cols = group_by_agg_raw([kc], vals, aggs, _ffi.FLAG_OUTPUT_DEVICE | _ffi.FLAG_KEYS_DENSE, ncat)
    dt = time.perf_counter() - t0
    L.plhip_result_free(h)
    if it > 2:
        best = min(best, dt)
print(f"group-by on the codes ({L.plhip_last_strategy().decode()}): {best * 1e3:.3f} ms", flush=True)
L.plhip_dictionary_free(handle)
