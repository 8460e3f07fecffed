import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import polars_amd as pa
from polars_amd import _ffi
from polars_amd.column import Column
rng = np.random.default_rng(5)
n = (1 << 20) + 12_345
for base in (0xFFFF0000, 0x80000000, 0x7FFF0000, 1000):
    k = (base + rng.integers(0, 40_000, n)).astype(np.uint32)
    u = rng.integers(0, 100, n).astype(np.int32)
    w = np.round(rng.uniform(0, 1, n), 6)
    h, cols = pa.group_by_agg_raw([Column("k", k)], [Column("u", u), Column("w", w)], [(0, _ffi.AGG_SUM), (1, _ffi.AGG_SUM)], 0, 0)
    print(hex(base), pa.last_strategy(), _ffi.lib().plhip_result_num_rows(h), flush=True)
    _ffi.lib().plhip_result_free(h)
