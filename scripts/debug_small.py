import sys
sys.path.insert(0, ".")
import numpy as np
import polars_amd as pa
from polars_amd.column import Column
from tests.test_parity_gpu import run_product
k = np.array([1, 1, 1, 1, 2], dtype=np.int32)
for vdt in (np.uint32, np.int32, np.int64, np.float64):
    for strat in ("direct", "tilesort"):
        a = np.array([1, 0, 0, 1, 1]).astype(vdt); b = np.array([0, 1, 0, 0, 0]).astype(vdt)
        out = run_product([Column("k", k)], [Column("a", a), Column("b", b)], [(0, "sum"), (1, "sum")], False, strat)
        print(np.dtype(vdt).name, strat, pa.last_strategy(), {c: out[c].to_list() for c in out.columns})
        out = run_product([Column("k", k)], [Column("a", a)], [(0, "sum"), (0, "count")], False, strat)
        print("   one col:", {c: out[c].to_list() for c in out.columns})
