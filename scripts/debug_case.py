"""Runs one reference known-answer case through a forced strategy and prints the frame (development aid)."""
import sys
sys.path.insert(0, ".")
import polars_amd as pa
from tests import util
from tests.test_parity_gpu import run_product
name, strategy = sys.argv[1], sys.argv[2]
KA = util.load_known_answers()
case = [c for c in KA["group_by"] if c["name"] == name][0]
keys = [util.make_column(k) for k in case["keys"]]
values = [util.make_column(v) for v in case["values"]]
for rep in range(3):
    out = run_product(keys, values, [tuple(a) for a in case["aggs"]], case["maintain_order"], strategy)
    print(pa.last_strategy(), {c: out[c].to_list() for c in out.columns})
print("expected", case["expected"])
