#!/bin/bash
# kernels of the sharded step on one rank (forced exchange): scripts/gpu_q3x_stats.sh <tag>
export PLHIP_BENCH_FORCE_EXCHANGE=1
bash scripts/gpu_stats.sh $1 --workload q3 --steps 100 > /dev/null
python - <<PY
import csv,re
rows=list(csv.DictReader(open("gpurun_out/$1/stats/stats_kernel_stats.csv")))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:22]:
    n=re.sub(r"^void ","",r["Name"]).replace("(anonymous namespace)::","")[:60]
    print(f"{n:60s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
grep "^{" gpurun_out/$1/stats.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['sharded_step_ms'], d['shuffle'])"
