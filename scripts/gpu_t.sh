#!/bin/bash
bash scripts/gpu_tests.sh 1000 || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3l.json 2> gpurun_out/bench_r3l.err || { tail -5 gpurun_out/bench_r3l.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r3l.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["parity"]["ok"] if isinstance(d.get("parity"), dict) else d.get("parity"))
print({k: (round(v.get("ms_per_step"), 4), v.get("parity", {}).get("ok")) for k, v in d.get("configs", {}).items()})
PY
bash scripts/gpu_stats.sh r3l_q3 --steps 300 --warmup 20 --configs none --no-parity || exit 1
bash scripts/gpu_stats.sh r3l_c3 --workload c3 --steps 20 --warmup 3 --configs none --no-parity || exit 1
bash scripts/gpu_stats.sh r3l_c5 --workload c5 --steps 5 --warmup 1 --configs none --no-parity || exit 1
