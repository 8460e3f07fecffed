#!/bin/bash
bash scripts/gpu_tests.sh 1000 || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_r3k.json 2> gpurun_out/bench_r3k.err || { tail -5 gpurun_out/bench_r3k.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r3k.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["parity"]["ok"] if isinstance(d.get("parity"), dict) else d.get("parity"))
print({k: (round(v.get("ms_per_step"), 4), v.get("parity", {}).get("ok")) for k, v in d.get("configs", {}).items()})
PY
timeout -k 10 300 python -u scripts/h2oai_harness.py --rows 10000000 > gpurun_out/harness_1e7.json 2> gpurun_out/harness_1e7.err; rc=$?
grep -v amdgpu.ids gpurun_out/harness_1e7.err | tail -10; exit $rc
