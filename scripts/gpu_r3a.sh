#!/bin/bash
# round 3, first GPU pass: new tests (exact fold, bench launcher), then the default bench line with its extra configs
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_exact_fold_gpu.py tests/test_bench_launcher_gpu.py -x -q --timeout 400 --timeout-method thread > gpurun_out/r3a_tests.log 2>&1
rc=$?; tail -15 gpurun_out/r3a_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 580 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err
rc=$?; tail -3 gpurun_out/r3a_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3a_bench.json"))
print("q3", d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["all_kernels_frac"], d["parity"]["ok"])
for k,v in d.get("configs",{}).items():
    print(k, {a: (round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("ms_per_step","strategy","kernel","kernel_frac","all_kernels_frac","step_frac","wall_s","skipped","error")}, (v.get("parity") or {}).get("ok"))
PY
exit $rc
