#!/bin/bash
# SQ counters of one bench command (one pass, 8 SQ counters): scripts/gpu_pmc_sq.sh <tag> <bench args...>
tag=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $out/sq -o sq -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/sq.log 2>&1 || { tail -5 $out/sq.log; exit 1; }
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$out/sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if not k.startswith("k_"): continue
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
