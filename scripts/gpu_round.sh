#!/bin/bash
# One measurement round on the GPU box: parity tests, the default bench line, rocprofv3 kernel stats of the same
# command, and the two HBM-traffic counter passes.  Outputs land under gpurun_out/<tag>/.
tag=${1:-r1}
out=gpurun_out/$tag; mkdir -p $out
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $out/pytest_gpu.log 2>&1 || { tail -20 $out/pytest_gpu.log; exit 1; }
tail -1 $out/pytest_gpu.log
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cat $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -o stats -- python $R/bench.py --no-cpu-baseline > $R/$out/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/pmc_fetch -o fetch -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$out/pmc_fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$out/pmc_write -o write -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$out/pmc_write.log 2>&1 || exit 1
cd $R
echo done
