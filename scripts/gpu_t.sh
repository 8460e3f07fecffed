#!/bin/bash
mkdir -p gpurun_out
for w in 2 3; do
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29540 + w)) scripts/rehearse_ranks.py > gpurun_out/r3d_rehearse_${w}ranks_one_gpu.log 2>&1 || { tail -20 gpurun_out/r3d_rehearse_${w}ranks_one_gpu.log; exit 1; }
  grep "ok: dense beyond" gpurun_out/r3d_rehearse_${w}ranks_one_gpu.log
done
