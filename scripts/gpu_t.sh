#!/bin/bash
# round 3: the rehearsal logs of 2 and 3 ranks sharing one GPU, regenerated with the round's library
mkdir -p gpurun_out
for w in 2 3; do
  timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port $((29540 + w)) scripts/rehearse_ranks.py > gpurun_out/r3c_rehearse_${w}ranks_one_gpu.log 2>&1 || { tail -20 gpurun_out/r3c_rehearse_${w}ranks_one_gpu.log; exit 1; }
  grep -c "ok:" gpurun_out/r3c_rehearse_${w}ranks_one_gpu.log
done
