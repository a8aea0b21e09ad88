#!/bin/bash
# gpurun -- 'bash tools/r02_sweep.sh'
mkdir -p gpurun_out
: > gpurun_out/group_sweep.log
for m in cheetah hopper walker humanoid; do
  timeout -k 10 400 python tools/gpu_group_sweep.py $m >> gpurun_out/group_sweep.log 2>&1 || exit 1
done
cat gpurun_out/group_sweep.log
