#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python tools/gpu_ablate.py cheetah 8192 - -fgpu-flush-denormals-to-zero -fapprox-func > gpurun_out/flagtest.log 2>&1
cat gpurun_out/flagtest.log
