#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python tools/gpu_ablate.py cheetah 8192 - -fgpu-flush-denormals-to-zero - -fgpu-flush-denormals-to-zero > gpurun_out/flagtest.log 2>&1
cat gpurun_out/flagtest.log
timeout -k 10 500 python tools/gpu_coop_variants.py "" "-fgpu-flush-denormals-to-zero" > gpurun_out/flagtest2.log 2>&1
cat gpurun_out/flagtest2.log
