#!/bin/bash
# gpurun -- 'bash tools/r02_coop_prof.sh'
mkdir -p gpurun_out
export DMC_EXTRA_FLAGS=-DDMC_COOP_PROFILE
timeout -k 10 500 python tools/gpu_coop_prof.py humanoid walk 1024 > gpurun_out/coop_prof_1024.log 2>&1 && \
timeout -k 10 300 python tools/gpu_coop_prof.py humanoid walk 8192 > gpurun_out/coop_prof_8192.log 2>&1
tail -40 gpurun_out/coop_prof_1024.log
