#!/bin/bash
# gpurun -- 'bash tools/r02_variants.sh "<flags1>" "<flags2>" ...'
mkdir -p gpurun_out
timeout -k 10 600 python tools/gpu_coop_variants.py "$@" > gpurun_out/variants.log 2>&1 && \
DMC_EXTRA_FLAGS=-DDMC_COOP_PROFILE timeout -k 10 300 python tools/gpu_coop_prof.py humanoid walk 1024 > gpurun_out/coop_prof_1024.log 2>&1
cat gpurun_out/variants.log gpurun_out/coop_prof_1024.log
