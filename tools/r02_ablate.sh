#!/bin/bash
# Solver ablation of the one-env-per-lane kernel, 1 and 4 wavefronts per workgroup.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python tools/gpu_ablate.py cheetah 8192 - W4 -DDMC_ABLATE_SOLVER W4,-DDMC_ABLATE_SOLVER -DDMC_SOLVER_PROFILE W4,-DDMC_SOLVER_PROFILE > gpurun_out/ablate.log 2>&1
cat gpurun_out/ablate.log
