#!/bin/bash
# GPU pass: solver ablation, GPU tests, bench.
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 500 python tools/gpu_ablate.py cheetah 8192 - -DDMC_ABLATE_SOLVER -DDMC_SOLVER_PROFILE > gpurun_out/ablate.log 2>&1 &&
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed at its limit"; exit $rc; fi
timeout -k 10 300 python tools/gpu_primitives_unrolled.py > gpurun_out/primitives_unrolled.log 2>&1
cat gpurun_out/primitives_unrolled.log
timeout -k 10 300 python bench.py > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/bench.log | cut -c1-600
