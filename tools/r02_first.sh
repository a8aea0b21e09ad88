#!/bin/bash
# Round-2 first GPU pass: LDS atomic cost, solver ablation, GPU tests, precision
# study, default bench line.  Steps are chained with && (a killed step ends the call).
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
./tools/micro/lds_atomic_bench > gpurun_out/lds_atomic.log 2>&1 &&
timeout -k 10 400 python tools/gpu_ablate.py cheetah 8192 - -DDMC_ROWPAR=0 -DDMC_ABLATE_SOLVER -DDMC_SOLVER_PROFILE > gpurun_out/ablate.log 2>&1 &&
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "pytest killed at its limit"; exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/bench.log 2>&1 &&
timeout -k 10 900 python tools/gpu_precision_study.py > gpurun_out/precision.log 2>&1
tail -3 gpurun_out/bench.log
