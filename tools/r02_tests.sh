#!/bin/bash
# gpurun -- 'bash tools/r02_tests.sh'   GPU tests, then two short bench lines
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -5 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then grep -n "Error\|error\|FAILED\|assert" gpurun_out/pytest_gpu.log | tail -20; exit $rc; fi
timeout -k 10 300 python bench.py --domain humanoid --task walk --batch 1024 > gpurun_out/bench_h1024.log 2>&1 &&
tail -1 gpurun_out/bench_h1024.log | cut -c1-700
