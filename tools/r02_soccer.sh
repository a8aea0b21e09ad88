#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_soccer_model.py -m gpu -x -q -s > gpurun_out/pytest_soccer.log 2>&1
tail -15 gpurun_out/pytest_soccer.log
