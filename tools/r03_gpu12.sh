#!/bin/bash
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_soccer_model.py -m gpu -x -q -s -k "other_team_sizes" > $O/team_sizes.log 2>&1; rc=$?
echo "pytest rc=$rc"; grep -E "OBSERVED|passed|failed|Error|assert" $O/team_sizes.log | tail -12
