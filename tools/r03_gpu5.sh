#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
for leg in twin teacher free timing native; do
  MALLOC_CHECK_=3 timeout -k 10 300 python -X faulthandler tools/debug/soccer_bench_bisect.py $leg > $O/$leg.log 2>&1; echo "$leg rc=$?"; grep -v amdgpu.ids $O/$leg.log | tail -4 | cut -c1-200
done
echo done
