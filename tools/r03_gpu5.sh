#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
for q in loud quiet; do timeout -k 10 400 python tools/debug/pitch_profile.py $q > $O/profile_$q.log 2>&1; grep -v amdgpu.ids $O/profile_$q.log | cut -c1-150; done
echo done
