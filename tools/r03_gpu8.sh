#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 300 python tools/debug/pitch_profile.py loud --team --solver > $O/pitch_solver_profile_team.txt 2>&1; cat $O/pitch_solver_profile_team.txt | tail -12
