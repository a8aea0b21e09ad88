#!/bin/bash
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 300 python tools/debug/pitch_bench_profile.py stage > $O/pitch_bench_stage_profile.txt 2>&1; tail -10 $O/pitch_bench_stage_profile.txt
