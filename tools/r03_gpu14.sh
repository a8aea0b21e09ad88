#!/bin/bash
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 400 python tools/debug/soccer_env_wall.py 1024 > $O/soccer_env_wall.txt 2>&1; grep -v amdgpu.ids $O/soccer_env_wall.txt | head -50
