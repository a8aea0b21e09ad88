#!/bin/bash
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 800 python tools/debug/soccer_soak.py 512 600 200 > $O/soccer_soak.txt 2>&1; grep -v amdgpu.ids $O/soccer_soak.txt | tail -8
