#!/bin/bash
timeout -k 10 300 python tools/debug/pitch_bench_profile.py tiles 2>&1 | tail -12
