#!/bin/bash
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 300 python tools/debug/pitch_bench_profile.py solver > $O/pitch_bench_solver_profile.txt 2>&1; tail -10 $O/pitch_bench_solver_profile.txt
timeout -k 10 600 python bench.py --domain soccer --task 2v2 --batch 1024 --steps 20 --warmup 3 --no-compliant-leg --no-cpu-baseline > $O/bench_soccer_team.json 2> $O/bench_soccer_team.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03o/bench_soccer_team.json').read().strip().splitlines()[-1])
print('bench', d['value'], 'env-steps/s', d['ms_per_step'], 'ms/step kernel', d['roofline'].get('kernel_ms_avg'))
PY
