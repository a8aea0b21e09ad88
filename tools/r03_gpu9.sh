#!/bin/bash
O=gpurun_out/r03i; mkdir -p $O
for B in 256 512 768 1024 2048; do
timeout -k 10 300 python bench.py --domain soccer --task 2v2 --batch $B --steps 10 --warmup 2 --no-compliant-leg --no-cpu-baseline > $O/bench_soccer_b$B.json 2> $O/bench_soccer_b$B.err || { echo fail $B; tail -3 $O/bench_soccer_b$B.err; }
python - <<PY
import json
d=json.loads(open('$O/bench_soccer_b$B.json').read().strip().splitlines()[-1])
print('B=$B', '%.0f env-steps/s'%d['value'], '%.2f ms/step'%d['ms_per_step'], 'kernel %.2f'%d['roofline'].get('kernel_ms_avg'))
PY
done
