#!/bin/bash
# Round 3, GPU call 4: the soccer bench leg with stage markers (it died with a
# host heap error in call 3), then the semi-rolled fp64 cheetah build.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
DMC_BENCH_PROGRESS=1 MALLOC_CHECK_=3 timeout -k 10 600 python -X faulthandler bench.py --domain soccer --task 2v2 --batch 1024 --steps 6 --warmup 2 --no-compliant-leg > $O/bench_soccer_2v2_b1024_f32.json 2> $O/bench_soccer_2v2_b1024_f32.err; echo "soccer rc=$?"
tail -25 $O/bench_soccer_2v2_b1024_f32.err; cut -c1-600 $O/bench_soccer_2v2_b1024_f32.json
timeout -k 10 300 python bench.py --precision f64 --no-compliant-leg --steps 300 > $O/bench_cheetah_run_b8192_f64.json 2> $O/bench_cheetah_f64.err || exit 1
python3 -c "
import json
d=json.loads([l for l in open('$O/bench_cheetah_run_b8192_f64.json') if l.startswith('{')][-1])
print('cheetah f64', '%.4g' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms_avg'], d['config']['code_object'], d['config']['kernel_shape'])"
echo done
