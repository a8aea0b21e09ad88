#!/bin/bash
# round 3, call 7: team build v2 (solver vectors / geom mirror in LDS, chain-based contact rows)
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_soccer_model.py -m gpu -x -q -s -k "pitch and team" > $O/pitch_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; grep -E "OBSERVED|passed|failed|Error|error" $O/pitch_tests.log | tail -20
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/debug/pitch_profile.py loud --team > $O/pitch_profile_team_loud.txt 2>&1 && cat $O/pitch_profile_team_loud.txt &&
timeout -k 10 300 python tools/debug/pitch_profile.py loud --team --f64 > $O/pitch_profile_team_loud_f64.txt 2>&1 && cat $O/pitch_profile_team_loud_f64.txt
timeout -k 10 600 python bench.py --domain soccer --task 2v2 --batch 1024 --steps 20 --warmup 3 --no-compliant-leg > $O/bench_soccer_team.json 2> $O/bench_soccer_team.err; echo bench rc=$?; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03g/bench_soccer_team.json').read().strip().splitlines()[-1])
print('bench', d['value'], 'env-steps/s', d['ms_per_step'], 'ms/step kernel', d['roofline'].get('kernel_ms_avg'), 'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('qpos_rel_err',{}).get('teacher_forced'))
PY

timeout -k 10 300 python tools/debug/pitch_bench_profile.py stage > $O/pitch_bench_stage_profile.txt 2>&1; tail -10 $O/pitch_bench_stage_profile.txt
