#!/bin/bash
# round 3, call 6: the team build of the 2v2 pitch -- parity tests, stage profile, bench
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_soccer_model.py -m gpu -x -q -s -k "pitch or environment_plays" > $O/pitch_tests.log 2>&1; rc=$?
echo "pytest rc=$rc"; grep -E "OBSERVED|passed|failed|Error|error" $O/pitch_tests.log | tail -20
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/debug/pitch_profile.py loud --team > $O/pitch_profile_team_loud.txt 2>&1 && cat $O/pitch_profile_team_loud.txt &&
timeout -k 10 300 python tools/debug/pitch_profile.py quiet --team > $O/pitch_profile_team_quiet.txt 2>&1 && cat $O/pitch_profile_team_quiet.txt &&
timeout -k 10 300 python tools/debug/pitch_profile.py loud --team --f64 > $O/pitch_profile_team_loud_f64.txt 2>&1 && cat $O/pitch_profile_team_loud_f64.txt &&
timeout -k 10 600 python bench.py --domain soccer --task 2v2 --batch 1024 --steps 20 --warmup 3 > $O/bench_soccer_team.json 2> $O/bench_soccer_team.err; echo bench rc=$?; tail -c 1500 $O/bench_soccer_team.json
