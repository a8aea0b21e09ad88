#!/bin/bash
# Round 3: the soccer evidence again after `soccer.load` got its contact capacity of 40 per
# player (a new code object): GPU tests of the soccer files, rocprofv3 stats + PMC passes,
# the bench line, stage / solver profiles on the bench's states, the soak.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O $R/gpurun_out/measure
cd $R
timeout -k 10 900 python -m pytest tests/test_soccer_model.py tests/test_soccer_task.py -m gpu -q -x -rA > $O/gputests_soccer.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/gputests_soccer.log
[ $rc -eq 0 ] || { grep -E "^E|FAILED" $O/gputests_soccer.log | head -20; exit 1; }
timeout -k 10 800 python tools/debug/soccer_soak.py 512 600 > $O/soccer_soak.txt 2>&1; grep -v amdgpu.ids $O/soccer_soak.txt | tail -5
cd /tmp; export TMPDIR=/tmp
for spec in "soccer 2v2 1024 f32 30"; do
  set -- $spec; tag=$1_$2_b$3_$4
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o s -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps $5 --warmup 5 --no-cpu-baseline --no-compliant-leg > $O/bench_under_rocprof_$tag.json 2> $O/bench_under_rocprof_$tag.err || { tail -5 $O/bench_under_rocprof_$tag.err; exit 1; }
  python3 $R/tools/rocprof_summary.py stats $O/stats_$tag > $O/stats_$tag.json
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    name=$(echo $set | cut -d" " -f1)
    timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${name}_$tag -o p -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps $(( $5 / 5 )) --warmup 3 --no-cpu-baseline --no-compliant-leg > $O/pmc_${name}_$tag.log 2>&1 || { tail -5 $O/pmc_${name}_$tag.log; exit 1; }
  done
  python3 $R/tools/collect_counters.py $O $tag $O/bench_under_rocprof_$tag.json > $O/counters_$tag.log 2>&1
  cat $O/stats_$tag.json; tail -1 $O/counters_$tag.log | cut -c1-300
done
cp $R/gpurun_out/measure/counters_*.json $O/ 2>/dev/null
cp $R/gpurun_out/measure/counters_*.json $R/profiles/ 2>/dev/null
cd $R
DMC_BENCH_PROGRESS=1 timeout -k 10 600 python bench.py --domain soccer --task 2v2 --batch 1024 --steps 30 --warmup 3 > $O/bench_soccer_2v2_b1024_f32.json 2> $O/bench_soccer_2v2_b1024_f32.err || { echo soccer bench failed; tail -8 $O/bench_soccer_2v2_b1024_f32.err; }
cut -c1-300 $O/bench_soccer_2v2_b1024_f32.json
timeout -k 10 300 python tools/debug/pitch_bench_profile.py solver > $O/pitch_bench_solver_profile.txt 2>&1
timeout -k 10 300 python tools/debug/pitch_bench_profile.py stage > $O/pitch_bench_stage_profile.txt 2>&1
tail -8 $O/pitch_bench_solver_profile.txt; tail -10 $O/pitch_bench_stage_profile.txt
find $O -name "*.csv" -size +2M -delete
echo done
