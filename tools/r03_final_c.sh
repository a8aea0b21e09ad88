#!/bin/bash
# Round 3 final evidence after the team build (every code object was re-keyed by
# the kernel source): GPU tests, smoke, rocprofv3 kernel stats + PMC passes for
# the headline code objects and the soccer pitch, default bench line, soccer line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O $R/gpurun_out/measure
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q -x -rA > $O/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/gputests.log
tail -2 $O/gputests.log; grep OBSERVED $O/gputests.log > $O/observed.txt
[ $rc -eq 0 ] || { grep -E "^E|FAILED" $O/gputests.log | head -20; exit 1; }
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && grep "smoke ok" $O/smoke.log || { echo smoke failed; tail -5 $O/smoke.log; exit 1; }
cd /tmp; export TMPDIR=/tmp
for spec in "cheetah run 8192 f32 300" "soccer 2v2 1024 f32 30" "humanoid walk 1024 f32 300" "cheetah run 8192 f64 300" "humanoid walk 1024 f64 300"; do
  set -- $spec; tag=$1_$2_b$3_$4
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o s -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps $5 --warmup 5 --no-cpu-baseline --no-compliant-leg > $O/bench_under_rocprof_$tag.json 2> $O/bench_under_rocprof_$tag.err || { tail -5 $O/bench_under_rocprof_$tag.err; exit 1; }
  python3 $R/tools/rocprof_summary.py stats $O/stats_$tag > $O/stats_$tag.json
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    name=$(echo $set | cut -d" " -f1)
    timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${name}_$tag -o p -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps $(( $5 / 5 )) --warmup 3 --no-cpu-baseline --no-compliant-leg > $O/pmc_${name}_$tag.log 2>&1 || { tail -5 $O/pmc_${name}_$tag.log; exit 1; }
  done
  python3 $R/tools/collect_counters.py $O $tag $O/bench_under_rocprof_$tag.json > $O/counters_$tag.log 2>&1
  cat $O/stats_$tag.json; tail -1 $O/counters_$tag.log | cut -c1-400
done
cp $R/gpurun_out/measure/counters_*.json $O/ 2>/dev/null
cp $R/gpurun_out/measure/counters_*.json $R/profiles/ 2>/dev/null   # the lines below carry them
cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
cut -c1-400 $O/bench_default.json
DMC_BENCH_PROGRESS=1 timeout -k 10 600 python bench.py --domain soccer --task 2v2 --batch 1024 --steps 30 --warmup 3 > $O/bench_soccer_2v2_b1024_f32.json 2> $O/bench_soccer_2v2_b1024_f32.err || { echo soccer bench failed; tail -8 $O/bench_soccer_2v2_b1024_f32.err; }
cut -c1-300 $O/bench_soccer_2v2_b1024_f32.json
for q in loud quiet; do timeout -k 10 400 python tools/debug/pitch_profile.py $q --team > $O/pitch_stage_profile_team_$q.txt 2>&1; done
timeout -k 10 300 python tools/debug/pitch_bench_profile.py solver > $O/pitch_bench_solver_profile.txt 2>&1
timeout -k 10 300 python tools/debug/pitch_bench_profile.py stage > $O/pitch_bench_stage_profile.txt 2>&1
tail -8 $O/pitch_bench_solver_profile.txt; tail -10 $O/pitch_bench_stage_profile.txt
find $O -name "*.csv" -size +2M -delete
echo done
