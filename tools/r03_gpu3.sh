#!/bin/bash
# Round 3, GPU call 3: GPU tests (incl. box narrowphase), default bench line,
# rocprofv3 kernel stats + PMC passes for the headline code objects (fp32 and
# fp64; cheetah-run 8192 and the humanoid-walk 1024 shard), then bench lines of
# the other configurations.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O $R/gpurun_out/measure
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x -rA > $O/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $O/gputests.log
tail -3 $O/gputests.log; grep OBSERVED $O/gputests.log > $O/observed.txt
[ $rc -eq 0 ] || { grep -E "^E|FAILED" $O/gputests.log | head -20; exit 1; }
cd /tmp; export TMPDIR=/tmp
for spec in "cheetah run 8192 f32" "humanoid walk 1024 f32" "cheetah run 8192 f64" "humanoid walk 1024 f64"; do
  set -- $spec; tag=$1_$2_b$3_$4
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o s -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof_$tag.json 2> $O/bench_under_rocprof_$tag.err || exit 1
  python3 $R/tools/rocprof_summary.py stats $O/stats_$tag > $O/stats_$tag.json
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    name=$(echo $set | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${name}_$tag -o p -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps 60 --warmup 5 --no-cpu-baseline > $O/pmc_${name}_$tag.log 2>&1 || exit 1
  done
  python3 $R/tools/collect_counters.py $O $tag $O/bench_under_rocprof_$tag.json > $O/counters_$tag.log 2>&1
  cat $O/stats_$tag.json; tail -1 $O/counters_$tag.log | cut -c1-500
done
cp $R/gpurun_out/measure/counters_*.json $O/ 2>/dev/null
cp $R/gpurun_out/measure/counters_*.json $R/profiles/ 2>/dev/null   # so that the bench lines below carry them
cd $R
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
for spec in "cheetah run 8192 f64" "cheetah run 65536 f32" "cheetah run 262144 f32" "cartpole swingup 4096 f32" "cartpole swingup 4096 f64" "cartpole swingup 4096 mixed" "humanoid walk 1024 f32" "humanoid walk 1024 f64" "humanoid walk 8192 f32" "walker walk 8192 f32" "walker walk 8192 f64" "hopper hop 8192 f32" "hopper hop 8192 f64"; do
  set -- $spec
  timeout -k 10 400 python bench.py --domain $1 --task $2 --batch $3 --precision $4 --no-compliant-leg > $O/bench_$1_$2_b$3_$4.json 2> $O/bench_$1_$2_b$3_$4.err || exit 1
done
timeout -k 10 900 python bench.py --domain soccer --task 2v2 --batch 1024 --steps 10 --warmup 2 --no-compliant-leg > $O/bench_soccer_2v2_b1024_f32.json 2> $O/bench_soccer_2v2_b1024_f32.err || { echo soccer bench failed; tail -5 $O/bench_soccer_2v2_b1024_f32.err; }
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
  if 'under_rocprof' in f: continue
  d=json.loads([l for l in open(f) if l.startswith('{')][-1])
  print(f.split('/')[-1], '%.4g env-steps/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms_avg'], d['config']['kernel_shape'][:28], 'cpu %.3g' % d.get('cpu_baseline',{}).get('value',0), 'tol', (d.get('tolerance') or {}).get('share_of_envs'))
PY
find $O -name "*.csv" -size +2M -delete
echo done
