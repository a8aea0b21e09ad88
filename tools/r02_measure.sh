#!/bin/bash
# Round-2 evidence in one gpurun call: rocprofv3 --kernel-trace --stats and the
# PMC passes (HBM bytes, VALU instructions / busy cycles) for the headline
# workloads, then bench lines for every configuration.  Outputs under
# gpurun_out/measure/ (copied into profiles/ afterwards).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/measure; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for spec in "cheetah run 8192" "humanoid walk 1024"; do
  set -- $spec; tag=$1_$2_b$3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o s -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof_$tag.json 2> $O/bench_under_rocprof_$tag.err || exit 1
  python3 $R/tools/rocprof_summary.py stats $O/stats_$tag > $O/stats_$tag.json
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    name=$(echo $set | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${name}_$tag -o p -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --steps 60 --warmup 5 --no-cpu-baseline > $O/pmc_${name}_$tag.log 2>&1 || exit 1
  done
  python3 $R/tools/collect_counters.py $O $tag $O/bench_under_rocprof_$tag.json > $O/counters_$tag.log 2>&1
  cat $O/stats_$tag.json; tail -1 $O/counters_$tag.log | cut -c1-600
done
cd $R
cp $O/counters_*_b*.json $R/profiles/ 2>/dev/null   # so that the bench lines below carry them
for spec in "cheetah run 8192 f32" "cheetah run 8192 f64" "cheetah run 65536 f32" "cheetah run 262144 f32" "cartpole swingup 4096 f32" "cartpole swingup 4096 f64" "cartpole swingup 4096 mixed" "humanoid walk 1024 f32" "humanoid walk 1024 f64" "humanoid walk 8192 f32" "walker walk 8192 f32" "hopper hop 8192 f32" "acrobot swingup 8192 f32" "reacher easy 8192 f32" "point_mass easy 8192 f32" "pendulum swingup 8192 f32"; do
  set -- $spec
  timeout -k 10 400 python bench.py --domain $1 --task $2 --batch $3 --precision $4 > $O/bench_$1_$2_b$3_$4.json 2> $O/bench_$1_$2_b$3_$4.err || exit 1
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/bench_$1_$2_b$3_$4.json') if l.startswith('{')][-1])
print('$1 $2 $3 $4', '%.3g env-steps/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms_avg'], 'cpu %.3g' % d['cpu_baseline']['value'], d['cpu_baseline']['qpos_rel_err']['free_run'].get('step_1000'))"
done
find $O -name "*.csv" -size +2M -delete
