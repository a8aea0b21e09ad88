#!/bin/bash
# Round 3, GPU call 1: GPU test suite, smoke, the spill-hazard discriminator
# builds, the default bench line (now with the fp64 tolerance-compliant leg),
# and rocprofv3 evidence for the fp64 builds (kernel stats + PMC passes).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x -rA > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log
tail -3 $O/gputests.log
grep OBSERVED $O/gputests.log > $O/observed.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 && tail -2 $O/smoke.log || { echo smoke failed; tail -5 $O/smoke.log; }
timeout -k 10 600 python tools/spill_hazard/variants.py run > $O/spill_hazard_variants.txt 2>&1; cat $O/spill_hazard_variants.txt | cut -c1-400
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo bench failed; tail -5 $O/bench_default.err; exit 1; }
python3 -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1])
print('default', '%.3g' % d['value'], 'kernel', d['roofline']['kernel_ms_avg'], 'tol', d.get('tolerance'))
print('compliant', json.dumps(d.get('tolerance_compliant'))[:900])"
cd /tmp; export TMPDIR=/tmp
for spec in "cheetah run 8192 f64" "humanoid walk 1024 f64"; do
  set -- $spec; tag=$1_$2_b$3_$4
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o s -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof_$tag.json 2> $O/bench_under_rocprof_$tag.err || exit 1
  python3 $R/tools/rocprof_summary.py stats $O/stats_$tag > $O/stats_$tag.json
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
    name=$(echo $set | cut -d" " -f1)
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/pmc_${name}_$tag -o p -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --precision $4 --steps 60 --warmup 5 --no-cpu-baseline > $O/pmc_${name}_$tag.log 2>&1 || exit 1
  done
  mkdir -p $R/gpurun_out/measure
  python3 $R/tools/collect_counters.py $O $tag $O/bench_under_rocprof_$tag.json > $O/counters_$tag.log 2>&1
  cat $O/stats_$tag.json; tail -1 $O/counters_$tag.log | cut -c1-700
done
cp $R/gpurun_out/measure/counters_*.json $O/ 2>/dev/null
find $O -name "*.csv" -size +2M -delete
echo done
