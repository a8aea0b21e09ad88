#!/bin/bash
# Round 3, GPU call 2: GPU tests, planar row-merge ablation, fp64 kernel-shape
# sweep, spill-hazard forensics (safe variants first, the allocation-changing
# ones last, one process each, stop at the first failure).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -x -rA > $O/gputests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gputests.log
tail -3 $O/gputests.log; grep OBSERVED $O/gputests.log > $O/observed.txt
for spec in "cheetah run" "hopper hop" "walker walk"; do
  set -- $spec
  timeout -k 10 300 python bench.py --domain $1 --task $2 --no-cpu-baseline --steps 600 > $O/bench_$1_merge.json 2> $O/bench_$1_merge.err || exit 1
  DMC_EXTRA_FLAGS=-DDMC_NO_PLANAR_MERGE timeout -k 10 300 python bench.py --domain $1 --task $2 --no-cpu-baseline --steps 600 > $O/bench_$1_nomerge.json 2> $O/bench_$1_nomerge.err || exit 1
  python3 -c "
import json
for t in ('merge','nomerge'):
  d=json.loads([l for l in open('$O/bench_$1_%s.json'%t) if l.startswith('{')][-1])
  print('$1', t, '%.4g env-steps/s' % d['value'], 'kernel %.5f ms' % d['roofline']['kernel_ms_avg'], d['config']['code_object'])"
done
for n in cheetah walker hopper; do
  DMC_SWEEP_PRECISION=f64 timeout -k 10 300 python tools/gpu_group_sweep.py $n 1024 4096 8192 32768 >> $O/f64_shape_sweep.txt 2>&1 || exit 1
done
cat $O/f64_shape_sweep.txt | grep -v amdgpu.ids
timeout -k 10 300 python tools/spill_hazard/variants.py run ifchain "=select" forcezero snop > $O/spill_hazard_safe.txt 2>&1 || { tail -5 $O/spill_hazard_safe.txt; exit 1; }
cut -c1-330 $O/spill_hazard_safe.txt | grep -v amdgpu.ids
cp gpurun_out/spill_hazard_dump.npz $O/spill_hazard_dump_safe.npz
for v in prealloc disable-ssc dce-in-ra opt-exec-mask rewrite-partial vgpr-to-agpr; do
  timeout -k 10 120 python tools/spill_hazard/variants.py run "$v" > $O/spill_hazard_$v.txt 2>&1 || { echo "variant $v failed"; tail -3 $O/spill_hazard_$v.txt; exit 1; }
  grep "select" $O/spill_hazard_$v.txt | cut -c1-330
done
echo done
