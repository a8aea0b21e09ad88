#!/bin/bash
# Round 3 final evidence, call B: bench lines of the other configurations, the
# soccer pitch (bench line, stage profile), launcher rehearsal at one rank.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
for spec in "cheetah run 8192 f64" "cheetah run 65536 f32" "cheetah run 262144 f32" "cartpole swingup 4096 f32" "cartpole swingup 4096 f64" "cartpole swingup 4096 mixed" "humanoid walk 1024 f32" "humanoid walk 1024 f64" "humanoid walk 8192 f32" "walker walk 8192 f32" "walker walk 8192 f64" "hopper hop 8192 f32" "hopper hop 8192 f64" "acrobot swingup 8192 f32" "pendulum swingup 8192 f32"; do
  set -- $spec
  timeout -k 10 400 python bench.py --domain $1 --task $2 --batch $3 --precision $4 --no-compliant-leg > $O/bench_$1_$2_b$3_$4.json 2> $O/bench_$1_$2_b$3_$4.err || { echo "$spec failed"; tail -3 $O/bench_$1_$2_b$3_$4.err; exit 1; }
done
DMC_BENCH_PROGRESS=1 timeout -k 10 600 python bench.py --domain soccer --task 2v2 --batch 1024 --steps 8 --warmup 2 --no-compliant-leg > $O/bench_soccer_2v2_b1024_f32.json 2> $O/bench_soccer_2v2_b1024_f32.err || { echo soccer bench failed; tail -8 $O/bench_soccer_2v2_b1024_f32.err; }
for q in loud quiet; do timeout -k 10 400 python tools/debug/pitch_profile.py $q > $O/pitch_stage_profile_$q.txt 2>&1; done
# the launcher path on the one-GPU box: N = 1 under the script's own launcher env
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 timeout -k 10 300 python bench.py --gpus 1 --steps 200 --no-cpu-baseline > $O/bench_one_rank_rccl.json 2> $O/bench_one_rank_rccl.err || { echo one-rank failed; tail -3 $O/bench_one_rank_rccl.err; }
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
  try:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
  except Exception as e:
    print(f.split('/')[-1], 'NO LINE'); continue
  cb=d.get('cpu_baseline',{})
  print(f.split('/')[-1], '%.4g env-steps/s' % d['value'], 'kernel %.4f ms' % d['roofline']['kernel_ms_avg'], d['config']['kernel_shape'][:26], 'cpu %.3g' % cb.get('value',0), 'tol', (d.get('tolerance') or {}).get('share_of_envs'), 'warn', d.get('envs_with_warnings'))
PY
grep -h "B=\|mean" $O/pitch_stage_profile_loud.txt | cut -c1-120
echo done
