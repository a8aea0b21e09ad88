set -e
# One gpurun call: GPU tests, rocprofv3 stats + PMC for the two headline workloads, bench lines
# for every domain.  Outputs under gpurun_out/final/ (copied into profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd $R && timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
cd /tmp; export TMPDIR=/tmp
for spec in "cheetah run 8192" "humanoid walk 8192"; do
  set -- $spec; tag=$1_$2_b$3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -o s -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --steps 300 --warmup 20 --no-cpu-baseline > $O/bench_under_rocprof_$tag.log 2>&1
  python3 $R/tools/rocprof_summary.py stats $O/stats_$tag > $O/stats_$tag.json
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$tag -o p -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --steps 60 --warmup 5 --no-cpu-baseline > $O/pmc_${c}_$tag.log 2>&1
    python3 $R/tools/rocprof_summary.py pmc $O/pmc_${c}_$tag > $O/pmc_${c}_$tag.json
  done
done
cd $R
for spec in "cheetah run 8192" "cheetah run 65536" "humanoid walk 8192" "humanoid walk 1024" "walker walk 8192" "cartpole swingup 4096" "hopper hop 8192" "hopper hop 4096" "acrobot swingup 8192" "reacher easy 8192" "point_mass easy 8192" "pendulum swingup 8192"; do
  set -- $spec
  timeout -k 10 300 python bench.py --domain $1 --task $2 --batch $3 > $O/bench_$1_$2_b$3.json 2> $O/bench_$1_$2_b$3.err
  tail -c 400 $O/bench_$1_$2_b$3.json | head -c 200; echo
done
cat $O/stats_*.json $O/pmc_*.json
find $O -name "*.csv" -size +2M -delete
