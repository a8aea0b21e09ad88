set -e
# SQ counter passes for one bench workload (arguments: domain task batch); summaries under gpurun_out/sq/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/sq; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
tag=$1_$2_b$3
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  name=$(echo $set | cut -d" " -f1)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/${name}_$tag -o p -- python3 $R/bench.py --domain $1 --task $2 --batch $3 --steps 60 --warmup 5 --no-cpu-baseline > $O/${name}_$tag.log 2>&1
  python3 $R/tools/rocprof_summary.py pmc $O/${name}_$tag > $O/${name}_$tag.json
  cat $O/${name}_$tag.json
done
find $O -name "*.csv" -size +2M -delete
