#!/bin/bash
# gpurun -- 'bash tools/r02_final_check.sh'   what the driver runs at round end
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/smoke.log 2>&1 || { tail -20 gpurun_out/smoke.log; exit 1; }
tail -2 gpurun_out/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/bench_default.log 2>&1 || { tail -20 gpurun_out/bench_default.log; exit 1; }
tail -1 gpurun_out/bench_default.log | cut -c1-300
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/bench_torchrun1.log 2>&1 || { tail -20 gpurun_out/bench_torchrun1.log; exit 1; }
tail -1 gpurun_out/bench_torchrun1.log | cut -c1-300
timeout -k 10 300 python bench.py --domain humanoid --task walk --global-batch 8192 --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_h8192.log 2>&1 || { tail -20 gpurun_out/bench_h8192.log; exit 1; }
tail -1 gpurun_out/bench_h8192.log | cut -c1-300
