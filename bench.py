"""Headline benchmark: env-steps/s of the batched MI355X physics step.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one control step of the whole batch: one launch of the fused kernel
(n_sub_steps physics steps + task observation + reward) on actions that are
already resident in HBM, including the episode resets that fall inside the
timed region (cheetah: 200 settle steps per reset).  Workload at N=1 is the
configuration BASELINE.json's metric is quoted on: cheetah-run, batch 8192,
fp32.  N>1 shards independent envs, 8192 per GPU (weak scaling); the only
collective is an RCCL all-gather of episode returns on the reporting path,
after the timed region.

Rank 0 prints ONE JSON line with `roofline` (HIP events on the batch's own
stream) and, at N=1, `cpu_baseline` (the fp64 CPU oracle = "port"; libmujoco
itself is a closed binary that is not available here).
"""

import argparse
import json
import os
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
  sys.path.insert(0, _ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md, chip-level parameters


def algorithmic_bytes_per_env_step(info, real_size):
  """state in, state + observation + reward out."""
  reads = info.nq + info.nv + info.nu + info.nv          # qpos qvel ctrl warm
  writes = info.nq + info.nv + info.nv + info.nobs + 1   # + obs + reward
  return (reads + writes)*real_size


def measured_traffic(domain, task, batch, precision):
  """HBM bytes per launch from the committed PMC summary of this workload.

  rocprofv3 counters cannot be collected from inside the timed run; they come
  from separate `--pmc` passes of this same script (profiles/README.md) and
  are only reported for the exact workload they were measured on.
  """
  path = os.path.join(_ROOT, 'profiles', 'r01_pmc_%s_%s_b%d_%s.json'
                      % (domain, task, batch, precision))
  try:
    with open(path) as f:
      return json.load(f)['traffic_bytes_per_launch']
  except (OSError, ValueError, KeyError):
    return None


def usable_cores():
  """Host cores this process may actually use (affinity and cgroup quota)."""
  n = os.cpu_count() or 1
  try:
    n = min(n, len(os.sched_getaffinity(0)))
  except AttributeError:
    pass
  try:
    with open('/sys/fs/cgroup/cpu.max') as f:
      quota, period = f.read().split()
    if quota != 'max':
      n = min(n, max(1, int(float(quota)/float(period))))
  except (OSError, ValueError):
    pass
  return n


def cpu_baseline(domain, task, nsub, budget_s=12.0, gpu_batch=None):
  """Times the fp64 oracle (OpenMP over envs) on a bounded sample.

  With `gpu_batch` (the benchmarked batch) the same leg also reports the
  second half of BASELINE's metric, "qpos rel-err vs CPU mj_step": 64 states
  taken from the benchmarked batch are stepped by the benchmarked code object
  and by the oracle, teacher-forced, under the same controls.
  """
  from dm_control_amd import suite as _suite  # host logic only (model compile)
  from dm_control_amd.mjcf import compiler
  from oracle import oracle
  mod = getattr(_suite, domain)
  xml, assets = mod.get_model_and_assets()
  model = compiler.from_xml_string(xml, assets)
  # rebuild the oracle natively for this host (the in-tree .so is generic x86-64)
  lib = None
  try:
    out = os.path.join('/tmp', 'libmjoracle_native_%d.so' % os.getpid())
    oracle.build(force=True, cflags=['-O3', '-march=native', '-fopenmp', '-fPIC',
                                     '-std=c99', '-ffp-contract=off'], out=out)
    lib = oracle.load(out)
  except Exception:  # pylint: disable=broad-except
    lib = oracle.load()
  om = oracle.OracleModel(model, lib)
  cores = usable_cores()
  nenv = 64*cores
  datas = [oracle.OracleData(om) for _ in range(nenv)]
  rs = np.random.RandomState(0)
  lim = model.jnt_limited.astype(bool)
  for d in datas:
    if domain == 'cheetah':
      lo, hi = model.jnt_range[lim].T
      d.qpos[lim] = rs.uniform(lo, hi)
    d.step1()
  ctrl = rs.uniform(-1, 1, (nenv, model.nu))
  used = oracle.batch_step(om, datas, ctrl, nsub, cores)      # warm-up
  t0 = time.time()
  oracle.batch_step(om, datas, ctrl, nsub, cores)
  probe = max(time.time() - t0, 1e-4)
  reps = int(max(5, min(20000, budget_s/probe)))
  t0 = time.time()
  for r in range(reps):
    if r % 16 == 0:
      ctrl = rs.uniform(-1, 1, (nenv, model.nu))
    oracle.batch_step(om, datas, ctrl, nsub, cores)
  dt = time.time() - t0
  out = {
      'value': nenv*reps/dt, 'unit': 'env-steps/s', 'cores': int(used),
      'kind': 'port',
      'sample': '%s-%s: %d envs x %d control steps (%d physics steps each), '
                'fp64 C restatement of mj_step, OpenMP over envs, %.1f s'
                % (domain, task, nenv, reps, nsub, dt)}
  if gpu_batch is not None:
    out['qpos_rel_err'] = _rel_err_sample(gpu_batch, om, oracle, nsub)
  return out


def _rel_err_sample(gpu_batch, om, oracle, nsub, nenv=64, steps=20):
  """max|q_gpu - q_cpu| / max(1, max|q_cpu|) per env and control step."""
  from dm_control_amd import wrapper
  W = wrapper
  model = om.model
  q0 = gpu_batch.read(W.FIELD_QPOS).T[:nenv].astype(np.float64)
  v0 = gpu_batch.read(W.FIELD_QVEL).T[:nenv].astype(np.float64)
  nenv = len(q0)
  hb = W.HipBatch(gpu_batch.model, nenv)       # same code object as the bench
  if model.nu and gpu_batch.model.info.ntaskdata:
    hb.write(W.FIELD_TASKDATA, gpu_batch.read(W.FIELD_TASKDATA)[:, :nenv])
  datas = [oracle.OracleData(om) for _ in range(nenv)]
  for i, d in enumerate(datas):
    d.qpos[:] = q0[i]
    d.qvel[:] = v0[i]
    d.step1()
  rs = np.random.RandomState(1)
  errs = []
  for _ in range(steps):
    oq = np.array([d.qpos.copy() for d in datas])
    ov = np.array([d.qvel.copy() for d in datas])
    ow = np.array([d.qacc_warmstart.copy() for d in datas])
    hb.set_state(oq.T, ov.T, ow.T)
    ctrl = rs.uniform(-1, 1, (nenv, model.nu))
    hb.step_host(ctrl, nsub)
    q = hb.read(W.FIELD_QPOS).T.astype(np.float64)
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrl[i]
      for _ in range(nsub):
        d.physics_step()
    nq = np.array([d.qpos.copy() for d in datas])
    errs.append(np.abs(q - nq).max(axis=1)/np.maximum(1, np.abs(nq).max(axis=1)))
  hb.free()
  e = np.concatenate(errs)
  return {'median': float(np.median(e)), 'p99': float(np.percentile(e, 99)),
          'max': float(e.max()),
          'sample': 'teacher-forced, %d envs from the benchmarked batch x %d '
                    'control steps, same code object, U(-1,1) controls'
                    % (nenv, steps)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=1000)
  ap.add_argument('--warmup', type=int, default=50)
  ap.add_argument('--domain', default='cheetah')
  ap.add_argument('--task', default='run')
  ap.add_argument('--batch', type=int, default=8192, help='envs per GPU')
  ap.add_argument('--precision', default='f32', choices=['f32', 'f64'])
  ap.add_argument('--no-cpu-baseline', action='store_true')
  args = ap.parse_args()

  import torch
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit('launch with torch.distributed.run for --gpus > 1')
  # under torchrun (RANK set) the RCCL path is taken even for one rank, so the
  # process-group / all-gather code is exercised on a single-GPU box as well
  distributed = world > 1 or 'RANK' in os.environ
  if distributed:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    torch.cuda.set_device(local_rank)
    dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

  from dm_control_amd import suite, wrapper
  env = suite.load(args.domain, args.task, task_kwargs={'random': 1000 + rank},
                   environment_kwargs={'batch_size': args.batch,
                                       'device': local_rank,
                                       'precision': args.precision,
                                       'device_init': True})
  physics, task = env.physics, env.task
  batch = physics.batch
  info = batch.model.info
  nsub = env._n_sub_steps                      # pylint: disable=protected-access
  step_limit = env._step_limit                 # pylint: disable=protected-access
  tdtype = torch.float32 if args.precision == 'f32' else torch.float64
  dev = torch.device('cuda', local_rank)
  gen = torch.Generator(device=dev)
  gen.manual_seed(rank)
  # 16 steps of actions U(-1,1), resident in HBM, [t][env][nu]
  pool = torch.rand(16, args.batch, info.nu, device=dev, dtype=tdtype,
                    generator=gen)*2 - 1
  torch.cuda.synchronize(dev)

  state = {'count': 0, 'ev_ms': 0.0, 'ev_launches': 0, 'timing': False}

  def reset_episode():
    with physics.reset_context():
      task.initialize_episode(physics)
    state['count'] = 0

  def run(nsteps, timed):
    # One launch per control step; the launches of up to 16 steps are issued by
    # one C call (dmc_batch_step_n) so that the small models are not bound by
    # the interpreter's per-step overhead.
    done = 0
    while done < nsteps:
      if state['count'] >= step_limit:
        if timed and state['timing']:
          ms, n = batch.timer_stop()
          state['ev_ms'] += ms; state['ev_launches'] += n
          state['timing'] = False
        reset_episode()
      if timed and not state['timing']:
        batch.timer_start()
        state['timing'] = True
      chunk = int(min(16, nsteps - done, step_limit - state['count']))
      batch.step_device_n(pool.data_ptr(), 1, info.nu, args.batch*info.nu,
                          chunk, nsub)
      state['count'] += chunk
      done += chunk
    if timed and state['timing']:
      ms, n = batch.timer_stop()
      state['ev_ms'] += ms; state['ev_launches'] += n
      state['timing'] = False

  reset_episode()
  run(args.warmup, False)
  batch.sync()
  torch.cuda.synchronize(dev)
  if distributed:
    dist.barrier()
  t0 = time.perf_counter()
  run(args.steps, True)
  batch.sync()
  torch.cuda.synchronize(dev)
  elapsed = time.perf_counter() - t0
  if distributed:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    dist.barrier()

  # reporting path: all-gather of per-env episode returns over RCCL/xGMI
  returns = torch.from_numpy(
      batch.read(wrapper.FIELD_RETURN).astype(np.float32)).to(dev)
  if distributed:
    gathered = [torch.empty_like(returns) for _ in range(world)]
    dist.all_gather(gathered, returns)
    returns = torch.cat(gathered)
  warn = batch.read(wrapper.FIELD_WARN)

  if rank == 0:
    total_envs = args.batch*world
    value = total_envs*args.steps/elapsed
    kernel_ms = state['ev_ms']/max(1, state['ev_launches'])
    bytes_per_launch = algorithmic_bytes_per_env_step(
        info, info.real_size)*args.batch
    achieved = bytes_per_launch/(kernel_ms*1e-3)/1e9 if kernel_ms > 0 else 0.0
    line = {
        'metric': 'env-steps/sec', 'value': value, 'unit': 'env-steps/s',
        'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': elapsed/args.steps*1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': args.precision, 'data': 'synthetic',
        'config': {
            'workload': '%s-%s batch=%d per GPU (dm_control.suite, %d physics '
                        'substeps per env-step, episode resets included)'
                        % (args.domain, args.task, args.batch, nsub),
            'global_batch': total_envs, 'parallelism': 'env-shard x%d' % world,
            'actions': 'U(-1,1), device-resident'},
        'roofline': {
            'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': achieved/HBM_PEAK_GBS,
            'traffic': measured_traffic(args.domain, args.task, args.batch,
                                        args.precision),
            'traffic_unit': 'bytes/launch (rocprofv3 FETCH_SIZE+WRITE_SIZE, '
                            'separate passes, see profiles/)',
            'kernel': 'dmc_step',
            'kernel_ms_avg': kernel_ms,
            'algorithmic_bytes_per_launch': bytes_per_launch},
        'physics_steps_per_s': value*nsub,   # excl. the settle steps of episode resets
        'mean_episode_return': float(returns.mean().item()),
        'envs_with_warnings': int((warn != 0).sum()),
    }
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(args.domain, args.task, nsub,
                                          gpu_batch=batch)
    print(json.dumps(line))
  if distributed:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
