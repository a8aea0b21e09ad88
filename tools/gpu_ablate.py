"""Scratch: ablation timing of the one-lane cheetah kernel (DMC_ABLATE_* builds)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
from dm_control_amd import build, suite, wrapper as W
name = sys.argv[1] if len(sys.argv) > 1 else 'cheetah'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
model = helpers.load_model(name)
nsub = {'cheetah': 1, 'walker': 10, 'hopper': 4, 'cartpole': 1}[name]
# realistic states: run the real env for a while, then copy its state
env = suite.load(name, {'cheetah': 'run', 'walker': 'walk', 'hopper': 'hop', 'cartpole': 'swingup'}[name], task_kwargs={'random': 1},
                 environment_kwargs={'batch_size': B, 'device_init': True, 'build_mode': 'auto'})
env.reset()
rs = np.random.RandomState(0)
for t in range(100):
  env.physics.set_control(rs.uniform(-1, 1, (B, model.nu))); env.physics.step(nsub, check=False)
q = env.physics.batch.read(W.FIELD_QPOS); v = env.physics.batch.read(W.FIELD_QVEL); w = env.physics.batch.read(W.FIELD_WARMSTART)
acts = [rs.uniform(-1, 1, (B, model.nu)) for _ in range(4)]
FLAGSETS = [(), ('-DDMC_ABLATE_OBS',), ('-DDMC_ABLATE_SOLVER',), ('-DDMC_ABLATE_CONTACT',),
            ('-DDMC_ABLATE_OBS', '-DDMC_ABLATE_SOLVER', '-DDMC_ABLATE_CONTACT')]
if len(sys.argv) > 3:
  FLAGSETS = [tuple(f.split(',')) if f != '-' else () for f in sys.argv[3:]]
for flags in FLAGSETS:
  path = build.build_model(model, helpers.TASKS[name], 'f32', extra_flags=flags)
  hm = W.HipModel(path); hb = W.HipBatch(hm, B)
  ts = []
  for rep in range(15):
    hb.set_state(q, v, w)
    hb.step_host(acts[0], nsub)
    hb.set_state(q, v, w)
    hb.sync(); hb.timer_start()
    hb.step_host(None, nsub)
    ms, n = hb.timer_stop(); ts.append(ms/n)
  st = hb.read(W.FIELD_STATS)
  if '-DDMC_SOLVER_PROFILE' in flags:
    pr = hb.read(W.FIELD_OBS)[:, :6]
    wave = pr[:, :5].reshape(-1, 64, 5).max(axis=1)     # a wave takes as long as its slowest lane
    tot = pr[:, :5].sum(axis=1).reshape(-1, 64).max(axis=1)
    print('solver phases per wave, us (hess+grad, chol+solve, Mv+Jv, line search, update):',
          np.round(wave.mean(axis=0)/100, 2), 'sum %.2f' % (wave.mean(axis=0).sum()/100),
          '| solver total per wave: mean %.1f p90 %.1f max %.1f us' % (
              tot.mean()/100, np.percentile(tot, 90)/100, tot.max()/100),
          '| extra line-search passes per env: mean %.2f max %d' % (pr[:, 5].mean(), pr[:, 5].max()))
  print('%-50s %.4f ms  (iters mean %.2f, per-wave max mean %.2f, batch max %d; nefc mean %.1f, per-wave max mean %.1f, batch max %d)' % (
      ' '.join(flags) or 'full',
      float(np.median(ts)), st[2].mean(), st[2].reshape(-1, 64).max(axis=1).mean(), st[2].max(),
      st[1].mean(), st[1].reshape(-1, 64).max(axis=1).mean(), st[1].max()), flush=True)
  hb.free(); hm.free()
