"""Scratch: parity + timing of the several-lanes-per-env build (mode='coop')."""
import os
import sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
import test_gpu_parity as T
from dm_control_amd import suite, wrapper

def parity(name, prec, nsub):
  helpers.MODES[name] = 'coop'
  e = T._teacher_forced(name, prec, nenv=64, steps=8, nsub=nsub)
  print('parity %s %s coop: median %.3g p99 %.3g max %.3g' % (name, prec, np.median(e), np.percentile(e, 99), e.max()), flush=True)

def timing(name, task, B, mode):
  env = suite.load(name, task, task_kwargs={'random': 1},
                   environment_kwargs={'batch_size': B, 'device_init': True, 'build_mode': mode})
  p = env.physics; b = p.batch
  env.reset()
  nsub = env._n_sub_steps
  rs = np.random.RandomState(0)
  acts = [rs.uniform(-1, 1, (B, p.model.nu)) for _ in range(8)]
  for t in range(20):
    p.set_control(acts[t % 8]); p.step(nsub, check=False)
  st = b.read(wrapper.FIELD_STATS)
  b.sync(); b.timer_start()
  for t in range(50):
    p.set_control(acts[t % 8]); p.step(nsub, check=False)
  ms, n = b.timer_stop()
  print('%s-%s B=%d %s: %.3f ms/launch -> %.3f M env-steps/s | iters mean %.2f nefc mean %.1f max %d warn %d' % (
      name, task, B, mode, ms/n, B/(ms/n)/1e3, st[2].mean(), st[1].mean(), st[1].max(),
      int((b.read(wrapper.FIELD_WARN) != 0).sum())), flush=True)
  env.physics.free()

what = sys.argv[1]
if what == 'parity':
  for name, prec, nsub in [('cheetah', 'f64', 1), ('humanoid', 'f64', 5), ('cheetah', 'f32', 1), ('humanoid', 'f32', 5), ('cartpole', 'f32', 1), ('walker', 'f32', 10)]:
    parity(name, prec, nsub)
else:
  for name, task, B, mode in [('humanoid', 'walk', 1024, 'coop'), ('humanoid', 'walk', 8192, 'coop'),
                              ('humanoid', 'walk', 1024, 'unrolled'), ('cheetah', 'run', 8192, 'coop'),
                              ('walker', 'walk', 8192, 'coop')]:
    timing(name, task, B, mode)
