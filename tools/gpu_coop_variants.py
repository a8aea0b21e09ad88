"""Kernel time of the several-lanes-per-env humanoid build under extra -D flags.

  python tools/gpu_coop_variants.py "" "-DDMC_COOP_CB=8" ...
Every variant is checked against the first one (qpos after the timed steps).
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dm_control_amd import suite, wrapper

def run(flags, B, name='humanoid', task='walk', prec='f32'):
  os.environ['DMC_EXTRA_FLAGS'] = flags
  env = suite.load(name, task, task_kwargs={'random': 1},
                   environment_kwargs={'batch_size': B, 'device_init': True,
                                       'build_mode': 'coop', 'precision': prec})
  p = env.physics; b = p.batch
  env.reset()
  nsub = env._n_sub_steps
  rs = np.random.RandomState(0)
  acts = [rs.uniform(-1, 1, (B, p.model.nu)) for _ in range(8)]
  for t in range(20):
    p.set_control(acts[t % 8]); p.step(nsub, check=False)
  q20 = b.read(wrapper.FIELD_QPOS).copy()
  b.sync(); b.timer_start()
  for t in range(50):
    p.set_control(acts[t % 8]); p.step(nsub, check=False)
  ms, n = b.timer_stop()
  st = b.read(wrapper.FIELD_STATS)
  env.physics.free()
  return ms/n, q20, st

ref = {}
for prec in ('f32',):
  for flags in sys.argv[1:]:
    line = '%-40s' % (flags or '(none)')
    for B in (1024, 8192):
      ms, q, st = run(flags, B, prec=prec)
      if (B, prec) not in ref: ref[(B, prec)] = q
      err = np.abs(q - ref[(B, prec)]).max()
      line += '  B=%d %.4f ms (%.2f M/s) dq %.1e iters %.2f' % (B, ms, B/ms/1e3, err, st[2].mean())
    print(line, flush=True)
