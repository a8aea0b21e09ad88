"""Scratch: one-lane vs several-lanes-per-env build at small batch sizes."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dm_control_amd import suite, wrapper
for name, task in [('cheetah', 'run'), ('walker', 'walk'), ('cartpole', 'swingup')]:
  for B in (256, 1024, 2048, 4096, 8192):
    line = '%s-%s B=%d:' % (name, task, B)
    for mode in ('auto', 'coop'):
      env = suite.load(name, task, task_kwargs={'random': 1},
                       environment_kwargs={'batch_size': B, 'device_init': True, 'build_mode': mode})
      p = env.physics; b = p.batch
      env.reset()
      nsub = env._n_sub_steps
      rs = np.random.RandomState(0)
      acts = [rs.uniform(-1, 1, (B, p.model.nu)) for _ in range(8)]
      for t in range(20):
        p.set_control(acts[t % 8]); p.step(nsub, check=False)
      b.sync(); b.timer_start()
      for t in range(100):
        p.set_control(acts[t % 8]); p.step(nsub, check=False)
      ms, n = b.timer_stop()
      line += '  %s %.4f ms %.2f M/s' % (mode, ms/n, B/(ms/n)/1e3)
      p.free()
    print(line, flush=True)
