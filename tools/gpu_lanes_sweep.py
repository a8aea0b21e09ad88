"""Scratch: envs per workgroup (DMC_LANES) x LDS budget sweep of the one-lane kernel."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
from dm_control_amd import build, suite, wrapper as W
name, task, nsub = sys.argv[1], sys.argv[2], int(sys.argv[3])
VARIANTS = [(64, 128), (32, 64), (32, 36), (16, 36), (16, 24)]
model = helpers.load_model(name)
for B in [int(b) for b in sys.argv[4:]]:
  env = suite.load(name, task, task_kwargs={'random': 1},
                   environment_kwargs={'batch_size': B, 'device_init': True, 'build_mode': 'auto'})
  env.reset()
  rs = np.random.RandomState(0)
  for t in range(100):
    env.physics.set_control(rs.uniform(-1, 1, (B, model.nu))); env.physics.step(nsub, check=False)
  b0 = env.physics.batch
  q, v, w = b0.read(W.FIELD_QPOS), b0.read(W.FIELD_QVEL), b0.read(W.FIELD_WARMSTART)
  if b0.model.info.env_major: pass
  env.physics.free()
  act = rs.uniform(-1, 1, (B, model.nu))
  line = '%s-%s B=%d:' % (name, task, B)
  for lanes, kb in VARIANTS:
    path = build.build_model(model, helpers.TASKS[name], 'f32', lanes=lanes, lds_budget=kb*1024)
    hm = W.HipModel(path); hb = W.HipBatch(hm, B)
    ts = []
    for rep in range(12):
      hb.set_state(q, v, w); hb.step_host(act, nsub)
      hb.set_state(q, v, w); hb.sync(); hb.timer_start(); hb.step_host(None, nsub)
      ms, n = hb.timer_stop(); ts.append(ms/n)
    line += '  L%d/%dK %.4f' % (lanes, kb, float(np.median(ts)))
    hb.free(); hm.free()
  print(line, flush=True)
