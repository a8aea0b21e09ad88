"""Scratch: group size sweep of the several-lanes-per-env kernel."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
from dm_control_amd import build, wrapper as W
name = sys.argv[1]
model = helpers.load_model(name)
nsub = {'cheetah': 1, 'walker': 10, 'hopper': 4, 'humanoid': 5}[name]
for B in (1024, 4096, 8192, 16384):
  line = '%s B=%d:' % (name, B)
  for mode, group in (('auto', 0), ('coop', 64), ('coop', 32), ('coop', 16)):
    try:
      path = build.build_model(model, helpers.TASKS[name], 'f32', mode=mode, group=group or 64)
    except Exception as e:   # LDS does not fit
      line += '  %s/%d n/a' % (mode, group); continue
    hm = W.HipModel(path); hb = W.HipBatch(hm, B)
    qpos, qvel = helpers.initial_states(model, name, B, seed=1)
    hb.set_state(qpos.T, qvel.T)
    rs = np.random.RandomState(0)
    acts = [rs.uniform(-1, 1, (B, model.nu)) for _ in range(4)]
    for t in range(30): hb.step_host(acts[t % 4], nsub)
    hb.sync(); hb.timer_start()
    for t in range(60): hb.step_host(None, nsub)
    ms, n = hb.timer_stop()
    line += '  %s/%d %.4f ms %.1f M/s' % (mode, group, ms/n, B/(ms/n)/1e3)
    hb.free(); hm.free()
  print(line, flush=True)
