"""Kernel-only sweep over the step kernels of one model: one env per lane, one
env per 64 lanes with and without the second (row-building) wavefront, one env
per 32 lanes.  Data behind `Physics._COOP_POLICY`.

  python tools/gpu_group_sweep.py walker [batch sizes]
"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
from dm_control_amd import build, wrapper as W
name = sys.argv[1]
BUILD_ONLY = '--build-only' in sys.argv      # pre-build here, run on the GPU box
batches = [int(x) for x in sys.argv[2:] if not x.startswith('--')] or [256, 1024, 2048, 4096, 8192]
model = helpers.load_model(name)
nsub = {'cheetah': 1, 'walker': 10, 'hopper': 4, 'humanoid': 5}[name]
PREC = os.environ.get('DMC_SWEEP_PRECISION', 'f32')
VARIANTS = (('one env per lane', 'auto', 64, ()),
            ('64 lanes x 2 waves', 'coop', 128, ()),
            ('64 lanes', 'coop', 64, ()),
            ('32 lanes', 'coop', 32, ()),
            ('16 lanes', 'coop', 16, ()),
            ('8 lanes', 'coop', 8, ()))
for B in batches:
  line = '%s %s B=%d:' % (name, PREC, B)
  for label, mode, group, flags in VARIANTS:
    if name == 'humanoid' and mode == 'auto' and B > 1024:
      continue
    try:
      path = build.build_model(model, helpers.TASKS[name], PREC, mode=mode, group=group,
                               extra_flags=flags)
    except Exception as e:   # LDS does not fit / beyond the spill budget
      line += '  | %s n/a' % label; continue
    if BUILD_ONLY:
      line += '  | %s built' % label; continue
    hm = W.HipModel(path); hb = W.HipBatch(hm, B)
    qpos, qvel = helpers.initial_states(model, name, B, seed=1)
    hb.set_state(qpos.T, qvel.T)
    rs = np.random.RandomState(0)
    acts = [rs.uniform(-1, 1, (B, model.nu)) for _ in range(4)]
    for t in range(30): hb.step_host(acts[t % 4], nsub)
    hb.sync(); hb.timer_start()
    for t in range(60): hb.step_host(None, nsub)
    ms, n = hb.timer_stop()
    line += '  | %s %.4f ms %.2f M/s' % (label, ms/n, B/(ms/n)/1e3)
    hb.free(); hm.free()
  print(line, flush=True)
