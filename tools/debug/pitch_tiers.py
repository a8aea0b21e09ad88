"""Semi-rolled vs strictly rolled build of the 2v2 pitch on the GPU: warning
bits and per-step error vs the oracle, test states and kick-off states."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_soccer_model as S
from dm_control_amd import build, wrapper as W
from oracle import oracle
quiet = sys.argv[1] == 'quiet'
prec = sys.argv[2]
m = S._pitch_model(quiet)
path = build.build_model(m, 0, prec, ncon_max=64, mode='rolled')
print('code object', os.path.basename(path), 'strict' if os.environ.get('DMC_ROLLED_STRICT') == '1' else 'semi')
hm = W.HipModel(path); hb = W.HipBatch(hm, 6)
qpos, qvel = S._pitch_states(m, 6, np.random.RandomState(7))
e, rows = S._teacher_forced(m, hb, qpos, qvel, 2, np.random.RandomState(1), W)
print('test states: warn', hb.read(W.FIELD_WARN), 'err max %.2e' % e.max())
hb.reset()
q0 = np.tile(m.qpos0, (6, 1)); v0 = np.zeros((6, m.nv))
e, rows = S._teacher_forced(m, hb, q0, v0, 2, np.random.RandomState(1), W)
print('qpos0 states: warn', hb.read(W.FIELD_WARN), 'err max %.2e' % e.max(), flush=True)
# kernel time of a control step (5 substeps), falling players under random torques
for B in (64, 1024):
  hb2 = W.HipBatch(hm, B)
  qp = np.tile(m.qpos0, (B, 1)); qp[:, [63*k + 2 for k in range(4)]] = 0.9
  hb2.set_state(qp.T, np.zeros((m.nv, B)))
  rs = np.random.RandomState(0)
  for t in range(2):
    hb2.step_host(rs.uniform(-1, 1, (B, m.nu)), 5)
  hb2.sync(); hb2.timer_start()
  for t in range(3):
    hb2.step_host(None, 5)
  ms, n = hb2.timer_stop()
  st = hb2.read(W.FIELD_STATS)
  print('B=%d: %.1f ms per control step (5 substeps); ncon mean %.1f max %d, nefc mean %.0f, iters mean %.1f max %d; warn %d'
        % (B, ms/n, st[0].mean(), st[0].max(), st[1].mean(), st[2].mean(), st[2].max(),
           int((hb2.read(W.FIELD_WARN) != 0).sum())), flush=True)
  hb2.free()
