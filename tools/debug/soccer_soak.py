"""Soak of `locomotion.soccer.load(2)` on the team build: random actions for many control
steps (episodes restart at the time limit / after goals), then: warning bits, finiteness, where the
players and the ball are, constraint statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from dm_control_amd.locomotion import soccer
from dm_control_amd import wrapper as W
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 600
NCON = int(sys.argv[3]) if len(sys.argv) > 3 else None
kw = {'batch_size': B}
if NCON: kw['ncon_max'] = NCON
env = soccer.load(2, random_state=7, time_limit=10.0, environment_kwargs=kw)
env.physics._warnings_cause_exception = False        # report, do not raise
ts = env.reset()
rs = np.random.RandomState(1)
t0 = time.time()
max_ncon = max_nefc = 0
goals = 0
for t in range(STEPS):
  ts = env.step([rs.uniform(-1, 1, (B, 56)) for _ in range(4)])
  st = env.physics.batch.read(W.FIELD_STATS)
  max_ncon = max(max_ncon, int(st[0].max())); max_nefc = max(max_nefc, int(st[1].max()))
  goals += int(np.sum(np.abs(np.asarray(ts.reward[0])) > 0))
  if ts.last():
    ts = env.reset()
q = np.asarray(env.physics.data.qpos)
warn = np.asarray(env.physics.batch.read(W.FIELD_WARN)).ravel()
z = q[:, [63*k + 2 for k in range(4)]]
ball = q[:, 252:255]
print('%d pitches x %d control steps in %.0f s' % (B, STEPS, time.time() - t0))
print('warning bits set in %d pitches (mask OR = %d); finite: %s' % (int((warn != 0).sum()), int(np.bitwise_or.reduce(warn)), bool(np.isfinite(q).all())))
print('root heights: min %.3f max %.3f; ball height min %.3f max %.3f; |ball xy| max %.2f' % (z.min(), z.max(), ball[:, 2].min(), ball[:, 2].max(), np.abs(ball[:, :2]).max()))
print('contacts per pitch up to %d, rows up to %d (capacity %d / %d); goals scored %d' % (max_ncon, max_nefc, env.physics.model_info.ncon_max if hasattr(env.physics, 'model_info') else 64, -1, goals))
