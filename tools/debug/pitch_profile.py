"""Stage profile of one control step of the 2v2 pitch (-DDMC_STEP_PROFILE build):
mean / max over the pitches of the time per stage of forward(), 5 substeps.
usage: pitch_profile.py quiet|loud [--team] [--f64] [--build-only]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import test_soccer_model as S
from dm_control_amd import build, wrapper as W
m = S._pitch_model(sys.argv[1] == 'quiet')
MODE = 'team' if '--team' in sys.argv else 'rolled'
PREC = 'f64' if '--f64' in sys.argv else 'f32'
SOLVER = '--solver' in sys.argv      # phases of the (team) Newton solver instead of the step's stages
path = build.build_model(m, 0, PREC, ncon_max=64, mode=MODE,
                         extra_flags=('-DDMC_SOLVER_PROFILE=1',) if SOLVER else ('-DDMC_STEP_PROFILE=1',))
if '--build-only' in sys.argv:
  print(os.path.basename(path)); sys.exit(0)
hm = W.HipModel(path)
B = 1024
hb = W.HipBatch(hm, B)
hb.set_aux_outputs(not SOLVER)
qp = np.tile(m.qpos0, (B, 1)); qp[:, [63*k + 2 for k in range(4)]] = 0.9
rs = np.random.RandomState(0)
# spread the players like a kick-off: far apart on the 23 x 17 m pitch
for k in range(4):
  qp[:, 63*k:63*k + 2] = rs.uniform([-6, -5], [6, 5], (B, 2)) + [[-3, 0], [3, 0], [-3, 3], [3, -3]][k]
hb.set_state(qp.T, np.zeros((m.nv, B)))
for t in range(2):
  hb.step_host(rs.uniform(-1, 1, (B, m.nu)), 5)
hb.sync(); hb.timer_start()
hb.step_host(None, 5)
ms, n = hb.timer_stop()
if SOLVER:
  obs = np.asarray(hb.read(W.FIELD_OBS)).reshape(B, -1)[:, :8].T.astype(np.float64)
  print('%s %s B=%d: %.2f ms per control step (solver profile)' % (MODE, PREC, B, ms/n))
  names = ['pass A + gradient', 'Hessian tiles + factor', 'solve', 'M*search, q1 q2', 'pass B (Jv)', 'line search']
  for k in range(6):
    print('  %-24s mean %9.1f us  max %9.1f us' % (names[k], obs[k].mean()/100.0, obs[k].max()/100.0))
  print('  iterations with a step: mean %.1f max %d; extra line-search evaluations: mean %.1f max %d'
        % (obs[6].mean(), obs[6].max(), obs[7].mean(), obs[7].max()))
  sys.exit(0)
prof = hb.read(W.FIELD_XPOS)[:8].astype(np.float64)/100.0     # us (100 MHz)
names = ['kinematics+com', 'crb+factor M', 'com_vel+smooth', 'limit rows', 'contact rows', 'warm start+Newton', '-', '-']
print('%s %s B=%d: %.2f ms per control step' % (MODE, PREC, B, ms/n))
for k in range(6):
  print('  %-20s mean %9.1f us  max %9.1f us' % (names[k], prof[k].mean(), prof[k].max()))
print('  sum of stages mean %.1f ms (x lanes run in lock step: a wave takes its slowest lane)' % (prof[:6].sum(axis=0).mean()/1e3))
st = hb.read(W.FIELD_STATS)
print('  ncon mean %.1f, nefc mean %.0f, iters (last substep) mean %.1f' % (st[0].mean(), st[1].mean(), st[2].mean()))
