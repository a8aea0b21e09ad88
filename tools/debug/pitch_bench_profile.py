"""Per-pitch solver / stage times of the team build on the states bench.py visits:
`soccer.load(2)` with 1024 pitches and U(-1,1) actions.
usage: pitch_bench_profile.py solver|stage|tree [--build-only]   (sets $DMC_EXTRA_FLAGS itself)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
WHAT = sys.argv[1]
os.environ['DMC_EXTRA_FLAGS'] = {'solver': '-DDMC_SOLVER_PROFILE=1', 'stage': '-DDMC_STEP_PROFILE=1',
                                 'tree': '-DDMC_STEP_PROFILE=1 -DDMC_TREE_PROFILE=1'}[WHAT]
import numpy as np
from dm_control_amd.locomotion import soccer
from dm_control_amd import wrapper as W
B = 1024
if '--build-only' in sys.argv:
  from dm_control_amd import build
  from dm_control_amd.locomotion.models import soccer as scene
  from dm_control_amd.mjcf import compiler
  geometry = soccer.PitchGeometry(soccer.area_to_size(soccer.MINI_FOOTBALL_MIN_AREA_PER_HUMANOID*4), soccer.MINI_FOOTBALL_GOAL_SIZE)
  xml = scene.build(4, with_ball=True, pitch_size=geometry.size, ball=scene.REGULATION_BALL, goal_size=soccer.MINI_FOOTBALL_GOAL_SIZE)
  print(os.path.basename(build.build_model(compiler.from_xml_string(xml), 0, 'f32', ncon_max=160, mode='team')))
  sys.exit(0)
env = soccer.load(2, random_state=1, environment_kwargs={'batch_size': B})
env.reset()
rs = np.random.RandomState(0)
hb = env.physics.batch
if WHAT != 'solver':
  hb.set_aux_outputs(True)
for t in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
  env.step([rs.uniform(-1, 1, (B, 56)) for _ in range(4)])
hb.sync()
st = hb.read(W.FIELD_STATS)
if WHAT == 'solver':
  obs = np.asarray(hb.read(W.FIELD_OBS)).reshape(B, -1)[:, :8].T.astype(np.float64)
  names = ['pass A + gradient', 'Hessian tiles + factor', 'solve', 'M*search, q1 q2', 'pass B (Jv)', 'line search']
  tot = obs[:6].sum(axis=0)/100.0
  for k in range(6):
    print('  %-24s mean %9.1f us  max %9.1f us' % (names[k], obs[k].mean()/100.0, obs[k].max()/100.0))
  print('  solver total per pitch: mean %.0f us p90 %.0f max %.0f' % (tot.mean(), np.percentile(tot, 90), tot.max()))
  print('  iterations with a step: mean %.1f max %d' % (obs[6].mean(), obs[6].max()))
  worst = int(np.argmax(tot))
  print('  slowest pitch %d: ncon %d nefc %d; phases %s' % (worst, st[0][worst], st[1][worst], np.round(obs[:6, worst]/100.0)))
else:
  prof = hb.read(W.FIELD_XPOS)[:8].astype(np.float64)/100.0
  names = ['tree recursions', 'factor M', 'qacc_smooth', 'limit rows', 'contact rows', 'warm start+Newton']
  if WHAT == 'tree':      # lane 0 = the first tree's lane: the shares of its recursions
    names = ['kinematics', 'com_pos', 'com_vel', 'crb + rows of M', 'RNE + actuation', '-']
  for k in range(6):
    print('  %-20s mean %9.1f us  max %9.1f us' % (names[k], prof[k].mean(), prof[k].max()))
  tot = prof[:6].sum(axis=0)
  print('  forward total per pitch: mean %.0f us p90 %.0f max %.0f' % (tot.mean(), np.percentile(tot, 90), tot.max()))
  print('  after forward (Euler, integration): mean %.0f us max %.0f; all substeps: mean %.0f us max %.0f'
        % (prof[6].mean(), prof[6].max(), prof[7].mean(), prof[7].max()))
  hb.sync(); hb.timer_start(); env.step([rs.uniform(-1, 1, (B, 56)) for _ in range(4)]); ms, n = hb.timer_stop()
  print('  one control step, kernel: %.2f ms' % (ms/max(n, 1)))
print('  ncon mean %.1f max %d, nefc mean %.0f max %d' % (st[0].mean(), st[0].max(), st[1].mean(), st[1].max()))
