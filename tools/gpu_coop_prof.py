"""Scratch: per-phase time of the coop kernel (DMC_COOP_PROFILE build)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
from dm_control_amd import suite, wrapper, build
name, task, B = sys.argv[1], sys.argv[2], int(sys.argv[3])
PH = 'KIN COM CRB FACM VEL SMOOTH LIMIT DETECT CROWS FINISH WARM HESS FACH SOLVE LS UPD EULER OBS'.split()
env = suite.load(name, task, task_kwargs={'random': 1},
                 environment_kwargs={'batch_size': B, 'device_init': True, 'build_mode': 'coop'})
p = env.physics; b = p.batch
env.reset()
nsub = env._n_sub_steps
rs = np.random.RandomState(0)
acc = np.zeros(len(PH)); its = 0
for t in range(30):
  p.set_control(rs.uniform(-1, 1, (B, p.model.nu))); p.step(nsub, check=False)
  if t >= 10:
    obs = b.read(wrapper.FIELD_OBS)
    acc += np.pad(obs[:, :len(PH)].mean(axis=0), (0, max(0, len(PH) - obs.shape[1]))); its += b.read(wrapper.FIELD_STATS)[2].mean()
acc /= 20
tot = acc.sum()
print('%s B=%d: total %.1f us per launch per wave (%d substeps), newton iters/substep %.2f' % (name, B, tot/100, nsub, its/20))
for k, v in zip(PH, acc):
  print('  %-7s %8.1f us  %5.1f %%' % (k, v/100, 100*v/tot))
