"""Per-phase time of the several-lanes-per-env kernel (-DDMC_COOP_PROFILE build)
plus the placement and life time of every wave of the last launch.

  DMC_EXTRA_FLAGS=-DDMC_COOP_PROFILE python tools/gpu_coop_prof.py humanoid walk 1024
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
os.environ.setdefault('DMC_EXTRA_FLAGS', '-DDMC_COOP_PROFILE')
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
N = 20
for t in range(10 + N):
  p.set_control(rs.uniform(-1, 1, (B, p.model.nu))); p.step(nsub, check=False)
  if t >= 10:
    obs = b.read(wrapper.FIELD_OBS)
    stats = b.read(wrapper.FIELD_STATS)
    acc += np.pad(obs[:, :len(PH)].mean(axis=0), (0, max(0, len(PH) - obs.shape[1]))); its += stats[2].mean()
acc /= N
tot = acc.sum()
print('%s B=%d: total %.1f us per launch per wave (%d substeps), newton iters of the last substep %.2f' % (name, B, tot/100, nsub, its/N))
for k, v in zip(PH, acc):
  print('  %-7s %8.1f us  %5.1f %%' % (k, v/100, 100*v/tot))
if obs.shape[1] >= len(PH) + 4:
  # the last launch: life time of every wave and where it ran
  per = obs[:, :len(PH)].sum(axis=1)/100
  t0, t1 = obs[:, len(PH)].astype(np.int64), obs[:, len(PH) + 1].astype(np.int64)
  t1 = np.where(t1 < t0, t1 + (1 << 20), t1)
  base = t0.min()
  start, end = (t0 - base)/100.0, (t1 - base)/100.0
  hw, xcc = obs[:, len(PH) + 2].astype(np.int64), obs[:, len(PH) + 3].astype(np.int64)
  simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
  q = lambda x: 'min %.0f  p50 %.0f  mean %.0f  p90 %.0f  p99 %.0f  max %.0f' % (
      x.min(), np.median(x), x.mean(), np.percentile(x, 90), np.percentile(x, 99), x.max())
  print('last launch, us: wave life time  ' + q(end - start))
  print('                 wave start       ' + q(start))
  print('                 wave end         ' + q(end))
  print('                 sum of phases    ' + q(per))
  slot = ((xcc*8 + se)*2 + sh)*16 + cu
  simd_slot = slot*4 + simd
  n_cu = np.bincount(np.unique(slot, return_inverse=True)[1])
  n_simd = np.bincount(np.unique(simd_slot, return_inverse=True)[1])
  print('placement: %d distinct CUs (waves per CU: %s), %d distinct SIMDs (waves per SIMD: %s)' % (
      len(n_cu), dict(zip(*np.unique(n_cu, return_counts=True))),
      len(n_simd), dict(zip(*np.unique(n_simd, return_counts=True)))))
  it = stats[2]; nefc = stats[1]
  life = end - start
  print('corr(life time, nefc) %.2f   corr(life time, iters of last substep) %.2f' % (
      np.corrcoef(life, nefc)[0, 1], np.corrcoef(life, it)[0, 1]))
  late = start > 50
  print('waves starting later than 50 us after the first: %d' % late.sum())
  for lo, hi in ((0, 25), (25, 50), (50, 75), (75, 90), (90, 100)):
    a_, b_ = np.percentile(life, lo), np.percentile(life, hi)
    m = (life >= a_) & (life <= b_)
    print('  life-time percentile %3d-%3d: %.0f-%.0f us, nefc mean %.1f, ncon mean %.1f' % (
        lo, hi, a_, b_, nefc[m].mean(), stats[0][m].mean()))
  ph = obs[:, :len(PH)]/100
  lo_m = life <= np.percentile(life, 25); hi_m = life >= np.percentile(life, 90)
  print('phase time of the last launch, us: fastest quarter | slowest tenth | difference')
  for k, name_ in enumerate(PH):
    print('  %-7s %7.1f %7.1f %7.1f' % (name_, ph[lo_m, k].mean(), ph[hi_m, k].mean(), ph[hi_m, k].mean() - ph[lo_m, k].mean()))
  print('iterations of the last substep: histogram', np.bincount(it.astype(int)))
  print('  slowest tenth: iters mean %.2f; fastest quarter: %.2f' % (it[hi_m].mean(), it[lo_m].mean()))
