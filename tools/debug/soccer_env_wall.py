"""Wall time of `soccer.load(2).step(actions)` (physics launch + host-side game logic and
observables) against the physics launch alone, 1024 pitches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from dm_control_amd.locomotion import soccer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = soccer.load(2, random_state=1, environment_kwargs={'batch_size': B})
env.reset()
rs = np.random.RandomState(0)
acts = [[rs.uniform(-1, 1, (B, 56)).astype(np.float32) for _ in range(4)] for _ in range(4)]
for t in range(3):
  env.step(acts[t % 4])
hb = env.physics.batch
hb.sync()
t0 = time.time(); hb.timer_start()
N = 10
for t in range(N):
  ts = env.step(acts[t % 4])
hb.sync()
ms, n = hb.timer_stop()
wall = (time.time() - t0)/N*1e3
print('B=%d: env.step wall %.1f ms per control step; physics kernel %.1f ms; host task layer + transfers %.1f ms'
      % (B, wall, ms/max(n, 1), wall - ms/max(n, 1)))
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for t in range(3):
  env.step(acts[t % 4])
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(14)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:40]))
