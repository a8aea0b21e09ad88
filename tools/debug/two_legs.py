"""Does a second Physics in the same process run slower?  (bench.py's fp64 leg after the fp32 one)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from dm_control_amd.locomotion import soccer
B = 1024
def leg(prec, free=True, steps=4):
  env = soccer.load(2, random_state=1, environment_kwargs={'batch_size': B, 'precision': prec})
  env.reset()
  rs = np.random.RandomState(0)
  hb = env.physics.batch
  for t in range(2):
    env.step([rs.uniform(-1, 1, (B, 56)) for _ in range(4)])
  hb.sync(); hb.timer_start()
  for t in range(steps):
    env.step([rs.uniform(-1, 1, (B, 56)) for _ in range(4)])
  ms, n = hb.timer_stop()
  print('%s: %.1f ms per control step (%d launches)' % (prec, ms/max(n, 1), n), flush=True)
  if free:
    env.physics.free()
  return env
order = sys.argv[1:] or ['f32', 'f64']
keep = []
for p in order:
  keep.append(leg(p.rstrip('+'), free=not p.endswith('+')))
