"""Which leg of bench.py's CPU-baseline section corrupts the host heap for the
soccer pitch (GPU call 3/4: `free(): invalid next size` when the oracle data of
cpu_baseline() are released)?  Runs ONE leg per process (argv[1])."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench

leg = sys.argv[1]
env = bench.load_env('soccer', '2v2', 1000, {'batch_size': 64})
physics = env.physics
env.reset()
rs = np.random.RandomState(0)
for _ in range(3):
  env.step([rs.uniform(-1, 1, (64, 56)) for _ in range(4)])
from oracle import oracle
model = physics.model
if leg == 'native':
  out = os.path.join('/tmp', 'libmjoracle_native_%d.so' % os.getpid())
  oracle.build(force=True, cflags=['-O3', '-march=native', '-fopenmp', '-fPIC', '-std=c99',
                                   '-ffp-contract=off'], out=out)
  lib = oracle.load(out)
else:
  lib = oracle.load()
om = oracle.OracleModel(model, lib)
if leg in ('timing', 'native'):
  datas = [oracle.OracleData(om) for _ in range(32)]
  for d in datas:
    d.step1()
  for r in range(20):
    oracle.batch_step(om, datas, rs.uniform(-1, 1, (32, model.nu)), 5, 16)
  del datas
elif leg == 'teacher':
  print(bench._rel_err_sample(physics.batch, om, oracle, 5, nenv=8, steps=3)['max'])
elif leg == 'free':
  print(bench._free_run_sample(physics.batch, om, oracle, 5, 16, nenv=8, steps=6)['sample'])
elif leg == 'twin':
  hb = bench._twin_batch(physics.batch, 8)
  q0, v0, w0 = bench._start_states(physics.batch, 8)
  hb.set_state(q0.T, v0.T, w0.T)
  hb.step_host(rs.uniform(-1, 1, (8, model.nu)), 5)
  print(hb.read(1).shape)
  hb.free()
del om
print('leg', leg, 'ok', flush=True)
