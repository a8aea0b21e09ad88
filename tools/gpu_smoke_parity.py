"""Scratch GPU parity probe (not a test): device vs oracle, teacher-forced."""
import os
import sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import helpers
from dm_control_amd import build, wrapper, codegen
from oracle import oracle

def run(name, prec, nenv=128, T=30, nsub=1):
  m = helpers.load_model(name)
  co = build.build_model(m, helpers.TASKS[name], prec)
  hm = wrapper.HipModel(co)
  hb = wrapper.HipBatch(hm, nenv)
  om = oracle.OracleModel(m)
  ods = [oracle.OracleData(om) for _ in range(nenv)]
  qpos, qvel = helpers.initial_states(m, name, nenv, 0)
  hb.set_state(qpos.T, qvel.T)
  hb.forward()
  for i, d in enumerate(ods):
    d.qpos[:] = qpos[i]; d.qvel[:] = qvel[i]; d.step1()
  rs = np.random.RandomState(1)
  worst_tf = 0
  for t in range(T):
    ctrl = rs.uniform(-1, 1, (nenv, m.nu))
    hb.step_host(ctrl, nsub)
    q = hb.read(wrapper.FIELD_QPOS).T.astype(np.float64); v = hb.read(wrapper.FIELD_QVEL).T.astype(np.float64)
    stats = hb.read(wrapper.FIELD_STATS)
    for i, d in enumerate(ods):
      d.ctrl[:] = ctrl[i]
      for _ in range(nsub): d.physics_step()
    oq = np.array([d.qpos.copy() for d in ods]); ov = np.array([d.qvel.copy() for d in ods])
    eq = helpers.rel_err(q, oq); ev = helpers.rel_err(v, ov)
    if t < 3 or t == T-1:
      print(name, prec, 't', t, 'free-run err qpos %.3e qvel %.3e' % (eq.max(), ev.max()),
            'ncon dev', stats[0].mean(), 'orc', np.mean([d.ncon for d in ods]),
            'iters dev', stats[2].mean(), 'orc', np.mean([d.solver_iter for d in ods]))
  warn = hb.read(wrapper.FIELD_WARN)
  print(name, prec, 'warn any', warn.any(), 'reward', hb.read(wrapper.FIELD_REWARD)[:4], 'obs0', hb.read(wrapper.FIELD_OBS)[0][:6])
  # timing
  ctrl = rs.uniform(-1, 1, (nenv, m.nu))
  hb.step_host(ctrl, nsub); hb.sync()
  t0 = time.time()
  for _ in range(20): hb.step_host(None, nsub)
  hb.sync(); dt = (time.time()-t0)/20
  print(name, prec, 'nenv', nenv, 'ms/step %.3f' % (dt*1e3))

if __name__ == '__main__':
  for name in sys.argv[1].split(','):
    for prec in sys.argv[2].split(','):
      run(name, prec, nenv=int(sys.argv[3]) if len(sys.argv) > 3 else 128, nsub=5 if name=='humanoid' else 1)
