import sys, numpy as np
sys.path.insert(0, 'tests')
import helpers
from dm_control_amd import build, wrapper
from oracle import oracle
np.set_printoptions(precision=6, suppress=True, linewidth=200)
name, prec, nenv = 'cheetah', 'f32', 128
m = helpers.load_model(name)
hm = wrapper.HipModel(build.build_model(m, helpers.TASKS[name], prec))
hb = wrapper.HipBatch(hm, nenv)
om = oracle.OracleModel(m)
ods = [oracle.OracleData(om) for _ in range(nenv)]
qpos, qvel = helpers.initial_states(m, name, nenv, 0)
for i, d in enumerate(ods):
  d.qpos[:] = qpos[i]; d.qvel[:] = qvel[i]; d.step1()
rs = np.random.RandomState(1)
for t in range(4):
  ctrl = rs.uniform(-1, 1, (nenv, m.nu))
  # teacher forcing: device starts from the oracle state (incl. warmstart)
  oq = np.array([d.qpos.copy() for d in ods]); ov = np.array([d.qvel.copy() for d in ods]); ow = np.array([d.qacc_warmstart.copy() for d in ods])
  hb.set_state(oq.T, ov.T, ow.T)
  hb.step_host(ctrl, 1)
  q = hb.read(wrapper.FIELD_QPOS).T.astype(np.float64); v = hb.read(wrapper.FIELD_QVEL).T.astype(np.float64)
  qa = hb.read(wrapper.FIELD_QACC).T.astype(np.float64)
  stats = hb.read(wrapper.FIELD_STATS)
  pre = []
  for i, d in enumerate(ods):
    d.ctrl[:] = ctrl[i]
    pre.append((d.ncon, d.nefc, [d.contact(c) for c in range(d.ncon)], d.efc_pos[:d.nefc].copy()))
    d.physics_step()
  nq = np.array([d.qpos.copy() for d in ods]); nv_ = np.array([d.qvel.copy() for d in ods]); oqa = np.array([d.qacc.copy() for d in ods])
  ev = helpers.rel_err(v, nv_)
  w = int(np.argmax(ev))
  print('t', t, 'teacher-forced max qvel err %.3e at env %d; median %.3e' % (ev.max(), w, np.median(ev)))
  bad = np.where(ev > 1e-3)[0]
  print('  envs > 1e-3:', bad)
  for w in bad[:3]:
    print('  env', w, 'dev ncon/nefc/iters', stats[:, w], 'oracle pre ncon/nefc', pre[w][0], pre[w][1], 'orc iters', ods[w].solver_iter)
    for c in pre[w][2]: print('    contact g', c['geom1'], c['geom2'], 'dist %.6g' % c['dist'])
    print('    efc_pos', pre[w][3])
    print('    qacc dev', qa[w]); print('    qacc orc', oqa[w])
