"""the fp64 UNROLLED build of the 20-dof PRIMITIVES model
computed one dof's velocity wrongly by g*h on the GPU in round 1 (hidden by the
spill-aware `auto` mode).  One confirming run: unrolled vs rolled vs oracle,
teacher-forced, with the acceleration-stage outputs next to each other so that
a wrong value can be localised (qacc from the device, qacc / qfrc_* from the
oracle at the same state)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import kat_models  # noqa: E402
from dm_control_amd import build, wrapper as W  # noqa: E402
from dm_control_amd.mjcf import compiler  # noqa: E402
from oracle import oracle  # noqa: E402

model = compiler.from_xml_string(kat_models.PRIMITIVES)
nenv = 32
rs = np.random.RandomState(2)
qpos = np.tile(model.qpos0, (nenv, 1))
qvel = 0.2*rs.randn(nenv, model.nv)
qvel[0] = 0
for j in range(model.njnt):
  a = model.jnt_qposadr[j]
  if model.jnt_type[j] == 0:
    quat = np.array([1.0, 0, 0, 0]) + 0.2*rs.randn(nenv, 4)
    quat[0] = [1, 0, 0, 0]
    qpos[:, a + 3:a + 7] = quat/np.linalg.norm(quat, axis=1, keepdims=True)
    qpos[1:, a + 2] += rs.uniform(0, 0.05, nenv - 1)
om = oracle.OracleModel(model)
for mode in ('unrolled', 'rolled'):
  path = build.build_model(model, 0, 'f64', mode=mode)
  hm = W.HipModel(path)
  hb = W.HipBatch(hm, nenv)
  hb.set_aux_outputs(True)
  datas = [oracle.OracleData(om) for _ in range(nenv)]
  for i, d in enumerate(datas):
    d.qpos[:] = qpos[i]; d.qvel[:] = qvel[i]; d.step1()
  worst = np.zeros(model.nv)
  worst_acc = np.zeros(model.nv)
  first = None
  for t in range(60):
    oq = np.array([d.qpos.copy() for d in datas]); ov = np.array([d.qvel.copy() for d in datas])
    ow = np.array([d.qacc_warmstart.copy() for d in datas])
    hb.set_state(oq.T, ov.T, ow.T)
    hb.step_host(None, 1)
    v = hb.read(W.FIELD_QVEL).T
    acc = hb.read(W.FIELD_QACC).T
    oacc = []
    for d in datas:
      d.step2()
      oacc.append(d.qacc.copy())
      d.step1()
    nv = np.array([d.qvel.copy() for d in datas])
    ev = np.abs(v - nv).max(axis=0)
    ea = np.abs(acc - np.array(oacc)).max(axis=0)
    worst = np.maximum(worst, ev); worst_acc = np.maximum(worst_acc, ea)
    if first is None and ev.max() > 1e-7:
      i = int(np.argmax(np.abs(v - nv).max(axis=1)))
      first = (t, int(np.argmax(ev)), float(ev.max()), i,
               'device qacc %s' % np.round(acc[i], 6), 'oracle qacc %s' % np.round(oacc[i], 6),
               'ncon %d nefc %d' % (datas[i].ncon, datas[i].nefc))
  print('%-9s max |dqvel| per dof %s' % (mode, np.array2string(worst, precision=2)))
  print('%-9s max |dqacc| per dof %s' % (mode, np.array2string(worst_acc, precision=2)))
  print('%-9s first divergence > 1e-7: %s' % (mode, first), flush=True)
  hb.free(); hm.free()
