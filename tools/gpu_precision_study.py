"""Free-run precision study: which build meets the
north-star tolerance (qpos rel-err <= 1e-4 over 1000 steps vs the CPU step)?

Runs on the GPU box.  For cart-pole swing-up and cheetah-run, 256 envs from
task-like start states, U(-1,1) actions, 1000 free-running control steps,
against the fp64 oracle; variants of the one-env-per-lane code object:
  f32        the benchmarked build
  f32+comp   fp32 arithmetic, qpos/qvel carried as fp64 (hi, lo) pairs
  f32+r64    fp32 arithmetic with the fp64 build's unmodified stopping rules
  f32+both
  f64
Reports the error distribution at 100 and 1000 steps and the first step at
which an env's contact count differs from the oracle's.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import helpers  # noqa: E402
from dm_control_amd import build, wrapper as W  # noqa: E402
from oracle import oracle  # noqa: E402


def task_like_states(model, name, nenv, rs):
  qpos = np.tile(model.qpos0, (nenv, 1))
  qvel = np.zeros((nenv, model.nv))
  if name == 'cheetah':
    lim = model.jnt_limited.astype(bool)
    lo, hi = model.jnt_range[lim].T
    qpos[:, lim] = rs.uniform(0.3*lo, 0.3*hi, (nenv, lim.sum()))
  else:
    qpos[:, 1] = np.pi + 0.01*rs.randn(nenv)
    qvel[:] = 0.01*rs.randn(nenv, model.nv)
  return qpos, qvel


def summarise(e):
  return {'median': float(np.median(e)), 'p90': float(np.percentile(e, 90)),
          'p99': float(np.percentile(e, 99)), 'max': float(e.max()),
          'frac_le_1e-4': float(np.mean(e <= 1e-4))}


def main():
  nenv, steps = 256, 1000
  marks = (100, 1000)
  variants = [('f32', 'f32', ()), ('f32+comp', 'f32', ('-DDMC_STATE_COMP=1',)),
              ('f32+r64', 'f32', ('-DDMC_F64_RULES=1',)),
              ('f32+both', 'f32', ('-DDMC_STATE_COMP=1', '-DDMC_F64_RULES=1')),
              ('f64', 'f64', ())]
  out = {}
  for name in ('cartpole', 'cheetah'):
    model = helpers.load_model(name)
    rs = np.random.RandomState(0)
    qpos, qvel = task_like_states(model, name, nenv, rs)
    ctrls = rs.uniform(-1, 1, (steps, nenv, model.nu))
    om = oracle.OracleModel(model)
    datas = [oracle.OracleData(om) for _ in range(nenv)]
    for i, d in enumerate(datas):
      d.qpos[:] = qpos[i]; d.qvel[:] = qvel[i]; d.step1()
    ref, ncon_ref = {}, np.zeros((steps, nenv), np.int32)
    for t in range(steps):
      ncon_ref[t] = [d.ncon for d in datas]      # contacts the step acts on
      oracle.batch_step(om, datas, ctrls[t], 1, 0)
      if t + 1 in marks:
        ref[t + 1] = np.array([d.qpos.copy() for d in datas])
    out[name] = {}
    for label, precision, flags in variants:
      path = build.build_model(model, helpers.TASKS[name], precision,
                               extra_flags=flags, mode='auto')
      hm = W.HipModel(path)
      hb = W.HipBatch(hm, nenv)
      hb.set_state(qpos.T, qvel.T)
      first = np.full(nenv, steps + 1)
      res = {}
      for t in range(steps):
        hb.step_host(ctrls[t], 1)
        ncon = hb.read(W.FIELD_STATS)[0]
        diff = (ncon != ncon_ref[t]) & (first > steps)
        first[diff] = t + 1
        if t + 1 in marks:
          q = hb.read(W.FIELD_QPOS).T.astype(np.float64)
          res['step%d' % (t + 1)] = summarise(helpers.rel_err(q, ref[t + 1]))
      res['first_contact_mismatch'] = {
          'envs_with_mismatch': int((first <= steps).sum()),
          'earliest_step': int(first.min()) if (first <= steps).any() else None,
          'median_step_among_mismatched': (
              float(np.median(first[first <= steps])) if (first <= steps).any() else None)}
      res['warn'] = int(hb.read(W.FIELD_WARN).any())
      out[name][label] = res
      print(name, label, json.dumps(res), flush=True)
      hb.free(); hm.free()
  os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
  with open(os.path.join(ROOT, 'gpurun_out', 'precision_study.json'), 'w') as f:
    json.dump(out, f, indent=1)


if __name__ == '__main__':
  main()
