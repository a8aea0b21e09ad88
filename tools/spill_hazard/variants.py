"""Discriminator builds for the wrong result of the over-budget fp64 unrolled
build of the 20-dof PRIMITIVES model.

  python tools/spill_hazard/variants.py build   # here (no GPU): compile all variants
  python tools/spill_hazard/variants.py run     # on the GPU box: one teacher-forced
                                                # step per variant, qacc of the arm
                                                # (dofs 18, 19) next to the oracle's

Every variant is the -DDMC_SELECT_SLOTS source (the semantics-preserving edit
that makes the result wrong) plus ONE code-generation switch.  Switches that do
not change register allocation (waitcnt-forcezero: every wait is a full wait;
snop-padding: a wait state before every instruction) separate "a wait or hazard
is missing" from "a spilled value is clobbered".
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import kat_models  # noqa: E402
from dm_control_amd import build  # noqa: E402
from dm_control_amd.mjcf import compiler  # noqa: E402

SEL = ('-DDMC_SELECT_SLOTS',)
VARIANTS = [
    ('ifchain (shipped source)', ()),
    ('select', SEL),
    ('select + waitcnt-forcezero', SEL + ('-mllvm', '-amdgpu-waitcnt-forcezero')),
    ('select + snop-padding=1', SEL + ('-mllvm', '-amdgpu-snop-padding=1')),
    # built but NOT run: with SGPR spills sent to memory this kernel raised a GPU
    # memory access fault (gpurun_out/r03a, round 3) -- never launch it again
    ('select + spill-sgpr-to-vgpr=0 [FAULTS: build only]',
     SEL + ('-mllvm', '-amdgpu-spill-sgpr-to-vgpr=0')),
    ('select + prealloc-sgpr-spill-vgprs', SEL + ('-mllvm', '-amdgpu-prealloc-sgpr-spill-vgprs')),
    ('select + spill-vgpr-to-agpr=0', SEL + ('-mllvm', '-amdgpu-spill-vgpr-to-agpr=0')),
    ('select + disable-ssc', SEL + ('-mllvm', '-disable-ssc')),
    ('select + dce-in-ra=0', SEL + ('-mllvm', '-amdgpu-dce-in-ra=0')),
    ('select + opt-exec-mask-pre-ra=0', SEL + ('-mllvm', '-amdgpu-opt-exec-mask-pre-ra=0')),
    ('select + rewrite-partial-reg-uses=0',
     SEL + ('-mllvm', '-amdgpu-enable-rewrite-partial-reg-uses=0')),
]


def main():
  os.environ['DMC_ALLOW_OVERBUDGET'] = '1'
  model = compiler.from_xml_string(kat_models.PRIMITIVES)
  what = sys.argv[1] if len(sys.argv) > 1 else 'build'
  paths = []
  for name, flags in VARIANTS:
    try:
      paths.append((name, build.build_model(model, 0, 'f64', mode='unrolled',
                                            extra_flags=flags)))
    except RuntimeError as e:
      print('%-40s does not build: %s' % (name, str(e).strip().splitlines()[-1][:120]))
  if what == 'build':
    for name, p in paths:
      print('%-40s %s' % (name, os.path.basename(p)))
    return
  from dm_control_amd import wrapper as W
  from oracle import oracle
  nenv = 8
  rs = np.random.RandomState(2)
  qpos = np.tile(model.qpos0, (nenv, 1))
  qvel = 0.2*rs.randn(nenv, model.nv)
  qpos[:, 20:] += 0.3*rs.randn(nenv, model.nq - 20)    # the arm's hinges
  om = oracle.OracleModel(model)
  datas = [oracle.OracleData(om) for _ in range(nenv)]
  for i, d in enumerate(datas):
    d.qpos[:] = qpos[i]; d.qvel[:] = qvel[i]; d.step1()
  ref = []
  for d in datas:
    d.step2()
    ref.append(d.qacc.copy())
  ref = np.array(ref)
  print('oracle qacc[18:20] env0 %r' % ref[0, 18:].tolist())
  dump = {'qpos': qpos, 'qvel': qvel, 'oracle_qacc': ref,
          'oracle_qM': np.array([d.qM.copy() for d in datas]) if hasattr(datas[0], 'qM') else 0,
          'oracle_qfrc_smooth': np.array([np.asarray(d.qfrc_smooth).copy() for d in datas])
          if hasattr(datas[0], 'qfrc_smooth') else 0}
  only = sys.argv[2:]          # optional: substrings of the variants to run
  for name, p in paths:
    if 'FAULTS' in name or (only and not any(
        name == o[1:] if o.startswith('=') else o in name for o in only)):
      continue
    hm = W.HipModel(p)
    hb = W.HipBatch(hm, nenv)
    hb.set_aux_outputs(True)
    hb.set_state(qpos.T, qvel.T, np.zeros((model.nv, nenv)))
    hb.step_host(None, 1)
    acc = hb.read(W.FIELD_QACC).T
    err = np.abs(acc - ref).max(axis=0)
    print('%-40s max|dqacc| dofs<18 %.1e  dof18 %.3e  dof19 %.3e  device qacc[18:20] env0 %r ratio18 %r'
          % (name, err[:18].max(), err[18], err[19], acc[0, 18:].tolist(),
             (acc[:, 18]/ref[:, 18]).tolist()[:3]), flush=True)
    dump['qacc ' + name] = acc.copy()
    hb.free(); hm.free()
  out = os.path.join(ROOT, 'gpurun_out', 'spill_hazard_dump.npz')
  os.makedirs(os.path.dirname(out), exist_ok=True)
  np.savez(out, **dump)


if __name__ == '__main__':
  main()
