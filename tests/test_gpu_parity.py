"""GPU parity: the HIP step (through the C ABI) against the fp64 CPU oracle.

Oracle = oracle/mjstep.c, pinned by the reference's known-answer tests
(tests/test_oracle_kat.py).  There is no golden qpos/qvel trajectory in the
reference and libmujoco cannot run here, so trajectory parity is stated
against that oracle ("parity unpinned" beyond the KATs).

Tolerances (relative error = max|a-b| / max(1, max|b|) over a state vector):
  fp64 build  teacher-forced 1 step   <= 1e-9      free-run 100 steps <= 1e-6
  fp32 build  teacher-forced 1 step: per model, ~10x what was observed on
              MI355X in round 2 (FP32_PER_STEP below; medians 9e-9 ... 1e-6)
              free-run 1000 steps: see test_north_star_1000_step_free_run
Contact-rich chaotic systems (cheetah, humanoid) are compared teacher-forced
in fp32; free-running fp32 vs fp64 trajectories separate exponentially after
the first contact-set change, as for any two fp32/fp64 runs of MuJoCo itself.
Configurations where two capsule axes intersect (closest distance 0, contact
normal defined by rounding in ANY implementation) are excluded from the
per-step fp32 statistics.
"""

import contextlib

import numpy as np
import pytest

import helpers
import kat_models
import task_formulas
from dm_control_amd import build
from dm_control_amd import codegen
from dm_control_amd import suite
from dm_control_amd import wrapper
from dm_control_amd.mjcf import compiler
from dm_control_amd.rl import control
from oracle import oracle

pytestmark = pytest.mark.gpu

W = wrapper

# fp32 teacher-forced per-step error (median, p99, max) asserted per model:
# about 10x the values observed on MI355X (round 2, both kernels; observed
# medians: cartpole 3.6e-8, cheetah 1.2e-7, humanoid 9.6e-7, walker 6.6e-7,
# pendulum 3.5e-8, acrobot 4.3e-8, hopper 2.9e-7, reacher 1.1e-7, point_mass
# 8.7e-9; observed maxima 1.1e-7 ... 1.2e-4, and 1.8e-3 for one humanoid sample
# of the two-envs-per-wave build: the maximum is set by single steps in which
# a constraint row sits within rounding of its activation threshold)
FP32_PER_STEP = {
    'cartpole': (4e-7, 1e-6, 2e-6), 'cheetah': (1.5e-6, 3e-5, 2e-4),
    'humanoid': (1e-5, 1.5e-4, 5e-3), 'walker': (8e-6, 8e-5, 5e-4),
    'pendulum': (4e-7, 1.2e-6, 2e-6), 'acrobot': (5e-7, 1.2e-6, 3e-6),
    'hopper': (3e-6, 5e-5, 3e-4), 'reacher': (1.2e-6, 1.2e-5, 3e-5),
    'point_mass': (1e-7, 2.5e-6, 2e-5)}


def _assert_fp32_per_step(name, e):
  med, p99, top = FP32_PER_STEP[name]
  assert np.median(e) <= med, (name, np.median(e))
  assert np.percentile(e, 99) <= p99, (name, np.percentile(e, 99))
  assert e.max() <= top, (name, e.max())


def _device_batch(model, task, precision, nenv, mode='auto', lds_budget=None,
                  group=64):
  hm = W.HipModel(build.build_model(model, task, precision, mode=mode,
                                    lds_budget=lds_budget, group=group))
  return hm, W.HipBatch(hm, nenv)


def _oracle_envs(model, qpos, qvel):
  om = oracle.OracleModel(model)
  datas = [oracle.OracleData(om) for _ in range(len(qpos))]
  for i, d in enumerate(datas):
    d.qpos[:] = qpos[i]
    d.qvel[:] = qvel[i]
    d.step1()
  return om, datas


def _degenerate(d, model):
  """True if a capsule-capsule contact has (numerically) intersecting axes."""
  for c in range(d.ncon):
    con = d.contact(c)
    g1, g2 = con['geom1'], con['geom2']
    if model.geom_type[g1] == 3 and model.geom_type[g2] == 3:
      if con['dist'] < -(model.geom_size[g1, 0] + model.geom_size[g2, 0]) + 1e-4:
        return True
  return False


def _teacher_forced(name, precision, nenv, steps, nsub, lds_budget=None,
                    mode=None, group=64):
  model = helpers.load_model(name)
  hm, hb = _device_batch(model, helpers.TASKS[name], precision, nenv,
                         mode or helpers.MODES[name], lds_budget, group)
  qpos, qvel = helpers.initial_states(model, name, nenv, seed=7)
  om, datas = _oracle_envs(model, qpos, qvel)
  rs = np.random.RandomState(11)
  errs = []
  for _ in range(steps):
    oq = np.array([d.qpos.copy() for d in datas])
    ov = np.array([d.qvel.copy() for d in datas])
    ow = np.array([d.qacc_warmstart.copy() for d in datas])
    skip = np.array([_degenerate(d, model) for d in datas])
    hb.set_state(oq.T, ov.T, ow.T)
    ctrl = rs.uniform(-1, 1, (nenv, model.nu))
    hb.step_host(ctrl, nsub)
    q = hb.read(W.FIELD_QPOS).T.astype(np.float64)
    v = hb.read(W.FIELD_QVEL).T.astype(np.float64)
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrl[i]
      for _ in range(nsub):
        skip[i] |= _degenerate(d, model)
        d.physics_step()
    nq = np.array([d.qpos.copy() for d in datas])
    nv = np.array([d.qvel.copy() for d in datas])
    e = np.maximum(helpers.rel_err(q, nq), helpers.rel_err(v, nv))
    errs.append(e[~skip])
  assert not hb.read(W.FIELD_WARN).any()
  return np.concatenate(errs)


@pytest.mark.parametrize('name,nsub', [('cartpole', 1), ('cheetah', 1),
                                       ('humanoid', 5), ('walker', 10),
                                       ('pendulum', 1), ('acrobot', 1),
                                       ('hopper', 4), ('reacher', 1),
                                       ('point_mass', 1)])
def test_fp64_build_matches_oracle_per_step(name, nsub):
  e = _teacher_forced(name, 'f64', nenv=64, steps=12, nsub=nsub)
  assert e.max() <= 1e-9, e.max()


@pytest.mark.parametrize('name,nsub', [('cartpole', 1), ('cheetah', 1),
                                       ('humanoid', 5), ('walker', 10),
                                       ('pendulum', 1), ('acrobot', 1),
                                       ('hopper', 4), ('reacher', 1),
                                       ('point_mass', 1)])
def test_fp32_build_matches_oracle_per_step(name, nsub):
  e = _teacher_forced(name, 'f32', nenv=128, steps=12, nsub=nsub)
  print('OBSERVED fp32 per-step %s: median %.2e p99 %.2e max %.2e'
        % (name, np.median(e), np.percentile(e, 99), e.max()))
  _assert_fp32_per_step(name, e)


@pytest.mark.parametrize('name', ['cartpole', 'cheetah'])
def test_mixed_build_matches_oracle_per_step(name):
  """precision='mixed' (fp32 arithmetic, qpos/qvel carried as fp64 (high, low)
  pairs; one-env-per-lane kernel): a step from an uploaded state is an fp32
  step -- same per-step bounds as the fp32 build."""
  e = _teacher_forced(name, 'mixed', nenv=128, steps=12, nsub=1)
  print('OBSERVED mixed per-step %s: median %.2e p99 %.2e max %.2e'
        % (name, np.median(e), np.percentile(e, 99), e.max()))
  _assert_fp32_per_step(name, e)


@pytest.mark.parametrize('name,nsub,group', [
    ('cartpole', 1, 64), ('cheetah', 1, 64), ('walker', 10, 64),
    ('hopper', 4, 64), ('point_mass', 1, 64), ('cheetah', 1, 32),
    ('humanoid', 5, 32), ('humanoid', 5, 64), ('humanoid', 5, 128),
    ('cheetah', 1, 128), ('walker', 10, 128), ('hopper', 4, 128)])
def test_several_lanes_per_env_build_matches_oracle(name, nsub, group):
  """csrc/dmc_coop.hip (mode='coop': a group of lanes per env, working set in
  LDS) on models whose default is the one-lane kernel, with two envs per wave
  (group 32), and with the second wavefront that builds the constraint rows
  and factorises M + h D meanwhile (group 128, the humanoid's fp32 default).
  Odd batch: the last workgroup is partially filled."""
  if group >= 64:
    e = _teacher_forced(name, 'f64', nenv=33, steps=10, nsub=nsub, mode='coop',
                        group=group)
    assert e.max() <= 1e-9, e.max()
  e = _teacher_forced(name, 'f32', nenv=65, steps=10, nsub=nsub, mode='coop',
                      group=group)
  print('OBSERVED fp32 per-step coop %s G=%d: median %.2e p99 %.2e max %.2e'
        % (name, group, np.median(e), np.percentile(e, 99), e.max()))
  _assert_fp32_per_step(name, e)


@pytest.mark.parametrize('mode,group', [('auto', 64), ('unrolled', 64), ('coop', 64),
                                        ('coop', 128)])
def test_touch_sensors_match_oracle(mode, group):
  """mjSENS_TOUCH (hopper toe / heel): the device reading after one physics
  step against the oracle's contact forces and the same ray-in-zone rule,
  evaluated between mj_step2 and mj_step1 as mj_sensorAcc does."""
  model = helpers.load_model('hopper')
  nenv = 48
  # the fp64 unrolled one-lane build spills 369 SGPRs: `auto` ships the rolled
  # form; the unrolled one is compared too, under the test-only override
  with (build.allow_overbudget() if mode == 'unrolled' else contextlib.nullcontext()):
    hm, hb = _device_batch(model, helpers.TASKS['hopper'], 'f64', nenv, mode, group=group)
  qpos, qvel = helpers.initial_states(model, 'hopper', nenv, seed=2)
  om, datas = _oracle_envs(model, qpos, qvel)
  rs = np.random.RandomState(0)
  names = ['touch_toe', 'touch_heel']
  adr = [int(model.sensor_adr[model.names['sensor'].index(n)]) for n in names]
  seen = 0
  for _ in range(30):
    oq = np.array([d.qpos.copy() for d in datas])
    ov = np.array([d.qvel.copy() for d in datas])
    ow = np.array([d.qacc_warmstart.copy() for d in datas])
    hb.set_state(oq.T, ov.T, ow.T)
    ctrl = rs.uniform(-1, 1, (nenv, model.nu))
    hb.step_host(ctrl, 1)
    got = hb.read(W.FIELD_SENSORDATA).astype(np.float64)
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrl[i]
      d.step2()
      want = [helpers.oracle_touch(model, d, n) for n in names]
      d.step1()
      for k in range(2):
        np.testing.assert_allclose(got[adr[k], i], want[k], rtol=1e-6, atol=1e-6)
        seen += want[k] > 0
  assert seen > 50      # feet were on the floor
  hb.free()


@pytest.mark.parametrize('kernel', [{}, {'build_mode': 'coop', 'group': 128}])
def test_touch_after_reset_of_a_hopper_in_contact(kernel):
  """`after_reset` (what produces the FIRST TimeStep's observation): the
  reference runs mj_forward with actuation disabled there (engine.py:283-295),
  so the touch sensors hold the contact forces of the start pose.  Poses with
  the foot in the floor, through `Physics.after_reset` -- the launch that also
  counts contacts -- against the oracle's forward pass."""
  nenv = 64
  env = suite.load('hopper', 'stand', task_kwargs={'random': 3},
                   environment_kwargs=dict(kernel, batch_size=nenv, precision='f64'))
  ts = env.reset()
  assert ts.first() and ts.observation['touch'].shape == (nenv, 2)
  physics, task = env.physics, env.task
  model = physics.model
  qpos, qvel = helpers.initial_states(model, 'hopper', nenv, seed=2)
  physics.set_state(np.hstack([qpos, qvel]))
  physics.after_reset()
  touch = task.get_observation(physics)['touch']
  want = np.zeros((nenv, 2))
  ncon = np.zeros(nenv, int)
  for i in range(nenv):
    p = oracle.OraclePhysics(model)
    p.reset()
    p.data.qpos[:] = qpos[i]
    p.data.qvel[:] = qvel[i]
    p.data.ctrl[:] = 1
    p.after_reset()
    ncon[i] = p.data.ncon
    want[i] = [helpers.oracle_touch(model, p.data, n)
               for n in ('touch_toe', 'touch_heel')]
  assert (want > 0).sum() > nenv//4            # many of these poses press on the floor
  np.testing.assert_allclose(touch, np.log1p(want), rtol=1e-6, atol=1e-9)
  np.testing.assert_array_equal(np.asarray(physics.data.ncon), ncon)
  physics.free()


@pytest.mark.parametrize('lds_budget', [64*1024, 36*1024])
def test_high_occupancy_variants_match_oracle(lds_budget):
  """The code objects `build.lds_budget_for` picks for batches > 16384 envs
  (smaller LDS row store, more rows in the HBM overflow tier) give the same
  step: fp32 cheetah per-step parity, and the same short free run as the
  default build up to fp32 rounding (the compiler contracts a few
  multiply-adds differently between the variants, so not bit-identical) in
  all but a rare env."""
  e = _teacher_forced('cheetah', 'f32', nenv=128, steps=12, nsub=1,
                      lds_budget=lds_budget)
  _assert_fp32_per_step('cheetah', e)
  model = helpers.load_model('cheetah')
  qpos, qvel = helpers.initial_states(model, 'cheetah', 256, seed=3)
  ctrl = np.random.RandomState(5).uniform(-1, 1, (12, 256, model.nu))
  out = []
  for budget in (None, lds_budget):
    hm, hb = _device_batch(model, helpers.TASKS['cheetah'], 'f32', 256,
                           lds_budget=budget)
    hb.set_state(qpos.T, qvel.T)
    for t in range(12):
      hb.step_host(ctrl[t], 1)
    out.append((hb.read(W.FIELD_QPOS), hb.read(W.FIELD_QVEL)))
  # per env; an env in which a contact row sits within rounding of its threshold
  # takes a different branch in one of the builds and separates (observed: 1 of
  # 256 in some draws) -- at most 1 % of the envs may, the rest agree to rounding
  dq = np.abs(out[0][0] - out[1][0]).max(axis=0)
  dv = np.abs(out[0][1] - out[1][1]).max(axis=0)
  assert (dq > 2e-5).sum() <= 2 and (dv > 2e-3).sum() <= 2, (np.sort(dq)[-4:], np.sort(dv)[-4:])
  assert np.median(np.abs(out[0][0] - out[1][0])) <= 1e-7
  assert build.lds_budget_for(8192) > build.lds_budget_for(32768) > \
      build.lds_budget_for(65536)


def _free_run(name, precision, nenv, steps, nsub):
  model = helpers.load_model(name)
  hm, hb = _device_batch(model, helpers.TASKS[name], precision, nenv,
                         helpers.MODES[name])
  qpos, qvel = helpers.initial_states(model, name, nenv, seed=3)
  om, datas = _oracle_envs(model, qpos, qvel)
  hb.set_state(qpos.T, qvel.T)
  rs = np.random.RandomState(5)
  skip = np.zeros(nenv, bool)
  for _ in range(steps):
    ctrl = rs.uniform(-1, 1, (nenv, model.nu))
    hb.step_host(ctrl, nsub)
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrl[i]
      for _ in range(nsub):
        skip[i] |= _degenerate(d, model)
        d.physics_step()
  q = hb.read(W.FIELD_QPOS).T.astype(np.float64)
  v = hb.read(W.FIELD_QVEL).T.astype(np.float64)
  nq = np.array([d.qpos.copy() for d in datas])
  nv = np.array([d.qvel.copy() for d in datas])
  return helpers.rel_err(q, nq)[~skip], helpers.rel_err(v, nv)[~skip]


@pytest.mark.parametrize('name,nsub,steps', [('cartpole', 1, 1000),
                                             ('cheetah', 1, 100),
                                             ('humanoid', 5, 20)])
def test_fp64_free_run(name, nsub, steps):
  """Free-running fp64 device trajectories stay on the oracle's."""
  eq, ev = _free_run(name, 'f64', 32, steps, nsub)
  assert np.median(eq) <= 1e-9 and eq.max() <= 1e-6, (np.median(eq), eq.max())
  assert ev.max() <= 1e-5, ev.max()


def test_fp32_cartpole_free_run_1000_steps():
  """BASELINE configs[1]: the smooth system where 1000-step parity is
  meaningful in fp32 (hard part 2)."""
  eq, ev = _free_run('cartpole', 'f32', 256, 1000, 1)
  print('OBSERVED cartpole fp32 1000-step free run: qpos rel err median %.2e p90 %.2e '
        'max %.2e' % (np.median(eq), np.percentile(eq, 90), eq.max()))
  # random torques drive some poles slowly through the upright (unstable)
  # equilibrium, where any rounding difference is amplified: the bulk of the
  # batch stays at fp32 resolution, the tail is reported, not hidden.
  # (observed on MI355X, round 2: median 1.0e-5, p90 1.5e-4, max 0.2)
  assert np.median(eq) <= 1e-4, np.median(eq)
  assert np.percentile(eq, 90) <= 1.5e-3, np.percentile(eq, 90)
  assert np.mean(eq <= 1e-4) >= 0.75, np.mean(eq <= 1e-4)
  assert eq.max() <= 0.3, eq.max()


def test_mixed_cartpole_free_run_1000_steps():
  """The same 256 adversarial starts and torques with precision='mixed': the
  low state words remove the state-rounding part of the fp32 error (the study
  of DESIGN 4.3: p99 3.4e-4 -> 2.7e-5 on task-like starts), so the bulk must
  be no worse than the fp32 build's and the same asserts hold."""
  eq, _ = _free_run('cartpole', 'mixed', 256, 1000, 1)
  ef, _ = _free_run('cartpole', 'f32', 256, 1000, 1)
  print('OBSERVED cartpole mixed 1000-step free run: qpos rel err median %.2e p90 %.2e '
        'max %.2e share<=1e-4 %.3f (fp32: %.2e %.2e %.2e %.3f)' % (
            np.median(eq), np.percentile(eq, 90), eq.max(), np.mean(eq <= 1e-4),
            np.median(ef), np.percentile(ef, 90), ef.max(), np.mean(ef <= 1e-4)))
  # observed on MI355X (round 3): mixed median 2.6e-6, p90 3.5e-5, 94.5 % within
  # 1e-4; fp32 on the same starts 9.8e-6, 1.7e-4, 86.3 %
  assert np.median(eq) <= max(1.5*np.median(ef), 1e-6)
  assert np.mean(eq <= 1e-4) >= np.mean(ef <= 1e-4) - 0.02
  assert np.median(eq) <= 3e-5 and np.percentile(eq, 90) <= 3.5e-4
  assert np.mean(eq <= 1e-4) >= 0.85


@pytest.mark.parametrize('key,mode', [(k, 'auto') for k in sorted(kat_models.GPU_MODELS)] +
                         [('primitives', 'unrolled'), ('primitives', 'rolled')])
def test_known_answer_models_on_device(key, mode):
  """Box/sphere/capsule primitives and free joints through the HIP path.

  The 20-dof `primitives` model is also run in both explicit build modes.  Its
  fp64 UNROLLED code object spills ~2600 registers -- far beyond the budget
  within which the product ships such a build (`build_model(mode='unrolled')`
  raises; asserted below) -- and builds of this kind have twice produced a
  wrong trajectory on the GPU after semantics-preserving source edits
  (tools/spill_hazard/).  It is built here with the explicit
  override was a canary whose outcome was reported, not asserted: with the
  sources of round 2 it was exact, mid round 3 it was wrong again (rel. error 1.2
  after 150 steps), and with the final sources of round 3 the process ABORTED
  inside the step (profiles/r03_overbudget_canary_abort.txt).  A build that can
  take the process down is not run on a shared GPU again: what is left of the
  canary is the assertion that the product refuses the build."""
  model = compiler.from_xml_string(kat_models.GPU_MODELS[key])
  nenv = 32
  if (key, mode) == ('primitives', 'unrolled'):
    import os
    assert os.environ.get('DMC_ALLOW_OVERBUDGET') != '1'
    with pytest.raises(RuntimeError, match='spills'):   # the product path refuses it
      build.build_model(model, codegen.TASK_NONE, 'f64', mode='unrolled')
    return
  hm, hb = _device_batch(model, codegen.TASK_NONE, 'f64', nenv, mode)
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
  om, datas = _oracle_envs(model, qpos, qvel)
  hb.set_state(qpos.T, qvel.T)
  touched = False
  for _ in range(150):
    ctrl = rs.uniform(-1, 1, (nenv, max(model.nu, 1)))[:, :model.nu]
    hb.step_host(ctrl if model.nu else None, 1)
    for i, d in enumerate(datas):
      if model.nu:
        d.ctrl[:] = ctrl[i]
      d.physics_step()
      touched |= d.ncon > 0
  q = hb.read(W.FIELD_QPOS).T[:, :model.nq]
  nq = np.array([d.qpos.copy() for d in datas])
  assert helpers.rel_err(q, nq).max() <= 1e-6
  assert touched                              # the contact path was exercised
  if key == 'readme_box':
    # K1 on the device: env 0 starts at rest, settles at the README height
    hb.set_aux_outputs(True)
    for _ in range(350):
      hb.step_host(None, 1)
    z = hb.read(W.FIELD_XPOS).T[0].reshape(-1, 3)[1, 2]
    assert abs((z - 0.1) - 0.19996362 + 0.1) < 1e-7 or abs(z - 0.19996362) < 1e-7


def test_full_size_properties_cheetah_8192():
  """BASELINE configs[2] size: properties that need no oracle.

  determinism (bitwise, suite_test.py:169-185), batch independence (an env's
  result does not depend on its lane/neighbours), reward in [0, 1]
  (suite_test.py:89-94), finite observations matching the spec (:149-167),
  observation == state and reward == formula(read-back speed).
  """
  nenv, steps = 8192, 40
  model = helpers.load_model('cheetah')
  qpos, qvel = helpers.initial_states(model, 'cheetah', nenv, seed=1)
  rs = np.random.RandomState(9)
  ctrls = rs.uniform(-1, 1, (steps, nenv, model.nu)).astype(np.float32)
  perm = rs.permutation(nenv)
  outs = []
  for order in (None, None, perm):
    hm, hb = _device_batch(model, codegen.TASK_CHEETAH, 'f32', nenv)
    q0, v0 = (qpos, qvel) if order is None else (qpos[order], qvel[order])
    hb.set_state(q0.T, v0.T)
    for t in range(steps):
      c = ctrls[t] if order is None else ctrls[t][order]
      hb.step_host(c, 1)
      r = hb.read(W.FIELD_REWARD)
      assert np.all((r >= 0) & (r <= 1))
    outs.append((hb.read(W.FIELD_QPOS), hb.read(W.FIELD_QVEL),
                 hb.read(W.FIELD_OBS), hb.read(W.FIELD_REWARD),
                 hb.read(W.FIELD_SENSORDATA), hb.read(W.FIELD_WARN)))
    hb.free()
  a, b, c = outs
  for x, y in zip(a, b):
    assert np.array_equal(x, y)                       # bitwise determinism
  assert np.array_equal(a[0][:, perm], c[0])          # lane independence
  assert np.array_equal(a[3][perm], c[3])
  qp, qv, obs, rew, sens, warn = a
  assert not warn.any() and np.isfinite(obs).all()
  np.testing.assert_array_equal(obs[:, :8], qp[1:].T)
  np.testing.assert_array_equal(obs[:, 8:], qv.T)
  expect = np.array([task_formulas.cheetah_reward(float(s)) for s in sens[0]])
  np.testing.assert_allclose(rew, expect, atol=1e-6)


def test_fused_task_outputs_match_reference_formulas():
  """Device reward/observation vs the golden-validated host formulas."""
  for domain, task, nenv in (('cartpole', 'swingup', 64),
                             ('cartpole', 'balance_sparse', 64),
                             ('humanoid', 'walk', 32),
                             ('humanoid', 'stand', 32),
                             ('walker', 'run', 32), ('walker', 'stand', 32),
                             ('pendulum', 'swingup', 64),
                             ('acrobot', 'swingup', 64),
                             ('acrobot', 'swingup_sparse', 64),
                             ('hopper', 'stand', 32), ('hopper', 'hop', 32),
                             ('reacher', 'easy', 64), ('reacher', 'hard', 64),
                             ('point_mass', 'easy', 64)):
    env = suite.load(domain, task, task_kwargs={'random': 4},
                     environment_kwargs={'batch_size': nenv})
    physics = env.physics
    spec = env.action_spec()
    rs = np.random.RandomState(0)
    ts = env.reset()
    assert ts.first() and ts.reward is None and ts.discount is None
    for _ in range(5):
      action = rs.uniform(spec.minimum, spec.maximum, (nenv,) + spec.shape)
      ts = env.step(action)
    assert ts.mid() and ts.reward.shape == (nenv,)
    assert np.array_equal(ts.discount, np.ones(nenv))
    xmat = np.asarray(physics.data.xmat).reshape(nenv, -1, 9)
    xpos = np.asarray(physics.data.xpos).reshape(nenv, -1, 3)
    qvel = np.asarray(physics.data.qvel)
    qpos = np.asarray(physics.data.qpos)
    ctrl = np.asarray(physics.data.ctrl)
    for i in range(nenv):
      if domain == 'cartpole':
        want = task_formulas.cartpole_reward(
            qpos[i, 0], xmat[i, 2:, 8], ctrl[i, 0], qvel[i, 1:],
            sparse='sparse' in task)
        np.testing.assert_allclose(ts.observation['position'][i],
                                   [qpos[i, 0], xmat[i, 2, 8], xmat[i, 2, 2]],
                                   atol=1e-6)
      elif domain == 'walker':
        m = physics.model
        torso = m.name2id('torso', 'body')
        want = task_formulas.walker_reward(
            xpos[i, torso, 2], xmat[i, torso, 8],
            np.asarray(physics.data.sensordata)[i, 0],
            {'run': 8, 'stand': 0}[task])
        np.testing.assert_allclose(ts.observation['orientations'][i],
                                   xmat[i, 1:][:, [0, 2]].ravel(), atol=1e-6)
        np.testing.assert_allclose(ts.observation['height'][i],
                                   xpos[i, torso, 2])
      elif domain == 'point_mass':
        dist = np.linalg.norm(np.array([0, 0, .01]) - xpos[i, 1])
        want = task_formulas.point_mass_reward(dist, ctrl[i])
        np.testing.assert_allclose(physics.mass_to_target_dist()[i], dist, atol=1e-7)
      elif domain == 'reacher':
        m = physics.model
        finger = xpos[i, m.name2id('finger', 'body'), :2]
        target = physics.target_position()[i]
        assert 0.05 - 1e-6 <= np.linalg.norm(target) <= 0.2 + 1e-6
        want = task_formulas.reacher_reward(
            np.linalg.norm(target - finger), {'easy': .05, 'hard': .015}[task])
        np.testing.assert_allclose(ts.observation['to_target'][i], target - finger,
                                   atol=1e-6)
        np.testing.assert_allclose(physics.finger_to_target_dist()[i],
                                   np.linalg.norm(target - finger), atol=1e-6)
        if abs(np.linalg.norm(target - finger) - ({'easy': .05, 'hard': .015}[task] + .01)) < 1e-5:
          continue     # on the rim of the indicator reward
      elif domain == 'hopper':
        sens = np.asarray(physics.data.sensordata)[i]
        want = task_formulas.hopper_reward(physics.height()[i], sens[0], ctrl[i],
                                           task == 'hop')
        np.testing.assert_allclose(ts.observation['touch'][i],
                                   np.log1p(sens[3:5]), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ts.observation['position'][i],
                                   np.asarray(physics.data.qpos)[i, 1:])
      elif domain == 'acrobot':
        tip = xpos[i, 2] + xmat[i, 2].reshape(3, 3).dot([0, 0, 1.0])
        dist = np.linalg.norm(np.array([0, 0, 4.0]) - tip)
        want = task_formulas.acrobot_reward(dist, task == 'swingup_sparse')
        np.testing.assert_allclose(physics.to_target()[i], dist, rtol=1e-6)
        np.testing.assert_allclose(
            ts.observation['orientations'][i],
            [xmat[i, 1, 2], xmat[i, 2, 2], xmat[i, 1, 8], xmat[i, 2, 8]],
            atol=1e-6)
      elif domain == 'pendulum':
        want = task_formulas.pendulum_reward(xmat[i, 1, 8])
        np.testing.assert_allclose(ts.observation['orientation'][i],
                                   [xmat[i, 1, 8], xmat[i, 1, 2]], atol=1e-6)
      else:
        m = physics.model
        com = np.asarray(physics.data.sensordata)[i, :3]
        want = task_formulas.humanoid_reward(
            xpos[i, m.name2id('head', 'body'), 2],
            xmat[i, m.name2id('torso', 'body'), 8], ctrl[i], com,
            1 if task == 'walk' else 0)
        torso = m.name2id('torso', 'body')
        r = xmat[i, torso].reshape(3, 3)
        ext = []
        for side in ('left_', 'right_'):
          for limb in ('hand', 'foot'):
            d = xpos[i, m.name2id(side + limb, 'body')] - xpos[i, torso]
            ext.append(d.dot(r))
        np.testing.assert_allclose(ts.observation['extremities'][i],
                                   np.hstack(ext), atol=2e-6)
        np.testing.assert_allclose(ts.observation['head_height'][i],
                                   xpos[i, m.name2id('head', 'body'), 2])
        np.testing.assert_allclose(ts.observation['torso_vertical'][i],
                                   r[2], atol=1e-6)
      np.testing.assert_allclose(ts.reward[i], want, rtol=2e-4, atol=1e-7)
    physics.free()


def test_suite_environment_contract_unbatched():
  """suite.load drop-in: unbatched env behaves like the reference's
  (loader_test.py:23-42, suite_test.py:149-167, control.py:110-123)."""
  env = suite.load('cartpole', 'balance', task_kwargs={'random': 0,
                                                       'time_limit': 0.05})
  assert isinstance(env, control.Environment)
  spec = env.action_spec()
  assert spec.shape == (1,) and spec.minimum[0] == -1 and spec.maximum[0] == 1
  ts = env.reset()
  obs_spec = env.observation_spec()
  assert list(ts.observation) == ['position', 'velocity']
  assert ts.observation['position'].shape == (3,) == obs_spec['position'].shape
  assert ts.observation['velocity'].dtype == np.float64
  types = []
  for _ in range(7):
    ts = env.step(np.zeros(1))
    types.append(int(ts.step_type))
    if ts.reward is not None:
      assert isinstance(ts.reward, float) and 0 <= ts.reward <= 1
  assert types == [1, 1, 1, 1, 2, 0, 1]      # 5 steps to the limit, then reset
  flat = suite.load('cheetah', 'run', task_kwargs={'random': 1},
                    environment_kwargs={'flat_observation': True})
  ts = flat.reset()
  assert ts.observation['observations'].shape == (17,)
  assert flat.physics.time() == 0.0          # 200 settle steps, then time = 0
  assert abs(np.asarray(flat.physics.data.qvel)).max() < 5.0
  with pytest.raises(ValueError):
    suite.load('cheetah', 'run', environment_kwargs={
        'n_sub_steps': 2, 'control_timestep': 0.02})


def test_initial_state_recipes_and_seeding():
  """Same seed, same episode (suite_test.py:169-185, 280-288); host recipes
  follow the reference's RandomState call order."""
  def first_state(seed, **kw):
    env = suite.load('cartpole', 'swingup', task_kwargs={'random': seed},
                     environment_kwargs=dict(batch_size=4, **kw))
    env.reset()
    s = np.asarray(env.physics.get_state())
    env.physics.free()
    return s
  a, b, c = first_state(5), first_state(5), first_state(6)
  assert np.array_equal(a, b) and not np.array_equal(a, c)
  rs = np.random.RandomState(5)   # cartpole.py:186-194 call order, instance 0
  expect = [.01*rs.randn(), np.pi + .01*rs.randn()]
  rs.randn(0)
  expect += list(0.01*rs.randn(2))
  np.testing.assert_allclose(a[0], expect, rtol=1e-6)
  d = first_state(5, device_init=True)
  assert abs(d[:, 1] - np.pi).max() < 0.1 and len(set(d[:, 0])) == 4
  env = suite.load('humanoid', 'stand', task_kwargs={'random': 2},
                   environment_kwargs={'batch_size': 16, 'device_init': True})
  env.reset()
  assert not np.any(env.physics.data.ncon)        # collision-free start
  quat = np.asarray(env.physics.data.qpos)[:, 3:7]
  np.testing.assert_allclose(np.linalg.norm(quat, axis=1), 1, atol=1e-5)
  env.physics.free()


def test_bad_state_sets_warning_bits_and_raises():
  """engine_test.py:400-436: divergence -> PhysicsError naming the warning."""
  env = suite.load('cartpole', 'balance', task_kwargs={'random': 0},
                   environment_kwargs={'batch_size': 8})
  env.reset()
  physics = env.physics
  state = np.asarray(physics.get_state())
  state[3, 0] = np.nan
  physics.set_state(state)
  with pytest.raises(control.PhysicsError) as err:
    physics.step()
  assert 'mjWARN_BADQPOS' in str(err.value)
  mask = physics.data.warning_mask
  assert mask[3] == 16 and not mask[[0, 1, 2, 4, 5, 6, 7]].any()
  np.testing.assert_array_equal(np.asarray(physics.data.qpos)[3],
                                physics.model.qpos0)   # mj_resetData semantics
  with physics.suppress_physics_errors():
    physics.set_control(np.full((8, 1), np.nan))
    physics.step()
  assert physics.data.warning_mask[0] & 128           # mjWARN_BADCTRL
  physics.free()


def test_get_set_state_copy_round_trip():
  """engine_test.py:462-484: copy, then co-step -> identical states."""
  env = suite.load('cheetah', 'run', task_kwargs={'random': 3},
                   environment_kwargs={'batch_size': 16, 'device_init': True})
  env.reset()
  p1 = env.physics
  p2 = p1.copy()
  rs = np.random.RandomState(0)
  for _ in range(5):
    a = rs.uniform(-1, 1, (16, 6))
    for p in (p1, p2):
      p.set_control(a)
      p.step()
  assert np.array_equal(np.asarray(p1.get_state()), np.asarray(p2.get_state()))
  with pytest.raises(ValueError):
    p1.set_state(np.zeros((16, 3)))
  p1.free()
  p2.free()


def test_checkpoint_round_trip_continues_bit_for_bit(tmp_path):
  """save_checkpoint / load_checkpoint (.npz of qpos, qvel, warm start, time):
  a fresh batch restored from the file continues exactly like the original."""
  def make():
    return suite.load('cheetah', 'run', task_kwargs={'random': 3},
                      environment_kwargs={'batch_size': 96})
  env = make()
  env.reset()
  rs = np.random.RandomState(1)
  acts = rs.uniform(-1, 1, (30, 96, 6))
  for a in acts[:12]:
    env.step(a)
  path = str(tmp_path/'state.npz')
  env.physics.save_checkpoint(path)
  for a in acts[12:]:
    ts = env.step(a)
  other = make()
  other.reset()
  other.physics.load_checkpoint(path)
  for a in acts[12:]:
    ts2 = other.step(a)
  for k in ts.observation:
    np.testing.assert_array_equal(ts.observation[k], ts2.observation[k])
  np.testing.assert_array_equal(ts.reward, ts2.reward)
  np.testing.assert_array_equal(np.asarray(env.physics.data.time),
                                np.asarray(other.physics.data.time))
  small = suite.load('cheetah', 'run', environment_kwargs={'batch_size': 4})
  with pytest.raises(ValueError):
    small.physics.load_checkpoint(path)
  wide = suite.load('cheetah', 'run', environment_kwargs={
      'batch_size': 96, 'precision': 'f64'})
  with pytest.raises(ValueError):
    wide.physics.load_checkpoint(path)              # precision mismatch
  for e in (env, other, small, wide):
    e.physics.free()


def test_checkpoint_of_a_mixed_precision_batch_and_old_format_files(tmp_path):
  """precision='mixed': the checkpoint holds the fp32 words of the state, the
  low words restart at zero, so the restored batch starts from the same fp32
  state and continues to fp32 rounding (documented in save_checkpoint).  A file
  without the newer fields is refused with a message naming them."""
  def make():
    return suite.load('cartpole', 'swingup', task_kwargs={'random': 5},
                      environment_kwargs={'batch_size': 64, 'precision': 'mixed'})
  env = make()
  env.reset()
  rs = np.random.RandomState(1)
  acts = rs.uniform(-1, 1, (40, 64, 1))
  for a in acts[:20]:
    env.step(a)
  path = str(tmp_path/'mixed.npz')
  env.physics.save_checkpoint(path, step_count=env.step_count)
  q_saved = np.asarray(env.physics.data.qpos).copy()
  other = make()
  other.reset()
  assert other.physics.load_checkpoint(path) == 20
  np.testing.assert_array_equal(np.asarray(other.physics.data.qpos), q_saved)
  for a in acts[20:]:
    ts, ts2 = env.step(a), other.step(a)
  for k in ts.observation:
    np.testing.assert_allclose(ts.observation[k], ts2.observation[k], atol=2e-5)
  with np.load(path) as z:
    old = {k: z[k] for k in ('model_hash', 'precision', 'step_count', 'qpos',
                             'qvel', 'qacc_warmstart', 'time')}
  np.savez(str(tmp_path/'old.npz'), **old)
  with pytest.raises(ValueError, match='lacks ctrl'):
    other.physics.load_checkpoint(str(tmp_path/'old.npz'))
  env.physics.free()
  other.physics.free()


@pytest.mark.parametrize('domain,task', [('reacher', 'hard'), ('point_mass', 'hard')])
def test_checkpoint_restores_task_data_control_and_bookkeeping(domain, task, tmp_path):
  """The per-instance task data (reacher target, point_mass actuation
  directions) is part of the dynamics and of the reward: a batch created with
  ANOTHER seed continues exactly like the original once the checkpoint is
  loaded; the last control (re-applied by a step without a new action), the
  episode return and the step count come back too."""
  n = 48
  env = suite.load(domain, task, task_kwargs={'random': 3},
                   environment_kwargs={'batch_size': n})
  env.reset()
  rs = np.random.RandomState(1)
  acts = rs.uniform(-1, 1, (20, n, env.physics.model.nu))
  for a in acts[:8]:
    env.step(a)
  path = str(tmp_path/'state')                      # no extension given
  env.physics.save_checkpoint(path, step_count=env.step_count)
  other = suite.load(domain, task, task_kwargs={'random': 99},
                     environment_kwargs={'batch_size': n})
  other.reset()
  assert not np.array_equal(env.physics.batch.read(W.FIELD_TASKDATA),
                            other.physics.batch.read(W.FIELD_TASKDATA))
  other.step_count = other.physics.load_checkpoint(path)
  assert other.step_count == 8
  for f in (W.FIELD_TASKDATA, W.FIELD_CTRL, W.FIELD_RETURN, W.FIELD_QPOS):
    np.testing.assert_array_equal(env.physics.batch.read(f),
                                  other.physics.batch.read(f))
  # a physics step without a new control re-applies the stored one
  env.physics.step()
  other.physics.step()
  np.testing.assert_array_equal(np.asarray(env.physics.data.qpos),
                                np.asarray(other.physics.data.qpos))
  for a in acts[8:]:
    ts, ts2 = env.step(a), other.step(a)
  for k in ts.observation:
    np.testing.assert_array_equal(ts.observation[k], ts2.observation[k])
  np.testing.assert_array_equal(ts.reward, ts2.reward)
  np.testing.assert_array_equal(env.physics.batch.read(W.FIELD_RETURN),
                                other.physics.batch.read(W.FIELD_RETURN))
  for e in (env, other):
    e.physics.free()


def test_per_instance_task_data_write_and_device_init():
  """DMC_FIELD_TASKDATA (the reacher's target, which the reference rewrites in
  model.geom_pos): host writes round-trip through dmc_batch_write / read in the
  [k][nenv] presentation, and the device-side initialiser draws the same ring
  distribution (reacher.py:100-104)."""
  env = suite.load('reacher', 'easy', task_kwargs={'random': 5},
                   environment_kwargs={'batch_size': 256})
  env.reset()
  batch = env.physics.batch
  t = np.random.RandomState(0).uniform(-0.2, 0.2, (2, 256))
  batch.write(W.FIELD_TASKDATA, t)
  np.testing.assert_allclose(batch.read(W.FIELD_TASKDATA), t, atol=1e-7)
  with pytest.raises(W.Error):
    batch.write(W.FIELD_OBS, np.zeros((256, 6)))      # not writable
  with pytest.raises(ValueError):
    batch.write(W.FIELD_TASKDATA, np.zeros((3, 256)))
  ts = env.step(np.zeros((256, 2)))
  finger = np.asarray(env.physics.data.xpos).reshape(256, -1, 3)[:, 3, :2]
  np.testing.assert_allclose(ts.observation['to_target'], t.T - finger, atol=1e-6)
  env.physics.free()
  dev = suite.load('reacher', 'hard', task_kwargs={'random': 5},
                   environment_kwargs={'batch_size': 4096, 'device_init': True})
  dev.reset()
  r = np.linalg.norm(dev.physics.target_position(), axis=1)
  assert r.min() >= 0.05 - 1e-6 and r.max() <= 0.2 + 1e-6
  assert abs(r.mean() - 0.125) < 0.01 and r.std() > 0.03
  ang = np.arctan2(*dev.physics.target_position().T)
  assert abs(ang.mean()) < 0.15 and ang.std() > 1.5
  dev.physics.free()


def test_point_mass_hard_per_instance_actuation_directions():
  """point_mass 'hard' redraws the tendon coefficients (model.wrap_prm in the
  reference) per episode: each instance is stepped against an oracle model
  compiled with that instance's coefficients; the device-side initialiser
  draws unit directions that are not too parallel."""
  import copy
  env = suite.load('point_mass', 'hard', task_kwargs={'random': 2},
                   environment_kwargs={'batch_size': 8, 'precision': 'f64'})
  env.reset()
  physics = env.physics
  dirs = physics.actuation_directions()
  assert dirs.shape == (8, 2, 2)
  np.testing.assert_allclose(np.linalg.norm(dirs, axis=2), 1, atol=1e-12)
  assert np.all(np.abs(np.einsum('ek,ek->e', dirs[:, 0], dirs[:, 1])) <= 0.9)
  assert np.abs(dirs[0] - np.eye(2)).max() > 1e-3
  q0 = np.asarray(physics.data.qpos).copy()
  datas = []
  for i in range(8):
    mi = copy.copy(physics.model)
    mi.wrap_prm = dirs[i].ravel().copy()
    d = oracle.OracleData(oracle.OracleModel(mi))
    d.qpos[:] = q0[i]
    d.step1()
    datas.append(d)
  rs = np.random.RandomState(0)
  for _ in range(25):
    a = rs.uniform(-1, 1, (8, 2))
    env.step(a)
    for i, d in enumerate(datas):
      d.ctrl[:] = a[i]
      d.physics_step()
  got = np.asarray(physics.data.qpos)
  np.testing.assert_allclose(got, [d.qpos for d in datas], atol=1e-12)
  physics.free()
  dev = suite.load('point_mass', 'hard', task_kwargs={'random': 2},
                   environment_kwargs={'batch_size': 2048, 'device_init': True})
  dev.reset()
  dd = dev.physics.actuation_directions()
  np.testing.assert_allclose(np.linalg.norm(dd, axis=2), 1, atol=1e-5)
  assert np.all(np.abs(np.einsum('ek,ek->e', dd[:, 0], dd[:, 1])) <= 0.9 + 1e-5)
  assert np.abs(dd[:, 0].mean(axis=0)).max() < 0.08      # isotropic
  easy = suite.load('point_mass', 'easy', environment_kwargs={'batch_size': 4})
  easy.reset()
  np.testing.assert_array_equal(easy.physics.actuation_directions(),
                                np.tile(np.eye(2), (4, 1, 1)))
  for e in (dev, easy):
    e.physics.free()


def test_step_sequence_equals_single_steps():
  """dmc_batch_step_n (a pre-computed action sequence in device memory, one
  launch per control step issued from C) gives exactly the single-step path."""
  import torch
  model = helpers.load_model('cheetah')
  qpos, qvel = helpers.initial_states(model, 'cheetah', 512, seed=4)
  acts = torch.rand(7, 512, model.nu, device='cuda') * 2 - 1
  out = []
  for mode in ('single', 'sequence'):
    hm, hb = _device_batch(model, codegen.TASK_CHEETAH, 'f32', 512)
    hb.set_state(qpos.T, qvel.T)
    if mode == 'single':
      for t in range(7):
        hb.step_device(acts[t].data_ptr(), 1, model.nu, 1)
    else:
      hb.step_device_n(acts.data_ptr(), 1, model.nu, 512*model.nu, 7, 1)
    out.append((hb.read(W.FIELD_QPOS), hb.read(W.FIELD_OBS), hb.read(W.FIELD_RETURN)))
    hb.free()
  for a, b in zip(*out):
    np.testing.assert_array_equal(a, b)


def test_derived_frames_match_oracle():
  """data.xipos / geom_xpos / geom_xmat / site_xpos are derived on the host from
  the exported body frames; compared with the oracle's kinematics at the same
  state, plus the named indexing the suite helpers use."""
  env = suite.load('acrobot', 'swingup', task_kwargs={'random': 3},
                   environment_kwargs={'batch_size': 6, 'precision': 'f64'})
  env.reset()
  for _ in range(5):
    env.step(np.zeros((6, 1)))
  physics = env.physics
  m = physics.model
  om = oracle.OracleModel(m)
  q, v = np.asarray(physics.data.qpos), np.asarray(physics.data.qvel)
  for i in range(6):
    d = oracle.OracleData(om)
    d.qpos[:] = q[i]
    d.qvel[:] = v[i]
    d.step1()
    np.testing.assert_allclose(np.asarray(physics.data.xipos)[i], d.xipos.ravel(), atol=1e-12)
    np.testing.assert_allclose(np.asarray(physics.data.ximat)[i], d.ximat.ravel(), atol=1e-12)
    np.testing.assert_allclose(np.asarray(physics.data.geom_xpos)[i], d.geom_xpos.ravel(),
                               atol=1e-12)
    np.testing.assert_allclose(np.asarray(physics.data.geom_xmat)[i], d.geom_xmat.ravel(),
                               atol=1e-12)
  tip = physics.named.data.site_xpos['tip']
  lower = np.asarray(physics.data.xpos).reshape(6, -1, 3)[:, 2]
  zaxis = np.asarray(physics.data.xmat).reshape(6, -1, 3, 3)[:, 2, :, 2]
  np.testing.assert_allclose(tip, lower + zaxis, atol=1e-12)
  np.testing.assert_allclose(physics.named.data.site_xpos['target'],
                             np.tile([0, 0, 4.0], (6, 1)))
  assert physics.named.data.geom_xpos['lower_arm', 'z'].shape == (6,)
  with pytest.raises(ValueError):
    physics.data.geom_xpos._put(0)
  physics.free()


def test_in_process_compile_and_load_from_memory(monkeypatch):
  """`dmc_model_compile` + `dmc_model_load_data` (include/dmc_hip.h): a model
  that was never pre-built is compiled inside the C-ABI library (HIP runtime
  compilation, no hipcc executable, no file) and loaded from the memory image;
  it steps like the oracle and like the toolchain-built code object.
  `Physics.from_xml_string` takes the same route when hipcc is absent
  (`$DMC_BUILD_BACKEND=hiprtc` here)."""
  xml = kat_models.GPU_MODELS['primitives'].replace('size=".08"', 'size=".081"')
  model = compiler.from_xml_string(xml)       # a model no build step has seen
  nenv = 32
  hm = W.HipModel.from_code(build.code_object_bytes(model, 0, 'f64'))
  hb = W.HipBatch(hm, nenv)
  rs = np.random.RandomState(4)
  qpos = np.tile(model.qpos0, (nenv, 1))
  qpos[:, 2], qpos[:, 9], qpos[:, 16] = 0.11, 0.2, 0.3
  qpos[:, 2] += rs.uniform(0, 0.02, nenv)
  qvel = 0.2*rs.randn(nenv, model.nv)
  om, datas = _oracle_envs(model, qpos, qvel)
  hb.set_state(qpos.T, qvel.T)
  touched = False
  for _ in range(60):
    ctrl = rs.uniform(-1, 1, (nenv, model.nu))
    hb.step_host(ctrl, 1)
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrl[i]
      d.physics_step()
      touched |= d.ncon > 0
  q = hb.read(W.FIELD_QPOS).T
  assert touched
  assert helpers.rel_err(q, np.array([d.qpos.copy() for d in datas])).max() <= 1e-6
  hb.free()
  # the high-level route, with the toolchain driver switched off
  monkeypatch.setenv('DMC_BUILD_BACKEND', 'hiprtc')
  from dm_control_amd import engine
  physics = engine.Physics.from_xml_string(
      xml.replace('size=".081"', 'size=".082"'), batch_size=4, precision='f32')
  physics.reset()
  physics.set_control(np.zeros((4, model.nu)))
  physics.step(5)
  assert np.isfinite(np.asarray(physics.data.qpos)).all()
  physics.free()


def test_c_abi_argument_errors():
  lib = wrapper.get_lib()
  model = helpers.load_model('cartpole')
  hm, hb = _device_batch(model, codegen.TASK_CARTPOLE, 'f32', 4)
  assert lib.dmc_batch_read(hb.ptr, 99, None, 0) != 0
  buf = np.zeros(3, np.float32)
  assert lib.dmc_batch_read(hb.ptr, W.FIELD_QPOS, buf.ctypes.data, 12) != 0
  assert b'bytes' in lib.dmc_last_error()
  import ctypes
  out = ctypes.c_void_p()
  assert lib.dmc_batch_create(hm.ptr, 0, ctypes.byref(out)) != 0
  assert lib.dmc_model_load(b'/nonexistent.hsaco', 0, ctypes.byref(out)) != 0
  with pytest.raises(ValueError):
    hb.step_host(np.zeros((3, 1)))
  hb.step_host(None, 0)       # zero substeps: state unchanged, outputs refreshed
  np.testing.assert_array_equal(hb.read(W.FIELD_QPOS)[:, 0], model.qpos0)


def test_vec_env_numpy_and_torch_modes():
  """scripts/vec_env.py calling convention: stacked obs, auto-reset with
  `terminal_observation` (:346-352); torch mode aliases device buffers."""
  import torch
  from dm_control_amd import vec_env
  n = 64
  env = vec_env.VecEnv('cartpole', 'swingup', n, seed=3,
                       task_kwargs={'time_limit': 0.05})
  obs = env.reset()
  assert obs.shape == (n, 5) and env.action_dim == 1
  rs = np.random.RandomState(0)
  for t in range(5):
    obs, rew, done, infos = env.step(rs.uniform(-1, 1, (n, 1)))
    assert obs.shape == (n, 5) and rew.shape == (n,) and done.shape == (n,)
    assert done.all() == (t == 4)
  assert 'terminal_observation' in infos[0] and len(infos) == n
  assert not np.array_equal(infos[0]['terminal_observation'], obs[0])
  env.close()

  tenv = vec_env.VecEnv('cheetah', 'run', n, seed=3, torch_io=True)
  obs = tenv.reset()
  assert obs.is_cuda and tuple(obs.shape) == (n, 17)
  ref = vec_env.VecEnv('cheetah', 'run', n, seed=3,
                       environment_kwargs={'device_init': True})
  ref_obs = ref.reset()
  np.testing.assert_array_equal(obs.cpu().numpy().astype(np.float64), ref_obs)
  gen = torch.Generator(device='cuda').manual_seed(0)
  for _ in range(4):
    act = torch.rand(n, 6, device='cuda', generator=gen)*2 - 1
    obs, rew, done, infos = tenv.step(act)
    robs, rrew, rdone, _ = ref.step(act.cpu().numpy())
    torch.cuda.synchronize()
    # the same physics whether actions arrive by pointer or by host copy
    np.testing.assert_array_equal(obs.cpu().numpy().astype(np.float64), robs)
    np.testing.assert_array_equal(rew.cpu().numpy().astype(np.float64), rrew)
    assert not bool(done.any())
  # column-major action tensor: read through strides, no copy needed
  act = (torch.rand(6, n, device='cuda', generator=gen)*2 - 1).t()
  obs, _, _, _ = tenv.step(act)
  robs, _, _, _ = ref.step(act.cpu().numpy())
  np.testing.assert_array_equal(obs.cpu().numpy().astype(np.float64), robs)
  tenv.close()
  ref.close()


@pytest.mark.parametrize('name,mode,group,nsub,nenv', [
    ('cheetah', 'auto', 64, 1, 1), ('cheetah', 'auto', 64, 1, 63),
    ('cheetah', 'auto', 64, 1, 65), ('cheetah', 'auto', 64, 1, 1000),
    # several lanes per env: env = workgroup, permuted so that every XCD works
    # on a contiguous range of envs -- a bijection for any workgroup count
    ('humanoid', 'coop', 128, 5, 1), ('humanoid', 'coop', 128, 5, 7),
    ('humanoid', 'coop', 128, 5, 1003), ('cheetah', 'coop', 32, 1, 1001)])
def test_ragged_batch_sizes_match_full_batches(name, mode, group, nsub, nenv):
  """Partial last workgroup (the coalesced observation store and the early
  exit of surplus lanes) and any number of workgroups: env i of a ragged
  batch == env i of a 1024 batch, bit for bit."""
  model = helpers.load_model(name)
  qpos, qvel = helpers.initial_states(model, name, 1024, seed=5)
  rs = np.random.RandomState(3)
  ctrl = rs.uniform(-1, 1, (6, 1024, model.nu))
  outs = []
  for n in (1024, nenv):
    hm, hb = _device_batch(model, helpers.TASKS[name], 'f32', n, mode, group=group)
    hb.set_state(qpos[:n].T, qvel[:n].T)
    for t in range(6):
      hb.step_host(ctrl[t, :n], nsub)
    outs.append((hb.read(W.FIELD_QPOS)[:, :nenv], hb.read(W.FIELD_OBS)[:nenv],
                 hb.read(W.FIELD_REWARD)[:nenv], hb.read(W.FIELD_RETURN)[:nenv]))
    hb.free()
  for a, b in zip(*outs):
    np.testing.assert_array_equal(a, b)


def test_soak_full_episode_with_reset_cheetah_8192():
  """1000 control steps + the automatic reset at BASELINE size: no warnings,
  finite state, rewards in range, returns accumulated in-kernel equal the
  sum of step rewards, and the next episode starts from a settled pose."""
  import torch
  from dm_control_amd import vec_env
  n = 8192
  env = vec_env.VecEnv('cheetah', 'run', n, seed=11, torch_io=True)
  env.reset()
  gen = torch.Generator(device='cuda').manual_seed(1)
  total = torch.zeros(n, device='cuda')
  acts = [torch.rand(n, 6, device='cuda', generator=gen)*2 - 1 for _ in range(32)]
  batch = env.environment.physics.batch
  for t in range(1000):
    obs, rew, done, info = env.step(acts[t % 32])
    total += rew
    if t == 998:
      ret = torch.from_numpy(batch.read(W.FIELD_RETURN)).cuda() + 0
    if t < 999:
      assert not bool(done[0])
  assert bool(done.all()) and 'terminal_observation' in info
  torch.cuda.synchronize()
  assert not batch.read(W.FIELD_WARN).any()
  assert torch.isfinite(obs).all() and torch.isfinite(total).all()
  assert float(total.min()) >= 0 and float(total.max()) <= 1000
  # in-kernel episode return after 999 steps == host-side sum of those rewards
  np.testing.assert_allclose(ret.cpu().numpy(), (total - rew).cpu().numpy(),
                             rtol=1e-4, atol=1e-3)
  # after the auto-reset: time 0, return 0, cheetah resting (200 settle steps)
  assert not batch.read(W.FIELD_TIME).any()
  assert not batch.read(W.FIELD_RETURN).any()
  assert np.abs(batch.read(W.FIELD_QVEL)).max() < 8.0
  env.close()


def _task_like_states(model, name, nenv, rs):
  """States of the kind the tasks start from (not the adversarial ones of
  helpers.initial_states): cheetah near its rest pose, cart-pole hanging."""
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


@pytest.mark.parametrize('name', ['cartpole', 'cheetah'])
def test_north_star_1000_step_free_run(name):
  """BASELINE.json: qpos within 1e-4 rel-err of the CPU step over 1000 steps on
  identical action sequences.  Measured against the fp64 oracle, free running,
  U(-1,1) actions: fp64 build <= 1e-9 for every env; fp32 build
  cart-pole median <= 1e-5, p90 <= 1e-4 (poles lingering near the upright
  amplify rounding: max 7e-4 seen), cheetah median <= 1e-5 and >= 85 % of the
  envs <= 1e-4 (a contact-set difference separates the rest exponentially)."""
  nenv, steps = 32, 1000
  model = helpers.load_model(name)
  rs = np.random.RandomState(0)
  qpos, qvel = _task_like_states(model, name, nenv, rs)
  ctrls = rs.uniform(-1, 1, (steps, nenv, model.nu))
  om, datas = _oracle_envs(model, qpos, qvel)
  for t in range(steps):
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrls[t, i]
      d.physics_step()
  ref = np.array([d.qpos.copy() for d in datas])
  for precision in ('f64', 'f32'):
    hm, hb = _device_batch(model, helpers.TASKS[name], precision, nenv)
    hb.set_state(qpos.T, qvel.T)
    for t in range(steps):
      hb.step_host(ctrls[t], 1)
    e = helpers.rel_err(hb.read(W.FIELD_QPOS).T.astype(np.float64), ref)
    print('OBSERVED %s %s 1000-step free run: median %.2e p90 %.2e max %.2e'
          % (name, precision, np.median(e), np.percentile(e, 90), e.max()))
    assert not hb.read(W.FIELD_WARN).any()
    if precision == 'f64':
      assert e.max() <= 1e-9, e.max()
    elif name == 'cartpole':
      assert np.median(e) <= 1e-5, np.median(e)
      assert np.percentile(e, 90) <= 1e-4, np.percentile(e, 90)
      assert e.max() <= 5e-3, e.max()
    else:
      # (observed: median 9.6e-7, p90 2.7e-6; an env whose contact sequence
      # leaves the oracle's -- about 1 in 60, profiles/r02_precision_study_* --
      # separates exponentially, hence a share and not a maximum)
      assert np.median(e) <= 1e-5, np.median(e)
      assert np.percentile(e, 90) <= 3e-5, np.percentile(e, 90)
      assert np.mean(e <= 1e-4) >= 0.9, np.mean(e <= 1e-4)
    hb.free()


# control steps after which the fp32 several-lanes free run is still compared
# with a bound, and (median, max) asserted there
EARLY = {'humanoid': 10, 'walker': 10, 'hopper': 25}
# observed (median, max), MI355X round 3: humanoid 2.7e-6 / 1.5e-5 after 10 control
# steps, walker 1.8e-6 / 4.3e-6 after 10, hopper 5.8e-7 / 1.4e-6 after 25
FP32_EARLY = {'humanoid': (3e-5, 1.5e-4), 'walker': (2e-5, 5e-5), 'hopper': (6e-6, 1.5e-5)}


@pytest.mark.parametrize('name,nsub,steps', [('humanoid', 5, 60), ('walker', 10, 100),
                                             ('hopper', 4, 150)])
def test_free_run_trajectories_several_lanes_kernel(name, nsub, steps):
  """Free-running control steps of the contact-rich 3-D / planar walkers on the
  several-lanes-per-env kernel against the oracle, identical U(-1,1) action
  sequences, upright task-like start (standing on / falling onto the floor).
  fp64: the whole trajectory stays on the oracle's (<= 1e-5 after 300-600
  physics steps with contact-set changes replayed identically; these systems
  are chaotic, rounding-order differences of 1e-16 grow by orders of magnitude
  per hundred steps, so longer horizons separate in any two implementations).
  fp32: reported only."""
  nenv = 16
  model = helpers.load_model(name)
  rs = np.random.RandomState(2)
  qpos = np.tile(model.qpos0, (nenv, 1))
  qvel = 0.01*rs.randn(nenv, model.nv)
  lim = np.array([bool(model.jnt_limited[j]) and model.jnt_type[j] == 3
                  for j in range(model.njnt)])
  adr = np.asarray(model.jnt_qposadr)[lim]
  lo, hi = np.asarray(model.jnt_range)[lim].T
  qpos[:, adr] += rs.uniform(0.05*lo, 0.05*hi, (nenv, lim.sum()))
  ctrls = rs.uniform(-1, 1, (steps, nenv, model.nu))
  om, datas = _oracle_envs(model, qpos, qvel)
  contacts = 0
  mid = None
  for t in range(steps):
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrls[t, i]
      for _ in range(nsub):
        d.physics_step()
      contacts += d.ncon > 0
    if t + 1 == EARLY[name]:
      mid = np.array([d.qpos.copy() for d in datas])
  assert contacts > steps*nenv//4           # the floor was involved
  ref = np.array([d.qpos.copy() for d in datas])
  for precision in ('f64', 'f32'):
    # group 128: one env per 64 lanes plus the row-building wavefront
    hm, hb = _device_batch(model, helpers.TASKS[name], precision, nenv, 'coop',
                           group=128)
    hb.set_state(qpos.T, qvel.T)
    early = None
    for t in range(steps):
      hb.step_host(ctrls[t], nsub)
      if t + 1 == EARLY[name]:
        early = helpers.rel_err(hb.read(W.FIELD_QPOS).T.astype(np.float64), mid)
    e = helpers.rel_err(hb.read(W.FIELD_QPOS).T.astype(np.float64), ref)
    print('OBSERVED %s %s free run (several lanes per env): after %d control steps '
          'median %.2e max %.2e; after %d median %.2e max %.2e'
          % (name, precision, EARLY[name], np.median(early), early.max(),
             steps, np.median(e), e.max()))
    assert not hb.read(W.FIELD_WARN).any()
    if precision == 'f64':
      assert e.max() <= 1e-5, e.max()
    else:
      # falling, contact-rich bodies under random torques are chaotic: fp32
      # rounding (per-step error ~1e-6) grows by orders of magnitude per hundred
      # physics steps.  What fp32 can be held to is the early part of the run
      # (bounds ~10x the values observed on MI355X, round 3) ...
      med, top = FP32_EARLY[name]
      assert np.median(early) <= med, np.median(early)
      assert early.max() <= top, early.max()
      # ... and, at the full horizon, finiteness plus the median where it still
      # means something (observed: humanoid 3.5e-3, hopper 4.2e-3; the walker
      # has decorrelated by then: 0.13)
      assert np.isfinite(e).all()
      limit = {'humanoid': 4e-2, 'hopper': 5e-2}.get(name)
      if limit is not None:
        assert np.median(e) <= limit, np.median(e)
    hb.free()


@pytest.mark.parametrize('domain,task,nenv,steps', [('humanoid', 'walk', 2048, 400),
                                                    ('hopper', 'hop', 4096, 600),
                                                    ('walker', 'run', 4096, 500)])
def test_soak_several_lanes_kernel(domain, task, nenv, steps):
  """Long random-action rollouts on the several-lanes-per-env kernel (bodies
  pile up on the floor, many simultaneous contacts): no warnings (contact /
  constraint buffers never overflow, no NaN resets), finite state, rewards in
  [0, 1], in-kernel returns equal the sum of step rewards."""
  import torch
  from dm_control_amd import vec_env
  env = vec_env.VecEnv(domain, task, nenv, seed=3, torch_io=True)
  assert env.environment.physics._build_mode == 'coop'    # pylint: disable=protected-access
  env.reset()
  gen = torch.Generator(device='cuda').manual_seed(1)
  nu = env.environment.physics.model.nu
  acts = [torch.rand(nenv, nu, device='cuda', generator=gen)*2 - 1 for _ in range(16)]
  total = torch.zeros(nenv, device='cuda')
  batch = env.environment.physics.batch
  for t in range(steps):
    obs, rew, done, _ = env.step(acts[t % 16])
    total += rew
    assert not bool(done.any())
  assert bool(torch.isfinite(obs).all()) and bool(((rew >= 0) & (rew <= 1)).all())
  assert not batch.read(W.FIELD_WARN).any()
  stats = batch.read(W.FIELD_STATS)
  info = batch.model.info
  assert stats[0].max() <= info.ncon_max and stats[1].max() <= info.nefc_max
  np.testing.assert_allclose(batch.read(W.FIELD_RETURN), total.cpu().numpy(),
                             rtol=1e-4, atol=1e-3)
  print('%s-%s %d envs x %d steps: ncon max %d, nefc max %d, Newton iterations max %d'
        % (domain, task, nenv, steps, stats[0].max(), stats[1].max(), stats[2].max()))
