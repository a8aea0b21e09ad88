"""Soccer, stage 1: the physics prerequisites only.

The frozen scenes of dm_control_amd/locomotion/models -- position-controlled
CMU humanoid on a fixed pitch, and the regulation ball -- against the reference
data where the tree is present (parameter tables vs humanoid_CMU_V2019.xml),
against closed forms (position actuators, weight on the floor, condim-6 /
priority contact of the ball), and on the device against the oracle
(`-m gpu`).  Stage 2 groundwork: scenes with several walkers (nv > 64) compile, generate a
kernel header (multi-word dof sets) and step on the oracle; the checks below
are CPU-side.  Composer, observables, mocap initialiser and per-episode
recompilation are not built.
"""

import os

import numpy as np
import pytest

import helpers
from dm_control_amd import codegen
from dm_control_amd.locomotion.models import cmu_humanoid_table as T
from dm_control_amd.locomotion.models import soccer
from dm_control_amd.mjcf import compiler
from oracle import oracle

REF_XML = ('/root/reference/dm_control/locomotion/walkers/assets/'
           'humanoid_CMU_V2019.xml')


def _walker_model():
  return compiler.from_xml_string(soccer.build(1, with_ball=False))


def _ball_model():
  return compiler.from_xml_string(soccer.build(0, with_ball=True))


@pytest.mark.skipif(not os.path.exists(REF_XML), reason='reference tree absent')
def test_tables_compile_like_the_reference_xml():
  ref = compiler.from_xml_path(REF_XML)
  m = _walker_model()
  assert (m.nq, m.nv, m.nu) == (ref.nq + 7, ref.nv + 6, 56)
  for name in [b[0] for b in T.BODIES]:
    i, j = m.name2id('walker0/' + name, 'body'), ref.name2id(name, 'body')
    np.testing.assert_allclose(m.body_pos[i], ref.body_pos[j], atol=1e-12)
    np.testing.assert_allclose(m.body_quat[i], ref.body_quat[j], atol=1e-12)
    if name not in ('lhand', 'rhand'):      # ellipsoid -> equal-volume sphere
      np.testing.assert_allclose(m.body_inertia[i], ref.body_inertia[j], rtol=1e-10)
    np.testing.assert_allclose(m.body_mass[i], ref.body_mass[j], rtol=1e-10)
  for name in [j[0] for j in T.JOINTS]:
    i, j = m.name2id('walker0/' + name, 'joint'), ref.name2id(name, 'joint')
    np.testing.assert_allclose(m.jnt_axis[i], ref.jnt_axis[j], atol=1e-12)
    np.testing.assert_allclose(m.jnt_range[i], ref.jnt_range[j], atol=1e-12)
    np.testing.assert_allclose(m.jnt_stiffness[i], ref.jnt_stiffness[j])
    di, dj = m.jnt_dofadr[i], ref.jnt_dofadr[j]
    np.testing.assert_allclose(m.dof_damping[di], ref.dof_damping[dj])
    np.testing.assert_allclose(m.dof_armature[di], ref.dof_armature[dj])
  assert m.nexclude == ref.nexclude == len(T.EXCLUDES)


def test_position_actuators_follow_the_scaled_actuator_formula():
  """walkers/scaled_actuators.py: ctrl -1 / +1 ask for the ends of the joint
  range; force = kp (target - q), clipped to the force range; MuJoCo's own
  ctrl clamp applies before."""
  m = _walker_model()
  p = oracle.OraclePhysics(m)
  p.reset()
  rs = np.random.RandomState(0)
  q = m.qpos0.copy()
  hinge = {name: (rng, i) for i, (name, _, _, rng, _) in enumerate(T.JOINTS)}
  for name, (rng, i) in hinge.items():
    q[7 + i] = rs.uniform(*rng)
  ctrl = rs.uniform(-1.3, 1.3, m.nu)
  p.data.qpos[:] = q
  p.data.ctrl[:] = ctrl
  p.forward()
  for a, (name, forcerange, kp) in enumerate(T.POSITION_ACTUATORS):
    (lo, hi), i = hinge[name]
    c = np.clip(ctrl[a], -1, 1)
    target = lo + (hi - lo)*(c + 1)/2
    want = np.clip(kp*(target - q[7 + i]), *forcerange)
    dof = m.jnt_dofadr[m.name2id('walker0/' + name, 'joint')]
    np.testing.assert_allclose(p.data.qfrc_actuator[dof], want, rtol=1e-9, atol=1e-9)
  assert abs(p.data.qfrc_actuator[:6]).max() == 0        # nothing acts on the root


def test_walker_comes_to_rest_on_its_weight():
  m = _walker_model()
  assert codegen.capacities(m, codegen.collision_pairs(m))[0] >= 16
  p = oracle.OraclePhysics(m)
  p.reset()
  p.data.qpos[2] = 0.4                     # dropped from a crouch height
  p.forward()
  for _ in range(1500):
    p.step()
  assert not p.data.warning.any() and np.isfinite(p.data.qpos).all()
  assert abs(p.data.qvel).max() < 0.05
  p.data.step2()
  normal = sum(p.data.contact_force(i)[0, 0] for i in range(p.data.ncon))
  p.data.step1()
  weight = 9.81*m.body_mass.sum()
  assert p.data.ncon >= 3
  assert abs(normal - weight) < 0.02*weight, (normal, weight)


def test_ball_contact_is_condim_six_with_its_own_friction():
  """soccer_ball.py:88-96: condim 6 and priority 1 -- the ball's parameters
  win over the floor's: ten pyramid rows per contact, friction (0.7, 0.075,
  0.075); a spinning, sliding ball is slowed down by all three of them."""
  m = _ball_model()
  pairs = codegen.collision_pairs(m)
  ground = m.name2id('ground', 'geom')
  ball = m.name2id('ball', 'geom')
  mx = codegen.mix_pair(m, ground, ball)
  assert mx['dim'] == 6
  # contact friction = (slide, slide, spin, roll, roll) of the ball's (slide, spin, roll)
  np.testing.assert_allclose(mx['friction'], [0.7, 0.7, 0.075, 0.075, 0.075])
  assert (ground, ball) in pairs or (ball, ground) in pairs
  p = oracle.OraclePhysics(m)
  p.reset()
  p.data.qpos[2] = soccer.BALL['radius']
  p.data.qvel[:] = [2.0, 0, 0, 0, 0, 8.0]           # sliding along x, spinning about z
  p.forward()
  nefc_seen = 0
  for _ in range(400):
    p.step()
    nefc_seen = max(nefc_seen, p.data.nefc)
  assert nefc_seen == 10
  assert not p.data.warning.any()
  assert abs(p.data.qvel[5]) < 1.0          # torsional friction stopped the spin
  assert 0 < p.data.qvel[0] < 2.0           # rolling now, slower than it slid
  # rolling without slipping: v = omega x r  ->  v_x = omega_y * R
  assert abs(p.data.qvel[0] - p.data.qvel[4]*soccer.BALL['radius']) < 0.05


# ---------------------------------------------------------------------------
# device (fp64 for the tight comparison, fp32 at full contact capacity)
# ---------------------------------------------------------------------------
def test_two_by_two_scene_sizes_and_header():
  """BASELINE configs[4] sizes: nq 259, nv 254, nu 224; capacities
  as soccer/task.py:105-108; `disable_walker_contacts` (task.py:29-33) leaves
  walker-pitch and walker-ball pairs only; the generated header carries dof sets
  of eight 32-bit words per body."""
  xml = soccer.build(4)
  assert 'nconmax="800"' in xml and 'njmax="800"' in xml
  m = compiler.from_xml_string(xml)
  assert (m.nq, m.nv, m.nu, m.nbody) == (259, 254, 224, 130)
  allpairs = codegen.collision_pairs(m)
  quiet = compiler.from_xml_string(soccer.build(4, disable_walker_contacts=True))
  pairs = codegen.collision_pairs(quiet)
  assert len(pairs) < len(allpairs)/10
  ball = quiet.names['geom'].index('ball')
  statics = {quiet.names['geom'].index(n) for n in
             ('ground', 'wall_nx', 'wall_px', 'wall_ny', 'wall_py')}
  for g1, g2 in pairs:     # every surviving pair involves the pitch or the ball
    assert {g1, g2} & (statics | {ball})
  header = codegen.generate_header(quiet, codegen.TASK_NONE)
  assert 'constexpr int NMASKW = 8;' in header and 'constexpr int NV = 254;' in header


def test_scene_of_two_walkers_steps_like_two_single_walker_scenes():
  """Two walkers that do not touch share nothing but the Newton solver's step
  length and stopping rule (M is block diagonal, their rows are disjoint).
  Teacher-forced per step the whole-scene solve must equal the two
  single-walker solves to far below the parity bar -- the basis for advancing a
  scene island by island: observed max 9e-14 over 300 steps with
  contacts and joint limits active, although the iteration counts differ in
  half of the steps.  (Free-running trajectories separate, as any two runs of a
  falling humanoid that differ by 1e-14 do.)"""
  pos = soccer.default_positions(2)
  two = compiler.from_xml_string(soccer.build(2, with_ball=False))
  ones = [compiler.from_xml_string(soccer.build(1, with_ball=False, walker_positions=[p]))
          for p in pos]
  nq1, nv1, nu1 = ones[0].nq, ones[0].nv, ones[0].nu
  assert (two.nq, two.nv, two.nu) == (2*nq1, 2*nv1, 2*nu1)
  rs = np.random.RandomState(0)
  d2 = oracle.OracleData(oracle.OracleModel(two))
  d1 = [oracle.OracleData(oracle.OracleModel(m)) for m in ones]
  for k in range(2):
    d2.qpos[k*nq1 + 2] = 0.95
    d2.qvel[k*nv1:(k + 1)*nv1] = 0.3*rs.randn(nv1)
  d2.step1()
  worst, contacts, differ = 0.0, 0, 0
  for _ in range(120):
    c = rs.uniform(-1, 1, 2*nu1)
    for k, d in enumerate(d1):
      d.qpos[:] = d2.qpos[k*nq1:(k + 1)*nq1]
      d.qvel[:] = d2.qvel[k*nv1:(k + 1)*nv1]
      d.qacc_warmstart[:] = d2.qacc_warmstart[k*nv1:(k + 1)*nv1]
      d.step1()
      d.ctrl[:] = c[k*nu1:(k + 1)*nu1]
    assert d2.ncon == d1[0].ncon + d1[1].ncon and d2.nefc == d1[0].nefc + d1[1].nefc
    contacts += d2.ncon
    d2.ctrl[:] = c
    d2.physics_step()
    for k, d in enumerate(d1):
      d.physics_step()
      v = d2.qvel[k*nv1:(k + 1)*nv1]
      worst = max(worst, np.abs(v - d.qvel).max()/max(1.0, np.abs(d.qvel).max()))
    differ += d2.solver_iter != max(d.solver_iter for d in d1)
  assert contacts > 50 and not d2.warning.any()
  assert worst <= 1e-11, worst


def _states(m, nenv, rs, height):
  qpos = np.tile(m.qpos0, (nenv, 1))
  for i, (name, _, _, rng, _) in enumerate(T.JOINTS):
    lo, hi = rng
    qpos[:, 7 + i] = rs.uniform(lo + 0.1*(hi - lo), hi - 0.1*(hi - lo), nenv)
  qpos[:, 7:] = 0.5*qpos[:, 7:] + 0.5*m.qpos0[7:]
  quat = np.array([0.7071067811865476, 0.7071067811865476, 0, 0]) + 0.1*rs.randn(nenv, 4)
  qpos[:, 3:7] = quat/np.linalg.norm(quat, axis=1, keepdims=True)
  qpos[:, 2] = height
  qvel = 0.3*rs.randn(nenv, m.nv)
  return qpos, qvel


def _teacher_forced(m, hb, qpos, qvel, steps, rs, W):
  om = oracle.OracleModel(m)
  datas = [oracle.OracleData(om) for _ in range(len(qpos))]
  for i, d in enumerate(datas):
    d.qpos[:] = qpos[i]
    d.qvel[:] = qvel[i]
    d.step1()
  errs, rows = [], 0
  _teacher_forced.max_ncon = 0
  for _ in range(steps):
    oq = np.array([d.qpos.copy() for d in datas])
    ov = np.array([d.qvel.copy() for d in datas])
    ow = np.array([d.qacc_warmstart.copy() for d in datas])
    hb.set_state(oq.T, ov.T, ow.T)
    ctrl = rs.uniform(-1, 1, (len(qpos), m.nu))
    hb.step_host(ctrl, 1)
    q = hb.read(W.FIELD_QPOS).T.astype(np.float64)
    v = hb.read(W.FIELD_QVEL).T.astype(np.float64)
    for i, d in enumerate(datas):
      d.ctrl[:] = ctrl[i]
      rows += d.nefc
      _teacher_forced.max_ncon = max(_teacher_forced.max_ncon, d.ncon)
      d.physics_step()
    nq = np.array([d.qpos.copy() for d in datas])
    nv = np.array([d.qvel.copy() for d in datas])
    errs.append(np.maximum(helpers.rel_err(q, nq), helpers.rel_err(v, nv)))
  return np.concatenate(errs), rows


@pytest.mark.gpu
def test_walker_scene_on_device_matches_oracle():
  """62-dof position-controlled walker on the several-lanes-per-env kernel.
  fp64: the working set of this model only fits in LDS with room for four
  contacts, so the tight comparison runs on airborne and toe-touching poses
  (joint limits, affine actuators with force clipping, stiff joints, 62-dof
  CRBA / RNE / Cholesky); fp32 at full capacity from standing and lying poses."""
  from dm_control_amd import build, wrapper as W
  m = _walker_model()
  rs = np.random.RandomState(0)
  with build.allow_overbudget():   # 62 dofs on one lane group: thousands of spills
    hm = W.HipModel(build.build_model(m, 0, 'f64', mode='coop', ncon_max=4))
  hb = W.HipBatch(hm, 16)
  qpos, qvel = _states(m, 16, rs, height=1.6)
  qpos[8:, 2] = 1.13                          # toes at the floor
  e, rows = _teacher_forced(m, hb, qpos, qvel, 8, rs, W)
  assert rows > 16*8*3                        # limit rows (and a few contacts) in play
  assert not hb.read(W.FIELD_WARN).any()
  assert e.max() <= 1e-9, e.max()
  hb.free()
  with build.allow_overbudget():
    hm = W.HipModel(build.build_model(m, 0, 'f32', mode='coop'))
  hb = W.HipBatch(hm, 32)
  qpos, qvel = _states(m, 32, rs, height=1.05)
  qpos[16:, 2] = 0.25                         # lying / crouching in the floor
  e, rows = _teacher_forced(m, hb, qpos, qvel, 8, rs, W)
  print('OBSERVED fp32 per-step soccer walker: median %.2e p99 %.2e max %.2e'
        % (np.median(e), np.percentile(e, 99), e.max()))
  assert np.median(e) <= 2e-5 and np.percentile(e, 99) <= 2e-3
  # soak: 300 steps of random position targets, no warnings, finite
  for _ in range(300):
    hb.step_host(rs.uniform(-1, 1, (32, m.nu)), 1)
  assert not hb.read(W.FIELD_WARN).any()
  assert np.isfinite(hb.read(W.FIELD_QPOS)).all()
  hb.free()


@pytest.mark.gpu
def test_ball_scene_on_device_matches_oracle():
  from dm_control_amd import build, wrapper as W
  m = _ball_model()
  rs = np.random.RandomState(1)
  n = 64
  qpos = np.tile(m.qpos0, (n, 1))
  qpos[:, 2] = soccer.BALL['radius'] + rs.uniform(-0.003, 0.05, n)
  qpos[:, :2] = rs.uniform(-5, 5, (n, 2))
  qpos[::4, 0] = soccer.PITCH_SIZE[0] - soccer.BALL['radius'] + 0.001      # at a wall
  quat = rs.randn(n, 4)
  qpos[:, 3:7] = quat/np.linalg.norm(quat, axis=1, keepdims=True)
  qvel = rs.randn(n, 6)*[3, 3, 1, 5, 5, 5]
  hm = W.HipModel(build.build_model(m, 0, 'f64'))
  hb = W.HipBatch(hm, n)
  e, rows = _teacher_forced(m, hb, qpos, qvel, 12, rs, W)
  assert rows > 100*10//2                     # ten rows per ball contact
  assert not hb.read(W.FIELD_WARN).any()
  assert e.max() <= 1e-9, e.max()
  hb.free()


def _pitch_model(quiet):
  from dm_control_amd.locomotion import soccer as soccer_env
  return compiler.from_xml_string(soccer.build(
      4, pitch_size=soccer_env.area_to_size(400.0), disable_walker_contacts=quiet,
      ball=soccer.REGULATION_BALL, goal_size=soccer_env.MINI_FOOTBALL_GOAL_SIZE))


def _pitch_states(m, nenv, rs):
  """Players standing on / sunk a little into the pitch or lying on it, joints
  off their zero pose, the ball resting against a foot of player 0 (so that a
  ball-walker contact couples two trees) or in the air."""
  qpos = np.tile(m.qpos0, (nenv, 1))
  qvel = 0.2*rs.randn(nenv, m.nv)
  for e in range(nenv):
    for k in range(4):
      a = 63*k                    # walkers first, the ball last
      qpos[e, a + 7:a + 63] += 0.15*rs.randn(56)
      if (e + k) % 3 == 2:        # lying on the back / side
        qpos[e, a + 2] = 0.25
        qpos[e, a + 3:a + 7] = [1, 0, 0, 0] + 0.05*rs.randn(4)
        qpos[e, a + 3:a + 7] /= np.linalg.norm(qpos[e, a + 3:a + 7])
      else:                       # standing, feet at / slightly in the ground
        qpos[e, a + 2] = rs.uniform(0.84, 0.92)   # (feet reach the ground below 0.93)
    foot = qpos[e, 0:2]
    qpos[e, 252:254] = foot + (rs.uniform(-0.25, 0.25, 2) if e % 2 == 0 else [3.0, 1.0])
    qpos[e, 254] = 0.115 if e % 2 == 0 else 0.6
  return qpos, qvel


@pytest.mark.gpu
def test_single_walker_fp64_at_full_contact_capacity():
  """The 62-dof walker in fp64 from standing and lying poses with every contact
  row in play (the several-lanes fp64 build above only has room for four
  contacts): the one-env-per-lane kernel in its rolled form, rows beyond the
  LDS tier in the HBM workspace."""
  from dm_control_amd import build, wrapper as W
  m = _walker_model()
  rs = np.random.RandomState(4)
  hm = W.HipModel(build.build_model(m, 0, 'f64', mode='rolled'))
  hb = W.HipBatch(hm, 12)
  qpos, qvel = _states(m, 12, rs, height=0.88)
  qpos[6:, 2] = 0.2
  qpos[6:, 3:7] = [1, 0, 0, 0]
  e, rows = _teacher_forced(m, hb, qpos, qvel, 8, rs, W)
  print('OBSERVED soccer walker f64 rolled, full capacity: max %.2e, ncon up to %d' % (
      e.max(), _teacher_forced.max_ncon))
  assert _teacher_forced.max_ncon > 4 and rows > 12*8*10
  assert not hb.read(W.FIELD_WARN).any()
  assert e.max() <= 1e-9, e.max()
  hb.free()


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['team', 'rolled'])
@pytest.mark.parametrize('quiet', [True, False])
def test_two_by_two_pitch_on_device_matches_oracle(quiet, mode):
  """BASELINE configs[4]: the 2v2 pitch (nq 259, nv 254, nu 224; four CMU
  humanoids, regulation ball, two goal frames) as ONE model on the device, the
  four packed 254 x 254 matrices and the geom-pose mirror of each pitch in the
  HBM workspace: `team` = one wavefront per pitch (what `soccer.load` runs),
  `rolled` = one pitch per lane.  fp64 per step against the oracle at full
  contact capacity, with walker-pitch, walker-ball and (quiet=False)
  walker-walker / self contacts; fp32 statistics."""
  from dm_control_amd import build, wrapper as W
  m = _pitch_model(quiet)
  assert (m.nq, m.nv, m.nu) == (259, 254, 224)
  rs = np.random.RandomState(7)
  nenv = 6
  qpos, qvel = _pitch_states(m, nenv, rs)
  if not quiet:                   # two players into each other
    qpos[:, 63:65] = qpos[:, 0:2] + [0.25, 0.1]
  hm = W.HipModel(build.build_model(m, 0, 'f64', ncon_max=64, mode=mode))
  hb = W.HipBatch(hm, nenv)
  e, rows = _teacher_forced(m, hb, qpos, qvel, 5, np.random.RandomState(1), W)
  print('OBSERVED 2v2 pitch f64 %s (quiet=%s): per-step max %.2e; ncon up to %d, %d rows in all'
        % (mode, quiet, e.max(), _teacher_forced.max_ncon, rows))
  assert not hb.read(W.FIELD_WARN).any()
  assert _teacher_forced.max_ncon >= 30 and rows > nenv*5*250
  assert e.max() <= 1e-9, e.max()
  hb.free()
  hm32 = W.HipModel(build.build_model(m, 0, 'f32', ncon_max=64, mode=mode))
  hb = W.HipBatch(hm32, nenv)
  e32, _ = _teacher_forced(m, hb, qpos, qvel, 5, np.random.RandomState(1), W)
  print('OBSERVED 2v2 pitch f32 %s (quiet=%s): per-step median %.2e p90 %.2e max %.2e'
        % (mode, quiet, np.median(e32), np.percentile(e32, 90), e32.max()))
  assert np.median(e32) <= 2e-4 and np.isfinite(e32).all()
  hb.free()


@pytest.mark.gpu
@pytest.mark.parametrize('team_size', [1, 3])
def test_other_team_sizes_on_the_team_build(team_size):
  """1v1 (two trees + ball, nv 130) and 3v3 (six trees + ball, nv 378, more
  trees than lane groups) through `soccer.load` on the one-wavefront-per-pitch
  build: fp64 per step against the oracle from standing / sunk-in poses, then the
  environment plays a few control steps without warnings."""
  from dm_control_amd.locomotion import soccer as soccer_env
  from dm_control_amd import wrapper as W
  nenv = 3
  env = soccer_env.load(team_size, random_state=5,
                        environment_kwargs={'batch_size': nenv, 'precision': 'f64'})
  m = env.physics.model
  nw = 2*team_size
  assert (m.nq, m.nv) == (63*nw + 7, 62*nw + 6)
  assert 'team mode' in env.physics.kernel_shape
  rs = np.random.RandomState(11)
  qpos = np.tile(m.qpos0, (nenv, 1))
  qvel = 0.2*rs.randn(nenv, m.nv)
  for e in range(nenv):
    for k in range(nw):
      qpos[e, 63*k + 7:63*k + 63] += 0.15*rs.randn(56)
      qpos[e, 63*k + 2] = rs.uniform(0.84, 0.92)
    qpos[e, 63*nw:63*nw + 2] = qpos[e, 0:2] + rs.uniform(-0.2, 0.2, 2)   # the ball at player 0's feet
    qpos[e, 63*nw + 2] = 0.115
  e, rows = _teacher_forced(m, env.physics.batch, qpos, qvel, 3, np.random.RandomState(2), W)
  print('OBSERVED %dv%d pitch f64 team: per-step max %.2e; ncon up to %d, %d rows in all'
        % (team_size, team_size, e.max(), _teacher_forced.max_ncon, rows))
  assert _teacher_forced.max_ncon >= 2*nw and e.max() <= 1e-9, e.max()
  env.reset()
  for _ in range(4):
    ts = env.step([rs.uniform(-1, 1, (nenv, 56)) for _ in range(nw)])
  assert len(ts.reward) == nw and not np.asarray(env.physics.data.warning_mask).any()
  env.physics.free()


@pytest.mark.gpu
def test_soccer_environment_plays():
  """`locomotion.soccer.load(2)` end to end on the device: 48 pitches, random
  actions for 40 control steps (200 physics steps): no warning bits, finite
  state, one observation dict and one reward per player, actions as a list."""
  from dm_control_amd.locomotion import soccer as soccer_env
  env = soccer_env.load(2, random_state=3, disable_walker_contacts=True,
                        environment_kwargs={'batch_size': 48})
  assert len(env.action_spec()) == 4 and env.action_spec()[0].shape == (56,)
  ts = env.reset()
  assert ts.first() and len(ts.observation) == 4
  rs = np.random.RandomState(0)
  for _ in range(40):
    ts = env.step([rs.uniform(-1, 1, (48, 56)) for _ in range(4)])
  assert len(ts.reward) == 4 and ts.reward[0].shape == (48,)
  assert ts.discount.shape == (48,)
  for obs in ts.observation:
    for key, value in obs.items():
      assert np.isfinite(value).all(), key
  assert obs['ball_ego_position'].shape == (48, 3)
  assert not np.asarray(env.physics.data.warning_mask).any()
  z = np.asarray(env.physics.data.qpos)[:, [63*k + 2 for k in range(4)]]
  assert (z > 0.05).all() and (z < 1.6).all()
  env.physics.free()


def test_oracle_at_its_row_capacity_flags_and_stays_in_bounds():
  """mjWARN_CNSTRFULL instead of a write past the row arrays when a pyramidal
  contact is cut short by the capacity (the bench's CPU leg ran into exactly
  that with four fallen humanoids and the C default of 600 rows)."""
  m = compiler.from_xml_string(soccer.build(2, with_ball=False))
  om = oracle.OracleModel(m)
  om.set_int('nefcmax', 70)
  d = oracle.OracleData(om)
  d.qpos[2] = d.qpos[65] = 0.08           # both walkers flat on the ground: many contacts
  d.qpos[3:7] = d.qpos[66:70] = [1, 0, 0, 0]
  d.step1()
  top = 0
  for _ in range(12):
    d.physics_step()
    top = max(top, d.nefc)
  assert top == 70 and d.warning[2] > 0 and np.isfinite(d.qpos).all()


def test_tables_of_the_team_build():
  """What codegen adds for scenes of several trees (csrc/dmc_kernels.hip, team mode):
  the pair list grouped into runs that one bounding test can skip, and every tree
  cut into chains without branches ("segments") on levels."""
  import re
  from dm_control_amd import codegen
  m = _pitch_model(False)
  text = codegen.generate_header(m, ncon_max=64)
  def table(name):
    return [int(x) for x in re.search(r'%s\[\] = \{([^}]*)\}' % name, text).group(1).split(',')]
  def const(name):
    return int(re.search(r'constexpr int %s = (-?\d+)' % name, text).group(1))
  # segments: a partition of the bodies into runs of consecutive bodies, each a chain
  lo, hi, hub = table('seg_body_lo'), table('seg_body_hi'), table('seg_hub_body')
  assert const('NSEG') == len(lo) == 45 and const('NSEGLEVEL') == 4      # 11 per walker + the ball
  covered = [b for a, e in zip(lo, hi) for b in range(a, e)]
  assert covered == list(range(1, m.nbody))
  parent = [int(p) for p in m.body_parentid]
  level = {}
  for sid, (a, e, h) in enumerate(zip(lo, hi, hub)):
    assert parent[a] == h                                   # the segment hangs from its hub
    assert all(parent[b] == b - 1 for b in range(a + 1, e))   # a chain inside
    seg_of_hub = next((k for k in range(sid) if lo[k] <= h < hi[k]), None)
    level[sid] = 0 if h == 0 else level[seg_of_hub] + 1
  per_level = table('lvl_seg')
  width = const('MAXSEGPERLEVEL')
  ntree = len(table('tree_body_lo'))
  seen = set()
  for t in range(ntree):
    for lv in range(4):
      for sid in per_level[(t*4 + lv)*width:(t*4 + lv + 1)*width]:
        if sid >= 0:
          assert level[sid] == lv and table('tree_body_lo')[t] <= lo[sid] < table('tree_body_hi')[t]
          seen.add(sid)
  assert seen == set(range(45))
  hubs = sorted(set(h for h in hub if h > 0))
  assert const('NHUB') == len(hubs) and [table('body_hub')[h] for h in hubs] == list(range(len(hubs)))
  # pair runs: the same pairs as MuJoCo's filters leave, every keyed run homogeneous
  pairs = codegen.collision_pairs(m)
  assert len(pairs) == 18883 and len(set(pairs)) == len(pairs)
  first, length, keyed = table('run_first'), table('run_len'), table('run_keyed')
  assert sum(length) == len(pairs) and first[0] == 0
  assert all(first[k + 1] == first[k] + length[k] for k in range(len(first) - 1))
  t1, t2, wg = table('pair_tree1'), table('pair_tree2'), table('pair_wgeom')
  for f, n, k in zip(first, length, keyed):
    kinds = set((min(t1[p], t2[p]), max(t1[p], t2[p]), wg[p]) for p in range(f, f + n))
    if k:
      assert len(kinds) == 1
      a, b, g = next(iter(kinds))
      assert (a >= 0 and b >= 0 and a != b and g < 0) or (a < 0 <= b and g >= 0)
    else:
      assert all(a == b or (a < 0 and b < 0) for a, b, _ in kinds)
  # 25 world geoms x 5 trees + 10 tree pairs can be skipped; the pairs inside the trees are one run
  assert sum(keyed) == 135 and len(first) == 136
  assert max(n for n, k in zip(length, keyed) if k) == 1849 and max(length) == 3292
