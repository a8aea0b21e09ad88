"""Reference-independent closed forms for the oracle AND the fp64 device build.

The reference holds no trajectory fixture for the suite models ("parity
unpinned" beyond K1-K9), and oracle and kernels share an author,
so a common misreading of MuJoCo's documentation would be invisible to the
device-vs-oracle tests.  These checks compare with physics instead (SURVEY.md
Appendix D), each on the oracle (CPU, always) and on the fp64 code object
(`-m gpu`):

  energy            undamped, unactuated, contact-free cheetah chain and acrobot:
                    RK4 keeps the mechanical energy to O(h^4), semi-implicit Euler
                    to O(h) without drift  -> RNE/Coriolis consistent with M
  momentum          the humanoid adrift (no gravity, no contacts) under random
                    motor torques, joint limits and damping -- all internal
                    forces -- keeps linear and angular momentum
                    -> free joint, quaternion integration, limit rows
  implicit damping  1-DoF spring-damper against the closed-form discrete map
                    v' = v + h (m + h b)^-1 (-k x - b v),  x' = x + h v'
  limit spring      cart pushed into the slider limit: steady penetration from
                    the force balance with solreflimit ".08 1" (cartpole.xml:25)
  narrowphase       capsule-capsule and plane-capsule distance / normal against
                    brute-force sampling of the two surfaces
  cart-pole ODE     1000 RK4 steps of the textbook equations of motion
  K2 with its sensor  the cube's touch sensor reads its weight
                    (wrapper/core_test.py:328-368)
"""

import math

import numpy as np
import pytest

import helpers
import kat_models
from dm_control_amd.mjcf import compiler
from oracle import oracle

G = 9.81
MODELS = kat_models.closed_form_models()


# ---------------------------------------------------------------------------
# two steppers with one interface: the oracle, and the fp64 device build whose
# invariants are evaluated by setting the oracle to the device's state
# ---------------------------------------------------------------------------
class OracleStepper:

  def __init__(self, model, qpos, qvel):
    self.model = model
    self.p = oracle.OraclePhysics(model)
    self.p.reset()
    self.p.data.qpos[:] = qpos
    self.p.data.qvel[:] = qvel
    self.p.forward()

  def step(self, ctrl=None):
    if ctrl is not None:
      self.p.data.ctrl[:] = ctrl
    self.p.step()

  def state(self):
    return self.p.data.qpos.copy(), self.p.data.qvel.copy()

  def probe(self):
    """Oracle data evaluated at the current state (kinematics, M, cvel)."""
    return self.p.data

  def sensordata(self):
    raise NotImplementedError


class DeviceStepper:
  """One instance on the fp64 code object, through the C ABI."""

  def __init__(self, model, qpos, qvel, mode):
    from dm_control_amd import build, wrapper
    self.W = wrapper
    self.model = model
    self.hm = wrapper.HipModel(build.build_model(model, 0, 'f64', mode=mode))
    self.hb = wrapper.HipBatch(self.hm, 1)
    self.hb.set_state(np.asarray(qpos)[:, None], np.asarray(qvel)[:, None])
    self._probe = oracle.OraclePhysics(model)
    self._probe.reset()

  def step(self, ctrl=None):
    c = None if ctrl is None else np.asarray(ctrl, np.float64)[None]
    self.hb.step_host(c, 1)

  def state(self):
    q = self.hb.read(self.W.FIELD_QPOS)[:self.model.nq, 0].astype(np.float64)
    v = self.hb.read(self.W.FIELD_QVEL)[:self.model.nv, 0].astype(np.float64)
    return q, v

  def probe(self):
    q, v = self.state()
    d = self._probe.data
    d.qpos[:] = q
    d.qvel[:] = v
    self._probe.forward()
    return d

  def sensordata(self):
    return self.hb.read(self.W.FIELD_SENSORDATA)[:, 0].astype(np.float64)


def _stepper(kind, name, qpos, qvel):
  model, mode = MODELS[name]
  if kind == 'oracle':
    return OracleStepper(model, qpos, qvel)
  return DeviceStepper(model, qpos, qvel, mode)


KINDS = ['oracle', pytest.param('device', marks=pytest.mark.gpu)]


def _energy(model, d):
  v = d.qvel
  kin = 0.5*v @ d.qM @ v
  pot = -sum(model.body_mass[b]*np.dot(model.opt.gravity, d.xipos[b])
             for b in range(model.nbody))
  return kin + pot


# ---------------------------------------------------------------------------
@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('name', ['cheetah_chain', 'acrobot'])
def test_energy_is_conserved_without_dissipation(name, kind):
  rs = np.random.RandomState(3)
  base, _ = MODELS[name + '_rk4']
  qpos = base.qpos0 + 0.4*rs.randn(base.nq)
  if name == 'cheetah_chain':
    qpos[1] = 2.0                 # the floor plays no role (contacts disabled)
  # (the acrobot's step is coarse for its chaotic swings: a short, calm run
  # keeps the Euler comparison in the asymptotic regime)
  qvel = {'cheetah_chain': 0.8, 'acrobot': 0.2}[name]*rs.randn(base.nv)
  horizon = {'cheetah_chain': 300, 'acrobot': 60}[name]*base.opt.timestep

  def energy_error(tag):
    model, _ = MODELS[name + '_' + tag]
    s = _stepper(kind, name + '_' + tag, qpos, qvel)
    e0 = _energy(model, s.probe())
    worst = 0.0
    for _ in range(int(round(horizon/model.opt.timestep))):
      s.step()
      e = _energy(model, s.probe())
      worst = max(worst, abs(e - e0))
    return worst/(abs(e0) + 1.0), abs(e - e0)/(abs(e0) + 1.0)
  worst, _ = energy_error('rk4')
  # (the acrobot's 1 m links swing fast for its 0.01 s step: O(h^4) is larger)
  assert worst < {'cheetah_chain': 2e-8, 'acrobot': 5e-5}[name], worst
  # semi-implicit Euler is first order: halving the step halves the energy error
  full, _ = energy_error('euler')
  half, _ = energy_error('euler_half')
  assert 50*worst < full < 0.2, full
  assert 1.6 < full/half < 2.4, full/half


def _momentum(model, d):
  """Linear momentum and angular momentum about the centre of mass."""
  com = d.subtree_com[1]            # subtree of the root body = everything
  p = np.zeros(3)
  ang = np.zeros(3)
  for b in range(1, model.nbody):
    m = model.body_mass[b]
    w = d.cvel[b, :3]
    root = d.subtree_com[1]         # cvel is about the subtree-root CoM
    v = d.cvel[b, 3:] + np.cross(w, d.xipos[b] - root)
    rot = d.ximat[b].reshape(3, 3)
    inertia = rot @ np.diag(model.body_inertia[b]) @ rot.T
    p += m*v
    ang += inertia @ w + m*np.cross(d.xipos[b] - com, v)
  return p, ang


@pytest.mark.parametrize('kind', KINDS)
def test_humanoid_adrift_keeps_its_momentum(kind):
  """The stage states of MuJoCo's RK4 are built with the exponential map of the
  free joint's quaternion, which makes the method second order in a tumbling
  body's orientation: the momentum error is small and falls by >= 3x when the
  step is halved (a force that does not belong -- gravity left on, a wrong
  limit or motor Jacobian -- would leave an error that does not)."""
  rs = np.random.RandomState(5)
  base, _ = MODELS['humanoid_adrift']
  qpos = base.qpos0.copy()
  for j in range(base.njnt):
    if base.jnt_limited[j]:
      lo, hi = base.jnt_range[j]
      qpos[base.jnt_qposadr[j]] = rs.uniform(0.9*lo, 0.9*hi)   # near the limits
  quat = rs.randn(4)
  qpos[3:7] = quat/np.linalg.norm(quat)
  qvel = rs.randn(base.nv)
  horizon = 200*base.opt.timestep
  ctrls = 0.03*rs.uniform(-1, 1, (200, base.nu))

  def drift(name):
    model, _ = MODELS[name]
    s = _stepper(kind, name, qpos, qvel)
    p0, l0 = _momentum(model, s.probe())
    assert np.linalg.norm(p0) > 10 and np.linalg.norm(l0) > 5
    n = int(round(horizon/model.opt.timestep))
    limits = 0
    for t in range(n):
      s.step(ctrls[t*200//n])
      if kind == 'oracle':
        limits += s.p.data.nefc > 0
    p1, l1 = _momentum(model, s.probe())
    assert limits > 20 or kind != 'oracle'      # joint-limit rows took part
    return (np.linalg.norm(p1 - p0)/np.linalg.norm(p0),
            np.linalg.norm(l1 - l0)/np.linalg.norm(l0))
  dp, dl = drift('humanoid_adrift')
  dp2, dl2 = drift('humanoid_adrift_half')
  assert dp < 2e-4 and dl < 2e-4, (dp, dl)
  assert dp2 < dp/3 and dl2 < dl/3, (dp, dp2, dl, dl2)


@pytest.mark.parametrize('kind', KINDS)
def test_implicit_joint_damping_matches_its_discrete_map(kind):
  model, _ = MODELS['oscillator']
  m, k, b, h = 2.0, 30.0, 1.5, 0.01
  assert abs(model.body_mass[1] - m) < 1e-12 and model.opt.timestep == h
  x, v = 0.3, -0.7
  s = _stepper(kind, 'oscillator', [x], [v])
  for _ in range(1000):
    v = v + h*(-k*x - b*v)/(m + h*b)
    x = x + h*v
    s.step()
  q, qd = s.state()
  np.testing.assert_allclose([q[0], qd[0]], [x, v], rtol=0, atol=1e-12)


@pytest.mark.parametrize('kind', KINDS)
def test_fixed_tendon_transmission_matches_its_discrete_map(kind):
  """Motors acting through fixed tendons that mix two sliders (the transmission
  of suite/point_mass.xml): the generalised force is sum_u gear_u ctrl_u C[u][j],
  controls are clamped to ctrlrange, and with the implicit joint damping of
  the Euler integrator each slider follows v' = (m v + h f_j)/(m + h d_j),
  q' = q + h v' exactly.  Independent of the oracle's reading of
  mj_transmission / mj_fwdActuation."""
  model, _ = MODELS['tendon_puck']
  m, h = 1.5, 0.01
  d = np.array([0.7, 0.2])
  gear = np.array([2.0, 0.5])
  C = np.array([[0.8, -0.6], [0.3, 0.9]])        # tendon u = sum_j C[u][j] q_j
  assert abs(model.body_mass[1] - m) < 1e-12 and model.opt.timestep == h
  q = np.array([0.1, -0.2])
  v = np.array([0.5, 0.3])
  s = _stepper(kind, 'tendon_puck', q, v)
  rs = np.random.RandomState(0)
  for _ in range(400):
    ctrl = rs.uniform(-1.5, 1.5, 2)                # beyond ctrlrange: clamped
    f = (gear*np.clip(ctrl, -1, 1)) @ C
    v = (m*v + h*f)/(m + h*d)
    q = q + h*v
    s.step(ctrl)
  qs, vs = s.state()
  np.testing.assert_allclose(qs, q, rtol=0, atol=1e-12)
  np.testing.assert_allclose(vs, v, rtol=0, atol=1e-12)


def _impedance(solimp, r):
  d0, dmax, width, mid, power = solimp
  x = min(abs(r)/width, 1.0)
  if x <= mid:
    y = x**power/mid**(power - 1)
  else:
    y = 1 - (1 - x)**power/(1 - mid)**(power - 1)
  return d0 + y*(dmax - d0)


@pytest.mark.parametrize('kind', KINDS)
def test_slider_limit_penetration_follows_solreflimit(kind):
  """Motor force F = gear*ctrl pushes the cart into the upper limit (1.8).  At
  rest the limit row carries f = F, and f = D k d(r) |r| with
  k = 1/(dmax^2 tc^2 zeta^2), D = d/((1 - d) invweight): solved for r here."""
  model, _ = MODELS['cartpole_at_limit']
  force = 10.0*0.6
  tc, zeta = 0.08, 1.0                           # suite/cartpole.xml:25
  solimp = (0.9, 0.95, 0.001, 0.5, 2.0)          # MuJoCo defaults
  mc, mp, l = 1.0, 0.1, 0.5
  ipole = model.body_inertia[2][0]
  mass = np.array([[mc + mp, -mp*l], [-mp*l, ipole + mp*l*l]])   # pole hanging
  mass0 = np.array([[mc + mp, mp*l], [mp*l, ipole + mp*l*l]])    # qpos0: upright
  invweight = np.linalg.inv(mass0)[0, 0]
  del mass
  kk = 1/(0.95**2*tc**2*zeta**2)

  def residual(r):
    d = _impedance(solimp, r)
    return kk*d*d*r/((1 - d)*invweight) - force
  lo, hi = 0.0, 0.1
  for _ in range(200):
    mid = 0.5*(lo + hi)
    lo, hi = (mid, hi) if residual(mid) < 0 else (lo, mid)
  pen = 0.5*(lo + hi)
  assert 1e-5 < pen < 1e-2
  s = _stepper(kind, 'cartpole_at_limit', [1.79, math.pi], [0.0, 0.0])
  for _ in range(6000):
    s.step([0.6])
  q, v = s.state()
  assert abs(v[0]) < 1e-7 and abs(v[1]) < 1e-5
  np.testing.assert_allclose(q[0] - 1.8, pen, rtol=1e-6)


def _segment_distance(p1, a1, h1, p2, a2, h2, n=1201):
  t = np.linspace(-1, 1, n)
  s1 = p1 + np.outer(t, a1*h1)
  s2 = p2 + np.outer(t, a2*h2)
  d = np.linalg.norm(s1[:, None] - s2[None], axis=2)
  i, j = np.unravel_index(np.argmin(d), d.shape)
  return d[i, j], s1[i], s2[j]


def _rand_quat(rs):
  q = rs.randn(4)
  return q/np.linalg.norm(q)


def test_capsule_narrowphase_against_brute_force():
  """Oracle contacts (dist, normal, position) of capsule-capsule and
  plane-capsule pairs vs dense sampling of the segments.  The device runs the
  same poses in test_gpu_closed_form_narrowphase (contact counts and the
  resulting accelerations agree with the oracle there)."""
  rs = np.random.RandomState(0)
  m = compiler.from_xml_string(kat_models.CAPSULE_PAIR)
  p = oracle.OraclePhysics(m)
  checked = 0
  for _ in range(60):
    p.reset()
    qa, qb = _rand_quat(rs), _rand_quat(rs)
    pa, pb = rs.uniform(-.4, .4, 3), rs.uniform(-.4, .4, 3) + [0, 0, .5]
    p.data.qpos[:] = np.concatenate([pa, qa, pb, qb])
    p.forward()
    axa = p.data.geom_xmat[0].reshape(3, 3)[:, 2]
    axb = p.data.geom_xmat[1].reshape(3, 3)[:, 2]
    seg, ca, cb = _segment_distance(pa, axa, 0.3, pb, axb, 0.2)
    assert p.data.ncon == 1
    con = p.data.contact(0)
    want = seg - 0.07 - 0.05
    assert abs(con['dist'] - want) < 2e-6 + 1e-3*abs(want)**0 * 2e-4
    if seg > 0.05:
      normal = (cb - ca)/np.linalg.norm(cb - ca)
      assert np.dot(con['frame'][0], normal) > 1 - 1e-4
      mid = ca + normal*(0.07 + 0.5*want)
      np.testing.assert_allclose(con['pos'], mid, atol=2e-3)
    checked += 1
  assert checked == 60
  # plane-capsule: one contact per end sphere, dist = height of the sphere - r
  m = compiler.from_xml_string(kat_models.CAPSULE_OVER_PLANE)
  p = oracle.OraclePhysics(m)
  for _ in range(40):
    p.reset()
    q = _rand_quat(rs)
    pos = rs.uniform(-.3, .3, 3) + [0, 0, .6]
    p.data.qpos[:] = np.concatenate([pos, q])
    p.forward()
    ax = p.data.geom_xmat[1].reshape(3, 3)[:, 2]
    ends = [pos + 0.3*ax, pos - 0.3*ax]
    assert p.data.ncon == 2
    got = sorted(p.data.contact(i)['dist'] for i in range(2))
    want = sorted(e[2] - 0.07 for e in ends)
    np.testing.assert_allclose(got, want, atol=1e-12)
    for i in range(2):
      np.testing.assert_allclose(p.data.contact(i)['frame'][0], [0, 0, 1], atol=1e-12)


def test_sphere_box_narrowphase_against_brute_force():
  """Oracle contact of a sphere and an oriented box (outside near faces, edges
  and corners, and with the centre inside the box) vs dense sampling of the box
  surface; the device runs sphere-box scenes in
  test_known_answer_models_on_device[balls_and_boxes]."""
  rs = np.random.RandomState(1)
  m = compiler.from_xml_string(kat_models.BALL_NEAR_BOX)
  p = oracle.OraclePhysics(m)
  box = m.name2id('box', 'geom')
  size = m.geom_size[box]
  g = np.linspace(-1, 1, 161)
  u, v = np.meshgrid(g, g)
  faces = []
  for axis in range(3):
    for sign in (-1, 1):
      pts = np.zeros(u.shape + (3,))
      pts[..., axis] = sign*size[axis]
      pts[..., (axis + 1) % 3] = u*size[(axis + 1) % 3]
      pts[..., (axis + 2) % 3] = v*size[(axis + 2) % 3]
      faces.append(pts.reshape(-1, 3))
  surface = np.concatenate(faces)
  inside_seen = 0
  for trial in range(80):
    p.reset()
    p.forward()
    bpos, bmat = p.data.geom_xpos[box].copy(), p.data.geom_xmat[box].reshape(3, 3).copy()
    if trial % 4 == 0:     # centre inside the box
      loc = rs.uniform(-0.9, 0.9, 3)*size
      inside_seen += 1
    else:
      loc = rs.uniform(-2, 2, 3)*size + rs.uniform(-0.1, 0.1, 3)
    centre = bpos + bmat @ loc
    p.data.qpos[:3] = centre
    p.forward()
    assert p.data.ncon == 1
    con = p.data.contact(0)
    world = bpos + surface @ bmat.T
    d = np.linalg.norm(world - centre, axis=1)
    k = int(np.argmin(d))
    is_inside = np.all(np.abs(loc) < size)
    want = (-d[k] if is_inside else d[k]) - 0.07
    # (the nearest grid point is off by up to ~0.7 grid spacings when the centre
    # sits on the surface, by spacing^2/(2 d) at distance d)
    spacing = 2*max(size)/160
    assert abs(con['dist'] - want) < 0.75*spacing, (trial, con['dist'], want)
    towards_box = (world[k] - centre)/d[k]
    if d[k] > 0.15:      # (far enough for the sampling grid to resolve the direction)
      cosang = np.dot(con['frame'][0], towards_box if not is_inside else -towards_box)
      assert cosang > 1 - 1e-3, (trial, cosang)
  assert inside_seen == 20


def _box_surface(size, n=121):
  g = np.linspace(-1, 1, n)
  u, v = np.meshgrid(g, g)
  faces = []
  for axis in range(3):
    for sign in (-1, 1):
      pts = np.zeros(u.shape + (3,))
      pts[..., axis] = sign*size[axis]
      pts[..., (axis + 1) % 3] = u*size[(axis + 1) % 3]
      pts[..., (axis + 2) % 3] = v*size[(axis + 2) % 3]
      faces.append(pts.reshape(-1, 3))
  return np.concatenate(faces)


def test_capsule_box_narrowphase_against_brute_force():
  """Our capsule-box construction (oracle/mjstep.c `capsule_box`; MuJoCo's own
  routine is in the closed binary: parity unpinned) vs dense sampling of the
  segment and of the box surface: the FIRST contact is the nearest point of
  the segment to the box (distance, direction), and when both end spheres
  reach the box a second contact reports the other end."""
  rs = np.random.RandomState(3)
  m = compiler.from_xml_string(kat_models.CAPSULE_NEAR_BOX)
  p = oracle.OraclePhysics(m)
  box, cap = m.name2id('box', 'geom'), m.name2id('cap', 'geom')
  size, (r, h) = m.geom_size[box], m.geom_size[cap][:2]
  surface = _box_surface(size, 81)
  ts = np.linspace(-h, h, 201)
  spacing = 2*max(size)/80
  for trial in range(30):
    p.reset()
    bq, cq = _rand_quat(rs), _rand_quat(rs)
    bpos = rs.uniform(-.2, .2, 3)
    p.data.qpos[:] = np.concatenate([bpos, bq, bpos + rs.uniform(-.7, .7, 3), cq])
    p.forward()
    bmat = p.data.geom_xmat[box].reshape(3, 3)
    cpos, axis = p.data.geom_xpos[cap], p.data.geom_xmat[cap].reshape(3, 3)[:, 2]
    seg = cpos + np.outer(ts, axis)
    loc = (seg - p.data.geom_xpos[box]) @ bmat              # box frame
    if np.any(np.all(np.abs(loc) < size, axis=1)):
      continue                                              # axis runs through the box
    world = p.data.geom_xpos[box] + surface @ bmat.T
    d = np.linalg.norm(seg[:, None] - world[None], axis=2)
    i, k = np.unravel_index(np.argmin(d), d.shape)
    assert p.data.ncon >= 1
    con = p.data.contact(0)
    want = d[i, k] - r
    assert abs(con['dist'] - want) < 0.8*spacing, (trial, con['dist'], want)
    if d[i, k] > 0.12:
      towards = (world[k] - seg[i])/d[i, k]
      assert np.dot(con['frame'][0], towards) > 1 - 2e-3, trial
    for j in range(p.data.ncon):                            # every contact: a real distance
      c = p.data.contact(j)
      on_axis = c['pos'] - c['frame'][0]*(r + 0.5*c['dist'])   # back to the segment
      t = np.dot(on_axis - cpos, axis)
      assert abs(t) <= h + 1e-9
      np.testing.assert_allclose(on_axis, cpos + t*axis, atol=1e-9)


def test_box_box_narrowphase_properties():
  """Our box-box construction (oracle/mjstep.c `box_box`: separating-axis test,
  the incident face clipped against the reference face, or the closest points of
  two edges; parity unpinned): no contact exactly when a sampled separating
  distance is positive; for overlapping boxes the reported depth is the least
  penetration over the face axes, every contact point lies within both boxes'
  reach and the normal points from box 1 to box 2."""
  rs = np.random.RandomState(4)
  m = compiler.from_xml_string(kat_models.BOX_NEAR_BOX)
  p = oracle.OraclePhysics(m)
  sa, sb = m.geom_size[0], m.geom_size[1]
  corners = np.array([[i, j, k] for i in (-1, 1) for j in (-1, 1) for k in (-1, 1)], float)
  hits = misses = 0
  for trial in range(300):
    p.reset()
    pa = rs.uniform(-.1, .1, 3)
    pb = pa + rs.uniform(-.55, .55, 3)
    p.data.qpos[:] = np.concatenate([pa, _rand_quat(rs), pb, _rand_quat(rs)])
    p.forward()
    ma, mb = (p.data.geom_xmat[g].reshape(3, 3) for g in (0, 1))
    ca, cb = pa + (corners*sa) @ ma.T, pb + (corners*sb) @ mb.T
    # separating-axis theorem by brute force over the 15 axes
    axes = [ma[:, i] for i in range(3)] + [mb[:, i] for i in range(3)]
    axes += [np.cross(ma[:, i], mb[:, j]) for i in range(3) for j in range(3)]
    sep = -np.inf
    for ax in axes:
      nrm = np.linalg.norm(ax)
      if nrm < 1e-6:
        continue
      ax = ax/nrm
      a, b = ca @ ax, cb @ ax
      sep = max(sep, b.min() - a.max(), a.min() - b.max())
    if sep > 1e-9:
      assert p.data.ncon == 0, (trial, sep)
      misses += 1
      continue
    if sep > -1e-4:
      continue                       # touching within rounding: either answer
    hits += 1
    assert 1 <= p.data.ncon <= 4, trial
    for j in range(p.data.ncon):
      c = p.data.contact(j)
      assert c['dist'] <= 1e-12 and c['dist'] >= sep - 1e-6 - 0.06, (trial, c['dist'], sep)
      assert np.dot(c['frame'][0], pb - pa) > -1e-9
      for pos, mat, size in ((pa, ma, sa), (pb, mb, sb)):   # near both boxes
        loc = (c['pos'] - pos) @ mat
        assert np.all(np.abs(loc) <= size + abs(c['dist']) + 1e-9), trial
    deepest = min(p.data.contact(j)['dist'] for j in range(p.data.ncon))
    assert deepest <= 0.5*sep + 1e-9     # at least one point about as deep as the overlap
  assert hits > 40 and misses > 40


def test_stacked_boxes_carry_the_weight_above_them():
  """K3 (core_test.py:461-484: sum of contact normal forces = weight) layer by
  layer for a stack floor / box / box / box / capsule: after settling, the
  normal forces across each interface add up to the weight of everything above
  it -- box-box (face contacts, one stack member rotated 30 degrees), capsule-
  box (a log lying on the top box) and plane-box."""
  m = compiler.from_xml_string(kat_models.STACKED_BOXES)
  p = oracle.OraclePhysics(m)
  p.reset()
  for _ in range(1500):
    p.step()
  # (the log may keep rolling slowly along the face: nothing resists that)
  assert not p.data.warning.any() and np.abs(p.data.qvel[:18]).max() < 1e-3
  assert abs(p.data.qvel[20]) < 1e-3
  names = ['floor', 'b0', 'b1', 'b2', 'log']
  ids = [m.name2id(n, 'geom') for n in names]
  mass = {n: float(m.body_mass[m.geom_bodyid[g]]) for n, g in zip(names[1:], ids[1:])}
  above = {('floor', 'b0'): sum(mass.values()),
           ('b0', 'b1'): mass['b1'] + mass['b2'] + mass['log'],
           ('b1', 'b2'): mass['b2'] + mass['log'], ('b2', 'log'): mass['log']}
  force = {k: 0.0 for k in above}
  counts = {k: 0 for k in above}
  for i in range(p.data.ncon):
    c = p.data.contact(i)
    key = tuple(sorted((names[ids.index(c['geom1'])], names[ids.index(c['geom2'])]),
                       key=names.index))
    assert key in above, key            # nothing else touches
    force[key] += p.data.contact_force(i)[0][0]
    counts[key] += 1
  for key, load in above.items():
    np.testing.assert_allclose(force[key], 9.81*load, rtol=2e-3, err_msg=str(key))
  assert counts[('floor', 'b0')] == 4 and counts[('b0', 'b1')] == 4
  assert counts[('b1', 'b2')] >= 3 and counts[('b2', 'log')] == 2
  # and the stack stands where it was built
  assert abs(p.data.qpos[2] - 0.1) < 1e-3 and abs(p.data.qpos[7*3 + 2] - 0.51) < 5e-3


def _textbook_cartpole_rhs(state, force, ipole):
  """Cart 1 kg, pole 0.1 kg with CoM at 0.5 m, theta = 0 upright, positive theta
  tips the pole towards +x (hinge axis +y); viscous damping on both joints."""
  mc, mp, l = 1.0, 0.1, 0.5
  bx, bt = 5e-4, 2e-6                            # suite/cartpole.xml
  x, th, xd, thd = state
  mass = np.array([[mc + mp, mp*l*math.cos(th)],
                   [mp*l*math.cos(th), ipole + mp*l*l]])
  rhs = np.array([force - bx*xd + mp*l*thd*thd*math.sin(th),
                  mp*G*l*math.sin(th) - bt*thd])
  acc = np.linalg.solve(mass, rhs)
  return np.array([xd, thd, acc[0], acc[1]])


@pytest.mark.parametrize('kind', KINDS)
def test_cartpole_follows_the_textbook_ode(kind):
  model = helpers.load_model('cartpole')
  ipole = model.body_inertia[2][0]
  h = model.opt.timestep
  rs = np.random.RandomState(1)
  ctrls = rs.uniform(-1, 1, 1000)
  y = np.array([0.1, 2.5, -0.2, 0.4])
  if kind == 'oracle':
    s = OracleStepper(model, y[:2], y[2:])
  else:
    s = DeviceStepper(model, y[:2], y[2:], 'auto')
  for c in ctrls:
    f = 10.0*c
    k1 = _textbook_cartpole_rhs(y, f, ipole)
    k2 = _textbook_cartpole_rhs(y + 0.5*h*k1, f, ipole)
    k3 = _textbook_cartpole_rhs(y + 0.5*h*k2, f, ipole)
    k4 = _textbook_cartpole_rhs(y + h*k3, f, ipole)
    y = y + h*(k1 + 2*k2 + 2*k3 + k4)/6
    s.step([c])
  q, v = s.state()
  assert abs(q[0]) < 1.7            # the slider limit stayed out of it
  np.testing.assert_allclose(np.concatenate([q, v]), y, rtol=0, atol=1e-8)


@pytest.mark.parametrize('kind', KINDS)
def test_k2_touch_sensor_reads_the_weight(kind):
  """wrapper/core_test.py:328-368 with the reference's own sensor: after 100
  steps the cube rests (|qvel| < 1e-4 there) and touch = 9.81 +- 0.005."""
  model, mode = MODELS['cube_with_touch']
  if kind == 'oracle':
    p = oracle.OraclePhysics(model)
    p.reset()
    for _ in range(99):
      p.data.step()
    p.data.step2()              # the sensor belongs to the acceleration stage
    touch = helpers.oracle_touch(model, p.data, 'touch_sensor')
    p.data.step1()
    qvel = p.data.qvel[0]
  else:
    s = DeviceStepper(model, model.qpos0, np.zeros(model.nv), mode)
    for _ in range(100):
      s.step()
    touch = s.sensordata()[0]
    qvel = s.state()[1][0]
  assert abs(qvel) < 0.5e-4
  assert abs(touch - 9.81) < 0.005


@pytest.mark.gpu
def test_gpu_closed_form_narrowphase():
  """The capsule and box poses of the brute-force tests on the device: the accelerations
  that the contacts produce (margin 5: every pose is 'in contact' with a soft
  reference) agree with the oracle's, i.e. the same distances, normals and
  points went into the rows."""
  from dm_control_amd import build, wrapper as W
  rs = np.random.RandomState(0)
  for xml, nfree, gap in ((kat_models.CAPSULE_PAIR, 2, .5), (kat_models.CAPSULE_OVER_PLANE, 1, .5),
                          # capsule-box and box-box (overlapping and apart; random poses
                          # meet faces, edges and corners)
                          (kat_models.CAPSULE_NEAR_BOX, 2, .25), (kat_models.BOX_NEAR_BOX, 2, .2)):
    m = compiler.from_xml_string(xml)
    n = 64
    qpos = np.zeros((n, m.nq))
    for i in range(n):
      parts = []
      for b in range(nfree):
        parts += [rs.uniform(-.4, .4, 3) + [0, 0, gap*(b + 1)], _rand_quat(rs)]
      qpos[i] = np.concatenate(parts)
    qvel = 0.2*rs.randn(n, m.nv)
    hm = W.HipModel(build.build_model(m, 0, 'f64'))
    hb = W.HipBatch(hm, n)
    hb.set_aux_outputs(True)
    hb.set_state(qpos.T, qvel.T)
    hb.step_host(None, 1)
    got_v = hb.read(W.FIELD_QVEL).T
    ncon = hb.read(W.FIELD_STATS)[0]
    om = oracle.OracleModel(m)
    for i in range(n):
      d = oracle.OracleData(om)
      d.qpos[:] = qpos[i]
      d.qvel[:] = qvel[i]
      d.step1()
      assert ncon[i] == d.ncon
      d.physics_step()
      np.testing.assert_allclose(got_v[i], d.qvel, rtol=0, atol=1e-9)
    hb.free()
