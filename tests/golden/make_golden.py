"""Generates tests/golden/*.json by IMPORTING the Python reference.

Run in the build container only (needs /root/reference; nothing here runs on
the GPU box):   python tests/golden/make_golden.py

What is imported and why:
  * dm_control.utils.rewards       -- importable as is; `tolerance` tables for
    the parameterisations the three tasks use + the rewards_test.py grid.
  * dm_control.rl.control, dm_control.suite.{base,cartpole,cheetah,humanoid}
    -- importable once the absent third-party packages (dm_env, absl, lxml) and
    the libmujoco-backed `dm_control.mujoco` are replaced by in-memory stub
    modules; their Task.get_reward / get_observation then run on canned physics
    readings, and control.Environment runs on a fake Physics/Task pair.
The outputs are data (inputs + expected outputs), committed as fixtures.
"""

import collections
import enum
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
  mod = types.ModuleType(name)
  mod.__dict__.update(attrs)
  sys.modules[name] = mod
  return mod


def install_stubs():
  # numpy 2 dropped aliases the 2020 reference still uses
  if not hasattr(np, 'float'):
    np.float = float
  if not hasattr(np, 'bool'):
    np.bool = bool

  class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2

  class TimeStep(collections.namedtuple(
      'TimeStep', 'step_type reward discount observation')):
    def first(self): return self.step_type == StepType.FIRST
    def mid(self): return self.step_type == StepType.MID
    def last(self): return self.step_type == StepType.LAST

  class Array:
    def __init__(self, shape, dtype, name=None):
      self.shape, self.dtype, self.name = tuple(shape), np.dtype(dtype), name

  class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
      super().__init__(shape, dtype, name)
      self.minimum, self.maximum = minimum, maximum

  class Environment:
    pass
  specs = _stub('dm_env.specs', Array=Array, BoundedArray=BoundedArray)
  _stub('dm_env', StepType=StepType, TimeStep=TimeStep,
        Environment=Environment, specs=specs)
  logging = _stub('absl.logging', info=lambda *a, **k: None,
                  warning=lambda *a, **k: None, warn=lambda *a, **k: None)
  flags = _stub('absl.flags', FLAGS=None)
  _stub('absl', logging=logging, flags=flags)
  etree = _stub('lxml.etree')
  _stub('lxml', etree=etree)

  class FakePhysicsBase:
    pass
  mj = _stub('dm_control.mujoco', Physics=FakePhysicsBase,
             action_spec=lambda physics: None)
  mjb = _stub('dm_control.mujoco.wrapper.mjbindings')
  wr = _stub('dm_control.mujoco.wrapper', mjbindings=mjb)
  mj.wrapper = wr
  sys.path.insert(0, REF)
  # import the three domain modules without running suite/__init__.py, which
  # pulls in every other domain (several need libmujoco enums at import time)
  pkg = _stub('dm_control.suite')
  pkg.__path__ = [os.path.join(REF, 'dm_control', 'suite')]


def rewards_golden():
  from dm_control.utils import rewards
  cases = []
  rs = np.random.RandomState(0)
  grid = [
      dict(bounds=(0.0, 0.0), margin=2.0, sigmoid='gaussian', vam=0.1),
      dict(bounds=(0.0, 0.0), margin=1.0, sigmoid='quadratic', vam=0.0),
      dict(bounds=(0.0, 0.0), margin=5.0, sigmoid='gaussian', vam=0.1),
      dict(bounds=(-.25, .25), margin=0.0, sigmoid='gaussian', vam=0.1),
      dict(bounds=(.995, 1.0), margin=0.0, sigmoid='gaussian', vam=0.1),
      dict(bounds=(10.0, float('inf')), margin=10.0, sigmoid='linear', vam=0.0),
      dict(bounds=(1.4, float('inf')), margin=0.35, sigmoid='gaussian', vam=0.1),
      dict(bounds=(0.9, float('inf')), margin=1.9, sigmoid='linear', vam=0.0),
      dict(bounds=(1.0, float('inf')), margin=1.0, sigmoid='linear', vam=0.0),
      dict(bounds=(0.0, 0.0), margin=2.0, sigmoid='gaussian', vam=0.1),
  ]
  for sig in ('gaussian', 'hyperbolic', 'long_tail', 'reciprocal', 'cosine',
              'linear', 'quadratic', 'tanh_squared'):
    grid.append(dict(bounds=(-0.5, 1.5), margin=0.7, sigmoid=sig, vam=0.25))
  for g in grid:
    xs = np.concatenate([rs.uniform(-4, 14, 12), [g['bounds'][0]],
                         [min(g['bounds'][1], 1e6)]])
    ys = rewards.tolerance(xs, bounds=g['bounds'], margin=g['margin'],
                           sigmoid=g['sigmoid'], value_at_margin=g['vam'])
    cases.append(dict(
        bounds=[g['bounds'][0], 'inf' if np.isinf(g['bounds'][1])
                else g['bounds'][1]],
        margin=g['margin'], sigmoid=g['sigmoid'], value_at_margin=g['vam'],
        x=xs.tolist(), y=np.asarray(ys).tolist()))
  return cases


def tasks_golden():
  from dm_control.suite import (acrobot, cartpole, cheetah, hopper, humanoid,
                                pendulum, point_mass, reacher, walker)
  rs = np.random.RandomState(1)
  out = {'cartpole': [], 'cheetah': [], 'humanoid': [], 'walker': [],
         'pendulum': [], 'acrobot': [], 'hopper': [], 'reacher': [], 'point_mass': []}

  for _ in range(24):
    x, cos, ctrl = rs.uniform(-2, 2), rs.uniform(-1, 1), rs.uniform(-1.5, 1.5)
    sin = np.sqrt(1 - cos*cos)*rs.choice([-1, 1])
    qvel = rs.uniform(-8, 8, 2)

    class P(cartpole.Physics):
      def cart_position(self): return x
      def angular_vel(self): return qvel[1:]
      def pole_angle_cosine(self): return np.array([cos])
      def bounded_position(self): return np.hstack((x, [cos, sin]))
      def control(self): return np.array([ctrl])
      def velocity(self): return qvel
    p = P()
    rec = dict(x=x, cos=cos, sin=sin, ctrl=ctrl, qvel=qvel.tolist())
    for sparse in (False, True):
      t = cartpole.Balance(swing_up=True, sparse=sparse, random=0)
      rec['reward_sparse' if sparse else 'reward_smooth'] = float(
          t.get_reward(p))
      obs = t.get_observation(p)
      rec['obs_keys'] = list(obs.keys())
      rec['obs_position'] = np.asarray(obs['position']).tolist()
      rec['obs_velocity'] = np.asarray(obs['velocity']).tolist()
    out['cartpole'].append(rec)

  for _ in range(16):
    speed = rs.uniform(-3, 14)

    class P(cheetah.Physics):
      def speed(self): return speed
    out['cheetah'].append(dict(
        speed=speed, reward=float(cheetah.Cheetah(random=0).get_reward(P()))))

  for _ in range(32):
    head, zz = rs.uniform(0.2, 1.8), rs.uniform(-1, 1)
    ctrl = rs.uniform(-1.3, 1.3, 21)
    com = rs.uniform(-3, 12, 3)

    class P(humanoid.Physics):
      def head_height(self): return head
      def torso_upright(self): return zz
      def control(self): return ctrl
      def center_of_mass_velocity(self): return com
    rec = dict(head_height=head, torso_upright=zz, ctrl=ctrl.tolist(),
               com_velocity=com.tolist())
    for speed in (0, 1, 10):
      t = humanoid.Humanoid(move_speed=speed, pure_state=False, random=0)
      rec['reward_speed_%d' % speed] = float(t.get_reward(P()))
    out['humanoid'].append(rec)

  for _ in range(32):
    height, zz, vel = rs.uniform(0.2, 1.6), rs.uniform(-1, 1), rs.uniform(-2, 10)

    class P(walker.Physics):
      def torso_height(self): return height
      def torso_upright(self): return zz
      def horizontal_velocity(self): return vel
    rec = dict(torso_height=height, torso_upright=zz, horizontal_velocity=vel)
    for speed in (0, 1, 8):
      t = walker.PlanarWalker(move_speed=speed, random=0)
      rec['reward_speed_%d' % speed] = float(t.get_reward(P()))
    out['walker'].append(rec)

  for _ in range(16):
    zz = rs.uniform(0.95, 1.0) if rs.rand() < 0.5 else rs.uniform(-1, 1)

    class P(pendulum.Physics):
      def pole_vertical(self): return zz
    out['pendulum'].append(dict(
        pole_vertical=zz, reward=float(pendulum.SwingUp(random=0).get_reward(P()))))

  # acrobot draws from its own stream so the records above stay as they were
  rs = np.random.RandomState(7)
  for _ in range(16):
    dist = rs.uniform(0, 0.3) if rs.rand() < 0.4 else rs.uniform(0, 6)
    hor, ver, vel = rs.uniform(-1, 1, 2), rs.uniform(-1, 1, 2), rs.uniform(-9, 9, 2)

    class P(acrobot.Physics):
      named = types.SimpleNamespace(model=types.SimpleNamespace(
          site_size={('target', 0): 0.2}))
      def to_target(self): return dist
      def horizontal(self): return hor
      def vertical(self): return ver
      def velocity(self): return vel
    rec = dict(to_target=dist, horizontal=hor.tolist(), vertical=ver.tolist(),
               velocity=vel.tolist())
    for sparse in (False, True):
      t = acrobot.Balance(sparse=sparse, random=0)
      rec['reward_sparse' if sparse else 'reward_smooth'] = float(t.get_reward(P()))
      obs = t.get_observation(P())
      rec['obs_keys'] = list(obs.keys())
      rec['obs_orientations'] = np.asarray(obs['orientations']).tolist()
    out['acrobot'].append(rec)

  for _ in range(24):
    height, speed = rs.uniform(0.1, 2.3), rs.uniform(-1, 4)
    ctrl, touch = rs.uniform(-1.2, 1.2, 4), rs.uniform(0, 300, 2)*(rs.rand(2) < 0.6)

    class P(hopper.Physics):
      def height(self): return height
      def speed(self): return speed
      def control(self): return ctrl
    rec = dict(height=height, speed=speed, ctrl=ctrl.tolist(),
               touch_raw=touch.tolist(), log1p_touch=np.log1p(touch).tolist())
    for hopping in (False, True):
      t = hopper.Hopper(hopping=hopping, random=0)
      rec['reward_hop' if hopping else 'reward_stand'] = float(t.get_reward(P()))
    out['hopper'].append(rec)

  class _Sizes:   # named.model.geom_size[['target', 'finger'], 0]
    def __init__(self, target): self._v = {'target': target, 'finger': 0.01}
    def __getitem__(self, key):
      names, col = key
      assert col == 0
      return np.array([self._v[n] for n in names])
  for _ in range(24):
    vec = rs.uniform(-0.08, 0.08, 2) if rs.rand() < 0.6 else rs.uniform(-0.3, 0.3, 2)
    rec = dict(to_target=vec.tolist(), dist=float(np.linalg.norm(vec)))
    for size in (0.05, 0.015):
      class P(reacher.Physics):
        named = types.SimpleNamespace(model=types.SimpleNamespace(
            geom_size=_Sizes(size)))
        def finger_to_target(self): return vec
      rec['reward_%g' % size] = float(
          reacher.Reacher(target_size=size, random=0).get_reward(P()))
    out['reacher'].append(rec)

  class _TargetSize:   # named.model.geom_size['target', 0]
    def __getitem__(self, key):
      assert key == ('target', 0)
      return 0.015
  for _ in range(24):
    dist = rs.uniform(0, 0.04) if rs.rand() < 0.6 else rs.uniform(0, 0.4)
    ctrl = rs.uniform(-1.2, 1.2, 2)

    class P(point_mass.Physics):
      named = types.SimpleNamespace(model=types.SimpleNamespace(
          geom_size=_TargetSize()))
      def mass_to_target_dist(self): return dist
      def control(self): return ctrl
    out['point_mass'].append(dict(
        dist=dist, ctrl=ctrl.tolist(),
        reward=float(point_mass.PointMass(randomize_gains=False,
                                          random=0).get_reward(P()))))
  return out


def control_golden():
  """Step-type / discount / reset sequence of control.Environment."""
  from dm_control.rl import control

  class FakePhysics(control.Physics):
    def __init__(self): self.steps = 0
    def step(self, n_sub_steps=1): self.steps += 1
    def time(self): return self.steps*0.01
    def timestep(self): return 0.01
    def reset(self): self.steps = 0
    def after_reset(self): pass
    def set_control(self, c): self.c = c

  class FakeTask(control.Task):
    def __init__(self, term_at=None): self.term_at, self.n = term_at, 0
    def initialize_episode(self, physics): self.n = 0
    def before_step(self, action, physics): physics.set_control(action)
    def after_step(self, physics): self.n += 1
    def action_spec(self, physics): return None
    def get_observation(self, physics):
      return collections.OrderedDict([('o', np.array([float(physics.steps)]))])
    def get_reward(self, physics): return 0.5*physics.steps
    def get_termination(self, physics):
      return 0.0 if self.term_at is not None and self.n >= self.term_at else None

  traces = []
  for kwargs, term_at, nsteps in (
      (dict(time_limit=0.05), None, 9),
      (dict(time_limit=0.06, n_sub_steps=2), None, 8),
      (dict(time_limit=1.0, control_timestep=0.03), 4, 9),
      (dict(), 3, 6)):
    physics = FakePhysics()
    env = control.Environment(physics, FakeTask(term_at), **kwargs)
    trace = []
    for i in range(nsteps):
      ts = env.step([0.0])
      trace.append(dict(step_type=int(ts.step_type), reward=ts.reward,
                        discount=ts.discount,
                        obs=float(ts.observation['o'][0]),
                        physics_steps=physics.steps))
    traces.append(dict(kwargs={k: v for k, v in kwargs.items()},
                       term_at=term_at, trace=trace))
  table = []
  for ctrl_dt, phys_dt in ((0.2, 0.1), (.111, .001), (100, 5), (0.03, 0.005),
                           (0.025, 0.005), (0.01, 0.01)):
    table.append(dict(control_timestep=ctrl_dt, physics_timestep=phys_dt,
                      n=control.compute_n_steps(ctrl_dt, phys_dt)))
  bad = []
  for ctrl_dt, phys_dt in ((0.1, 0.2), (0.15, 0.1)):
    try:
      control.compute_n_steps(ctrl_dt, phys_dt)
      bad.append(dict(control_timestep=ctrl_dt, physics_timestep=phys_dt,
                      raises=False))
    except ValueError:
      bad.append(dict(control_timestep=ctrl_dt, physics_timestep=phys_dt,
                      raises=True))
  obs = collections.OrderedDict([('b', np.arange(6.).reshape(2, 3)),
                                 ('a', np.array([7., 8.]))])
  flat = control.flatten_observation(obs)
  return dict(traces=traces, n_steps_table=table, n_steps_errors=bad,
              flatten=dict(keys=list(obs.keys()),
                           values=[v.tolist() for v in obs.values()],
                           flat=flat['observations'].tolist()))


def initial_states_golden():
  """qpos written by the reference's `initialize_episode` (before any physics
  runs) for seeds 0..3 of cheetah, walker, hopper, humanoid and cart-pole.

  The reference tasks are imported and run on a recording fake Physics that
  offers exactly what they touch: `model` joint tables (taken from the in-tree
  compiled models, which tests/test_compiler.py checks against the reference's
  XML files), `data.qpos/qvel/time/ncon`, `named.data.qpos[...]` views and
  no-op `step()` / `after_reset()`.  For the humanoid `ncon` reports a
  collision for the first `redraws` calls, so that the rejection loop's draw
  order is pinned as well.
  """
  sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
  sys.path.insert(0, os.path.dirname(OUT))
  import helpers
  mjb = sys.modules['dm_control.mujoco.wrapper.mjbindings']

  class mjtJoint:
    mjJNT_FREE, mjJNT_BALL, mjJNT_SLIDE, mjJNT_HINGE = 0, 1, 2, 3
  mjb.enums = types.SimpleNamespace(mjtJoint=mjtJoint)
  mjb.mjlib = None
  from dm_control.suite import cartpole, cheetah, hopper, humanoid, walker

  class NamedQpos:
    def __init__(self, model, qpos):
      self.m, self.q = model, qpos
    def _span(self, name):
      j = self.m.names['joint'].index(name)
      a = int(self.m.jnt_qposadr[j])
      n = {0: 7, 1: 4}.get(int(self.m.jnt_type[j]), 1)
      return a, n
    def __getitem__(self, name):
      if not isinstance(name, str):
        return self.q[name]
      a, n = self._span(name)
      return self.q[a:a + n]          # a view, as in mujoco/index.py
    def __setitem__(self, name, value):
      if not isinstance(name, str):
        self.q[name] = value
        return
      a, n = self._span(name)
      self.q[a:a + n] = value

  class FakeModel:
    def __init__(self, m):
      self.m = m
      for k in ('nq', 'nv', 'njnt', 'jnt_type', 'jnt_limited', 'jnt_range'):
        setattr(self, k, getattr(m, k))
    def id2name(self, i, kind):
      return self.m.names[kind][i]

  class FakePhysics:
    def __init__(self, m, redraws=0):
      self.model = FakeModel(m)
      self.data = types.SimpleNamespace(
          qpos=m.qpos0.copy(), qvel=np.zeros(m.nv), time=0.0, ncon=0)
      self.named = types.SimpleNamespace(
          data=types.SimpleNamespace(qpos=NamedQpos(m, self.data.qpos),
                                     qvel=self.data.qvel))
      self.steps, self.checks, self.redraws = 0, 0, redraws
    def step(self):
      self.steps += 1
    def after_reset(self):
      self.checks += 1
      self.data.ncon = 1 if self.checks <= self.redraws else 0

  out = {}
  cases = [('cheetah', cheetah.Cheetah, {}, 0),
           ('walker', walker.PlanarWalker, {'move_speed': 1}, 0),
           ('hopper', hopper.Hopper, {'hopping': True}, 0),
           ('humanoid', humanoid.Humanoid, {'move_speed': 1, 'pure_state': False}, 0),
           ('humanoid_two_redraws', humanoid.Humanoid,
            {'move_speed': 1, 'pure_state': False}, 2),
           ('cartpole_swingup', cartpole.Balance, {'swing_up': True, 'sparse': False}, 0),
           ('cartpole_balance', cartpole.Balance, {'swing_up': False, 'sparse': False}, 0)]
  for name, cls, kwargs, redraws in cases:
    model = helpers.load_model(name.split('_')[0])
    rows = []
    for seed in range(4):
      task = cls(random=seed, **kwargs)
      physics = FakePhysics(model, redraws)
      task.initialize_episode(physics)
      rows.append(dict(seed=seed, qpos=physics.data.qpos.tolist(),
                       qvel=physics.data.qvel.tolist(),
                       physics_steps=physics.steps))
    out[name] = rows
  return out


def main():
  install_stubs()
  with open(os.path.join(OUT, 'initial_states.json'), 'w') as f:
    json.dump(initial_states_golden(), f, indent=1)
  with open(os.path.join(OUT, 'rewards.json'), 'w') as f:
    json.dump(rewards_golden(), f, indent=1)
  with open(os.path.join(OUT, 'tasks.json'), 'w') as f:
    json.dump(tasks_golden(), f, indent=1)
  with open(os.path.join(OUT, 'control.json'), 'w') as f:
    json.dump(control_golden(), f, indent=1)
  print('golden fixtures written to', OUT)


if __name__ == '__main__':
  main()
