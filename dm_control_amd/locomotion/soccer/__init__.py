"""Batched multi-agent soccer on the MI355X step (BASELINE configs[4]).

Stands in for `dm_control.locomotion.soccer.load` (soccer/__init__.py:92-152)
with humanoid walkers: the same call returns an environment whose `step` takes
one action per player and returns per-player rewards and observations, for B
pitches at once.

What is built, and what it replaces:

  scene      `locomotion/models/soccer.py`: pitch (ground + four wall planes,
             pitch.py:404-424), two goal frames of ten capsules each
             (pitch.py:55-64, 165-198), the regulation ball (soccer_ball.py:27-
             47, condim 6, priority 1), 2 x team_size position-controlled CMU
             humanoids (walkers/cmu_humanoid.py), `disable_walker_contacts`
             (task.py:29-33).  One frozen MJCF per (team_size, pitch size).
  physics    the whole scene is one model (2v2: nv 254) stepped by the
             one-env-per-lane kernel in its rolled form with the packed
             matrices in the HBM workspace (csrc/dmc_kernels.hip, MAT_IN_WS).
  game logic `Task`: goal detection (the ball's centre inside a goal box:
             pitch.py:426-447, 574-580), reward +1 / -1 per player and discount
             0 on a goal (task.py:161-200), throw-in when the ball leaves the
             field rectangle (task.py:117-124, 205-206, pitch.py:449-457,
             582-583), the uniform initialiser with its keep-apart retry
             (initializers.py:32-120), time limit.
  observables per player, the names and formulas of `CoreObservablesAdder`
             (observables.py:59-240) for: proprioception (joints_pos,
             joints_vel, body_height, world_zaxis, prev_action), the ball, every
             other player (position, linear velocity, orientation) and the eight
             arena features, all in the player's egocentric frame
             (composer/entity.py:340-375: v . xmat).

Stated deviations: the pitch size is fixed per environment (`RandomizedPitch`
redraws it every episode and the reference recompiles the model,
composer/environment.py:321-333); goals are detected once per control step, not
per physics substep; walkers start upright at random places and headings (the
reference's `CMUMocapInitializer` needs a 488 MB download); end-effector and
sensor observables (`end_effectors_pos`, `sensors_*`, `stats_*`) are not
produced; BOXHEAD / ANT walkers and the field box are not built.
"""

import collections
import enum

import numpy as np

from dm_control_amd import _dm_env as dm_env
from dm_control_amd import engine
from dm_control_amd.locomotion.models import soccer as scene
from dm_control_amd.rl import control
from dm_control_amd.suite import common

specs = dm_env.specs

# pitch.py:716-724: mini-football goal and the area a humanoid is given
GOAL_LENGTH, GOAL_SIDE = 3.66, 1.22
MINI_FOOTBALL_GOAL_SIZE = (GOAL_SIDE/2, GOAL_LENGTH/2, GOAL_SIDE/2)
MINI_FOOTBALL_MIN_AREA_PER_HUMANOID = 100.0
MINI_FOOTBALL_MAX_AREA_PER_HUMANOID = 350.0
SPAWN_RATIO = 0.6             # initializers.py: share of the pitch used for kick-off
INIT_BALL_Z = 0.5
THROW_IN_BALL_Z = 0.5         # task.py:26
WALKER_HEIGHT = 1.05          # root height of the upright CMU humanoid in this scene
NQ_WALKER, NV_WALKER, NU_WALKER = 63, 62, 56


class WalkerType(enum.Enum):
  BOXHEAD = 0
  ANT = 1
  HUMANOID = 2


class Team(enum.Enum):
  HOME = 0
  AWAY = 1


def area_to_size(area, aspect_ratio=0.75):
  """(half length, half width) of a pitch of `area` (soccer/__init__.py:84-86)."""
  return tuple(np.sqrt([area/aspect_ratio, area*aspect_ratio])/2.0)


class PitchGeometry:
  """The detector boxes of `Pitch` (pitch.py:426-457) as plain arrays."""

  def __init__(self, size, goal_size, field_box_offset=0.0):
    self.size = tuple(float(v) for v in size)
    self.goal_size = tuple(float(v) for v in goal_size)
    lx, ly = self.size
    gd, gw, gh = self.goal_size
    home = np.array([-lx + gd + field_box_offset, 0.0, gh])
    away = np.array([lx - gd - field_box_offset, 0.0, gh])
    half = np.array([gd, gw, gh])
    self.home_goal = (home - half, home + half, home)      # lower, upper, mid
    self.away_goal = (away - half, away + half, away)
    fhalf = np.array([lx - 2*gd, ly - 2*gd])
    self.field = (-fhalf, fhalf)

  def in_goal(self, pos, goal):
    lower, upper, _ = goal
    return np.all((pos >= lower) & (pos <= upper), axis=-1)

  def off_court(self, pos):
    lower, upper = self.field
    xy = pos[..., :2]
    return ~np.all((xy >= lower) & (xy <= upper), axis=-1)


def _quat_to_mat(q):
  """[..., 4] unit quaternions (w, x, y, z) -> [..., 3, 3]."""
  w, x, y, z = (q[..., k] for k in range(4))
  m = np.empty(q.shape[:-1] + (3, 3))
  m[..., 0, 0] = 1 - 2*(y*y + z*z); m[..., 0, 1] = 2*(x*y - w*z); m[..., 0, 2] = 2*(x*z + w*y)
  m[..., 1, 0] = 2*(x*y + w*z); m[..., 1, 1] = 1 - 2*(x*x + z*z); m[..., 1, 2] = 2*(y*z - w*x)
  m[..., 2, 0] = 2*(x*z - w*y); m[..., 2, 1] = 2*(y*z + w*x); m[..., 2, 2] = 1 - 2*(x*x + y*y)
  return m


def _ego(vec, xmat):
  """`vec . xmat` per instance (composer/entity.py:368-371); 2-vectors use the
  upper-left 2 x 2 block."""
  k = vec.shape[-1]
  return np.einsum('...i,...ij->...j', vec, xmat[..., :k, :k])


class Physics(engine.Physics):
  """The whole pitch as one model; one wavefront per pitch ("team" build of
  csrc/dmc_kernels.hip: generic loops, the matrices of a pitch in the HBM
  workspace and one kinematic tree's block at a time in LDS)."""

  _BUILD_MODE = 'team'


class Task(control.Task):
  """Two teams of humanoids, a ball, two goals (soccer/task.py:36-214)."""

  def __init__(self, team_size, pitch, random=None, terminate_on_goal=True):
    if not isinstance(random, np.random.RandomState):
      random = np.random.RandomState(random)
    self._random = random
    self.team_size = int(team_size)
    self.num_players = 2*self.team_size
    # player order as soccer/__init__.py:72-81: the home team, then the away team
    self.teams = [Team.HOME]*self.team_size + [Team.AWAY]*self.team_size
    self.pitch = pitch
    self._terminate_on_goal = bool(terminate_on_goal)
    self._scored = None          # [B] -1 none, 0 home scored, 1 away scored (last step)
    self._prev_action = None

  # -- layout of the scene's state vectors ---------------------------------------
  # (walkers first, the ball last: locomotion/models/soccer.py)
  def _walker_q(self, k):
    return NQ_WALKER*k

  def _walker_v(self, k):
    return NV_WALKER*k

  @property
  def _ball_q(self):
    return NQ_WALKER*self.num_players

  @property
  def _ball_v(self):
    return NV_WALKER*self.num_players

  # -- episode -------------------------------------------------------------------
  def _draw_kickoff(self, rs):
    """Ball and players uniformly over the spawn range, random headings; redrawn
    while two of them are closer than they could stand without touching
    (initializers.py:77-120: there the test is `physics.forward()` contacts)."""
    span = np.asarray(self.pitch.size)*SPAWN_RATIO
    for _ in range(100):
      pts = rs.uniform(-span, span, size=(1 + self.num_players, 2))
      heading = rs.uniform(-np.pi, np.pi, size=self.num_players)
      d = np.linalg.norm(pts[:, None] - pts[None], axis=-1) + 10*np.eye(len(pts))
      if d.min() > 1.0:
        break
    return pts, heading

  def initialize_episode(self, physics, only=None):
    m = physics.model
    n = physics.batch_size or 1
    qpos = np.atleast_2d(np.asarray(physics.data.qpos, np.float64)).copy()
    qvel = np.atleast_2d(np.asarray(physics.data.qvel, np.float64)).copy()
    rows = range(n) if only is None else np.nonzero(only)[0]
    for e in rows:
      pts, heading = self._draw_kickoff(self._random)
      qpos[e] = m.qpos0
      qvel[e] = 0
      bq = self._ball_q
      qpos[e, bq:bq + 2] = pts[0]
      qpos[e, bq + 2] = INIT_BALL_Z
      for k in range(self.num_players):
        a = self._walker_q(k)
        qpos[e, a:a + 2] = pts[1 + k]
        qpos[e, a + 2] = WALKER_HEIGHT
        # upright (the scene's root orientation) turned by `heading` about z
        turn = np.array([np.cos(heading[k]/2), 0, 0, np.sin(heading[k]/2)])
        qpos[e, a + 3:a + 7] = _quat_mul(turn, m.qpos0[a + 3:a + 7])
    physics.set_state(np.concatenate([qpos, qvel], axis=1) if physics.batch_size
                      else np.concatenate([qpos[0], qvel[0]]))
    if only is None:
      self._scored = np.full(n, -1)
      self._prev_action = np.zeros((n, m.nu))

  def before_step(self, action, physics):
    """One action per player (task.py:202-206), throw-in for balls off court,
    kick-off again where a goal ended the last rally."""
    n = physics.batch_size or 1
    if isinstance(action, (list, tuple)):
      action = np.concatenate([np.atleast_2d(a) for a in action], axis=-1)
    action = np.asarray(action, np.float64).reshape(n, physics.model.nu)
    self._prev_action = action
    if self._scored is not None and (self._scored >= 0).any():
      self.initialize_episode(physics, only=self._scored >= 0)
      self._scored[:] = -1
    ball = np.atleast_2d(np.asarray(physics.data.qpos, np.float64))[
        :, self._ball_q:self._ball_q + 3]
    out = self.pitch.off_court(ball)
    if out.any():
      self._throw_in(physics, out)
    physics.set_control(action if physics.batch_size else action[0])

  def _throw_in(self, physics, mask):
    qpos = np.atleast_2d(np.asarray(physics.data.qpos, np.float64)).copy()
    qvel = np.atleast_2d(np.asarray(physics.data.qvel, np.float64)).copy()
    for e in np.nonzero(mask)[0]:
      shrink = self._random.uniform([0.7, 0.7], [0.9, 0.9])
      bq, bv = self._ball_q, self._ball_v
      qpos[e, bq:bq + 2] *= shrink
      qpos[e, bq + 2] = THROW_IN_BALL_Z
      qpos[e, bq + 3:bq + 7] = [1, 0, 0, 0]
      qvel[e, bv:bv + 6] = 0
    physics.set_state(np.concatenate([qpos, qvel], axis=1) if physics.batch_size
                      else np.concatenate([qpos[0], qvel[0]]))

  def after_step(self, physics):
    ball = np.atleast_2d(np.asarray(physics.data.qpos, np.float64))[
        :, self._ball_q:self._ball_q + 3]
    scored = np.full(len(ball), -1)
    scored[self.pitch.in_goal(ball, self.pitch.home_goal)] = 1    # away team scores
    scored[self.pitch.in_goal(ball, self.pitch.away_goal)] = 0
    self._scored = scored

  # -- outputs -------------------------------------------------------------------
  def get_reward(self, physics):
    """Per player [B] (or scalar): +1 scored, -1 conceded, 0 otherwise."""
    out = []
    for team in self.teams:
      r = np.zeros(len(self._scored), np.float32)
      r[self._scored == team.value] = 1.0
      r[(self._scored >= 0) & (self._scored != team.value)] = -1.0
      out.append(r if physics.batch_size else r[0])
    return out

  def get_termination(self, physics):
    """None while no pitch has seen a goal; otherwise the per-pitch discount
    (0 where a goal was scored, task.py:191-194).  With `terminate_on_goal` the
    pitches that scored kick off again at the next step (per-instance
    auto-reset, as scripts/vec_env.py:346-352 does per env); the batch as a whole
    ends at the time limit."""
    return None    # per-pitch restarts: see before_step; `discount()` carries the zeros

  def discount(self, physics):
    d = np.ones(len(self._scored), np.float32)
    if self._terminate_on_goal:
      d[self._scored >= 0] = 0.0
    return d if physics.batch_size else d[0]

  def action_spec(self, physics):
    return [specs.BoundedArray((NU_WALKER,), np.float64, -1.0, 1.0, name='action')
            for _ in range(self.num_players)]

  def get_reward_spec(self):
    return [specs.Array((), np.float32, name='reward') for _ in range(self.num_players)]

  def get_observation(self, physics):
    qpos = np.atleast_2d(np.asarray(physics.data.qpos, np.float64))
    qvel = np.atleast_2d(np.asarray(physics.data.qvel, np.float64))
    n = len(qpos)
    bq, bv = self._ball_q, self._ball_v
    ball_pos, ball_lin = qpos[:, bq:bq + 3], qvel[:, bv:bv + 3]
    ball_ang = _ego(qvel[:, bv + 3:bv + 6],
                    np.swapaxes(_quat_to_mat(qpos[:, bq + 3:bq + 7]), -1, -2))  # local -> world
    root_pos, root_mat, root_lin = [], [], []
    for k in range(self.num_players):
      a, v = self._walker_q(k), self._walker_v(k)
      root_pos.append(qpos[:, a:a + 3])
      root_mat.append(_quat_to_mat(qpos[:, a + 3:a + 7]))
      root_lin.append(qvel[:, v:v + 3])
    g = self.pitch
    features = [(g.home_goal[0][:2], 2), (g.home_goal[2], 3), (g.home_goal[1][:2], 2),
                (g.field[1], 2), (g.away_goal[1][:2], 2), (g.away_goal[2], 3),
                (g.away_goal[0][:2], 2), (g.field[0], 2)]
    names = ['team_goal_back_right', 'team_goal_mid', 'team_goal_front_left',
             'field_front_left', 'opponent_goal_back_left', 'opponent_goal_mid',
             'opponent_goal_front_right', 'field_back_right']
    prev = self._prev_action if self._prev_action is not None else np.zeros((n, 0))
    out = []
    for k, team in enumerate(self.teams):
      a, v = self._walker_q(k), self._walker_v(k)
      mat, pos, lin = root_mat[k], root_pos[k], root_lin[k]
      obs = control.BatchedObservation()
      obs.batch_size = physics.batch_size
      obs['joints_pos'] = qpos[:, a + 7:a + NQ_WALKER].copy()
      obs['joints_vel'] = qvel[:, v + 6:v + NV_WALKER].copy()
      obs['body_height'] = pos[:, 2].copy()
      obs['world_zaxis'] = mat[:, 2, :].copy()
      obs['prev_action'] = prev[:, NU_WALKER*k:NU_WALKER*(k + 1)].copy()
      obs['ball_ego_angular_velocity'] = _ego(ball_ang, mat)
      obs['ball_ego_position'] = _ego(ball_pos - pos, mat)
      obs['ball_ego_linear_velocity'] = _ego(ball_lin - lin, mat)
      mates = foes = 0
      for j, other in enumerate(self.teams):
        if j == k:
          continue
        if other == team:
          prefix, mates = 'teammate_%d' % mates, mates + 1
        else:
          prefix, foes = 'opponent_%d' % foes, foes + 1
        obs[prefix + '_ego_linear_velocity'] = _ego(root_lin[j] - lin, mat)
        obs[prefix + '_ego_position'] = _ego(root_pos[j] - pos, mat)
        obs[prefix + '_ego_orientation'] = np.einsum(
            '...ji,...jk->...ik', mat, root_mat[j]).reshape(n, 9)
      order = list(range(8)) if team == Team.HOME else list(range(4, 8)) + list(range(4))
      for name, idx in zip(names, order):
        feature, dim = features[idx]
        obs[name] = _ego(feature[None, :dim] - pos[:, :dim], mat)
      if physics.batch_size is None:
        for key in obs:
          obs[key] = obs[key][0]
      out.append(obs)
    return out


def _quat_mul(a, b):
  aw, ax, ay, az = a
  bw, bx, by, bz = b
  return np.array([aw*bw - ax*bx - ay*by - az*bz, aw*bx + ax*bw + ay*bz - az*by,
                   aw*by - ax*bz + ay*bw + az*bx, aw*bz + ax*by - ay*bx + az*bw])


class Environment(control.Environment):
  """`control.Environment` with the per-pitch discount of a goal."""

  def step(self, action):
    ts = super().step(action)
    if ts.first() or ts.discount is None:
      return ts
    return ts._replace(discount=self._task.discount(self._physics))


def load(team_size, time_limit=45., random_state=None, disable_walker_contacts=False,
         enable_field_box=False, keep_aspect_ratio=False, terminate_on_goal=True,
         walker_type=WalkerType.HUMANOID, pitch_size=None, control_timestep=0.025,
         environment_kwargs=None):
  """`team_size`-vs-`team_size` soccer (signature of soccer/__init__.py:92-99, the
  walker type defaulting to the one that is built).

  pitch_size: (half length, half width); default: the smallest mini-football
  pitch for this many humanoids (soccer/__init__.py:130-135).
  environment_kwargs: `batch_size`, `device`, `precision` for the batched Physics.
  """
  del keep_aspect_ratio      # the pitch does not change between episodes
  if not 1 <= team_size <= 11:
    raise ValueError('team_size must be between 1 and 11, got {}'.format(team_size))
  if walker_type != WalkerType.HUMANOID:
    raise ValueError('walker type {} is not built; WalkerType.HUMANOID is'.format(walker_type))
  if enable_field_box:
    raise ValueError('the field box is not built')
  num_walkers = 2*team_size
  if pitch_size is None:
    pitch_size = area_to_size(MINI_FOOTBALL_MIN_AREA_PER_HUMANOID*num_walkers)
  geometry = PitchGeometry(pitch_size, MINI_FOOTBALL_GOAL_SIZE)
  xml = scene.build(num_walkers, with_ball=True, pitch_size=geometry.size,
                    disable_walker_contacts=disable_walker_contacts,
                    ball=scene.REGULATION_BALL, goal_size=MINI_FOOTBALL_GOAL_SIZE)
  phys_kw, _, env_kw = common.split_kwargs(environment_kwargs)
  # contact capacity: 40 per player (the reference allocates 200 per player,
  # task.py:105-108).  Fallen humanoids with self-collisions on reached 27 contacts
  # per player in a 600-step soak of 512 pitches under random actions
  # (tools/debug/soccer_soak.py; 16 per player raised mjWARN_CONTACTFULL in 71 of
  # them); beyond the capacity the step raises mjWARN_CONTACTFULL like the reference
  phys_kw.setdefault('ncon_max', 40*num_walkers)
  physics = Physics.from_xml_string(xml, **phys_kw)
  task = Task(team_size, geometry, random=random_state,
              terminate_on_goal=terminate_on_goal)
  return Environment(physics, task, time_limit=time_limit,
                     control_timestep=control_timestep, **env_kw)
