"""Humanoid domain (cf. /root/reference/dm_control/suite/humanoid.py)."""

import numpy as np

from dm_control_amd import codegen
from dm_control_amd import engine
from dm_control_amd.mjcf import model as mdl
from dm_control_amd.rl import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite.randomizers import randomize_limited_and_rotational_joints
from dm_control_amd.utils import containers

_DEFAULT_TIME_LIMIT = 25
_CONTROL_TIMESTEP = .025
_WALK_SPEED = 1
_RUN_SPEED = 10
_MAX_REJECTION_ROUNDS = 200

SUITE = containers.TaggedTasks()


def get_model_and_assets():
  return common.read_model('humanoid.xml'), common.ASSETS


def _make_env(move_speed, pure_state, time_limit, random, environment_kwargs):
  phys_kw, task_kw, env_kw = common.split_kwargs(environment_kwargs)
  physics = Physics.from_xml_string(*get_model_and_assets(), **phys_kw)
  task = Humanoid(move_speed=move_speed, pure_state=pure_state, random=random,
                  **task_kw)
  physics.set_task_params(0, (float(move_speed),))
  return control.Environment(physics, task, time_limit=time_limit,
                             control_timestep=_CONTROL_TIMESTEP, **env_kw)


@SUITE.add('benchmarking')
def stand(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None):
  return _make_env(0, False, time_limit, random, environment_kwargs)


@SUITE.add('benchmarking')
def walk(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None):
  return _make_env(_WALK_SPEED, False, time_limit, random, environment_kwargs)


@SUITE.add('benchmarking')
def run(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None):
  return _make_env(_RUN_SPEED, False, time_limit, random, environment_kwargs)


@SUITE.add()
def run_pure_state(time_limit=_DEFAULT_TIME_LIMIT, random=None,
                   environment_kwargs=None):
  return _make_env(_RUN_SPEED, True, time_limit, random, environment_kwargs)


class Physics(engine.Physics):
  """Physics with the Humanoid helpers (humanoid.py:93-129)."""

  _TASK = codegen.TASK_HUMANOID
  # 27 dofs: one env per 64-lane group with its working set in LDS
  # (csrc/dmc_coop.hip); the one-lane build spills the 27 x 27 matrices
  _BUILD_MODE = 'coop'
  _GROUP = 128   # + a second wavefront per env for the constraint rows (fp32 and fp64)

  def torso_upright(self):
    return self.named.data.xmat['torso', 'zz']

  def head_height(self):
    return self.named.data.xpos['head', 'z']

  def center_of_mass_velocity(self):
    return self.named.data.sensordata['torso_subtreelinvel'].copy()

  def torso_vertical_orientation(self):
    return self.named.data.xmat['torso', ['zx', 'zy', 'zz']]

  def joint_angles(self):
    return self.data.qpos[..., 7:].copy()

  def extremities(self):
    nq = self.model.nq
    return self.fused_observation()[..., nq - 7 + 1:nq - 7 + 13]


class Humanoid(base.Task):
  """Stand / walk / run (humanoid.py:132-207)."""

  def __init__(self, move_speed, pure_state, random=None, device_init=False):
    self._move_speed = move_speed
    self._pure_state = pure_state
    super().__init__(random=random, device_init=device_init)

  def initialize_episode(self, physics):
    """Collision-free random initial configuration (humanoid.py:152-166)."""
    batch = physics.batch
    n = batch.nenv
    if self._device_init:
      batch.init_episode(self.device_seed())
      for _ in range(_MAX_REJECTION_ROUNDS):
        physics.after_reset()
        if not np.any(np.atleast_1d(physics.data.ncon) > 0):
          break
        batch.init_episode(self.device_seed(), only_colliding=True)
    else:
      streams = self.streams(physics)
      qpos = np.tile(physics.model.qpos0, (n, 1))
      penetrating = np.ones(n, bool)
      for _ in range(_MAX_REJECTION_ROUNDS):
        for i in np.nonzero(penetrating)[0]:
          randomize_limited_and_rotational_joints(physics.model, qpos[i],
                                                  streams[i])
        physics.data.qpos[:] = qpos[0] if physics.batch_size is None else qpos
        physics.after_reset()
        penetrating = np.atleast_1d(physics.data.ncon) > 0
        if not penetrating.any():
          break
    super().initialize_episode(physics)

  def get_observation(self, physics):
    m = physics.model
    if self._pure_state:
      obs = control.BatchedObservation()
      obs.batch_size = physics.batch_size
      obs['position'] = physics.position()
      obs['velocity'] = physics.velocity()
      return obs
    return self._obs_dict(physics, [
        ('joint_angles', m.nq - 7, False), ('head_height', 1, True),
        ('extremities', 12, False), ('torso_vertical', 3, False),
        ('com_velocity', 3, False), ('velocity', m.nv, False)])
