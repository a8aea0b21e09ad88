"""Reacher domain (cf. /root/reference/dm_control/suite/reacher.py).

The reference moves the target by rewriting `model.geom_pos['target']` every
episode; here the target's x, y are per-instance task data in HBM
(`DMC_FIELD_TASKDATA`), read by the fused observation / reward.
"""

import numpy as np

from dm_control_amd import codegen
from dm_control_amd import engine
from dm_control_amd import wrapper
from dm_control_amd.rl import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.suite import randomizers
from dm_control_amd.utils import containers

SUITE = containers.TaggedTasks()
_DEFAULT_TIME_LIMIT = 20
_BIG_TARGET = .05
_SMALL_TARGET = .015


def get_model_and_assets():
  return common.read_model('reacher.xml'), common.ASSETS


def _make(target_size, time_limit, random, environment_kwargs):
  phys_kw, task_kw, env_kw = common.split_kwargs(environment_kwargs)
  physics = Physics.from_xml_string(*get_model_and_assets(), **phys_kw)
  task = Reacher(target_size=target_size, random=random, **task_kw)
  finger = physics.model.geom_size[physics.model.name2id('finger', 'geom'), 0]
  physics.set_task_params(rparams=(target_size + finger,))
  return control.Environment(physics, task, time_limit=time_limit, **env_kw)


@SUITE.add('benchmarking', 'easy')
def easy(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None):
  """Reacher with a 5e-2 target (reacher.py:40-47)."""
  return _make(_BIG_TARGET, time_limit, random, environment_kwargs)


@SUITE.add('benchmarking')
def hard(time_limit=_DEFAULT_TIME_LIMIT, random=None, environment_kwargs=None):
  """Reacher with a 1.5e-2 target (reacher.py:50-57)."""
  return _make(_SMALL_TARGET, time_limit, random, environment_kwargs)


class Physics(engine.Physics):
  """Physics with the Reacher helpers (reacher.py:60-71).

  Fused observation layout: [qpos, target - finger (x, y), qvel].
  """

  _TASK = codegen.TASK_REACHER

  def finger_to_target(self):
    nq = self.model.nq
    return self.fused_observation()[..., nq:nq + 2]

  def finger_to_target_dist(self):
    return np.linalg.norm(self.finger_to_target(), axis=-1)

  def target_position(self):
    """x, y of the target of every instance."""
    t = self.batch.read(wrapper.FIELD_TASKDATA).T.astype(np.float64)
    return t[0] if self.batch_size is None else t


class Reacher(base.Task):
  """Reach the randomly placed target (reacher.py:74-122)."""

  def __init__(self, target_size, random=None, device_init=False):
    self._target_size = target_size
    super().__init__(random=random, device_init=device_init)

  def initialize_episode(self, physics):
    if self._device_init:
      physics.batch.init_episode(self.device_seed())
    else:
      # per instance, in the reference's order: joints, then angle, then radius
      qpos, target = [], []
      for rs in self.streams(physics):
        q = physics.model.qpos0.copy()
        randomizers.randomize_limited_and_rotational_joints(physics.model, q, rs)
        angle = rs.uniform(0, 2*np.pi)
        radius = rs.uniform(.05, .20)
        qpos.append(q)
        target.append((radius*np.sin(angle), radius*np.cos(angle)))
      qpos = np.array(qpos)
      physics.data.qpos[:] = qpos[0] if physics.batch_size is None else qpos
      physics.batch.write(wrapper.FIELD_TASKDATA, np.array(target).T)
    super().initialize_episode(physics)

  def get_observation(self, physics):
    m = physics.model
    return self._obs_dict(physics, [('position', m.nq, False),
                                    ('to_target', 2, False),
                                    ('velocity', m.nv, False)])
