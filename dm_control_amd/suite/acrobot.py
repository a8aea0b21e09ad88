"""Acrobot domain (cf. /root/reference/dm_control/suite/acrobot.py)."""

import numpy as np

from dm_control_amd import codegen
from dm_control_amd import engine
from dm_control_amd.rl import control
from dm_control_amd.suite import base
from dm_control_amd.suite import common
from dm_control_amd.utils import containers

_DEFAULT_TIME_LIMIT = 10
SUITE = containers.TaggedTasks()


def get_model_and_assets():
  return common.read_model('acrobot.xml'), common.ASSETS


def _make(sparse, time_limit, random, environment_kwargs):
  phys_kw, task_kw, env_kw = common.split_kwargs(environment_kwargs)
  physics = Physics.from_xml_string(*get_model_and_assets(), **phys_kw)
  task = Balance(sparse=sparse, random=random, **task_kw)
  physics.set_task_params(iparam=1 if sparse else 0)
  return control.Environment(physics, task, time_limit=time_limit, **env_kw)


@SUITE.add('benchmarking')
def swingup(time_limit=_DEFAULT_TIME_LIMIT, random=None,
            environment_kwargs=None):
  """Returns the Acrobot balance task (acrobot.py:37-45)."""
  return _make(False, time_limit, random, environment_kwargs)


@SUITE.add('benchmarking')
def swingup_sparse(time_limit=_DEFAULT_TIME_LIMIT, random=None,
                   environment_kwargs=None):
  """Returns the sparse-reward variant (acrobot.py:48-56)."""
  return _make(True, time_limit, random, environment_kwargs)


class Physics(engine.Physics):
  """Physics with the Acrobot helpers (acrobot.py:59-81).

  Fused observation layout: [xz(upper), xz(lower), zz(upper), zz(lower), qvel].
  """

  _TASK = codegen.TASK_ACROBOT

  def horizontal(self):
    return self.fused_observation()[..., 0:2]

  def vertical(self):
    return self.fused_observation()[..., 2:4]

  def orientations(self):
    return self.fused_observation()[..., 0:4]

  def to_target(self):
    """Distance from the tip site to the target site."""
    site = self.named.data.site_xpos
    return np.linalg.norm(site['target'] - site['tip'], axis=-1)


class Balance(base.Task):
  """Swing up and balance the two-link pole (acrobot.py:84-126)."""

  def __init__(self, sparse, random=None, device_init=False):
    self._sparse = sparse
    super().__init__(random=random, device_init=device_init)

  def initialize_episode(self, physics):
    if self._device_init:
      physics.batch.init_episode(self.device_seed())
    else:
      # shoulder and elbow uniform in [-pi, pi): acrobot.py:106-107
      qpos = np.stack([rs.uniform(-np.pi, np.pi, 2)
                       for rs in self.streams(physics)])
      physics.data.qpos[:] = qpos[0] if physics.batch_size is None else qpos
    super().initialize_episode(physics)

  def get_observation(self, physics):
    return self._obs_dict(physics, [('orientations', 4, False),
                                    ('velocity', 2, False)])
