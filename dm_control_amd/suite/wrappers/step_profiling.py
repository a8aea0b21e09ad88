"""Adds the cumulative physics-step time as an observation.

Role of the reference's suite/wrappers/mujoco_profiling.py: there MuJoCo's
internal timer is switched on and `data.timer[0]` = (cumulative seconds spent in
mj_step, number of calls) is appended under `step_timing`.  Here the pair comes
from HIP events around the step launches on the batch's stream
(`Physics.enable_profiling`), i.e. it is device time of the physics step.
"""

import collections

import numpy as np

from dm_control_amd import _dm_env as dm_env

specs = dm_env.specs
STATE_KEY = 'state'


def _as_mapping(value, is_dict):
  """The wrapped observation (or its spec) as an OrderedDict."""
  if is_dict:
    return collections.OrderedDict(value)
  return collections.OrderedDict([(STATE_KEY, value)])


class Wrapper(dm_env.Environment):
  """Environment + a `[seconds, calls]` observation of the step timer."""

  def __init__(self, env, observation_key='step_timing'):
    spec = env.observation_spec()
    self._is_dict = isinstance(spec, collections.abc.MutableMapping)
    if not self._is_dict and not isinstance(spec, specs.Array):
      raise ValueError('Unsupported observation spec structure.')
    taken = set(spec.keys()) if self._is_dict else {STATE_KEY}
    if observation_key in taken:
      raise ValueError(
          'Duplicate or reserved observation key {!r}.'.format(observation_key))
    self._key = observation_key
    self._env = env
    self._spec = _as_mapping(spec, self._is_dict)
    self._spec[observation_key] = specs.Array((2,), np.double,
                                              name=observation_key)
    env.physics.enable_profiling()

  def _with_timing(self, time_step):
    observation = _as_mapping(time_step.observation, self._is_dict)
    seconds, calls = self._env.physics.data.timer[0]
    observation[self._key] = np.array([seconds, calls], dtype=np.double)
    return time_step._replace(observation=observation)

  def reset(self):
    return self._with_timing(self._env.reset())

  def step(self, action):
    return self._with_timing(self._env.step(action))

  def observation_spec(self):
    return self._spec

  def action_spec(self):
    return self._env.action_spec()

  def __getattr__(self, name):
    return getattr(self._env, name)
