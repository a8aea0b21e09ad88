"""Adds the cumulative step time as an observation.

Counterpart of suite/wrappers/mujoco_profiling.py: the reference enables
MuJoCo's internal timer (`physics.enable_profiling()`) and appends
`data.timer[0]` = (cumulative seconds in mj_step, number of calls).  Here the
pair is measured with HIP events around the step launches on the batch's own
stream (`Physics.enable_profiling`), so it is device time of the physics step,
not host time.
"""

import collections

import numpy as np

from dm_control_amd import _dm_env as dm_env

specs = dm_env.specs
STATE_KEY = 'state'


class Wrapper(dm_env.Environment):
  """Wraps an environment and adds a `step_timing` observation."""

  def __init__(self, env, observation_key='step_timing'):
    wrapped = env.observation_spec()
    if isinstance(wrapped, specs.Array):
      self._observation_is_dict = False
      invalid_keys = set([STATE_KEY])
    elif isinstance(wrapped, collections.abc.MutableMapping):
      self._observation_is_dict = True
      invalid_keys = set(wrapped.keys())
    else:
      raise ValueError('Unsupported observation spec structure.')
    if observation_key in invalid_keys:
      raise ValueError(
          'Duplicate or reserved observation key {!r}.'.format(observation_key))
    if self._observation_is_dict:
      self._observation_spec = collections.OrderedDict(wrapped)
    else:
      self._observation_spec = collections.OrderedDict([(STATE_KEY, wrapped)])
    env.physics.enable_profiling()
    self._observation_spec[observation_key] = specs.Array(
        shape=(2,), dtype=np.double, name=observation_key)
    self._env = env
    self._observation_key = observation_key

  def reset(self):
    return self._add_profile_observation(self._env.reset())

  def step(self, action):
    return self._add_profile_observation(self._env.step(action))

  def observation_spec(self):
    return self._observation_spec

  def action_spec(self):
    return self._env.action_spec()

  def _add_profile_observation(self, time_step):
    if self._observation_is_dict:
      observation = collections.OrderedDict(time_step.observation)
    else:
      observation = collections.OrderedDict(
          [(STATE_KEY, time_step.observation)])
    timing = self._env.physics.data.timer[0]
    observation[self._observation_key] = np.array([timing[0], timing[1]],
                                                  dtype=np.double)
    return time_step._replace(observation=observation)

  def __getattr__(self, name):
    return getattr(self._env, name)
