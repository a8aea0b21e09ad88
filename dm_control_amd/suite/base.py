"""Base class of the suite tasks (cf. /root/reference/dm_control/suite/base.py).

Actions map directly to actuator controls (base.py:73-77).  Observation and
reward are produced by the step kernel; the task classes only slice the fused
vector into the reference's OrderedDict layout.  Initial states follow each
domain's `initialize_episode` recipe either on the host (numpy RandomState,
same call order as the reference, one stream per instance) or on the device
(`device_init=True`, counter-based RNG; same distributions).
"""

import numpy as np

from dm_control_amd import engine
from dm_control_amd.rl import control


class Task(control.Task):
  """Control Suite task over a batched Physics."""

  def __init__(self, random=None, device_init=False):
    if not isinstance(random, np.random.RandomState):
      random = np.random.RandomState(random)
    self._random = random
    self._visualize_reward = False
    self._device_init = bool(device_init)
    self._episode = 0
    self._streams = None

  @property
  def random(self):
    return self._random

  def streams(self, physics):
    """One RandomState per instance (instance 0 = `self.random`)."""
    n = physics.batch_size
    if n is None:
      return [self._random]
    if self._streams is None or len(self._streams) != n:
      # seeds for instances 1.. come from a COPY of stream 0, so instance 0
      # draws exactly what the reference's single env would draw
      fork = np.random.RandomState()
      fork.set_state(self._random.get_state())
      seeds = fork.randint(0, 2**31 - 1, size=n - 1) if n > 1 else []
      self._streams = [self._random] + [np.random.RandomState(int(s))
                                        for s in seeds]
    return self._streams

  def device_seed(self):
    self._episode += 1
    base = int(self._random.randint(0, 2**31 - 1))
    return (base << 20) + self._episode

  def action_spec(self, physics):
    return engine.action_spec(physics)

  def initialize_episode(self, physics):
    self.after_step(physics)

  def before_step(self, action, physics):
    action = getattr(action, 'continuous_actions', action)
    physics.set_control(action)

  def after_step(self, physics):
    """Reward colouring is a rendering-only side effect (base.py:79-83)."""

  @property
  def visualize_reward(self):
    return self._visualize_reward

  @visualize_reward.setter
  def visualize_reward(self, value):
    if not isinstance(value, bool):
      raise ValueError('Expected a boolean, got {}.'.format(type(value)))
    self._visualize_reward = value

  def get_reward(self, physics):
    return physics.fused_reward()

  def _obs_dict(self, physics, fields):
    """Slices the fused [B, nobs] vector into named leaves."""
    vec = physics.fused_observation()
    obs = control.BatchedObservation()
    obs.batch_size = physics.batch_size
    start = 0
    for name, size, scalar in fields:
      leaf = vec[..., start:start + size]
      if scalar:
        leaf = leaf[..., 0]
      obs[name] = np.array(leaf, dtype=np.float64)
      start += size
    return obs
