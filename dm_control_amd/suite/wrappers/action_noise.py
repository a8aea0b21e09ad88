"""Adds Gaussian noise to actions (cf. suite/wrappers/action_noise.py)."""

import numpy as np

from dm_control_amd import _dm_env as dm_env

_BOUNDS_MUST_BE_FINITE = (
    'All bounds in `env.action_spec()` must be finite, got: {action_spec}')


class Wrapper(dm_env.Environment):
  """Noise std = `scale` x (max - min) per action dimension, then clipped.

  The noise comes from the task's RandomState like in the reference; a batched
  action `[B, nu]` gets independent noise per instance (one draw of that shape).
  """

  def __init__(self, env, scale=0.01):
    action_spec = env.action_spec()
    if not (np.all(np.isfinite(action_spec.minimum)) and
            np.all(np.isfinite(action_spec.maximum))):
      raise ValueError(_BOUNDS_MUST_BE_FINITE.format(action_spec=action_spec))
    self._minimum = action_spec.minimum
    self._maximum = action_spec.maximum
    self._noise_std = scale*(action_spec.maximum - action_spec.minimum)
    self._env = env

  def step(self, action):
    action = np.asarray(action, dtype=np.float64)
    noise = self._env.task.random.normal(
        scale=np.broadcast_to(self._noise_std, action.shape))
    noisy_action = np.clip(action + noise, self._minimum, self._maximum)
    return self._env.step(noisy_action)

  def reset(self):
    return self._env.reset()

  def observation_spec(self):
    return self._env.observation_spec()

  def action_spec(self):
    return self._env.action_spec()

  def __getattr__(self, name):
    return getattr(self._env, name)
