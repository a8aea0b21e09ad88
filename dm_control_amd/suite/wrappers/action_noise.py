"""Adds Gaussian noise to actions (cf. suite/wrappers/action_noise.py)."""

import numpy as np

from dm_control_amd import _dm_env as dm_env

class Wrapper(dm_env.Environment):
  """Noise std = `scale` x (max - min) per action dimension, then clipped.

  The noise comes from the task's RandomState like in the reference; a batched
  action `[B, nu]` gets independent noise per instance (one draw of that shape).
  """

  def __init__(self, env, scale=0.01):
    spec = env.action_spec()
    lo = np.asarray(spec.minimum, dtype=np.float64)
    hi = np.asarray(spec.maximum, dtype=np.float64)
    if not np.isfinite(np.concatenate([lo.ravel(), hi.ravel()])).all():
      raise ValueError('action noise is scaled by the action range, which must '
                       'be finite; got minimum={}, maximum={}'.format(lo, hi))
    self._env = env
    self._minimum, self._maximum = lo, hi
    self._noise_std = float(scale)*(hi - lo)

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
