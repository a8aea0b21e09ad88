"""Affine re-scaling of the action interval (stands in for the reference's
suite/wrappers/action_scale.py: same constructor signature, same effect).

The agent acts in `[minimum, maximum]`; the wrapped env receives
`env_lo + (a - minimum) * (env_hi - env_lo) / (maximum - minimum)`.  The map is
elementwise over the last axis, so unbatched `[nu]` and batched `[B, nu]`
actions go through the same code.
"""

import numpy as np

from dm_control_amd import _dm_env as dm_env


# Message templates: the reference's tests format these module constants
# themselves (action_scale_test.py:119-160), so the names and the fields
# {name} / {bounds} / {shape} are interface; the wording is ours.
_ACTION_SPEC_MUST_BE_BOUNDED_ARRAY = (
    'action_scale needs an env whose action_spec() is one BoundedArray; this '
    'one returned {}')
_MUST_BE_FINITE = 'action_scale: `{name}` has non-finite entries: {bounds}'
_MUST_BROADCAST = ('action_scale: `{name}` = {bounds} does not fit actions of '
                   'shape {shape}')


def _checked_bound(values, label, shape):
  """`values` as an array that is finite everywhere and fits `shape`."""
  values = np.asarray(values)
  if not np.isfinite(values).all():
    raise ValueError(_MUST_BE_FINITE.format(name=label, bounds=values))
  try:
    fits = np.broadcast_shapes(values.shape, shape) == tuple(shape)
  except ValueError:
    fits = False
  if not fits:
    raise ValueError(_MUST_BROADCAST.format(name=label, bounds=values,
                                            shape=tuple(shape)))
  return values


class Wrapper(dm_env.Environment):
  """Lets an agent act in [minimum, maximum] instead of the env's own bounds."""

  def __init__(self, env, minimum, maximum):
    inner = env.action_spec()
    if not isinstance(inner, dm_env.specs.BoundedArray):
      raise ValueError(_ACTION_SPEC_MUST_BE_BOUNDED_ARRAY.format(inner))
    self._env = env
    self._lo = _checked_bound(inner.minimum, 'env.action_spec().minimum', inner.shape)
    span = _checked_bound(inner.maximum, 'env.action_spec().maximum',
                          inner.shape) - self._lo
    self._from = _checked_bound(minimum, 'minimum', inner.shape)
    upto = _checked_bound(maximum, 'maximum', inner.shape)
    self._gain = span/(upto - self._from)
    self._inner_dtype = inner.dtype
    self._spec = inner.replace(
        minimum=self._from, maximum=upto,
        dtype=np.result_type(self._from, upto, inner.dtype))

  def _to_env(self, action):
    scaled = self._lo + self._gain*(np.asarray(action) - self._from)
    return scaled.astype(self._inner_dtype, copy=False)

  def step(self, action):
    return self._env.step(self._to_env(action))

  def reset(self):
    return self._env.reset()

  def action_spec(self):
    return self._spec

  def observation_spec(self):
    return self._env.observation_spec()

  def __getattr__(self, attr):
    return getattr(self._env, attr)
