"""Host fp64 soft-indicator rewards.

Same contract as /root/reference/dm_control/utils/rewards.py:93-135 (`tolerance`)
and :25-90 (the eight sigmoids).  The device fuses the three sigmoids the suite
tasks use (csrc/dmc_kernels.hip `tolerance`); this host version serves task
code that runs on numpy state and is what tests/golden/rewards.json pins.
"""

import numpy as np

_DEFAULT_VALUE_AT_MARGIN = 0.1
_ZERO_OK = ('cosine', 'linear', 'quadratic')


def _sigmoids(x, value_at_1, sigmoid):
  lo_ok = 0 <= value_at_1 < 1 if sigmoid in _ZERO_OK else 0 < value_at_1 < 1
  if not lo_ok:
    if sigmoid in _ZERO_OK:
      raise ValueError('`value_at_1` must be nonnegative and smaller than 1, '
                       'got {}.'.format(value_at_1))
    raise ValueError('`value_at_1` must be strictly between 0 and 1, '
                     'got {}.'.format(value_at_1))
  x = np.asarray(x, dtype=np.float64)
  if sigmoid == 'gaussian':
    return np.exp(-0.5*(x*np.sqrt(-2*np.log(value_at_1)))**2)
  if sigmoid == 'hyperbolic':
    return 1/np.cosh(x*np.arccosh(1/value_at_1))
  if sigmoid == 'long_tail':
    return 1/((x*np.sqrt(1/value_at_1 - 1))**2 + 1)
  if sigmoid == 'reciprocal':
    return 1/(abs(x)*(1/value_at_1 - 1) + 1)
  if sigmoid == 'cosine':
    sx = x*(np.arccos(2*value_at_1 - 1)/np.pi)
    with np.errstate(invalid='ignore'):
      c = np.cos(np.pi*sx)
    return np.where(abs(sx) < 1, (1 + c)/2, 0.0)
  if sigmoid == 'linear':
    sx = x*(1 - value_at_1)
    return np.where(abs(sx) < 1, 1 - sx, 0.0)
  if sigmoid == 'quadratic':
    sx = x*np.sqrt(1 - value_at_1)
    return np.where(abs(sx) < 1, 1 - sx**2, 0.0)
  if sigmoid == 'tanh_squared':
    return 1 - np.tanh(x*np.arctanh(np.sqrt(1 - value_at_1)))**2
  raise ValueError('Unknown sigmoid type {!r}.'.format(sigmoid))


def tolerance(x, bounds=(0.0, 0.0), margin=0.0, sigmoid='gaussian',
              value_at_margin=_DEFAULT_VALUE_AT_MARGIN):
  """1 inside `bounds`, decaying sigmoidally with distance/margin outside."""
  lower, upper = bounds
  if lower > upper:
    raise ValueError('Lower bound must be <= upper bound.')
  if margin < 0:
    raise ValueError('`margin` must be non-negative.')
  in_bounds = np.logical_and(lower <= x, x <= upper)
  if margin == 0:
    value = np.where(in_bounds, 1.0, 0.0)
  else:
    d = np.where(x < lower, lower - x, x - upper)/margin
    value = np.where(in_bounds, 1.0, _sigmoids(d, value_at_margin, sigmoid))
  return float(value) if np.isscalar(x) else value
