"""`dm_env.specs` stand-in: `Array` and `BoundedArray`."""

import numpy as np


class Array:
  """Describes a numpy array or scalar shape and dtype."""

  def __init__(self, shape, dtype, name=None):
    self._shape = tuple(int(d) for d in shape)
    self._dtype = np.dtype(dtype)
    self._name = name

  shape = property(lambda self: self._shape)
  dtype = property(lambda self: self._dtype)
  name = property(lambda self: self._name)

  def __repr__(self):
    return 'Array(shape={}, dtype={}, name={})'.format(
        self.shape, repr(self.dtype), repr(self.name))

  def __eq__(self, other):
    return (isinstance(other, Array) and self.shape == other.shape
            and self.dtype == other.dtype and self.name == other.name)

  def validate(self, value):
    value = np.asarray(value)
    if value.shape != self.shape:
      raise ValueError('Expected shape %r but found %r'
                       % (self.shape, value.shape))
    if value.dtype != self.dtype:
      raise ValueError('Expected dtype %s but found %s'
                       % (self.dtype, value.dtype))
    return value

  def generate_value(self):
    return np.zeros(shape=self.shape, dtype=self.dtype)

  def replace(self, **kwargs):
    args = dict(shape=self.shape, dtype=self.dtype, name=self.name)
    args.update(kwargs)
    return type(self)(**args)


class BoundedArray(Array):
  """An `Array` with elementwise minimum and maximum."""

  def __init__(self, shape, dtype, minimum, maximum, name=None):
    super().__init__(shape, dtype, name)
    try:
      np.broadcast_to(minimum, shape=shape)
      np.broadcast_to(maximum, shape=shape)
    except ValueError as e:
      raise ValueError('minimum/maximum not compatible with shape: %s' % e)
    if np.any(np.asarray(minimum) > np.asarray(maximum)):
      raise ValueError('All values in `minimum` must be <= `maximum`.')
    self._minimum = np.array(minimum, dtype=self.dtype)
    self._minimum.setflags(write=False)
    self._maximum = np.array(maximum, dtype=self.dtype)
    self._maximum.setflags(write=False)

  minimum = property(lambda self: self._minimum)
  maximum = property(lambda self: self._maximum)

  def __repr__(self):
    return ('BoundedArray(shape={}, dtype={}, name={}, minimum={}, maximum={})'
            .format(self.shape, repr(self.dtype), repr(self.name),
                    self._minimum, self._maximum))

  def __eq__(self, other):
    return (isinstance(other, BoundedArray) and super().__eq__(other)
            and (self.minimum == other.minimum).all()
            and (self.maximum == other.maximum).all())

  def validate(self, value):
    value = super().validate(value)
    if (value < self.minimum).any() or (value > self.maximum).any():
      raise ValueError('Values were not all within bounds %s <= value <= %s'
                       % (self.minimum, self.maximum))
    return value

  def replace(self, **kwargs):
    args = dict(shape=self.shape, dtype=self.dtype, name=self.name,
                minimum=self.minimum, maximum=self.maximum)
    args.update(kwargs)
    return type(self)(**args)
