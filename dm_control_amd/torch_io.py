"""Zero-copy torch views of a batch's device fields.

PyTorch is plumbing here (device memory, streams): a field of a
`wrapper.HipBatch` is exposed as a `torch.Tensor` that aliases the HBM buffer
the step kernel writes (via `__cuda_array_interface__`), and the batch can be
told to launch on torch's current stream so that policy kernels and physics
steps are ordered without host synchronisation.  The views stay valid for the
lifetime of the batch handle (cf. the borrowed numpy views of the reference,
wrapper/util.py:171-221); their contents change with every step.
"""

import numpy as np

from dm_control_amd import wrapper


class _DeviceArray:
  """Minimal `__cuda_array_interface__` carrier for a raw device pointer."""

  def __init__(self, ptr, shape, dtype, owner):
    self._owner = owner   # keeps the batch (and its allocation) alive
    self.__cuda_array_interface__ = {
        'shape': tuple(int(s) for s in shape),
        'typestr': np.dtype(dtype).str,
        'data': (int(ptr), False),
        'version': 2,
        'strides': None,
    }


def field_tensor(batch, field):
  """torch tensor aliasing `field` of `batch` in its native layout."""
  import torch
  shape = batch._shape(field)      # pylint: disable=protected-access
  dtype = batch._dtype(field)      # pylint: disable=protected-access
  ptr = batch.device_ptr(field)
  if not ptr:
    raise wrapper.Error('field %d has no device buffer' % field)
  arr = _DeviceArray(ptr, shape, dtype, batch)
  dev = torch.device('cuda', batch.model.device_id)
  t = torch.as_tensor(arr, device=dev)
  t._dmc_owner = arr               # pylint: disable=protected-access
  return t


def use_current_stream(batch):
  """Launch the batch's kernels on torch's current stream of its device."""
  import torch
  with torch.cuda.device(batch.model.device_id):
    batch.set_stream(torch.cuda.current_stream().cuda_stream)
