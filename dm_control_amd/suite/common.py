"""Model files of the suite domains (cf. suite/common/__init__.py:22-34)."""

import importlib

# The reference ships rendering-only includes (materials/skybox/visual); the
# in-tree models are physics-only, so there are no assets to pass along.
ASSETS = {}

# Keys of `environment_kwargs` consumed by the batched Physics rather than by
# `control.Environment`.
PHYSICS_KWARGS = ('batch_size', 'device', 'precision', 'ncon_max',
                  'build_mode', 'group')
TASK_KWARGS = ('device_init',)


def read_model(model_filename, **kwargs):
  """Returns the MJCF of a domain as a string.

  The reference reads `<domain>.xml` from disk; here the physics-only model is
  generated from the parameter tables in `suite/models/<domain>.py`.
  """
  name = model_filename[:-4] if model_filename.endswith('.xml') else model_filename
  module = importlib.import_module('dm_control_amd.suite.models.' + name)
  return module.build(**kwargs)


def split_kwargs(environment_kwargs):
  """-> (physics kwargs, task kwargs, control.Environment kwargs)."""
  kw = dict(environment_kwargs or {})
  phys = {k: kw.pop(k) for k in PHYSICS_KWARGS if k in kw}
  task = {k: kw.pop(k) for k in TASK_KWARGS if k in kw}
  return phys, task, kw
