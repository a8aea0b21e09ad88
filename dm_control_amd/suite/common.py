"""Model files of the suite domains (cf. suite/common/__init__.py:22-34)."""

import os

_MODELS_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'models')

# The reference ships rendering-only includes (materials/skybox/visual); the
# in-tree models are physics-only, so there are no assets to pass along.
ASSETS = {}

# Keys of `environment_kwargs` consumed by the batched Physics rather than by
# `control.Environment`.
PHYSICS_KWARGS = ('batch_size', 'device', 'precision', 'ncon_max',
                  'build_mode')
TASK_KWARGS = ('device_init',)


def read_model(model_filename):
  """Returns the contents of a model XML file as a string."""
  with open(os.path.join(_MODELS_DIR, model_filename), 'r') as f:
    return f.read()


def split_kwargs(environment_kwargs):
  """-> (physics kwargs, task kwargs, control.Environment kwargs)."""
  kw = dict(environment_kwargs or {})
  phys = {k: kw.pop(k) for k in PHYSICS_KWARGS if k in kw}
  task = {k: kw.pop(k) for k in TASK_KWARGS if k in kw}
  return phys, task, kw
