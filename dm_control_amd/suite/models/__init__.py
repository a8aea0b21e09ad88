"""Physics-only models of the suite domains, as parameter tables.

Each module holds the physical parameters of one Control Suite domain (masses
via density/size, joint ranges, stiffness, gains, ...; values are those of the
reference's `dm_control/suite/<domain>.xml`) in flat Python tables and a
`build()` function that emits an MJCF string for `mjcf.compiler`.  Rendering
assets, lights, cameras, sites and sensors that no task on the step path reads
are not represented.  tests/test_compiler.py checks, where the reference tree
is present, that every model compiles to the same arrays as the reference file.
"""

import xml.etree.ElementTree as ET


def fmt(value):
  """Numbers / tuples -> MJCF attribute text."""
  if isinstance(value, str):
    return value
  if isinstance(value, bool):
    return 'true' if value else 'false'
  if isinstance(value, (tuple, list)):
    return ' '.join(fmt(v) for v in value)
  return repr(float(value)) if isinstance(value, float) else str(value)


def node(parent, tag, **attrs):
  """Adds <tag .../> under `parent`; `None` attributes are skipped;
  a trailing underscore lets Python keywords through (class_ -> class)."""
  clean = {k.rstrip('_'): fmt(v) for k, v in attrs.items() if v is not None}
  if parent is None:
    return ET.Element(tag, clean)
  return ET.SubElement(parent, tag, clean)


def document(model_name, timestep, integrator=None, flags=None,
             settotalmass=None):
  """-> (root, worldbody, actuator, sensor) of an empty MJCF document."""
  root = node(None, 'mujoco', model=model_name)
  if settotalmass is not None:
    node(root, 'compiler', settotalmass=settotalmass)
  option = node(root, 'option', timestep=timestep, integrator=integrator)
  if flags:
    node(option, 'flag', **flags)
  world = node(root, 'worldbody')
  return root, world, node(root, 'actuator'), node(root, 'sensor')


def to_string(root):
  return ET.tostring(root, encoding='unicode')
