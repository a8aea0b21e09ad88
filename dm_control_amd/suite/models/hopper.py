"""Planar hopper parameters (reference values: dm_control/suite/hopper.xml)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.005
HINGE = dict(type='hinge', axis=(0, 1, 0), limited=True, damping=.05,
             armature=.2)
ROOT = dict(limited=False, damping=0, armature=0, stiffness=0)
TOUCH_RADIUS = 0.05
# chain below the torso: (body, pos, joint, range (deg), geom fromto, radius, gear)
CHAIN = (
    ('pelvis', (0, 0, -.05), 'waist', (-30, 30), (0, 0, 0, 0, 0, -.15), 0.065, 30),
    ('thigh', (0, 0, -.2), 'hip', (-170, 10), (0, 0, 0, 0, 0, -.33), 0.04, 40),
    ('calf', (0, 0, -.33), 'knee', (5, 150), (0, 0, 0, 0, 0, -.32), 0.03, 30),
    ('foot', (0, 0, -.32), 'ankle', (-45, 45), (-.08, 0, 0, .17, 0, 0), 0.04, 10),
)
TORSO_GEOMS = (('torso', (0, 0, -.05, 0, 0, .2), 0.0653),
               ('nose', (.08, 0, .13, .15, 0, .14), 0.03))
TOUCH_SITES = (('touch_toe', (.17, 0, 0)), ('touch_heel', (-.08, 0, 0)))


def build():
  root, world, actuator, sensor = m.document('planar hopper', TIMESTEP)
  m.node(world, 'geom', name='floor', type='plane', conaffinity=1,
         pos=(48, 0, 0), size=(50, 1, .2))
  torso = m.node(world, 'body', name='torso', pos=(0, 0, 1))
  for name, kind, axis in (('rootx', 'slide', (1, 0, 0)),
                           ('rootz', 'slide', (0, 0, 1)),
                           ('rooty', 'hinge', (0, 1, 0))):
    m.node(torso, 'joint', name=name, type=kind, axis=axis, **ROOT)
  for name, fromto, radius in TORSO_GEOMS:
    m.node(torso, 'geom', name=name, type='capsule', fromto=fromto, size=radius)
  parent = torso
  for body, pos, joint, rng, fromto, radius, gear in CHAIN:
    parent = m.node(parent, 'body', name=body, pos=pos)
    m.node(parent, 'joint', name=joint, range=rng, **HINGE)
    m.node(parent, 'geom', name=body, type='capsule', fromto=fromto,
           size=radius)
    m.node(actuator, 'motor', name=joint, joint=joint, gear=gear,
           ctrllimited=True, ctrlrange=(-1, 1))
  for name, pos in TOUCH_SITES:
    m.node(parent, 'site', name=name, type='sphere', pos=pos, size=TOUCH_RADIUS)
  m.node(sensor, 'subtreelinvel', name='torso_subtreelinvel', body='torso')
  for name, _ in TOUCH_SITES:
    m.node(sensor, 'touch', name=name, site=name)
  return m.to_string(root)
