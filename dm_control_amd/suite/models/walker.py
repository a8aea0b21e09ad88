"""Planar walker parameters (reference values: dm_control/suite/walker.xml)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.0025
FRICTION = (.7, .1, .1)
JOINT = dict(damping=.1, armature=0.01, solimplimit=(0, .99, .01),
             axis=(0, -1, 0))
TORSO = dict(height=1.3, radius=0.07, half=0.3)
# per leg: (segment, body pos, joint anchor, range (deg), geom pos, geom zaxis,
#           radius, half length, motor gear)
LEG = (
    ('thigh', (0, None, -0.3), None, (-20, 100), (0, 0, -0.225), None, 0.05, 0.225, 100),
    ('leg', (0, 0, -0.7), (0, 0, 0.25), (-150, 0), None, None, 0.04, 0.25, 50),
    ('foot', (0.06, 0, -0.25), (-0.06, 0, 0), (-45, 45), None, (1, 0, 0), 0.05, 0.1, 20),
)
SIDES = (('right', -.05), ('left', .05))
JOINT_NAMES = {'thigh': 'hip', 'leg': 'knee', 'foot': 'ankle'}


def _capsule(parent, name, **kw):
  m.node(parent, 'geom', name=name, type='capsule', contype=1, conaffinity=0,
         friction=FRICTION, **kw)


def build():
  root, world, actuator, sensor = m.document('planar walker', TIMESTEP)
  # the reference's top-level geom default (friction) also covers the floor
  m.node(world, 'geom', name='floor', type='plane', conaffinity=1,
         friction=FRICTION, pos=(248, 0, 0), size=(250, .8, .2),
         zaxis=(0, 0, 1))
  torso = m.node(world, 'body', name='torso', pos=(0, 0, TORSO['height']))
  for name, kind, axis in (('rootz', 'slide', (0, 0, 1)),
                           ('rootx', 'slide', (1, 0, 0)),
                           ('rooty', 'hinge', (0, 1, 0))):
    m.node(torso, 'joint', name=name, type=kind, axis=axis,
           solimplimit=JOINT['solimplimit'])
  _capsule(torso, 'torso', size=(TORSO['radius'], TORSO['half']))
  for side, y in SIDES:
    parent = torso
    for seg, bpos, anchor, rng, gpos, zaxis, radius, half, gear in LEG:
      pos = tuple(y if v is None else v for v in bpos)
      body = m.node(parent, 'body', name='%s_%s' % (side, seg), pos=pos)
      jname = '%s_%s' % (side, JOINT_NAMES[seg])
      m.node(body, 'joint', name=jname, type='hinge', limited=True, range=rng,
             pos=anchor, **JOINT)
      _capsule(body, '%s_%s' % (side, seg), pos=gpos, zaxis=zaxis,
               size=(radius, half))
      m.node(actuator, 'motor', name=jname, joint=jname, gear=gear,
             ctrllimited=True, ctrlrange=(-1, 1))
      parent = body
  m.node(sensor, 'subtreelinvel', name='torso_subtreelinvel', body='torso')
  return m.to_string(root)
