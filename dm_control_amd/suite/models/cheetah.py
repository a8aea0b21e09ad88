"""Planar cheetah parameters (reference values: dm_control/suite/cheetah.xml)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.01
TOTAL_MASS = 14                       # compiler settotalmass
RADIUS = 0.046                        # every capsule
FRICTION = (.4, .1, .1)
JOINT = dict(armature=.1, axis=(0, 1, 0))
TORSO_HEIGHT = .7
# torso-fixed geoms: name, pos, pitch (deg), half length | fromto
TORSO_GEOMS = (
    dict(name='torso', fromto=(-.5, 0, 0, .5, 0, 0)),
    dict(name='head', pos=(.6, 0, .1), pitch=50, half=.15),
)
# body, parent, body pos, joint range (deg), stiffness, damping,
# geom pos, geom pitch (deg), geom half length, motor gear
LIMBS = (
    ('bthigh', 'torso', (-.5, 0, 0), (-30, 60), 240, 6, (.1, 0, -.13), -218, .145, 120),
    ('bshin', 'bthigh', (.16, 0, -.25), (-50, 50), 180, 4.5, (-.14, 0, -.07), -116, .15, 90),
    ('bfoot', 'bshin', (-.28, 0, -.14), (-230, 50), 120, 3, (.03, 0, -.097), -15, .094, 60),
    ('fthigh', 'torso', (.5, 0, 0), (-57, .40), 180, 4.5, (-.07, 0, -.12), 30, .133, 90),
    ('fshin', 'fthigh', (-.14, 0, -.24), (-70, 50), 120, 3, (.065, 0, -.09), -34, .106, 60),
    ('ffoot', 'fshin', (.13, 0, -.18), (-28, 28), 60, 1.5, (.045, 0, -.07), -34, .07, 30),
)
ROOT_JOINTS = (('rootx', 'slide', (1, 0, 0)), ('rootz', 'slide', (0, 0, 1)),
               ('rooty', 'hinge', (0, 1, 0)))


def _capsule(parent, name, **kw):
  m.node(parent, 'geom', name=name, type='capsule', contype=1, conaffinity=1,
         condim=3, friction=FRICTION, **kw)


def build():
  root, world, actuator, sensor = m.document('cheetah', TIMESTEP,
                                             settotalmass=TOTAL_MASS)
  m.node(world, 'geom', name='ground', type='plane', conaffinity=1,
         pos=(98, 0, 0), size=(100, .8, .5))
  bodies = {'torso': m.node(world, 'body', name='torso',
                            pos=(0, 0, TORSO_HEIGHT))}
  for name, kind, axis in ROOT_JOINTS:
    m.node(bodies['torso'], 'joint', name=name, type=kind, axis=axis)
  for g in TORSO_GEOMS:
    if 'fromto' in g:
      _capsule(bodies['torso'], g['name'], fromto=g['fromto'], size=RADIUS)
    else:
      _capsule(bodies['torso'], g['name'], pos=g['pos'],
               euler=(0, g['pitch'], 0), size=(RADIUS, g['half']))
  for (name, parent, pos, rng, stiff, damp, gpos, pitch, half, gear) in LIMBS:
    body = bodies[name] = m.node(bodies[parent], 'body', name=name, pos=pos)
    m.node(body, 'joint', name=name, type='hinge', limited=True, range=rng,
           stiffness=stiff, damping=damp, **JOINT)
    _capsule(body, name, pos=gpos, euler=(0, pitch, 0), size=(RADIUS, half))
    m.node(actuator, 'motor', name=name, joint=name, gear=gear,
           ctrllimited=True, ctrlrange=(-1, 1))
  m.node(sensor, 'subtreelinvel', name='torso_subtreelinvel', body='torso')
  return m.to_string(root)
