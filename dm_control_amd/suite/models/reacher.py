"""Two-link planar reacher parameters (reference values:
dm_control/suite/reacher.xml)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.02                      # contacts disabled
HINGE = dict(type='hinge', axis=(0, 0, 1), damping=0.01)
MOTOR = dict(gear=.05, ctrlrange=(-1, 1), ctrllimited=True)
ARENA = .3                           # half width of the walled square
WALL = dict(height=.02, thickness=.02)
LINK_RADIUS = .01
# (body, offset from parent, link length, joint, range in degrees or None)
LINKS = (('arm', (0, 0, .01), 0.12, 'shoulder', None),
         ('hand', (.12, 0, 0), 0.1, 'wrist', (-160, 160)))
FINGER = dict(offset=(.12, 0, 0), radius=.01)
TARGET = dict(pos=(0, 0, .01), radius=.05)   # x, y and radius are set per episode


def build():
  root, world, actuator, _ = m.document('two-link planar reacher', TIMESTEP,
                                        flags=dict(contact='disable'))
  m.node(world, 'geom', name='ground', type='plane', pos=(0, 0, 0),
         size=(ARENA, ARENA, 10))
  h, t = WALL['height'], WALL['thickness']
  for name, pos, zaxis, size in (
      ('wall_x', (-ARENA, 0, h), (1, 0, 0), (t, ARENA, h)),
      ('wall_y', (0, -ARENA, h), (0, 1, 0), (ARENA, t, h)),
      ('wall_neg_x', (ARENA, 0, h), (-1, 0, 0), (t, ARENA, h)),
      ('wall_neg_y', (0, ARENA, h), (0, -1, 0), (ARENA, t, h))):
    m.node(world, 'geom', name=name, type='plane', pos=pos, zaxis=zaxis, size=size)
  m.node(world, 'geom', name='root', type='cylinder',
         fromto=(0, 0, 0, 0, 0, 0.02), size=.011)
  parent = world
  for body, offset, length, joint, rng in LINKS:
    parent = m.node(parent, 'body', name=body, pos=offset)
    m.node(parent, 'geom', name=body, type='capsule',
           fromto=(0, 0, 0, length, 0, 0), size=LINK_RADIUS)
    m.node(parent, 'joint', name=joint, limited=rng is not None or None,
           range=rng, **HINGE)
    m.node(actuator, 'motor', name=joint, joint=joint, **MOTOR)
  finger = m.node(parent, 'body', name='finger', pos=FINGER['offset'])
  m.node(finger, 'geom', name='finger', type='sphere', size=FINGER['radius'])
  m.node(world, 'geom', name='target', type='sphere', pos=TARGET['pos'],
         size=TARGET['radius'])
  return m.to_string(root)
