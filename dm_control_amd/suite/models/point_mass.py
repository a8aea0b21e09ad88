"""Planar point mass parameters (reference values: dm_control/suite/point_mass.xml)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.02                      # contacts disabled
ARENA = .3
WALL = dict(height=.02, thickness=.02)
SLIDER = dict(type='slide', limited=True, range=(-.29, .29), damping=1,
              pos=(0, 0, 0))
MASS = dict(radius=.01, mass=.3, height=.01)
TARGET = dict(pos=(0, 0, .01), radius=.015)
MOTOR = dict(gear=.1, ctrlrange=(-1, 1), ctrllimited=True)
# each motor pulls a fixed tendon: (tendon, ((joint, coefficient), ...)); the
# "hard" task redraws the coefficients every episode
TENDONS = (('t1', (('root_x', 1), ('root_y', 0))),
           ('t2', (('root_x', 0), ('root_y', 1))))


def build():
  root, world, actuator, _ = m.document('planar point mass', TIMESTEP,
                                        flags=dict(contact='disable'))
  m.node(world, 'geom', name='ground', type='plane', pos=(0, 0, 0),
         size=(ARENA, ARENA, .1))
  h, t = WALL['height'], WALL['thickness']
  for name, pos, zaxis, size in (
      ('wall_x', (-ARENA, 0, h), (1, 0, 0), (t, ARENA, h)),
      ('wall_y', (0, -ARENA, h), (0, 1, 0), (ARENA, t, h)),
      ('wall_neg_x', (ARENA, 0, h), (-1, 0, 0), (t, ARENA, h)),
      ('wall_neg_y', (0, ARENA, h), (0, -1, 0), (ARENA, t, h))):
    m.node(world, 'geom', name=name, type='plane', pos=pos, zaxis=zaxis, size=size)
  body = m.node(world, 'body', name='pointmass', pos=(0, 0, MASS['height']))
  m.node(body, 'joint', name='root_x', axis=(1, 0, 0), **SLIDER)
  m.node(body, 'joint', name='root_y', axis=(0, 1, 0), **SLIDER)
  m.node(body, 'geom', name='pointmass', type='sphere', size=MASS['radius'],
         mass=MASS['mass'])
  m.node(world, 'geom', name='target', type='sphere', pos=TARGET['pos'],
         size=TARGET['radius'])
  tendon = m.node(root, 'tendon')
  for name, wraps in TENDONS:
    fixed = m.node(tendon, 'fixed', name=name)
    for joint, coef in wraps:
      m.node(fixed, 'joint', joint=joint, coef=coef)
    m.node(actuator, 'motor', name=name, tendon=name, **MOTOR)
  return m.to_string(root)
