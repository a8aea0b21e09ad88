"""Pendulum parameters (reference values: dm_control/suite/pendulum.xml)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.02
PIVOT_HEIGHT = .6
HINGE_DAMPING = 0.1
ROD = dict(length=0.5, radius=0.02)        # massless
BOB = dict(radius=0.05, mass=1.0)


def build():
  root, world, actuator, _ = m.document(
      'pendulum', TIMESTEP, flags=dict(contact='disable', energy='enable'))
  pole = m.node(world, 'body', name='pole', pos=(0, 0, PIVOT_HEIGHT))
  m.node(pole, 'joint', name='hinge', type='hinge', axis=(0, 1, 0),
         damping=HINGE_DAMPING)
  m.node(pole, 'geom', name='pole', type='capsule',
         fromto=(0, 0, 0, 0, 0, ROD['length']), size=ROD['radius'], mass=0)
  m.node(pole, 'geom', name='mass', type='sphere', pos=(0, 0, ROD['length']),
         size=BOB['radius'], mass=BOB['mass'])
  m.node(actuator, 'motor', name='torque', joint='hinge', gear=1,
         ctrlrange=(-1, 1), ctrllimited=True)
  return m.to_string(root)
