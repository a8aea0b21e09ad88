"""Acrobot parameters (reference values: dm_control/suite/acrobot.xml;
Coulom's variant of the two-link underactuated pendulum)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.01                      # RK4, constraints disabled
SHOULDER_HEIGHT = 2.0
LINK_LENGTH = 1.0
LINK_MASS = 1.0
RADIUS = dict(upper_arm=0.05, lower_arm=0.049)
JOINT_DAMPING = 0.05
ELBOW_GEAR = 2
TARGET = dict(pos=(0, 0, 4), radius=0.2)
TIP_RADIUS = 0.01
HUB = dict(half_width=.06, radius=0.051)    # massless decoration on the shoulder


def build():
  root, world, actuator, _ = m.document(
      'acrobot', TIMESTEP, integrator='RK4',
      flags=dict(constraint='disable', energy='enable'))
  m.node(world, 'geom', name='floor', type='plane', size=(3, 3, .2))
  m.node(world, 'site', name='target', type='sphere', pos=TARGET['pos'],
         size=TARGET['radius'])
  parent, offset = world, SHOULDER_HEIGHT
  for link, joint in (('upper_arm', 'shoulder'), ('lower_arm', 'elbow')):
    body = m.node(parent, 'body', name=link, pos=(0, 0, offset))
    m.node(body, 'joint', name=joint, type='hinge', axis=(0, 1, 0),
           damping=JOINT_DAMPING)
    if link == 'upper_arm':
      m.node(body, 'geom', name='upper_arm_decoration', type='cylinder',
             fromto=(0, -HUB['half_width'], 0, 0, HUB['half_width'], 0),
             size=HUB['radius'], mass=0)
    m.node(body, 'geom', name=link, type='capsule',
           fromto=(0, 0, 0, 0, 0, LINK_LENGTH), size=RADIUS[link],
           mass=LINK_MASS)
    parent, offset = body, LINK_LENGTH
  m.node(parent, 'site', name='tip', pos=(0, 0, LINK_LENGTH), size=TIP_RADIUS)
  m.node(actuator, 'motor', name='elbow', joint='elbow', gear=ELBOW_GEAR,
         ctrllimited=True, ctrlrange=(-1, 1))
  return m.to_string(root)
