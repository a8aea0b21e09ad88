"""Cart-pole parameters (reference values: dm_control/suite/cartpole.xml)."""

from dm_control_amd.suite import models as m

TIMESTEP = 0.01                       # RK4, contacts disabled
CART = dict(height=1.0, half_size=(0.2, 0.15, 0.1), mass=1.0)
SLIDER = dict(limit=1.8, solreflimit=(.08, 1), damping=5e-4)
POLE = dict(length=1.0, radius=0.045, mass=0.1, hinge_damping=2e-6)
MOTOR_GEAR = 10


def build(num_poles=1):
  """MJCF string of a cart with a chain of `num_poles` poles."""
  root, world, actuator, _ = m.document(
      'cart-pole', TIMESTEP, integrator='RK4',
      flags=dict(contact='disable', energy='enable'))
  cart = m.node(world, 'body', name='cart', pos=(0, 0, CART['height']))
  m.node(cart, 'joint', name='slider', type='slide', axis=(1, 0, 0),
         limited=True, range=(-SLIDER['limit'], SLIDER['limit']),
         solreflimit=SLIDER['solreflimit'], damping=SLIDER['damping'])
  m.node(cart, 'geom', name='cart', type='box', size=CART['half_size'],
         mass=CART['mass'])
  parent = cart
  for i in range(1, num_poles + 1):
    pole = m.node(parent, 'body', name='pole_%d' % i,
                  pos=(0, 0, POLE['length']) if i > 1 else None)
    m.node(pole, 'joint', name='hinge_%d' % i, type='hinge', axis=(0, 1, 0),
           damping=POLE['hinge_damping'])
    m.node(pole, 'geom', name='pole_%d' % i, type='capsule',
           fromto=(0, 0, 0, 0, 0, POLE['length']), size=POLE['radius'],
           mass=POLE['mass'])
    parent = pole
  m.node(actuator, 'motor', name='slide', joint='slider', gear=MOTOR_GEAR,
         ctrllimited=True, ctrlrange=(-1, 1))
  return m.to_string(root)
