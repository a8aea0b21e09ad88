"""Host fp64 restatement of the three tasks' reward/observation formulas.

Test infrastructure: validated against tests/golden/tasks.json (outputs of the
reference's own `get_reward` on canned readings) in test_host_logic.py, then
used by the GPU tests to check the device-fused rewards from read-back physics
quantities.  Formulas: suite/cartpole.py:204-225, suite/cheetah.py:87-93,
suite/humanoid.py:183-207 of the reference.
"""

import numpy as np

from dm_control_amd.utils import rewards


def cartpole_reward(x, cos, ctrl, angular_vel, sparse):
  cos = np.atleast_1d(cos)
  if sparse:
    return (rewards.tolerance(x, (-.25, .25)) *
            rewards.tolerance(cos, (.995, 1)).prod())
  upright = (cos + 1)/2
  centered = (1 + rewards.tolerance(x, margin=2))/2
  small_control = rewards.tolerance(np.atleast_1d(ctrl), margin=1,
                                    value_at_margin=0, sigmoid='quadratic')[0]
  small_control = (4 + small_control)/5
  small_velocity = rewards.tolerance(np.atleast_1d(angular_vel), margin=5).min()
  small_velocity = (1 + small_velocity)/2
  return upright.mean()*small_control*small_velocity*centered


def cheetah_reward(speed):
  return rewards.tolerance(speed, bounds=(10, float('inf')), margin=10,
                           value_at_margin=0, sigmoid='linear')


def humanoid_reward(head_height, torso_upright, ctrl, com_velocity, move_speed):
  standing = rewards.tolerance(head_height, bounds=(1.4, float('inf')),
                               margin=1.4/4)
  upright = rewards.tolerance(torso_upright, bounds=(0.9, float('inf')),
                              sigmoid='linear', margin=1.9, value_at_margin=0)
  stand_reward = standing*upright
  small_control = rewards.tolerance(np.asarray(ctrl), margin=1,
                                    value_at_margin=0,
                                    sigmoid='quadratic').mean()
  small_control = (4 + small_control)/5
  com_velocity = np.asarray(com_velocity)
  if move_speed == 0:
    dont_move = rewards.tolerance(com_velocity[[0, 1]], margin=2).mean()
    return small_control*stand_reward*dont_move
  speed = np.linalg.norm(com_velocity[[0, 1]])
  move = rewards.tolerance(speed, bounds=(move_speed, float('inf')),
                           margin=move_speed, value_at_margin=0,
                           sigmoid='linear')
  move = (5*move + 1)/6
  return small_control*stand_reward*move


def walker_reward(torso_height, torso_upright, horizontal_velocity, move_speed):
  """suite/walker.py:144-160."""
  standing = rewards.tolerance(torso_height, bounds=(1.2, float('inf')),
                               margin=1.2/2)
  upright = (1 + torso_upright)/2
  stand_reward = (3*standing + upright)/4
  if move_speed == 0:
    return stand_reward
  move = rewards.tolerance(horizontal_velocity,
                           bounds=(move_speed, float('inf')),
                           margin=move_speed/2, value_at_margin=0.5,
                           sigmoid='linear')
  return stand_reward*(5*move + 1)/6


def pendulum_reward(pole_vertical):
  """suite/pendulum.py:119-120."""
  return rewards.tolerance(pole_vertical, (np.cos(np.deg2rad(8)), 1))


def acrobot_reward(to_target, sparse, target_radius=0.2):
  """suite/acrobot.py:116-126."""
  return rewards.tolerance(to_target, bounds=(0, target_radius),
                           margin=0 if sparse else 1)


def hopper_reward(height, speed, ctrl, hopping):
  """suite/hopper.py:124-140."""
  standing = rewards.tolerance(height, (0.6, 2))
  if hopping:
    return standing*rewards.tolerance(speed, bounds=(2, float('inf')), margin=1,
                                      value_at_margin=0.5, sigmoid='linear')
  small_control = rewards.tolerance(np.asarray(ctrl), margin=1, value_at_margin=0,
                                    sigmoid='quadratic').mean()
  return standing*(small_control + 4)/5


def reacher_reward(finger_to_target_dist, target_size, finger_size=0.01):
  """suite/reacher.py:118-120."""
  return rewards.tolerance(finger_to_target_dist, (0, target_size + finger_size))


def point_mass_reward(mass_to_target_dist, ctrl, target_size=0.015):
  """suite/point_mass.py:122-130."""
  near = rewards.tolerance(mass_to_target_dist, bounds=(0, target_size),
                           margin=target_size)
  control_reward = rewards.tolerance(np.asarray(ctrl), margin=1, value_at_margin=0,
                                     sigmoid='quadratic').mean()
  return near*(control_reward + 4)/5
