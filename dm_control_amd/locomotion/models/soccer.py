"""Frozen soccer scene: fixed-size pitch, ball, N position-controlled humanoids.

Reference: locomotion/soccer/__init__.py:92-152 (loader), soccer/task.py:49,
95-108 (physics timestep 0.005, `njmax`/`nconmax` per player),
soccer/soccer_ball.py:29-96 (regulation ball: radius 0.35? no -- the defaults
of `SoccerBall._build`: radius 0.35 m, mass 0.045 kg, friction (0.7, 0.075,
0.075), condim 6, priority 1, solref (0.02, damp_ratio)), soccer/pitch.py
(ground plane and four wall planes around it).

Stage 1 scope (stated): the pitch is the ground plane plus the four wall planes
at a fixed size -- `RandomizedPitch` resizes the arena and the composer
recompiles the model every episode (composer/environment.py:321-333), which a
frozen code object cannot follow; goal posts, nets and the field box (box
geoms: the box narrowphase beyond plane-box is not built) are left out.
"""

from dm_control_amd.locomotion.models import cmu_humanoid
from dm_control_amd.suite import models as m

PHYSICS_TIMESTEP = 0.005           # soccer/task.py:105
BALL = dict(radius=0.35, mass=0.045, friction=(0.7, 0.075, 0.075), damp_ratio=1.0)
# `regulation_soccer_ball()` (soccer_ball.py:27-47), what soccer.load pairs with
# humanoid walkers (soccer/__init__.py:130-136): a size-5 ball
REGULATION_BALL = dict(radius=0.117, mass=0.45, friction=(0.7, 0.05, 0.04), damp_ratio=0.4)
PITCH_SIZE = (9.0, 6.0)            # half extents of the fixed arena (m)

# Goal frame as capsules in the unit cube facing +x (what pitch.py:55-64 lists as
# `fromto` sextuples): name -> (from corner, to corner)
GOAL_FRAME = (
    ('right_post', (1, -1, -1), (1, -1, 1)), ('left_post', (1, 1, -1), (1, 1, 1)),
    ('top_post', (1, -1, 1), (1, 1, 1)), ('right_base', (1, -1, -1), (-1, -1, -1)),
    ('left_base', (1, 1, -1), (-1, 1, -1)), ('back_base', (-1, -1, -1), (-1, 1, -1)),
    ('right_support', (-1, -1, -1), (.2, -1, 1)), ('right_top_support', (.2, -1, 1), (1, -1, 1)),
    ('left_support', (-1, 1, -1), (.2, 1, 1)), ('left_top_support', (.2, 1, 1), (1, 1, 1)))


def goal_frame_capsules(pos, size, direction):
  """(name, fromto, radius) of the ten capsules of one goal (pitch.py:165-198,
  236-276): the unit-cube frame mirrored in x and y for `direction` = -1, scaled
  by the goal's half sizes and moved to `pos`; radius 7 % of the mean half size,
  the crossbar 1 % thicker, the supports 25 % thinner."""
  radius = 0.07*sum(size)/3.0
  flip = (direction, direction, 1)
  out = []
  for name, a, b in GOAL_FRAME:
    fromto = tuple(pos[k] + flip[k]*a[k]*size[k] for k in range(3)) + \
             tuple(pos[k] + flip[k]*b[k]*size[k] for k in range(3))
    r = radius*(1.01 if 'top' in name else 1.0)*(0.75 if 'support' in name else 1.0)
    out.append((name, fromto, r))
  return out


def default_positions(num_walkers):
  """Where `build` puts the walkers: two rows facing each other across x = 0."""
  return [((-1 if i % 2 == 0 else 1)*(1.5 + i//2), 0.6*(i//2), 1.05)
          for i in range(num_walkers)]


def build(num_walkers=1, with_ball=True, pitch_size=PITCH_SIZE,
          nconmax_per_player=200, njmax_per_player=200,
          walker_positions=None, disable_walker_contacts=False, ball=None,
          goal_size=None, field_box_offset=0.0):
  """MJCF string of the scene.  Walkers stand in a row facing the ball.

  walker_positions: root positions, default `default_positions(num_walkers)`.
  disable_walker_contacts: the reference's option of the same name
  (soccer/task.py:29-33, 84-85): `contype=0` on every walker geom, so walkers
  touch the pitch and the ball (whose contype matches the walkers'
  conaffinity) but neither each other nor themselves.

  ball: parameter dict (`BALL` by default, `REGULATION_BALL` for humanoid games).
  goal_size: (depth, half width, half height) of the two goals; adds their
  frames (ten capsules each, fixed to the world) where Pitch._build puts them
  (pitch.py:426-447).  None: no goal frames.

  A 2v2 scene is nq 259, nv 254, nu 224: it compiles, generates
  a kernel header (dof sets are multi-word) and steps on the CPU oracle; the
  device kernels hold one scene's dense M / Hessian per lane or per lane
  group and do not fit a scene of that size yet."""
  root = m.node(None, 'mujoco', model='soccer_%d' % num_walkers)
  m.node(root, 'option', timestep=PHYSICS_TIMESTEP)
  m.node(root, 'size', nconmax=nconmax_per_player*max(1, num_walkers),
         njmax=njmax_per_player*max(1, num_walkers))
  cmu_humanoid.add_defaults(root)
  world = m.node(root, 'worldbody')
  actuator = m.node(root, 'actuator')
  contact = m.node(root, 'contact')
  # the reference pitch's plane geoms use MuJoCo's geom defaults, not the
  # walker's: they are written out here because this document has ONE default set
  plane = dict(type='plane', condim=3, friction=(1, 0.005, 0.0001),
               solref=(0.02, 1), solimp=(0.9, 0.95, 0.001))
  m.node(world, 'geom', name='ground', size=(pitch_size[0], pitch_size[1], 0.1), **plane)
  lx, ly = pitch_size
  walls = (('wall_ny', (0, -ly, 0), (-1, 0, 0, 0, 0, 1)), ('wall_py', (0, ly, 0), (1, 0, 0, 0, 0, 1)),
           ('wall_nx', (-lx, 0, 0), (0, 1, 0, 0, 0, 1)), ('wall_px', (lx, 0, 0), (0, -1, 0, 0, 0, 1)))
  for name, pos, xyaxes in walls:
    m.node(world, 'geom', name=name, pos=pos, xyaxes=xyaxes, size=(max(lx, ly), 3.0, 0.1),
           **plane)
  if goal_size is not None:
    posts = dict(type='capsule', condim=3, friction=(1, 0.005, 0.0001),
                 solref=(0.02, 1), solimp=(0.9, 0.95, 0.001))
    for prefix, direction, x in (('home_goal', 1, -lx + goal_size[0] + field_box_offset),
                                 ('away_goal', -1, lx - goal_size[0] - field_box_offset)):
      for name, fromto, r in goal_frame_capsules((x, 0.0, goal_size[2]), goal_size, direction):
        m.node(world, 'geom', name='%s/%s' % (prefix, name), fromto=fromto, size=(r,), **posts)
  if walker_positions is None:
    walker_positions = default_positions(num_walkers)
  for i in range(num_walkers):
    cmu_humanoid.add_walker(root, world, actuator, contact, prefix='walker%d/' % i,
                            pos=tuple(walker_positions[i]),
                            contype=0 if disable_walker_contacts else None)
  # The ball comes LAST in the kinematic order: a ball-walker contact then widens
  # the envelope of the ball's six Hessian rows only, not that of the walker's 62
  # (csrc/dmc_kernels.hip, envelope Cholesky of big scenes).
  if with_ball:
    prm = BALL if ball is None else ball
    body = m.node(world, 'body', name='ball', pos=(0, 0, prm['radius'] + 0.01))
    m.node(body, 'freejoint', name='ball_free')
    m.node(body, 'geom', name='ball', type='sphere', size=(prm['radius'],), condim=6,
           priority=1, mass=prm['mass'], friction=prm['friction'],
           solref=(0.02, prm['damp_ratio']), solimp=(0.9, 0.95, 0.001))
  return m.to_string(root)
