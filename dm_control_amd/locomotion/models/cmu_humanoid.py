"""Position-controlled CMU humanoid as an MJCF fragment.

Values: `cmu_humanoid_table.py` (the reference's humanoid_CMU_V2019.xml and the
gains of walkers/cmu_humanoid.py:53-110).  What the reference does in
`CMUHumanoidPositionControlled._build` (walkers/cmu_humanoid.py:360-398) and
`scaled_actuators.add_position_actuator` (walkers/scaled_actuators.py) is
restated here:

  * every joint gets a `general` actuator with affine bias: for the control
    range (-1, 1) mapped onto the joint range (lo, hi), slope = (hi - lo)/2,
    gainprm = kp*slope, biasprm = (kp*(lo + slope), -kp, 0), `forcelimited`
    with the table's force range;
  * the walker hangs on a free joint (walkers/base.py:71-72);
  * `<exclude>` pairs as in the XML.

Deviation, stated: the two ELLIPSOID hand geoms (humanoid_CMU_V2019.xml:147,187;
ellipsoid collisions need MuJoCo's general convex narrowphase, which is not
built) are replaced by spheres of equal volume, r = (a b c)^(1/3), i.e. equal
mass at the default density.  Sites, cameras, lights, sensors and the skin are
not represented.
"""

from dm_control_amd.locomotion.models import cmu_humanoid_table as T
from dm_control_amd.suite import models as m


def add_defaults(root):
  """<default> block of the reference XML (joint/geom defaults, stiffness classes)."""
  d = m.node(root, 'default')
  m.node(d, 'joint', limited=True, solimplimit=T.JOINT_DEFAULT['solimplimit'],
         stiffness=T.JOINT_DEFAULT['stiffness'], armature=T.JOINT_DEFAULT['armature'],
         damping=T.JOINT_DEFAULT['damping'])
  m.node(d, 'geom', condim=int(T.GEOM_DEFAULT['condim']), friction=T.GEOM_DEFAULT['friction'],
         solref=T.GEOM_DEFAULT['solref'], solimp=T.GEOM_DEFAULT['solimp'])
  m.node(d, 'general', ctrllimited=True, ctrlrange=(-1, 1), forcelimited=True)
  for name, attrs in T.JOINT_CLASSES.items():
    c = m.node(d, 'default', class_=name)
    m.node(c, 'joint', **attrs)


def add_walker(root, world, actuator, contact, prefix='', pos=(0, 0, 1.0),
               quat=(0.7071067811865476, 0.7071067811865476, 0, 0), contype=None):
  """Adds one walker under `world`; names get `prefix`.  `quat`: the CMU model's
  up axis is +y, the default orientation stands it up along +z."""
  nodes = {}
  frame = m.node(world, 'body', name=prefix + 'frame', pos=pos, quat=quat)
  m.node(frame, 'freejoint', name=prefix + 'root_free')
  nodes[None] = frame
  for name, parent, bpos, bquat in T.BODIES:
    nodes[name] = m.node(nodes[parent], 'body', name=prefix + name, pos=bpos, quat=bquat)
  ranges = {}
  for name, body, axis, rng, klass in T.JOINTS:
    m.node(nodes[body], 'joint', name=prefix + name, type='hinge', axis=axis,
           range=rng, class_=klass)
    ranges[name] = rng
  for name, body, kind, size, gpos, gquat in T.GEOMS:
    if kind == 'ellipsoid':     # equal-volume sphere, see the module docstring
      kind, size = 'sphere', ((size[0]*size[1]*size[2])**(1.0/3.0),)
    extra = {} if contype is None else {'contype': contype}
    m.node(nodes[body], 'geom', name=prefix + name, type=kind,
           size=tuple(s for s in size if s != 0) or size[:1], pos=gpos, quat=gquat,
           **extra)
  for name, forcerange, kp in T.POSITION_ACTUATORS:
    lo, hi = ranges[name]
    slope = (hi - lo)/2.0
    m.node(actuator, 'general', name=prefix + name, joint=prefix + name,
           biastype='affine', gainprm=(kp*slope,),
           biasprm=(kp*(lo + slope), -kp, 0.0), forcerange=forcerange)
  for b1, b2 in T.EXCLUDES:
    m.node(contact, 'exclude', body1=prefix + b1, body2=prefix + b2)
  return frame
