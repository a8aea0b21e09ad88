"""Compiled-model container (the counterpart of MuJoCo's `mjModel`).

The reference reaches its compiled model through `wrapper.MjModel`
(/root/reference/dm_control/mujoco/wrapper/core.py:444-627), whose attributes are
numpy views named after the `mjModel` C fields.  This build keeps the same field
names (`nq`, `body_parentid`, `jnt_range`, ...) so that task code written against
`physics.model.<field>` reads identically, but the arrays are plain owned numpy
arrays produced by `dm_control_amd.mjcf.compiler` (there is no libmujoco here).
"""

import hashlib

import numpy as np

# Enum values follow MuJoCo's public headers (mjmodel.h) so that integer codes
# stored in compiled models mean the same thing as in the reference.
JNT_FREE, JNT_BALL, JNT_SLIDE, JNT_HINGE = 0, 1, 2, 3
GEOM_PLANE, GEOM_HFIELD, GEOM_SPHERE, GEOM_CAPSULE = 0, 1, 2, 3
GEOM_ELLIPSOID, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 4, 5, 6, 7
INT_EULER, INT_RK4 = 0, 1
CONE_PYRAMIDAL, CONE_ELLIPTIC = 0, 1
SOLVER_PGS, SOLVER_CG, SOLVER_NEWTON = 0, 1, 2
GAIN_FIXED = 0
BIAS_NONE, BIAS_AFFINE = 0, 1
TRN_JOINT = 0
TRN_TENDON = 3          # mjTRN_TENDON

# mjtDisableBit
DSBL_CONSTRAINT = 1 << 0
DSBL_EQUALITY = 1 << 1
DSBL_FRICTIONLOSS = 1 << 2
DSBL_LIMIT = 1 << 3
DSBL_CONTACT = 1 << 4
DSBL_PASSIVE = 1 << 5
DSBL_GRAVITY = 1 << 6
DSBL_CLAMPCTRL = 1 << 7
DSBL_WARMSTART = 1 << 8
DSBL_FILTERPARENT = 1 << 9
DSBL_ACTUATION = 1 << 10
DSBL_REFSAFE = 1 << 11
DISABLE_NAMES = {
    'constraint': DSBL_CONSTRAINT, 'equality': DSBL_EQUALITY,
    'frictionloss': DSBL_FRICTIONLOSS, 'limit': DSBL_LIMIT,
    'contact': DSBL_CONTACT, 'passive': DSBL_PASSIVE, 'gravity': DSBL_GRAVITY,
    'clampctrl': DSBL_CLAMPCTRL, 'warmstart': DSBL_WARMSTART,
    'filterparent': DSBL_FILTERPARENT, 'actuation': DSBL_ACTUATION,
    'refsafe': DSBL_REFSAFE,
}
# mjtEnableBit (only `energy` appears in the target models; it is accepted and
# has no effect on the dynamics).
ENBL_OVERRIDE, ENBL_ENERGY = 1 << 0, 1 << 1
ENABLE_NAMES = {'override': ENBL_OVERRIDE, 'energy': ENBL_ENERGY,
                'fwdinv': 1 << 2, 'sensornoise': 1 << 3}

# mjtSensor codes for the sensor kinds that occur in the suite models, and the
# number of scalars each one contributes to `sensordata`.
SENSOR_TYPES = {
    'touch': (0, 1), 'accelerometer': (1, 3), 'velocimeter': (2, 3),
    'gyro': (3, 3), 'force': (4, 3), 'torque': (5, 3), 'magnetometer': (6, 3),
    'rangefinder': (7, 1), 'jointpos': (8, 1), 'jointvel': (9, 1),
    'tendonpos': (10, 1), 'tendonvel': (11, 1), 'actuatorpos': (12, 1),
    'actuatorvel': (13, 1), 'actuatorfrc': (14, 1), 'ballquat': (15, 4),
    'ballangvel': (16, 3), 'framepos': (25, 3), 'framequat': (26, 4),
    'framexaxis': (27, 3), 'frameyaxis': (28, 3), 'framezaxis': (29, 3),
    'framelinvel': (30, 3), 'frameangvel': (31, 3), 'framelinacc': (32, 3),
    'frameangacc': (33, 3), 'subtreecom': (34, 3), 'subtreelinvel': (35, 3),
    'subtreeangmom': (36, 3),
}
SENS_SUBTREELINVEL = 35
SENS_JOINTPOS, SENS_JOINTVEL = 8, 9
SENS_SUBTREECOM = 34

MJ_MINVAL = 1e-15
MJ_MAXVAL = 1e10
MJ_MINIMP = 1e-4
MJ_MAXIMP = 0.9999

WARNING_NAMES = ('mjWARN_INERTIA', 'mjWARN_CONTACTFULL', 'mjWARN_CNSTRFULL',
                 'mjWARN_VGEOMFULL', 'mjWARN_BADQPOS', 'mjWARN_BADQVEL',
                 'mjWARN_BADQACC', 'mjWARN_BADCTRL')

# (name, kind) for every field handed to native code.  kind: 'i' scalar int,
# 'd' scalar double, 'I' int array, 'D' double array.  The same table drives the
# oracle's ctypes loader and the device-header generator, so the two consumers
# cannot disagree about what a compiled model contains.
FIELDS = (
    ('nq', 'i'), ('nv', 'i'), ('nu', 'i'), ('nbody', 'i'), ('njnt', 'i'),
    ('ngeom', 'i'), ('nsensor', 'i'), ('nsensordata', 'i'), ('nexclude', 'i'),
    ('ntendon', 'i'), ('nwrap', 'i'),
    ('integrator', 'i'), ('cone', 'i'), ('solver', 'i'), ('iterations', 'i'),
    ('disableflags', 'i'), ('enableflags', 'i'),
    ('timestep', 'd'), ('tolerance', 'd'), ('impratio', 'd'),
    ('meaninertia', 'd'),
    ('gravity', 'D'),
    ('qpos0', 'D'), ('qpos_spring', 'D'),
    ('body_parentid', 'I'), ('body_rootid', 'I'), ('body_weldid', 'I'),
    ('body_jntnum', 'I'), ('body_jntadr', 'I'), ('body_dofnum', 'I'),
    ('body_dofadr', 'I'), ('body_geomnum', 'I'), ('body_geomadr', 'I'),
    ('body_pos', 'D'), ('body_quat', 'D'), ('body_ipos', 'D'),
    ('body_iquat', 'D'), ('body_mass', 'D'), ('body_subtreemass', 'D'),
    ('body_inertia', 'D'), ('body_invweight0', 'D'),
    ('jnt_type', 'I'), ('jnt_qposadr', 'I'), ('jnt_dofadr', 'I'),
    ('jnt_bodyid', 'I'), ('jnt_limited', 'I'), ('jnt_pos', 'D'),
    ('jnt_axis', 'D'), ('jnt_stiffness', 'D'), ('jnt_range', 'D'),
    ('jnt_margin', 'D'), ('jnt_solref', 'D'), ('jnt_solimp', 'D'),
    ('dof_bodyid', 'I'), ('dof_jntid', 'I'), ('dof_parentid', 'I'),
    ('dof_armature', 'D'), ('dof_damping', 'D'), ('dof_invweight0', 'D'),
    ('geom_type', 'I'), ('geom_contype', 'I'), ('geom_conaffinity', 'I'),
    ('geom_condim', 'I'), ('geom_bodyid', 'I'), ('geom_priority', 'I'),
    ('geom_size', 'D'), ('geom_pos', 'D'), ('geom_quat', 'D'),
    ('geom_friction', 'D'), ('geom_solmix', 'D'), ('geom_solref', 'D'),
    ('geom_solimp', 'D'), ('geom_margin', 'D'), ('geom_gap', 'D'),
    ('geom_rbound', 'D'),
    ('actuator_trntype', 'I'), ('actuator_trnid', 'I'),
    ('actuator_ctrllimited', 'I'), ('actuator_forcelimited', 'I'),
    ('actuator_gaintype', 'I'), ('actuator_biastype', 'I'),
    ('actuator_gear', 'D'), ('actuator_ctrlrange', 'D'),
    ('actuator_forcerange', 'D'), ('actuator_gainprm', 'D'),
    ('actuator_biasprm', 'D'),
    ('tendon_adr', 'I'), ('tendon_num', 'I'), ('wrap_objid', 'I'),
    ('wrap_prm', 'D'),
    ('sensor_type', 'I'), ('sensor_objid', 'I'), ('sensor_adr', 'I'),
    ('sensor_dim', 'I'),
    ('exclude_signature', 'I'),
)


class Opt:
  """`model.opt` namespace (timestep, gravity, integrator, ...)."""


_OPT_FIELDS = ('timestep', 'gravity', 'integrator', 'cone', 'solver',
               'iterations', 'tolerance', 'impratio', 'disableflags',
               'enableflags')


class Model:
  """Plain-numpy compiled model with `mjModel` field names.

  The `mjOption` fields live in `model.opt`; reading or assigning them at the
  top level (`model.timestep`) goes to the same object, so the native-field
  table (oracle) and the code generator (device) can never see two values.
  """

  def __init__(self):
    object.__setattr__(self, 'opt', Opt())
    self.names = {}  # objtype -> list of names (index = id)

  def __getattr__(self, name):   # only reached when normal lookup fails
    if name in _OPT_FIELDS and 'opt' in self.__dict__:
      return getattr(self.__dict__['opt'], name)
    raise AttributeError(name)

  def __setattr__(self, name, value):
    if name in _OPT_FIELDS:
      setattr(self.opt, name, value)
    else:
      object.__setattr__(self, name, value)

  # -- name lookups (wrapper/core.py:532-574) --------------------------------
  def name2id(self, name, object_type):
    try:
      return self.names[object_type].index(name)
    except (KeyError, ValueError):
      raise ValueError('Object of type {!r} with name {!r} does not exist.'
                       .format(object_type, name))

  def id2name(self, object_id, object_type):
    names = self.names.get(object_type, [])
    if not 0 <= object_id < len(names):
      raise ValueError('Object of type {!r} with ID {} does not exist.'
                       .format(object_type, object_id))
    return names[object_id] or ''

  def field(self, name):
    if name in _OPT_FIELDS:
      return getattr(self.opt, name)
    return getattr(self, name)

  def content_hash(self):
    """Stable digest of everything native code consumes."""
    h = hashlib.sha1()
    for name, kind in FIELDS:
      v = self.field(name)
      if kind in 'id':
        h.update(('%s=%r;' % (name, v)).encode())
      else:
        a = np.ascontiguousarray(
            v, dtype=np.int32 if kind == 'I' else np.float64)
        h.update(name.encode())
        h.update(a.tobytes())
    return h.hexdigest()[:16]
