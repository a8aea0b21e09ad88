"""ctypes front-end of the fp64 CPU oracle (oracle/mjstep.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of bench.py.  Nothing under dm_control_amd/ imports it.

The class layout mirrors how the reference reaches libmujoco
(/root/reference/dm_control/mujoco/wrapper/core.py: `MjModel` :444-627,
`MjData` :630-776; attribute access returns numpy *views* on native memory,
wrapper/util.py:171-221).
"""

import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, 'libmjoracle.so')
_lib = None


def build(force=False, cflags=None, out=None):
  """Compiles mjstep.c with gcc (recipe = oracle/Makefile)."""
  out = out or _LIB_PATH
  src = os.path.join(_DIR, 'mjstep.c')
  if (not force and os.path.exists(out)
      and os.path.getmtime(out) >= os.path.getmtime(src)):
    return out
  flags = cflags or ['-O2', '-fopenmp', '-fPIC', '-std=c99',
                     '-ffp-contract=off']
  subprocess.check_call(['gcc'] + flags + ['-shared', '-o', out, src, '-lm'])
  return out


def load(path=None):
  global _lib
  if _lib is not None and path is None:
    return _lib
  path = path or _LIB_PATH
  if not os.path.exists(path):
    build()
  lib = ctypes.CDLL(path)
  vp, cp, ci, cd = (ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int,
                    ctypes.c_double)
  pi, pd = ctypes.POINTER(ci), ctypes.POINTER(cd)
  sigs = {
      'mjo_model_new': (vp, []), 'mjo_model_free': (None, [vp]),
      'mjo_model_set_int': (ci, [vp, cp, ci]),
      'mjo_model_get_int': (ci, [vp, cp]),
      'mjo_model_set_double': (ci, [vp, cp, cd]),
      'mjo_model_set_iarr': (ci, [vp, cp, pi, ci]),
      'mjo_model_set_darr': (ci, [vp, cp, pd, ci]),
      'mjo_model_darr': (pd, [vp, cp]),
      'mjo_data_new': (vp, [vp]), 'mjo_data_free': (None, [vp]),
      'mjo_data_ptr': (pd, [vp, cp, pi]),
      'mjo_data_warnings': (pi, [vp]),
      'mjo_data_ncon': (ci, [vp]), 'mjo_data_nefc': (ci, [vp]),
      'mjo_data_solver_iter': (ci, [vp]),
      'mjo_data_set_time': (None, [vp, cd]), 'mjo_data_time': (cd, [vp]),
      'mjo_copy_data': (None, [vp, vp]),
      'mjo_reset_data': (None, [vp, vp]),
      'mjo_clear_warnings': (None, [vp]),
      'mjo_forward': (None, [vp, vp]), 'mjo_step': (None, [vp, vp]),
      'mjo_step1': (None, [vp, vp]), 'mjo_step2': (None, [vp, vp]),
      'mjo_physics_step': (None, [vp, vp]),
      'mjo_contact_force': (None, [vp, vp, ci, pd]),
      'mjo_contact_get': (None, [vp, ci, pd, pd, pd, pi]),
      'mjo_batch_step': (ci, [vp, ctypes.POINTER(vp), ci, pd, ci, ci]),
      'mjo_point_velocity': (None, [vp, vp, ci, pd, pd, pd]),
  }
  for name, (res, args) in sigs.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
  if path == _LIB_PATH:
    _lib = lib
  return lib


class OracleModel:
  """Native copy of a compiled `dm_control_amd.mjcf.model.Model`."""

  def __init__(self, model, lib=None):
    from dm_control_amd.mjcf import model as mdl
    self.lib = lib or load()
    self.model = model
    self.ptr = self.lib.mjo_model_new()
    for name, kind in mdl.FIELDS:
      v = model.field(name)
      key = name.encode()
      if kind == 'i':
        rc = self.lib.mjo_model_set_int(self.ptr, key, int(v))
      elif kind == 'd':
        rc = self.lib.mjo_model_set_double(self.ptr, key, float(v))
      elif kind == 'I':
        a = np.ascontiguousarray(v, dtype=np.int32).ravel()
        rc = self.lib.mjo_model_set_iarr(
            self.ptr, key, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
            a.size)
      else:
        a = np.ascontiguousarray(v, dtype=np.float64).ravel()
        rc = self.lib.mjo_model_set_darr(
            self.ptr, key, a.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
            a.size)
      if rc != 0:
        raise RuntimeError('oracle rejected model field %r' % name)
    # capacities: the C defaults (128 contacts, 600 rows) suit one walker; scenes
    # with several get room in proportion (the reference: 200 per player, task.py:105-108)
    if model.nv > 64:
      self.set_int('nconmax', 128 + model.nv)
      self.set_int('nefcmax', 600 + 4*model.nv)

  def set_int(self, name, value):
    if self.lib.mjo_model_set_int(self.ptr, name.encode(), int(value)) != 0:
      raise KeyError(name)

  def get_int(self, name):
    return self.lib.mjo_model_get_int(self.ptr, name.encode())

  def __del__(self):
    try:
      self.lib.mjo_model_free(self.ptr)
    except Exception:  # pylint: disable=broad-except
      pass


class OracleData:
  """One simulation instance; attributes are numpy views on native buffers."""

  _VIEWS = ('qpos', 'qvel', 'ctrl', 'qacc', 'qacc_warmstart', 'qfrc_applied',
            'xpos', 'xquat', 'xmat', 'xipos', 'ximat', 'geom_xpos',
            'geom_xmat', 'subtree_com', 'qM', 'qfrc_bias', 'qfrc_passive',
            'qfrc_actuator', 'qfrc_smooth', 'qacc_smooth', 'qfrc_constraint',
            'actuator_force', 'subtree_linvel', 'sensordata', 'efc_J',
            'efc_pos', 'efc_aref', 'efc_R', 'efc_D', 'efc_force', 'cvel',
            'cdof', 'efc_vel', 'efc_margin', 'efc_diagApprox')
  _SHAPES = {'xpos': (-1, 3), 'xquat': (-1, 4), 'xmat': (-1, 9),
             'xipos': (-1, 3), 'ximat': (-1, 9), 'geom_xpos': (-1, 3),
             'geom_xmat': (-1, 9), 'subtree_com': (-1, 3),
             'subtree_linvel': (-1, 3), 'cvel': (-1, 6), 'cdof': (-1, 6)}

  def __init__(self, omodel):
    self.omodel = omodel
    self.lib = omodel.lib
    self.ptr = self.lib.mjo_data_new(omodel.ptr)
    n = ctypes.c_int(0)
    for name in self._VIEWS:
      p = self.lib.mjo_data_ptr(self.ptr, name.encode(), ctypes.byref(n))
      if not p:
        raise KeyError(name)
      size = max(n.value, 0)
      arr = (np.ctypeslib.as_array(p, shape=(size,)) if size
             else np.zeros(0))
      if name in self._SHAPES:
        arr = arr.reshape(self._SHAPES[name])
      elif name == 'qM':
        arr = arr.reshape(omodel.model.nv, omodel.model.nv)
      setattr(self, name, arr)
    self.warning = np.ctypeslib.as_array(
        self.lib.mjo_data_warnings(self.ptr), shape=(8,))

  # state ---------------------------------------------------------------
  @property
  def time(self):
    return self.lib.mjo_data_time(self.ptr)

  @time.setter
  def time(self, t):
    self.lib.mjo_data_set_time(self.ptr, float(t))

  @property
  def ncon(self):
    return self.lib.mjo_data_ncon(self.ptr)

  @property
  def nefc(self):
    return self.lib.mjo_data_nefc(self.ptr)

  @property
  def solver_iter(self):
    return self.lib.mjo_data_solver_iter(self.ptr)

  def efc_J_matrix(self):
    nv = self.omodel.model.nv
    return self.efc_J[:self.nefc*nv].reshape(self.nefc, nv)

  def contact(self, i):
    dist = ctypes.c_double()
    pos = np.zeros(3)
    frame = np.zeros(9)
    geoms = np.zeros(3, np.int32)
    dp = ctypes.POINTER(ctypes.c_double)
    self.lib.mjo_contact_get(
        self.ptr, i, ctypes.byref(dist), pos.ctypes.data_as(dp),
        frame.ctypes.data_as(dp),
        geoms.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return dict(dist=dist.value, pos=pos, frame=frame.reshape(3, 3),
                geom1=int(geoms[0]), geom2=int(geoms[1]), dim=int(geoms[2]))

  def contact_force(self, i):
    """[normal, tangent1, tangent2], [torsion, roll1, roll2] (core.py:704-728)."""
    if not 0 <= i < self.ncon:
      raise ValueError('contact id out of range')
    out = np.zeros(6)
    self.lib.mjo_contact_force(
        self.omodel.ptr, self.ptr, i,
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return out.reshape(2, 3)

  def point_velocity(self, body, point):
    """World-frame (linvel, angvel) of a point fixed to `body`."""
    dp = ctypes.POINTER(ctypes.c_double)
    p = np.ascontiguousarray(point, np.float64)
    lin, ang = np.zeros(3), np.zeros(3)
    self.lib.mjo_point_velocity(self.omodel.ptr, self.ptr, int(body),
                                p.ctypes.data_as(dp), lin.ctypes.data_as(dp),
                                ang.ctypes.data_as(dp))
    return lin, ang

  # pipeline --------------------------------------------------------------
  def reset(self):
    self.lib.mjo_reset_data(self.omodel.ptr, self.ptr)

  def forward(self):
    self.lib.mjo_forward(self.omodel.ptr, self.ptr)

  def step(self):
    self.lib.mjo_step(self.omodel.ptr, self.ptr)

  def step1(self):
    self.lib.mjo_step1(self.omodel.ptr, self.ptr)

  def step2(self):
    self.lib.mjo_step2(self.omodel.ptr, self.ptr)

  def physics_step(self):
    """`Physics.step` of the reference (engine.py:149-166)."""
    self.lib.mjo_physics_step(self.omodel.ptr, self.ptr)

  def copy_from(self, other):
    self.lib.mjo_copy_data(self.ptr, other.ptr)

  def __del__(self):
    try:
      self.lib.mjo_data_free(self.ptr)
    except Exception:  # pylint: disable=broad-except
      pass


class OraclePhysics:
  """Single-instance `Physics` on the oracle (engine.py:86-573 subset)."""

  def __init__(self, model):
    self.model = model
    self.omodel = OracleModel(model)
    self.data = OracleData(self.omodel)

  @classmethod
  def from_xml_string(cls, xml, assets=None):
    from dm_control_amd.mjcf import compiler
    return cls(compiler.from_xml_string(xml, assets))

  def _forward_no_actuation(self):
    from dm_control_amd.mjcf import model as mdl
    flags = self.omodel.get_int('disableflags')
    self.omodel.set_int('disableflags', flags | mdl.DSBL_ACTUATION)
    try:
      self.data.forward()
    finally:
      self.omodel.set_int('disableflags', flags)

  def reset(self):
    self.data.reset()
    self._forward_no_actuation()

  def after_reset(self):
    self._forward_no_actuation()

  def forward(self):
    self.data.forward()

  def set_control(self, ctrl):
    np.copyto(self.data.ctrl, ctrl)

  def step(self):
    self.data.physics_step()

  def time(self):
    return self.data.time

  def timestep(self):
    return self.model.opt.timestep


def batch_step(omodel, datas, ctrl, nsub=1, nthreads=0):
  """Steps many instances with OpenMP; returns the thread count used."""
  arr = (ctypes.c_void_p*len(datas))(*[d.ptr for d in datas])
  c = np.ascontiguousarray(ctrl, dtype=np.float64)
  return omodel.lib.mjo_batch_step(
      omodel.ptr, arr, len(datas),
      c.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), nsub, nthreads)
