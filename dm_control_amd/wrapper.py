"""ctypes shim over libdmc_hip.so (include/dmc_hip.h).

Counterpart of the reference's wrapper layer:
  library discovery  /root/reference/dm_control/mujoco/wrapper/util.py:39-67,107-120
                     (`$MJLIB_PATH` -> here `$DMC_HIP_LIB`, else the in-tree build)
  error convention   wrapper/core.py:85-101,312-328 (`wrapper.Error`)
  MjModel / MjData   wrapper/core.py:444-776

There is no CPU fallback: if the library cannot be loaded, or no MI355X-class
device is present, construction raises `Error`.
"""

import ctypes
import os
import sys

import numpy as np

ENV_DMC_HIP_LIB = 'DMC_HIP_LIB'
_DEFAULT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                            'csrc', 'libdmc_hip.so')

# enum dmc_field
(FIELD_QPOS, FIELD_QVEL, FIELD_WARMSTART, FIELD_TIME, FIELD_CTRL, FIELD_OBS,
 FIELD_REWARD, FIELD_SENSORDATA, FIELD_XPOS, FIELD_XMAT, FIELD_QACC,
 FIELD_WARN, FIELD_STATS, FIELD_RETURN, FIELD_TASKDATA) = range(15)


class Error(Exception):
  """Base class for errors raised by the HIP runtime (cf. wrapper.Error)."""


class ModelInfo(ctypes.Structure):
  _fields_ = [(n, ctypes.c_int) for n in (
      'abi', 'real_size', 'nq', 'nv', 'nu', 'nbody', 'nobs', 'nsensordata',
      'ws_per_env', 'task', 'ncon_max', 'nefc_max', 'integrator', 'npair',
      'ntaskdata', 'envs_per_block', 'lanes_per_env', 'env_major')]


# every symbol declared in include/dmc_hip.h: (restype, argtypes)
_vp, _ci, _cll, _cs = (ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong,
                       ctypes.c_size_t)
SIGNATURES = {
    'dmc_version': (_ci, []),
    'dmc_last_error': (ctypes.c_char_p, []),
    'dmc_device_count': (_ci, []),
    'dmc_model_load': (_ci, [ctypes.c_char_p, _ci, ctypes.POINTER(_vp)]),
    'dmc_model_compile': (_ci, [
        ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p),
        ctypes.POINTER(ctypes.c_char_p), _ci, ctypes.POINTER(ctypes.c_char_p), _ci,
        ctypes.POINTER(_vp), ctypes.POINTER(_cs), ctypes.c_char_p, _cs]),
    'dmc_code_free': (None, [_vp]),
    'dmc_model_load_data': (_ci, [_vp, _cs, _ci, ctypes.POINTER(_vp)]),
    'dmc_model_get_info': (_ci, [_vp, ctypes.POINTER(ModelInfo)]),
    'dmc_model_free': (None, [_vp]),
    'dmc_batch_create': (_ci, [_vp, _ci, ctypes.POINTER(_vp)]),
    'dmc_batch_free': (None, [_vp]),
    'dmc_batch_nenv': (_ci, [_vp]),
    'dmc_batch_set_aux_outputs': (_ci, [_vp, _ci]),
    'dmc_batch_set_task_params': (
        _ci, [_vp, _ci, ctypes.POINTER(ctypes.c_double), _ci]),
    'dmc_batch_reset': (_ci, [_vp]),
    'dmc_batch_set_state': (_ci, [_vp, _vp, _vp, _vp, _vp]),
    'dmc_batch_write': (_ci, [_vp, _ci, _vp, _cs]),
    'dmc_batch_init_episode': (_ci, [_vp, ctypes.c_uint64, _ci]),
    'dmc_batch_forward': (_ci, [_vp, _ci]),
    'dmc_batch_step': (_ci, [_vp, _vp, _cll, _cll, _ci, _ci, _ci]),
    'dmc_batch_step_n': (_ci, [_vp, _vp, _cll, _cll, _cll, _ci, _ci, _ci]),
    'dmc_batch_read': (_ci, [_vp, _ci, _vp, _cs]),
    'dmc_batch_field_bytes': (_cs, [_vp, _ci]),
    'dmc_batch_device_ptr': (_vp, [_vp, _ci]),
    'dmc_batch_clear_warnings': (_ci, [_vp]),
    'dmc_batch_copy_state': (_ci, [_vp, _vp]),
    'dmc_batch_sync': (_ci, [_vp]),
    'dmc_batch_stream': (_vp, [_vp]),
    'dmc_batch_set_stream': (_ci, [_vp, _vp, _ci]),
    'dmc_batch_timer_start': (_ci, [_vp]),
    'dmc_batch_timer_stop': (_ci, [_vp, ctypes.POINTER(ctypes.c_double),
                                   ctypes.POINTER(_cll)]),
}

_lib = None


def get_lib_path():
  return os.environ.get(ENV_DMC_HIP_LIB) or _DEFAULT_LIB


def get_lib():
  """Loads libdmc_hip.so (cf. `util.get_mjlib`, wrapper/util.py:107-120)."""
  global _lib
  if _lib is not None:
    return _lib
  # PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as the
  # system one libdmc_hip.so links to).  Whichever is mapped first serves the
  # whole process, and torch fails to find devices when the system runtime got
  # there first, so let torch map its copy before ours when it is installed.
  if 'torch' not in sys.modules and not os.environ.get('DMC_NO_TORCH_PRELOAD'):
    try:
      import torch  # noqa: F401  pylint: disable=unused-import,g-import-not-at-top
    except ImportError:
      pass
  path = get_lib_path()
  if not os.path.exists(path):
    raise Error('HIP extension not found at {!r}; run '
                '`python -c "import __graft_entry__ as g; g.build()"` '
                '(there is no CPU fallback)'.format(path))
  try:
    lib = ctypes.cdll.LoadLibrary(path)
  except OSError as e:
    raise Error('cannot load {!r}: {}'.format(path, e))
  for name, (res, args) in SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
  _lib = lib
  return lib


def _check(rc):
  if rc != 0:
    raise Error(get_lib().dmc_last_error().decode('utf-8', 'replace'))


def compile_code_object(source, source_name, headers, options):
  """gfx950 code object (bytes) of `source` built in-process by
  `dmc_model_compile` -- HIP runtime compilation, no hipcc executable.

  headers: {include name: text}; options: compiler flags.  Needs no GPU."""
  lib = get_lib()
  names = list(headers)
  arr = ctypes.c_char_p*max(1, len(names))
  hn = arr(*[n.encode() for n in names])
  ht = arr(*[headers[n].encode() for n in names])
  opts = (ctypes.c_char_p*max(1, len(options)))(*[o.encode() for o in options])
  code, size = _vp(), _cs()
  log = ctypes.create_string_buffer(1 << 16)
  rc = lib.dmc_model_compile(source.encode(), source_name.encode(), hn, ht, len(names),
                             opts, len(options), ctypes.byref(code),
                             ctypes.byref(size), log, len(log))
  if rc != 0:
    raise Error('%s\n%s' % (lib.dmc_last_error().decode('utf-8', 'replace'),
                            log.value.decode('utf-8', 'replace')[-4000:]))
  try:
    return ctypes.string_at(code.value, size.value), log.value.decode('utf-8', 'replace')
  finally:
    lib.dmc_code_free(code)


class HipModel:
  """Loaded code object of one compiled model (cf. `MjModel`)."""

  def __init__(self, code_object_path, device_id=0, code=None):
    self._lib = get_lib()
    self.ptr = _vp()
    if code is not None:
      self._code = bytes(code)     # (the module keeps no reference to the image)
      _check(self._lib.dmc_model_load_data(self._code, len(self._code), device_id,
                                           ctypes.byref(self.ptr)))
    else:
      _check(self._lib.dmc_model_load(code_object_path.encode(), device_id,
                                      ctypes.byref(self.ptr)))
    self.info = ModelInfo()
    _check(self._lib.dmc_model_get_info(self.ptr, ctypes.byref(self.info)))
    self.device_id = device_id
    self.dtype = np.float32 if self.info.real_size == 4 else np.float64

  @classmethod
  def from_code(cls, code, device_id=0):
    """Loads a code object held in memory (`build.code_object_bytes`,
    `dmc_model_compile`) -- no file involved."""
    return cls(None, device_id, code=code)

  def free(self):
    if self.ptr:
      self._lib.dmc_model_free(self.ptr)
      self.ptr = _vp()

  def __del__(self):
    try:
      self.free()
    except Exception:  # pylint: disable=broad-except
      pass


class HipBatch:
  """HBM-resident state of `nenv` instances (cf. `MjData`)."""

  def __init__(self, model, nenv):
    self._lib = get_lib()
    self.model = model  # keeps the module alive (core.py:643)
    self.nenv = int(nenv)
    self.ptr = _vp()
    _check(self._lib.dmc_batch_create(model.ptr, self.nenv,
                                      ctypes.byref(self.ptr)))

  # -- shapes ---------------------------------------------------------------
  def _shape(self, field):
    i, n = self.model.info, self.nenv
    return {
        FIELD_QPOS: (max(i.nq, 1), n), FIELD_QVEL: (max(i.nv, 1), n),
        FIELD_WARMSTART: (max(i.nv, 1), n), FIELD_TIME: (n,),
        FIELD_CTRL: (max(i.nu, 1), n), FIELD_OBS: (n, max(i.nobs, 1)),
        FIELD_REWARD: (n,), FIELD_SENSORDATA: (max(i.nsensordata, 1), n),
        FIELD_XPOS: (i.nbody*3, n), FIELD_XMAT: (i.nbody*9, n),
        FIELD_QACC: (max(i.nv, 1), n), FIELD_WARN: (n,), FIELD_STATS: (3, n),
        FIELD_RETURN: (n,), FIELD_TASKDATA: (max(i.ntaskdata, 1), n),
    }[field]

  def _dtype(self, field):
    if field == FIELD_WARN:
      return np.uint32
    if field == FIELD_STATS:
      return np.int32
    return self.model.dtype

  def read(self, field):
    """Device -> host copy of a whole field, native [k][env] layout."""
    out = np.empty(self._shape(field), self._dtype(field))
    _check(self._lib.dmc_batch_read(self.ptr, field, out.ctypes.data,
                                    out.nbytes))
    return out

  def device_ptr(self, field):
    return self._lib.dmc_batch_device_ptr(self.ptr, field)

  # -- state ------------------------------------------------------------------
  def reset(self):
    _check(self._lib.dmc_batch_reset(self.ptr))

  def write(self, field, array):
    """Uploads a writable field given as [k][nenv] (any float dtype)."""
    a = np.ascontiguousarray(array, dtype=self._dtype(field))
    if a.shape != self._shape(field):
      raise ValueError('expected shape {}, got {}'.format(
          self._shape(field), a.shape))
    _check(self._lib.dmc_batch_write(self.ptr, field, a.ctypes.data, a.nbytes))

  def set_state(self, qpos=None, qvel=None, warmstart=None, time=None):
    """Uploads [k][nenv] arrays (any float dtype; converted to the batch's)."""
    keep = []

    def prep(a, field):
      if a is None:
        return None
      a = np.ascontiguousarray(a, dtype=self.model.dtype)
      if a.shape != self._shape(field):
        raise ValueError('expected shape {}, got {}'.format(
            self._shape(field), a.shape))
      keep.append(a)
      return a.ctypes.data
    _check(self._lib.dmc_batch_set_state(
        self.ptr, prep(qpos, FIELD_QPOS), prep(qvel, FIELD_QVEL),
        prep(warmstart, FIELD_WARMSTART), prep(time, FIELD_TIME)))

  def set_aux_outputs(self, enabled):
    _check(self._lib.dmc_batch_set_aux_outputs(self.ptr, int(bool(enabled))))

  def set_task_params(self, iparam=0, rparams=()):
    arr = (ctypes.c_double*4)(*(list(rparams) + [0.0]*4)[:4])
    _check(self._lib.dmc_batch_set_task_params(self.ptr, int(iparam), arr,
                                               len(rparams)))

  def init_episode(self, seed, only_colliding=False):
    _check(self._lib.dmc_batch_init_episode(self.ptr, int(seed) & (2**64 - 1),
                                            int(only_colliding)))

  def forward(self, count_contacts=False):
    _check(self._lib.dmc_batch_forward(self.ptr, int(count_contacts)))

  def step_host(self, ctrl, nsub=1, want_outputs=True, stale_first=False):
    """ctrl: host array [nenv, nu] (agent layout) or None.

    stale_first: DMC_STEP_STALE_FIRST (the first substep takes its acceleration
    from the reset state; the reference cheetah's first settle step)."""
    want_outputs = int(bool(want_outputs)) | (2 if stale_first else 0)
    if ctrl is None:
      _check(self._lib.dmc_batch_step(self.ptr, None, 0, 0, 0, nsub,
                                      want_outputs))
      return
    c = np.ascontiguousarray(ctrl, dtype=self.model.dtype)
    nu = self.model.info.nu
    if c.shape != (self.nenv, nu):
      raise ValueError('ctrl must have shape ({}, {}), got {}'.format(
          self.nenv, nu, c.shape))
    _check(self._lib.dmc_batch_step(self.ptr, c.ctypes.data, 1, nu, 0, nsub,
                                    int(want_outputs)))

  def step_device(self, ctrl_ptr, stride_k, stride_env, nsub=1,
                  want_outputs=True):
    """ctrl_ptr: device address; element (k, env) at k*stride_k+env*stride_env."""
    _check(self._lib.dmc_batch_step(self.ptr, ctrl_ptr, stride_k, stride_env,
                                    1, nsub, int(want_outputs)))

  def step_device_n(self, ctrl_ptr, stride_k, stride_env, stride_t, nsteps,
                    nsub=1, want_outputs=True):
    """`nsteps` control steps; step t reads ctrl_ptr + t*stride_t reals."""
    _check(self._lib.dmc_batch_step_n(self.ptr, ctrl_ptr, stride_k, stride_env,
                                      stride_t, nsteps, nsub, int(want_outputs)))

  def clear_warnings(self):
    _check(self._lib.dmc_batch_clear_warnings(self.ptr))

  def copy_state_from(self, other):
    _check(self._lib.dmc_batch_copy_state(self.ptr, other.ptr))

  def stream(self):
    return self._lib.dmc_batch_stream(self.ptr)

  def set_stream(self, stream_ptr, external=True):
    """Runs later launches on a caller-owned hipStream_t.

    `stream_ptr` 0/None with external=True is HIP's default (null) stream;
    external=False restores the batch's own stream.
    """
    _check(self._lib.dmc_batch_set_stream(self.ptr, stream_ptr or None,
                                          int(external)))

  def sync(self):
    _check(self._lib.dmc_batch_sync(self.ptr))

  def timer_start(self):
    _check(self._lib.dmc_batch_timer_start(self.ptr))

  def timer_stop(self):
    ms, n = ctypes.c_double(), _cll()
    _check(self._lib.dmc_batch_timer_stop(self.ptr, ctypes.byref(ms),
                                          ctypes.byref(n)))
    return ms.value, n.value

  def free(self):
    if self.ptr:
      self._lib.dmc_batch_free(self.ptr)
      self.ptr = _vp()

  def __del__(self):
    try:
      self.free()
    except Exception:  # pylint: disable=broad-except
      pass
