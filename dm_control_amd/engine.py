"""Batched `Physics` on the MI355X step kernels.

Mirror of the reference's `mujoco.Physics`
(/root/reference/dm_control/mujoco/engine.py:86-573) for B independent
instances that live in HBM.  Same method names and meaning:

  step           engine.py:149-166   (mj_step2|mj_step, then mj_step1) x n
  set_control    engine.py:141-147
  reset          engine.py:268-289   after_reset :291-295   forward :297-305
  check_invalid_state  :307-330      (mjtWarning bits -> PhysicsError)
  get_state/set_state  :217-245      copy :247-266
  action_spec    engine.py:1018-1028

With `batch_size=None` the object behaves like the reference's single
instance (arrays without a batch axis, python-float time); with an integer
every array gets a leading batch axis.  Field access (`physics.data.qpos`,
`physics.named.data.xpos['head', 'z']`) returns host copies; assignment writes
through to device memory.
"""

import collections
import contextlib
import logging

import numpy as np

from dm_control_amd import _dm_env as dm_env
from dm_control_amd import build
from dm_control_amd import codegen
from dm_control_amd import wrapper
from dm_control_amd.mjcf import compiler
from dm_control_amd.mjcf import model as mdl
from dm_control_amd.rl import control as _control

specs = dm_env.specs

_INVALID_PHYSICS_STATE = (
    'Physics state is invalid. Warning(s) raised: {warning_names}')

Contexts = collections.namedtuple('Contexts', ['gl', 'mujoco'])


class _Field:
  """Host view of one device field with write-through assignment.

  Reads copy device -> host; `field[idx] = v` is read-modify-write.  The batch
  axis leads (agent layout) and is dropped for an unbatched Physics.
  """

  def __init__(self, physics, field, width=None, writable=None, aux=False):
    self._p = physics
    self._f = field
    self._width = width
    self._writable = writable
    self._aux = aux

  def _get(self):
    p = self._p
    if self._aux:
      p._ensure_aux()
    a = p._batch.read(self._f)
    if self._f == wrapper.FIELD_OBS:
      out = a
    elif a.ndim == 2:
      out = a.T
      if self._width is not None:
        out = out[:, :self._width]
    else:
      out = a
    out = np.ascontiguousarray(out)
    if self._f not in (wrapper.FIELD_WARN, wrapper.FIELD_STATS):
      out = out.astype(np.float64)
    return out[0] if p._squeeze else out

  def _put(self, value):
    if self._writable is None:
      raise ValueError('this field is read-only')
    p = self._p
    v = np.asarray(value, dtype=np.float64)
    if p._squeeze:
      v = v[None]
    if v.ndim == 2:
      v = v.T
    p._batch.set_state(**{self._writable: v})
    p._dirty = True

  def __array__(self, dtype=None, copy=None):
    a = self._get()
    return a.astype(dtype) if dtype is not None else a

  def copy(self):
    return self._get()

  def __getitem__(self, idx):
    return self._get()[idx]

  def __setitem__(self, idx, value):
    a = self._get()
    a[idx] = value
    self._put(a)

  def __len__(self):
    return len(self._get())

  @property
  def shape(self):
    return self._get().shape

  def __repr__(self):
    return repr(self._get())


def _quat_to_mat(q):
  w, x, y, z = q
  return np.array([[w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y)],
                   [2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x)],
                   [2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z]])


class _Derived:
  """Read-only `mjData` field computed on the host from the body frames the
  kernel exports (xpos, xmat): frames of geoms, sites and body inertial frames
  are fixed offsets in their body (mj_kinematics), so they need no kernel
  output of their own."""

  def __init__(self, physics, bodyid, pos, quat, want_mat):
    self._p = physics
    self._body = np.asarray(bodyid, np.int64).reshape(-1)
    self._pos = np.asarray(pos, np.float64).reshape(-1, 3)
    self._rot = np.array([_quat_to_mat(q) for q in
                          np.asarray(quat, np.float64).reshape(-1, 4)]).reshape(-1, 3, 3)
    self._want_mat = want_mat

  def _get(self):
    p = self._p
    nb = p.model.nbody
    xpos = np.asarray(p.data.xpos).reshape(-1, nb, 3)
    xmat = np.asarray(p.data.xmat).reshape(-1, nb, 3, 3)
    if self._want_mat:
      out = np.einsum('ebij,bjk->ebik', xmat[:, self._body], self._rot)
      out = out.reshape(len(xpos), -1)
    else:
      out = xpos[:, self._body] + np.einsum('ebij,bj->ebi', xmat[:, self._body],
                                           self._pos)
      out = out.reshape(len(xpos), -1)
    return out[0] if p._squeeze else out

  def _put(self, value):
    raise ValueError('this field is derived from the body frames and read-only')

  def __array__(self, dtype=None, copy=None):
    a = self._get()
    return a if dtype is None else a.astype(dtype)

  def __getitem__(self, idx):
    return self._get()[idx]

  def __repr__(self):
    return repr(self._get())


class _NamedField:
  """`physics.named.data.<field>[row_name(, col_name)]` (mujoco/index.py)."""

  def __init__(self, field, rows, cols=None):
    self._field = field
    self._rows = rows      # name -> slice / int along the last-but-cols axis
    self._cols = cols      # optional name -> int

  def _index(self, key):
    if isinstance(key, tuple):
      r, c = key
    else:
      r, c = key, None
    def one(name, table):
      if isinstance(name, str):
        if name not in table:
          raise IndexError('invalid name {!r}'.format(name))
        return table[name]
      if isinstance(name, (list, tuple)) and name and isinstance(name[0], str):
        return [table[n] for n in name]
      return name
    return one(r, self._rows), (None if c is None else one(c, self._cols or {}))

  def __getitem__(self, key):
    r, c = self._index(key)
    a = self._field._get()
    if self._cols is not None:
      a = a.reshape(a.shape[:-1] + (-1, len(self._cols)))
      out = a[..., r, :]
      return out if c is None else out[..., c]
    return a[..., r]

  def __setitem__(self, key, value):
    r, c = self._index(key)
    a = self._field._get()
    if self._cols is not None:
      b = a.reshape(a.shape[:-1] + (-1, len(self._cols)))
      if c is None:
        b[..., r, :] = value
      else:
        b[..., r, c] = value
    else:
      a[..., r] = value
    self._field._put(a)


class _Data:
  """`physics.data`: host accessors named like `mjData` fields."""

  def __init__(self, physics):
    p, m = physics, physics.model
    W = wrapper
    self.qpos = _Field(p, W.FIELD_QPOS, m.nq, 'qpos')
    self.qvel = _Field(p, W.FIELD_QVEL, m.nv, 'qvel')
    self.qacc_warmstart = _Field(p, W.FIELD_WARMSTART, m.nv, 'warmstart')
    self.qacc = _Field(p, W.FIELD_QACC, m.nv, aux=True)
    self.ctrl = _Field(p, W.FIELD_CTRL, m.nu)
    self.sensordata = _Field(p, W.FIELD_SENSORDATA, m.nsensordata)
    self.xpos = _Field(p, W.FIELD_XPOS, aux=True)
    self.xmat = _Field(p, W.FIELD_XMAT, aux=True)
    self._time = _Field(p, W.FIELD_TIME, None, 'time')
    self._p = p
    nb = m.nbody
    ident = np.tile([1.0, 0, 0, 0], (nb, 1))
    self.xipos = _Derived(p, np.arange(nb), m.body_ipos, ident, False)
    self.ximat = _Derived(p, np.arange(nb), m.body_ipos, m.body_iquat, True)
    self.geom_xpos = _Derived(p, m.geom_bodyid, m.geom_pos, m.geom_quat, False)
    self.geom_xmat = _Derived(p, m.geom_bodyid, m.geom_pos, m.geom_quat, True)
    if getattr(m, 'nsite', 0):
      self.site_xpos = _Derived(p, m.site_bodyid, m.site_pos, m.site_quat, False)
      self.site_xmat = _Derived(p, m.site_bodyid, m.site_pos, m.site_quat, True)

  @property
  def time(self):
    t = self._time._get()
    return float(t) if self._p._squeeze else t

  @time.setter
  def time(self, value):
    n = self._p._batch.nenv
    self._p._batch.set_state(time=np.full(n, value, np.float64))

  @property
  def timer(self):
    """[[cumulative step seconds, step calls]] (cf. mjData.timer[0])."""
    return np.array([[self._p._profile_seconds, self._p._profile_calls]],
                    dtype=np.float64)

  @property
  def ncon(self):
    """Contacts of the most recent collision pass (after `forward`)."""
    s = self._p._batch.read(wrapper.FIELD_STATS)[0]
    return int(s[0]) if self._p._squeeze else s

  @property
  def warning_mask(self):
    w = self._p._batch.read(wrapper.FIELD_WARN)
    return int(w[0]) if self._p._squeeze else w


class _NamedData:

  def __init__(self, physics, data):
    m = physics.model
    jq, jv = {}, {}
    for j, name in enumerate(m.names['joint']):
      if not name:
        continue
      nq = {mdl.JNT_FREE: 7, mdl.JNT_BALL: 4}.get(int(m.jnt_type[j]), 1)
      nv = {mdl.JNT_FREE: 6, mdl.JNT_BALL: 3}.get(int(m.jnt_type[j]), 1)
      a, d = int(m.jnt_qposadr[j]), int(m.jnt_dofadr[j])
      jq[name] = slice(a, a + nq)
      jv[name] = slice(d, d + nv)
    bodies = {n: i for i, n in enumerate(m.names['body']) if n}
    acts = {n: i for i, n in enumerate(m.names.get('actuator', [])) if n}
    sens = {}
    for i, n in enumerate(m.names.get('sensor', [])):
      if n:
        a = int(m.sensor_adr[i])
        sens[n] = slice(a, a + int(m.sensor_dim[i]))
    xyz = {'x': 0, 'y': 1, 'z': 2}
    mat = {a + b: 3*i + j for i, a in enumerate('xyz') for j, b in
           enumerate('xyz')}
    self.qpos = _NamedField(data.qpos, jq)
    self.qvel = _NamedField(data.qvel, jv)
    self.ctrl = _NamedField(data.ctrl, acts)
    self.sensordata = _NamedField(data.sensordata, sens)
    self.xpos = _NamedField(data.xpos, bodies, xyz)
    self.xmat = _NamedField(data.xmat, bodies, mat)
    self.xipos = _NamedField(data.xipos, bodies, xyz)
    self.ximat = _NamedField(data.ximat, bodies, mat)
    geoms = {n: i for i, n in enumerate(m.names.get('geom', [])) if n}
    self.geom_xpos = _NamedField(data.geom_xpos, geoms, xyz)
    self.geom_xmat = _NamedField(data.geom_xmat, geoms, mat)
    if hasattr(data, 'site_xpos'):
      sites = {n: i for i, n in enumerate(m.names.get('site', [])) if n}
      self.site_xpos = _NamedField(data.site_xpos, sites, xyz)
      self.site_xmat = _NamedField(data.site_xmat, sites, mat)


class _NamedArray:
  """Read-only `named.model.<field>[row(s)(, col)]` view (mujoco/index.py)."""

  def __init__(self, array, rows):
    self._a = np.asarray(array)
    self._rows = rows

  def _row(self, key):
    if isinstance(key, str):
      if key not in self._rows:
        raise IndexError('invalid name {!r}'.format(key))
      return self._rows[key]
    if isinstance(key, (list, tuple)) and key and isinstance(key[0], str):
      return [self._row(k) for k in key]
    return key

  def __getitem__(self, key):
    if isinstance(key, tuple):
      return self._a[(self._row(key[0]),) + tuple(key[1:])]
    return self._a[self._row(key)]

  def __setitem__(self, key, value):
    raise ValueError(
        'the compiled model is immutable: what a task varies per episode is '
        'per-instance task data (DMC_FIELD_TASKDATA), see suite/reacher.py')

  def __array__(self, dtype=None, copy=None):
    return self._a if dtype is None else self._a.astype(dtype)

  def __repr__(self):
    return repr(self._a)


class _NamedModel:
  """`physics.named.model`: model arrays indexable by object names."""

  _KINDS = (('body_', 'body'), ('jnt_', 'joint'), ('geom_', 'geom'),
            ('site_', 'site'), ('actuator_', 'actuator'), ('sensor_', 'sensor'),
            ('tendon_', 'tendon'))

  def __init__(self, model):
    self._m = model

  def __getattr__(self, name):
    value = getattr(self._m, name)
    for prefix, kind in self._KINDS:
      if name.startswith(prefix) and isinstance(value, np.ndarray):
        rows = {n: i for i, n in enumerate(self._m.names.get(kind, [])) if n}
        return _NamedArray(value, rows)
    return value


class _Named:

  def __init__(self, physics, data):
    self.data = _NamedData(physics, data)
    self.model = _NamedModel(physics.model)


class Physics(_control.Physics):
  """Batched simulation of one compiled MJCF on one MI355X."""

  _TASK = codegen.TASK_NONE   # domain subclasses select the fused task
  _BUILD_MODE = 'auto'        # see build.build_model
  # When the mode is "auto", batches up to `max_batch` use the several-lanes-
  # per-env kernel (build mode "coop") with `group` lanes per env: one env per
  # lane needs >= 64 envs per CU to fill the chip, one env per wavefront fills
  # it at 4 per CU, two envs per wavefront at 8; 128 = one env per wavefront
  # plus a helper wavefront, fastest while the batch runs in one round.
  # ((max_batch, group), ...) in increasing max_batch; measured cross-overs per
  # domain (profiles/r02_kernel_shape_sweep.txt).
  _COOP_POLICY = ()
  # fp64 cross-overs differ: the one-env-per-lane fp64 build of a contact model
  # does not fit the register file (its per-lane arrays live in scratch: 134 MB
  # of scratch traffic per cheetah launch), so the several-lanes kernel stays
  # ahead up to larger batches (profiles/r03_f64_kernel_shape_sweep.txt).
  # None: same as _COOP_POLICY.
  _COOP_POLICY_F64 = None
  _GROUP = 64                 # lanes per env of build mode "coop" (128: two wavefronts)

  def __init__(self, model, batch_size=None, device=0, precision='f32',
               task=None, ncon_max=None, build_mode=None, group=None):
    self.model = model
    self._squeeze = batch_size is None
    self._batch_size = 1 if batch_size is None else int(batch_size)
    self._task_id = self._TASK if task is None else task
    self._precision = precision
    self._device = device
    self._ncon_max = ncon_max
    self._warnings_cause_exception = True
    self._pending_ctrl = None
    self._dirty = True
    self._aux_on = False
    self._profiling = False
    self._profile_seconds = 0.0
    self._profile_calls = 0
    self._build_mode = build_mode or self._BUILD_MODE
    self._group = group or self._GROUP
    if build_mode is None and self._build_mode == 'auto' and precision != 'mixed':
      policy = self._COOP_POLICY
      if precision == 'f64' and self._COOP_POLICY_F64 is not None:
        policy = self._COOP_POLICY_F64
      for max_batch, lanes in policy:
        if self._batch_size <= max_batch:
          self._build_mode, self._group = 'coop', lanes
          break
    path = build.build_model(
        model, self._task_id, precision, ncon_max, mode=self._build_mode,
        lds_budget=build.lds_budget_for(self._batch_size), group=self._group)
    self._code_object = path
    self._hip_model = wrapper.HipModel(path, device)
    self._batch = wrapper.HipBatch(self._hip_model, self._batch_size)
    self.data = _Data(self)
    self.named = _Named(self, self.data)
    self._warn_seen = np.zeros(self._batch_size, np.uint32)

  # -- constructors (engine.py:411-470) ------------------------------------------
  @classmethod
  def from_xml_string(cls, xml_string, assets=None, **kwargs):
    return cls(compiler.from_xml_string(xml_string, assets), **kwargs)

  @classmethod
  def from_xml_path(cls, file_path, **kwargs):
    return cls(compiler.from_xml_path(file_path), **kwargs)

  @classmethod
  def from_model(cls, model, **kwargs):
    return cls(model, **kwargs)

  # -- properties -----------------------------------------------------------------
  @property
  def batch_size(self):
    return None if self._squeeze else self._batch_size

  @property
  def batch(self):
    """The underlying `wrapper.HipBatch` (device pointers, raw fields)."""
    return self._batch

  @property
  def dtype(self):
    return self._hip_model.dtype

  @property
  def code_object(self):
    """Path of the gfx950 code object this batch runs (content-hashed name)."""
    return self._code_object

  @property
  def kernel_shape(self):
    info = self._hip_model.info
    if info.lanes_per_env > 64:
      return ('64 lanes per env + a second wavefront building the constraint '
              'rows (csrc/dmc_coop.hip)')
    if info.lanes_per_env > 1 and not info.env_major:
      # (only the team build of csrc/dmc_kernels.hip keeps [k][env] state with
      # several lanes per env)
      return ('one wavefront per env, matrices in the HBM workspace, a tree\'s block '
              'at a time in LDS (csrc/dmc_kernels.hip, team mode)')
    if info.lanes_per_env > 1:
      return '%d lanes per env (csrc/dmc_coop.hip)' % info.lanes_per_env
    return 'one env per lane (csrc/dmc_kernels.hip)'

  # -- stepping -----------------------------------------------------------------
  @contextlib.contextmanager
  def suppress_physics_errors(self):
    prev = self._warnings_cause_exception
    self._warnings_cause_exception = False
    try:
      yield
    finally:
      self._warnings_cause_exception = prev

  def enable_profiling(self):
    """Times every step launch with HIP events (engine.py:137-139)."""
    self._profiling = True

  def set_control(self, control):
    """Stores the control applied by subsequent steps (engine.py:141-147)."""
    c = np.asarray(control, dtype=np.float64)
    if c.ndim == 1:
      c = np.broadcast_to(c, (self._batch_size, self.model.nu))
    if c.shape != (self._batch_size, self.model.nu):
      raise ValueError('control must have shape ({}, {}), got {}'.format(
          self._batch_size, self.model.nu, c.shape))
    self._pending_ctrl = np.ascontiguousarray(c)

  def set_control_device(self, ptr, stride_k, stride_env):
    """Zero-copy control: a device address plus element strides (in reals)."""
    self._pending_ctrl = ('device', int(ptr), int(stride_k), int(stride_env))

  def step(self, n_sub_steps=1, outputs=True, check=True, stale_first=False):
    """`n_sub_steps` x (mj_step2|mj_step + mj_step1), one kernel launch.

    stale_first: the first substep is an `mj_step2` on the position/velocity
    stage left by `reset()` (qpos0), applied to the current state -- the order
    of operations of a task that rewrites qpos inside `reset_context` and steps
    without a forward pass (suite/cheetah.py:63-77)."""
    ctrl = self._pending_ctrl
    if self._profiling:
      self._batch.timer_start()
    if isinstance(ctrl, tuple):
      if stale_first:
        raise ValueError('stale_first applies to settle steps (no new control)')
      self._batch.step_device(ctrl[1], ctrl[2], ctrl[3], n_sub_steps, outputs)
    else:
      self._batch.step_host(ctrl, n_sub_steps, outputs, stale_first)
    if self._profiling:
      ms, _ = self._batch.timer_stop()
      self._profile_seconds += ms*1e-3
      self._profile_calls += n_sub_steps
    self._pending_ctrl = None   # ctrl now lives in data.ctrl on the device
    self._dirty = not outputs
    if check:
      self.check_invalid_state()

  def forward(self, count_contacts=False):
    """Recomputes derived quantities (observation inputs) without stepping."""
    self._batch.forward(count_contacts)
    self._dirty = False

  def reset(self):
    """mj_resetData + forward with actuation disabled (engine.py:268-289)."""
    self._batch.reset()
    self._warn_seen[:] = 0
    self._pending_ctrl = None
    self.forward()

  def after_reset(self):
    """engine.py:291-295."""
    self.forward(count_contacts=True)

  def check_invalid_state(self):
    """Raises PhysicsError for new mjtWarning bits (engine.py:307-330)."""
    w = self._batch.read(wrapper.FIELD_WARN)
    new = w & ~self._warn_seen
    self._warn_seen = w.copy()
    if new.any():
      bits = int(np.bitwise_or.reduce(new))
      names = [n for i, n in enumerate(mdl.WARNING_NAMES) if bits & (1 << i)]
      message = _INVALID_PHYSICS_STATE.format(warning_names=', '.join(names))
      if self._warnings_cause_exception:
        raise _control.PhysicsError(
            message + ' (envs {})'.format(np.nonzero(new)[0][:8].tolist()))
      logging.warning(message)

  check_divergence = check_invalid_state

  def _ensure_aux(self):
    """xpos/xmat/qacc are written only once somebody asked for them; from then
    on every step keeps them current (qacc: from the next step on)."""
    if not self._aux_on:
      self._aux_on = True
      self._batch.set_aux_outputs(True)
      self.forward()

  # -- fused task outputs ---------------------------------------------------------
  def _ensure_outputs(self):
    if self._dirty:
      self.forward()

  def fused_observation(self):
    """[B, nobs] (or [nobs]) observation vector written by the last launch."""
    self._ensure_outputs()
    return self._squeeze_out(
        self._batch.read(wrapper.FIELD_OBS).astype(np.float64))

  def fused_reward(self):
    self._ensure_outputs()
    r = self._batch.read(wrapper.FIELD_REWARD).astype(np.float64)
    return float(r[0]) if self._squeeze else r

  def _squeeze_out(self, a):
    return a[0] if self._squeeze else a

  # -- state accessors (engine.py:217-245, 520-573) ----------------------------
  def time(self):
    return self.data.time

  def timestep(self):
    return self.model.opt.timestep

  def control(self):
    if isinstance(self._pending_ctrl, np.ndarray):
      return self._squeeze_out(self._pending_ctrl.copy())
    return self.data.ctrl.copy()

  def position(self):
    return self.data.qpos.copy()

  def velocity(self):
    return self.data.qvel.copy()

  def activation(self):
    shape = (0,) if self._squeeze else (self._batch_size, 0)
    return np.zeros(shape)

  def state(self):
    return np.concatenate(self._physics_state_items(), axis=-1)

  def _physics_state_items(self):
    return [self.position(), self.velocity(), self.activation()]

  def get_state(self):
    return np.concatenate(self._physics_state_items(), axis=-1)

  def set_state(self, physics_state):
    physics_state = np.asarray(physics_state, np.float64)
    nq, nv = self.model.nq, self.model.nv
    expected = ((nq + nv,) if self._squeeze
                else (self._batch_size, nq + nv))
    if physics_state.shape != expected:
      raise ValueError('Input physics state has shape {}. Expected {}.'
                       .format(physics_state.shape, expected))
    s = physics_state.reshape(self._batch_size, nq + nv)
    self._batch.set_state(qpos=s[:, :nq].T, qvel=s[:, nq:].T)
    self._dirty = True

  def copy(self, share_model=True):
    del share_model  # the compiled model is immutable and always shared
    new = type(self).__new__(type(self))
    Physics.__init__(new, self.model,
                     None if self._squeeze else self._batch_size,
                     self._device, self._precision, self._task_id,
                     self._ncon_max, self._build_mode, self._group)
    new._batch.copy_state_from(self._batch)
    new._warn_seen = self._warn_seen.copy()
    new._dirty = self._dirty
    new._aux_on = self._aux_on
    return new

  # -- checkpoints ----------------------------------------------
  _CHECKPOINT_FIELDS = (
      ('qpos', wrapper.FIELD_QPOS), ('qvel', wrapper.FIELD_QVEL),
      ('qacc_warmstart', wrapper.FIELD_WARMSTART), ('time', wrapper.FIELD_TIME),
      ('ctrl', wrapper.FIELD_CTRL), ('taskdata', wrapper.FIELD_TASKDATA),
      ('episode_return', wrapper.FIELD_RETURN), ('warn', wrapper.FIELD_WARN))

  @staticmethod
  def _checkpoint_path(path):
    path = str(path)
    return path if path.endswith('.npz') else path + '.npz'

  def save_checkpoint(self, path, step_count=None):
    """Writes everything the next step depends on to `<path>.npz`.

    qpos, qvel, qacc_warmstart and time are what `mj_step` carries from one
    step to the next (the warm start seeds the Newton solver); ctrl is the
    control a step without a new action re-applies; the per-instance task data
    (reacher target, point_mass directions) is part of the dynamics and of the
    reward; episode_return and the warning mask complete the bookkeeping.  A
    restored f32 or f64 batch continues bit-for-bit under the same actions.  A
    precision='mixed' batch restores the fp32 words of qpos/qvel only (the low
    words of its fp64 state live in the kernel's workspace and restart at
    zero), so it continues to fp32 rounding, not bit-for-bit.  `step_count`
    (e.g. `Environment.step_count`) is stored for the caller's episode logic.
    """
    b = self._batch
    arrays = {name: b.read(field) for name, field in self._CHECKPOINT_FIELDS}
    np.savez(self._checkpoint_path(path),
             model_hash=np.array(self.model.content_hash()),
             precision=np.array(self._precision),
             step_count=np.array(-1 if step_count is None else int(step_count)),
             **arrays)

  def load_checkpoint(self, path):
    """Restores a state written by `save_checkpoint` (same model, precision and
    batch size); returns the stored step count (None if none was stored)."""
    with np.load(self._checkpoint_path(path), allow_pickle=False) as z:
      needed = ['model_hash', 'precision', 'step_count'] + [
          name for name, _ in self._CHECKPOINT_FIELDS]
      missing = [name for name in needed if name not in z.files]
      if missing:
        raise ValueError(
            'checkpoint {} lacks {} (written by an older version that stored '
            'qpos/qvel/qacc_warmstart/time only?)'.format(path, ', '.join(missing)))
      if str(z['model_hash']) != self.model.content_hash():
        raise ValueError('checkpoint was written for a different model')
      if str(z['precision']) != self._precision:
        raise ValueError('checkpoint was written by a {} build, this batch is {}'
                         .format(z['precision'], self._precision))
      if z['qpos'].shape != (self.model.nq, self._batch_size):
        raise ValueError('checkpoint holds {} instances, this batch {}'.format(
            z['qpos'].shape[-1], self._batch_size))
      for name, field in self._CHECKPOINT_FIELDS:
        self._batch.write(field, z[name])
      step_count = int(z['step_count'])
    self._warn_seen = self._batch.read(wrapper.FIELD_WARN).copy()
    self._pending_ctrl = None
    self._dirty = True
    return None if step_count < 0 else step_count

  def set_task_params(self, iparam=0, rparams=()):
    self._batch.set_task_params(iparam, rparams)

  def free(self):
    self._batch.free()
    self._hip_model.free()

  # context-manager parity with engine.py:392-409
  def __enter__(self):
    return self

  def __exit__(self, *unused):
    self.free()


def action_spec(physics):
  """`BoundedArray` matching the actuators (engine.py:1018-1028)."""
  m = physics.model
  num_actions = m.nu
  is_limited = m.actuator_ctrllimited.ravel().astype(bool)
  minima = np.full(num_actions, fill_value=-mdl.MJ_MAXVAL, dtype=np.float64)
  maxima = np.full(num_actions, fill_value=mdl.MJ_MAXVAL, dtype=np.float64)
  minima[is_limited], maxima[is_limited] = m.actuator_ctrlrange[is_limited].T
  return specs.BoundedArray(shape=(num_actions,), dtype=np.float64,
                            minimum=minima, maximum=maxima)
