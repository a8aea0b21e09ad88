"""Model -> generated C++ constants header for the HIP kernels.

The device code (csrc/dmc_kernels.hip) is hand-written once; what differs per
model is a table of compile-time constants: tree topology, inertias, the static
list of geom pairs that can ever collide (with their mixed contact parameters)
and the limited joints.  Emitting them as `static __device__ constexpr` arrays
lets hipcc fold topology into straight-line code for small models.

Host-side logic restated here (independently of oracle/mjstep.c):
  * collision filtering: contype/conaffinity, same weld group, parent-child,
    <exclude>  (SURVEY.md Appendix A "collision")
  * contact parameter mixing for equal priority: condim/friction = max,
    solref/solimp weighted by solmix, margin/gap = max
  * stiffness/damping of the constraint reference: b = 2/(dmax*tc),
    k = 1/(dmax^2 tc^2 zeta^2), tc >= 2*dt (refsafe)
"""

import numpy as np

from dm_control_amd.mjcf import model as mdl

TASK_NONE, TASK_CARTPOLE, TASK_CHEETAH, TASK_HUMANOID = 0, 1, 2, 3
TASK_WALKER, TASK_PENDULUM, TASK_ACROBOT, TASK_HOPPER = 4, 5, 6, 7
TASK_REACHER, TASK_POINTMASS = 8, 9
SENS_TOUCH = 0

_SUPPORTED_PAIRS = {
    (mdl.GEOM_PLANE, mdl.GEOM_SPHERE), (mdl.GEOM_PLANE, mdl.GEOM_CAPSULE),
    (mdl.GEOM_PLANE, mdl.GEOM_BOX), (mdl.GEOM_SPHERE, mdl.GEOM_SPHERE),
    (mdl.GEOM_SPHERE, mdl.GEOM_CAPSULE), (mdl.GEOM_CAPSULE, mdl.GEOM_CAPSULE),
    (mdl.GEOM_SPHERE, mdl.GEOM_BOX), (mdl.GEOM_CAPSULE, mdl.GEOM_BOX),
    (mdl.GEOM_BOX, mdl.GEOM_BOX),
}


class UnsupportedModelError(ValueError):
  pass


def _sanitise_solimp(s):
  return [float(np.clip(s[0], mdl.MJ_MINIMP, mdl.MJ_MAXIMP)),
          float(np.clip(s[1], mdl.MJ_MINIMP, mdl.MJ_MAXIMP)),
          float(max(0.0, s[2])),
          float(np.clip(s[3], mdl.MJ_MINIMP, mdl.MJ_MAXIMP)),
          float(max(1.0, s[4]))]


def _kb(solref, solimp, timestep, refsafe):
  tc, dr = float(solref[0]), float(solref[1])
  dmax = float(np.clip(solimp[1], mdl.MJ_MINIMP, mdl.MJ_MAXIMP))
  if tc > 0:
    if refsafe:
      tc = max(tc, 2*timestep)
    k = 1/max(mdl.MJ_MINVAL, dmax*dmax*tc*tc*dr*dr)
    b = 2/max(mdl.MJ_MINVAL, dmax*tc)
  else:
    k = -tc/max(mdl.MJ_MINVAL, dmax*dmax)
    b = -dr/max(mdl.MJ_MINVAL, dmax)
  return k, b


def collision_pairs(m):
  """Static list of geom pairs that pass MuJoCo's filters, type-ordered."""
  pairs = []
  if m.opt.disableflags & (mdl.DSBL_CONTACT | mdl.DSBL_CONSTRAINT):
    return pairs
  excl = set(int(s) for s in np.atleast_1d(m.exclude_signature)[:m.nexclude])
  for ga in range(m.ngeom):
    for gb in range(ga + 1, m.ngeom):
      ta, tb = int(m.geom_type[ga]), int(m.geom_type[gb])
      if ta == mdl.GEOM_PLANE and tb == mdl.GEOM_PLANE:
        continue
      if not ((m.geom_contype[ga] & m.geom_conaffinity[gb]) or
              (m.geom_contype[gb] & m.geom_conaffinity[ga])):
        continue
      b1, b2 = int(m.geom_bodyid[ga]), int(m.geom_bodyid[gb])
      w1, w2 = int(m.body_weldid[b1]), int(m.body_weldid[b2])
      if w1 == w2:
        continue
      if not (m.opt.disableflags & mdl.DSBL_FILTERPARENT) and w1 and w2:
        pw1 = int(m.body_weldid[m.body_parentid[w1]])
        pw2 = int(m.body_weldid[m.body_parentid[w2]])
        if w1 == pw2 or w2 == pw1:
          continue
      if (min(b1, b2) << 16) + max(b1, b2) in excl:
        continue
      g1, g2 = (ga, gb) if ta <= tb else (gb, ga)
      key = (int(m.geom_type[g1]), int(m.geom_type[g2]))
      if key not in _SUPPORTED_PAIRS:
        raise UnsupportedModelError(
            'collision between geom types %s is not implemented (geoms %r, %r)'
            % (key, m.names['geom'][g1], m.names['geom'][g2]))
      pairs.append((g1, g2))
  return _group_pairs_by_tree(m, pairs)


def _trees_of(m):
  """(root body ids, tree index of every body; -1: the world and what is welded to it)."""
  roots = sorted(set(int(r) for r in m.body_rootid[1:])) or [0]
  tree = [-1 if int(r) == 0 else roots.index(int(r)) for r in m.body_rootid]
  if tree:
    tree[0] = -1
  return roots, tree


def pair_run_key(m, tree, pair):
  """What the pairs of one broadphase run share: two different trees, or a world
  geom and a tree (None: pairs inside one tree, or between world geoms)."""
  ta, tb = tree[int(m.geom_bodyid[pair[0]])], tree[int(m.geom_bodyid[pair[1]])]
  if ta >= 0 and tb >= 0:
    return ('tt', min(ta, tb), max(ta, tb)) if ta != tb else None
  if ta < 0 and tb < 0:
    return None
  return ('wt', pair[0] if ta < 0 else pair[1], max(ta, tb))


def _group_pairs_by_tree(m, pairs):
  """Scenes of several kinematic trees (a soccer pitch): the pairs between the same
  two trees, and between one world geom and one tree, become consecutive -- a
  tree-level bounding test then skips a whole run at once.  (Geom order decided the
  list before: a walker's 43 geoms against another walker's came in 43 runs.)
  One-tree models keep the geom order."""
  roots, tree = _trees_of(m)
  ndof_trees = len(set(int(m.body_rootid[int(m.dof_bodyid[i])]) for i in range(m.nv)))
  if ndof_trees < 2:
    return pairs
  def order(pair):
    key = pair_run_key(m, tree, pair)
    if key is None:       # inside a tree: by tree, after the rest
      return (2, tree[int(m.geom_bodyid[pair[0]])], 0)
    return (0, key[1], key[2]) if key[0] == 'wt' else (1, key[1], key[2])
  return sorted(pairs, key=order)      # (stable: geom order inside a run)


def mix_pair(m, g1, g2):
  """Mixed contact parameters of a geom pair (equal or unequal priority)."""
  if m.geom_priority[g1] != m.geom_priority[g2]:
    gp = g1 if m.geom_priority[g1] > m.geom_priority[g2] else g2
    dim = int(m.geom_condim[gp])
    f = m.geom_friction[gp]
    solref = m.geom_solref[gp].copy()
    solimp = m.geom_solimp[gp].copy()
  else:
    dim = int(max(m.geom_condim[g1], m.geom_condim[g2]))
    f = np.maximum(m.geom_friction[g1], m.geom_friction[g2])
    s1, s2 = float(m.geom_solmix[g1]), float(m.geom_solmix[g2])
    if s1 >= mdl.MJ_MINVAL and s2 >= mdl.MJ_MINVAL:
      mix = s1/(s1 + s2)
    elif s1 < mdl.MJ_MINVAL and s2 < mdl.MJ_MINVAL:
      mix = 0.5
    else:
      mix = 0.0 if s1 < mdl.MJ_MINVAL else 1.0
    if m.geom_solref[g1][0] > 0 and m.geom_solref[g2][0] > 0:
      solref = mix*m.geom_solref[g1] + (1 - mix)*m.geom_solref[g2]
    else:
      solref = np.minimum(m.geom_solref[g1], m.geom_solref[g2])
    solimp = mix*m.geom_solimp[g1] + (1 - mix)*m.geom_solimp[g2]
  margin = float(max(m.geom_margin[g1], m.geom_margin[g2]))
  gap = float(max(m.geom_gap[g1], m.geom_gap[g2]))
  friction = [float(f[0]), float(f[0]), float(f[1]), float(f[2]), float(f[2])]
  return dict(dim=dim, friction=friction, solref=solref, solimp=solimp,
              margin=margin, includemargin=margin - gap)


def _fmt(vals, kind):
  vals = list(np.asarray(vals).ravel())
  if not vals:
    vals = [0]
  if kind == 'int':
    return ', '.join(str(int(v)) for v in vals)
  return ', '.join(repr(float(v)) for v in vals)


def task_bodies(m, task):
  if task in (TASK_CHEETAH, TASK_WALKER):
    return [m.name2id('torso', 'body')]
  if task == TASK_PENDULUM:
    return [m.name2id('pole', 'body')]
  if task == TASK_ACROBOT:
    return [m.name2id('upper_arm', 'body'), m.name2id('lower_arm', 'body')]
  if task == TASK_HOPPER:
    return [m.name2id('torso', 'body'), m.name2id('foot', 'body')]
  if task == TASK_REACHER:
    return [m.name2id('finger', 'body')]
  if task == TASK_POINTMASS:
    return [m.name2id('pointmass', 'body')]
  if task == TASK_HUMANOID:
    return [m.name2id(n, 'body') for n in
            ('torso', 'head', 'left_hand', 'left_foot', 'right_hand',
             'right_foot')]
  return [0]


def observation_size(m, task):
  if task == TASK_CARTPOLE:
    return 1 + 2*(m.nbody - 2) + m.nv
  if task == TASK_CHEETAH:
    return (m.nq - 1) + m.nv
  if task == TASK_HUMANOID:
    return (m.nq - 7) + 1 + 12 + 3 + 3 + m.nv
  if task == TASK_WALKER:
    return 2*(m.nbody - 1) + 1 + m.nv
  if task == TASK_PENDULUM:
    return 3
  if task == TASK_ACROBOT:
    return 4 + m.nv
  if task == TASK_HOPPER:
    return (m.nq - 1) + m.nv + 2
  if task == TASK_REACHER:
    return m.nq + 2 + m.nv
  if task == TASK_POINTMASS:
    return m.nq + m.nv
  return m.nq + m.nv


def task_data_size(task):
  """Per-instance task parameters (DMC_FIELD_TASKDATA rows)."""
  # reacher: target x, y; point_mass: the 4 tendon coefficients (wrap_prm)
  return {TASK_REACHER: 2, TASK_POINTMASS: 4}.get(task, 0)


def task_data_default(m, task):
  """Values the per-instance task data starts with (the compiled model's)."""
  if task == TASK_POINTMASS:
    return [float(v) for v in m.wrap_prm[:4]]
  return [0.0]*task_data_size(task)


def task_sites(m, task):
  """Sites a task's reward reads: [(body id, local pos[3], size[0])]."""
  names = {TASK_ACROBOT: ('tip', 'target'),
           TASK_REACHER: ('geom:finger', 'geom:target'),
           TASK_POINTMASS: ('geom:pointmass', 'geom:target')}.get(task, ())
  out = []
  for n in names:
    if n.startswith('geom:'):     # a geom frame used like a site
      i = m.name2id(n[5:], 'geom')
      out.append((int(m.geom_bodyid[i]), [float(v) for v in m.geom_pos[i]],
                  float(m.geom_size[i][0])))
    else:
      i = m.name2id(n, 'site')
      out.append((int(m.site_bodyid[i]), [float(v) for v in m.site_pos[i]],
                  float(m.site_size[i][0])))
  return out


def capacities(m, pairs, ncon_max=None):
  """Contact / constraint-row capacities of the per-env workspace."""
  per_pair = {mdl.GEOM_CAPSULE: 2, mdl.GEOM_BOX: 4}
  worst = 0
  worst_rows = 0
  mixed = [mix_pair(m, g1, g2) for g1, g2 in pairs]
  for (g1, g2), mx in zip(pairs, mixed):
    t1, t2 = int(m.geom_type[g1]), int(m.geom_type[g2])
    if t1 == mdl.GEOM_PLANE:
      n = per_pair.get(t2, 1)
    elif t2 == mdl.GEOM_BOX and t1 in (mdl.GEOM_CAPSULE, mdl.GEOM_BOX):
      n = 2 if t1 == mdl.GEOM_CAPSULE else 4     # capsule-box, box-box manifolds
    else:
      n = 1
    worst += n
    worst_rows += n*(1 if mx['dim'] == 1 else 2*(mx['dim'] - 1))
  nlimit = int(np.sum((np.asarray(m.jnt_limited) != 0) & np.isin(
      m.jnt_type, (mdl.JNT_HINGE, mdl.JNT_SLIDE)))) if m.njnt else 0
  if m.opt.disableflags & (mdl.DSBL_LIMIT | mdl.DSBL_CONSTRAINT):
    nlimit = 0
  if ncon_max is None:
    ncon_max = min(worst, 32)
  rows_per_con = max([1] + [1 if mx['dim'] == 1 else 2*(mx['dim'] - 1)
                            for mx in mixed])
  nefc_max = nlimit + min(worst_rows, ncon_max*rows_per_con)
  return max(ncon_max, 1), max(nefc_max, 1)


def planar_in_xz(m):
  """True if every dof moves its bodies inside the x-z plane: slides along an
  axis with no y component and hinges about +-y, on bodies whose frames are
  rotations about y only.  Then the Jacobian of any point along the world y
  direction is exactly zero, so a contact whose frame has a tangent along +-y
  gets two IDENTICAL pyramid rows J_n +- mu*0: the one-env-per-lane kernel
  stores them as one row of twice the weight (DMC_PLANAR_MERGE)."""
  import numpy as np
  if m.nv == 0:
    return False
  for j in range(m.njnt):
    axis = np.asarray(m.jnt_axis[j], float)
    if m.jnt_type[j] == mdl.JNT_SLIDE:
      if axis[1] != 0.0:
        return False
    elif m.jnt_type[j] == mdl.JNT_HINGE:
      if axis[0] != 0.0 or axis[2] != 0.0:
        return False
    else:
      return False
  for b in range(1, m.nbody):
    quat = m.body_quat[b]      # (the inertial frame does not enter a Jacobian)
    if float(quat[1]) != 0.0 or float(quat[3]) != 0.0:   # (w, x, y, z): about y only
      return False
  return True


def generate_header(m, task=TASK_NONE, ncon_max=None, unroll=None):
  """Returns the text of the constants header for model `m`."""
  if m.opt.cone != mdl.CONE_PYRAMIDAL:
    raise UnsupportedModelError('only pyramidal cones are implemented')
  if m.opt.solver != mdl.SOLVER_NEWTON:
    raise UnsupportedModelError('only the Newton solver is implemented')
  pairs = collision_pairs(m)
  mixed = [mix_pair(m, g1, g2) for g1, g2 in pairs]
  ncon_max, nefc_max = capacities(m, pairs, ncon_max)
  refsafe = not (m.opt.disableflags & mdl.DSBL_REFSAFE)
  dt = float(m.opt.timestep)
  if unroll is None:
    unroll = True

  limit_jnt = [j for j in range(m.njnt)
               if m.jnt_limited[j] and m.jnt_type[j] in (mdl.JNT_HINGE,
                                                         mdl.JNT_SLIDE)]
  if m.opt.disableflags & (mdl.DSBL_LIMIT | mdl.DSBL_CONSTRAINT):
    limit_jnt = []
  limit_k, limit_b, limit_solimp = [], [], []
  for j in limit_jnt:
    k, b = _kb(m.jnt_solref[j], m.jnt_solimp[j], dt, refsafe)
    limit_k.append(k)
    limit_b.append(b)
    limit_solimp.extend(_sanitise_solimp(m.jnt_solimp[j]))

  pair_k, pair_b, pair_solimp, pair_diag, pair_fric = [], [], [], [], []
  for (g1, g2), mx in zip(pairs, mixed):
    k, b = _kb(mx['solref'], mx['solimp'], dt, refsafe)
    pair_k.append(k)
    pair_b.append(b)
    pair_solimp.extend(_sanitise_solimp(mx['solimp']))
    b1, b2 = int(m.geom_bodyid[g1]), int(m.geom_bodyid[g2])
    tran = float(m.body_invweight0[b1, 0] + m.body_invweight0[b2, 0])
    rot = float(m.body_invweight0[b1, 1] + m.body_invweight0[b2, 1])
    fr = mx['friction']
    # diagApprox: [frictionless/normal, pyramid edge k=1..5]
    diag = [tran] + [tran + fr[k]*fr[k]*(tran if k < 2 else rot)
                     for k in range(5)]
    pair_diag.extend(diag)
    pair_fric.extend(fr)

  out = []
  w = out.append
  w('// GENERATED by dm_control_amd/codegen.py -- do not edit.')
  w('// model %r  hash %s  task %d' % (getattr(m, 'modelname', ''),
                                        m.content_hash(), task))
  w('#pragma once')
  w('#ifdef DMC_REAL_IS_DOUBLE')
  w('typedef double dmc_real;')
  w('#else')
  w('typedef float dmc_real;')
  w('#endif')
  w('#define DMC_UNROLL %s' % ('_Pragma("unroll")' if unroll else ''))
  w('#define DMC_GENERIC_BUILD %d' % (0 if unroll else 1))
  # the static pair list is unrolled only while it stays small; larger models
  # keep a rolled narrowphase loop over a per-lane geom-pose mirror
  import os
  max_unrolled_pairs = int(os.environ.get('DMC_UNROLL_PAIRS_MAX', '40'))
  unroll_pairs = unroll and len(pairs) <= max_unrolled_pairs
  w('#define DMC_UNROLL_PAIRS %s' % ('_Pragma("unroll")' if unroll_pairs else ''))
  w('#define DMC_PAIRS_UNROLLED %d' % (1 if unroll_pairs else 0))
  w('namespace dmc_model {')

  def ci(name, v):
    w('constexpr int %s = %d;' % (name, int(v)))

  def cd(name, v):
    w('constexpr double %s = %r;' % (name, float(v)))

  def ti(name, vals):
    w('static __device__ constexpr int %s[] = {%s};' % (name, _fmt(vals, 'int')))

  def tr(name, vals):
    w('static __device__ constexpr dmc_real %s[] = {%s};'
      % (name, _fmt(vals, 'real')))

  ci('NQ', m.nq); ci('NV', m.nv); ci('NU', m.nu); ci('NBODY', m.nbody)
  ci('NJNT', m.njnt); ci('NGEOM', m.ngeom); ci('NSENSOR', m.nsensor)
  ci('NSENSORDATA', m.nsensordata)
  ci('INTEGRATOR', m.opt.integrator); ci('DISABLEFLAGS', m.opt.disableflags)
  ci('ITERATIONS', m.opt.iterations)
  ci('NPAIR', len(pairs)); ci('NCON_MAX', ncon_max); ci('NEFC_MAX', nefc_max)
  ci('NLIMIT', len(limit_jnt)); ci('TASK', task)
  ci('PLANAR_XZ', 1 if planar_in_xz(m) else 0)
  ci('NOBS', observation_size(m, task))
  ci('NTASKDATA', task_data_size(task))
  tr('task_data_default', task_data_default(m, task) or [0])
  cd('timestep', dt); cd('tolerance_opt', m.opt.tolerance)
  cd('meaninertia', m.meaninertia)
  tr('gravity', m.opt.gravity)
  tr('qpos0', m.qpos0); tr('qpos_spring', m.qpos_spring)
  for name in ('body_parentid', 'body_rootid', 'body_jntnum', 'body_jntadr',
               'body_dofnum', 'body_dofadr', 'jnt_type', 'jnt_qposadr',
               'jnt_dofadr', 'jnt_bodyid', 'jnt_limited', 'dof_bodyid',
               'dof_parentid', 'geom_type', 'geom_bodyid', 'actuator_trnid',
               'actuator_ctrllimited', 'actuator_forcelimited',
               'actuator_biastype', 'sensor_type', 'sensor_objid',
               'sensor_adr'):
    ti(name, getattr(m, name))
  for name in ('body_pos', 'body_quat', 'body_ipos', 'body_iquat',
               'body_mass', 'body_subtreemass', 'body_inertia', 'jnt_pos',
               'jnt_axis', 'jnt_stiffness', 'jnt_range', 'jnt_margin',
               'dof_armature', 'dof_damping', 'dof_invweight0', 'geom_size',
               'geom_pos', 'geom_quat', 'geom_rbound', 'actuator_gear',
               'actuator_ctrlrange', 'actuator_forcerange',
               'actuator_gainprm', 'actuator_biasprm'):
    tr(name, getattr(m, name))
  # actuator transmissions flattened to (dof, qpos address, coefficient) lists:
  # one entry for a joint, one per wrapped joint for a fixed tendon
  w_adr, w_num, w_dof, w_qadr, w_coef = [], [], [], [], []
  for u in range(m.nu):
    w_adr.append(len(w_dof))
    if int(m.actuator_trntype[u]) == mdl.TRN_TENDON:
      t = int(m.actuator_trnid[u])
      items = [(int(m.wrap_objid[k]), float(m.wrap_prm[k]))
               for k in range(int(m.tendon_adr[t]),
                              int(m.tendon_adr[t] + m.tendon_num[t]))]
    else:
      items = [(int(m.actuator_trnid[u]), 1.0)]
    for j, coef in items:
      w_dof.append(int(m.jnt_dofadr[j]))
      w_qadr.append(int(m.jnt_qposadr[j]))
      w_coef.append(coef)
    w_num.append(len(w_dof) - w_adr[-1])
  ci('MAXWRAP', max(w_num + [1]))
  ti('act_wrap_adr', w_adr); ti('act_wrap_num', w_num)
  ti('act_wrap_dof', w_dof); ti('act_wrap_qadr', w_qadr)
  tr('act_wrap_coef', w_coef)
  ti('limit_jnt', limit_jnt)
  tr('limit_K', limit_k); tr('limit_B', limit_b)
  tr('limit_solimp', limit_solimp)
  ti('pair_g1', [p[0] for p in pairs]); ti('pair_g2', [p[1] for p in pairs])
  ti('pair_dim', [mx['dim'] for mx in mixed])
  tr('pair_margin', [mx['margin'] for mx in mixed])
  tr('pair_includemargin', [mx['includemargin'] for mx in mixed])
  tr('pair_friction', pair_fric); tr('pair_K', pair_k); tr('pair_B', pair_b)
  tr('pair_solimp', pair_solimp); tr('pair_diag', pair_diag)
  ti('task_body', (task_bodies(m, task) + [0]*6)[:6])
  sites = task_sites(m, task)
  # touch sensors: the zone is the sensor's site, a sphere or a box
  # (mj_sensorAcc, mjSENS_TOUCH; wrapper/core_test.py:329-344 uses a box site)
  touch = [i for i in range(m.nsensor) if int(m.sensor_type[i]) == SENS_TOUCH]
  touch_type, touch_size, touch_mat = [], [], []
  for i in touch:
    sid = int(m.sensor_objid[i])
    kind = (int(m.site_type[sid]) if getattr(m, 'site_type', None) is not None
            else mdl.GEOM_SPHERE)
    if kind not in (mdl.GEOM_SPHERE, mdl.GEOM_BOX):
      raise UnsupportedModelError('touch sensors need a spherical or box site')
    touch_type.append(kind)
    touch_size += [float(v) for v in m.site_size[sid]]
    qw, qx, qy, qz = [float(v) for v in m.site_quat[sid]]
    touch_mat += [qw*qw + qx*qx - qy*qy - qz*qz, 2*(qx*qy - qw*qz), 2*(qx*qz + qw*qy),
                  2*(qx*qy + qw*qz), qw*qw - qx*qx + qy*qy - qz*qz, 2*(qy*qz - qw*qx),
                  2*(qx*qz - qw*qy), 2*(qy*qz + qw*qx), qw*qw - qx*qx - qy*qy + qz*qz]
  ci('NTOUCH', len(touch))
  ti('touch_adr', [int(m.sensor_adr[i]) for i in touch])
  ti('touch_body', [int(m.site_bodyid[int(m.sensor_objid[i])]) for i in touch])
  ti('touch_type', touch_type)
  tr('touch_pos', [float(v) for i in touch
                   for v in m.site_pos[int(m.sensor_objid[i])]] or [0, 0, 0])
  tr('touch_size', touch_size or [0, 0, 0])      # sphere: radius first
  tr('touch_mat', touch_mat or [1, 0, 0, 0, 1, 0, 0, 0, 1])   # site frame in its body
  ti('task_site_body', [s[0] for s in sites] or [0])
  tr('task_site_pos', [v for s in sites for v in s[1]] or [0, 0, 0])
  tr('task_site_size', [s[2] for s in sites] or [0])
  # static ancestor chains (replace pointer-chasing loops over dof_parentid so
  # that fully unrolled code indexes per-lane arrays with compile-time indices)
  chains = []
  for b in range(m.nbody):
    bb = b
    while bb > 0 and m.body_dofnum[bb] == 0:
      bb = int(m.body_parentid[bb])
    chain = []
    if bb > 0:
      d = int(m.body_dofadr[bb] + m.body_dofnum[bb] - 1)
      while d >= 0:
        chain.append(d)
        d = int(m.dof_parentid[d])
    chains.append(chain)
  ancs = []
  for i in range(m.nv):
    a, d = [], int(m.dof_parentid[i])
    while d >= 0:
      a.append(d)
      d = int(m.dof_parentid[d])
    ancs.append(a)
  maxchain = max([1] + [len(c) for c in chains] + [len(a) for a in ancs])
  # per-body dof masks (bit j set: dof j moves the body), root list
  masks = []
  for c in chains:
    bits = 0
    for d in c:
      bits |= 1 << d
    masks.append(bits)
  roots = sorted(set(int(r) for r in m.body_rootid[1:])) or [0]
  ci('NROOT', len(roots))
  ti('root_body', roots)
  ti('body_rootidx', [roots.index(int(r)) if int(r) in roots else 0
                      for r in m.body_rootid])
  # first dof of the kinematic tree a dof belongs to: rows of M (and of the
  # Newton Hessian while no contact couples two trees) are zero left of it --
  # the envelope that the big-scene kernels confine their matrix loops to
  treeroot = []
  for i in range(m.nv):
    d = i
    while int(m.dof_parentid[d]) >= 0:
      d = int(m.dof_parentid[d])
    treeroot.append(d)
  ti('dof_treeroot', treeroot or [0])
  # the same as dof ranges [lo, hi] per tree (a tree's dofs are consecutive): the
  # diagonal blocks that the team build of a big scene factors one LDS tile at a time
  starts = sorted(set(treeroot))
  ends = [max(i for i, t in enumerate(treeroot) if t == s0) for s0 in starts]
  for s0, e0 in zip(starts, ends):
    assert all(treeroot[i] == s0 for i in range(s0, e0 + 1)), 'tree dofs not consecutive'
  ci('NDTREE', len(starts))
  ti('dtree_lo', starts or [0])
  ti('dtree_hi', ends or [0])
  ci('MAXTREEDOF', max([e0 - s0 + 1 for s0, e0 in zip(starts, ends)] or [0]))
  # [body][word]: bit (j & 31) of word (j >> 5) set <=> dof j moves the body
  # (multi-word, so scenes with several walkers -- nv > 64 -- compile too)
  nmaskw = max(1, (m.nv + 31)//32)
  ci('NMASKW', nmaskw)
  w('static __device__ constexpr unsigned body_dofmask[] = {%s};'
    % ', '.join('%du' % ((b >> (32*k)) & 0xffffffff)
                for b in masks for k in range(nmaskw)))
  # kinematic trees (roots = children of the world): per-step bounding spheres
  # of the trees let big scenes skip whole blocks of walker-walker pairs
  tree_of_body = [-1 if int(r) == 0 else roots.index(int(r)) for r in m.body_rootid]
  tree_of_body[0] = -1
  ci('NTREE', len(roots) if m.nbody > 1 else 0)
  # the trees as ranges of bodies, joints and dofs ([lo, hi), all consecutive): in
  # team builds one lane per tree runs the tree-local recursions
  tb_lo, tb_hi, tj_lo, tj_hi, td_lo, td_hi = [], [], [], [], [], []
  for t in range(len(roots) if m.nbody > 1 else 0):
    bodies = [b for b in range(1, m.nbody) if tree_of_body[b] == t]
    assert bodies == list(range(bodies[0], bodies[-1] + 1)), 'tree bodies not consecutive'
    tb_lo.append(bodies[0]); tb_hi.append(bodies[-1] + 1)
    jn = [int(m.body_jntadr[b]) + k for b in bodies for k in range(int(m.body_jntnum[b]))]
    dn = [int(m.body_dofadr[b]) + k for b in bodies for k in range(int(m.body_dofnum[b]))]
    assert jn == list(range(jn[0], jn[-1] + 1)) if jn else True
    assert dn == list(range(dn[0], dn[-1] + 1)) if dn else True
    tj_lo.append(jn[0] if jn else 0); tj_hi.append(jn[-1] + 1 if jn else 0)
    td_lo.append(dn[0] if dn else 0); td_hi.append(dn[-1] + 1 if dn else 0)
  for name, vals in (('tree_body_lo', tb_lo), ('tree_body_hi', tb_hi), ('tree_jnt_lo', tj_lo),
                     ('tree_jnt_hi', tj_hi), ('tree_dof_lo', td_lo), ('tree_dof_hi', td_hi)):
    ti(name, vals or [0])
  # Segments: maximal runs of consecutive bodies of a tree that form a chain without
  # branches (a leg, an arm, the spine); a segment hangs from a body of an earlier
  # one (its "hub") and the recursions of the segments of one level of a tree are
  # independent -- team builds run them on different lanes.
  seg_lo, seg_hi, seg_tree, seg_level, seg_hub = [], [], [], [], []
  seg_of_body = [-1]*m.nbody
  nchild = [0]*m.nbody
  for b in range(1, m.nbody):
    nchild[int(m.body_parentid[b])] += 1
  for t in range(len(tb_lo)):
    b = tb_lo[t]
    while b < tb_hi[t]:
      start = b
      parent = int(m.body_parentid[b])
      b += 1
      # (the run goes on while a body is the ONLY child of the body before it)
      while (b < tb_hi[t] and int(m.body_parentid[b]) == b - 1
             and nchild[b - 1] == 1):
        b += 1
      sid = len(seg_lo)
      for k in range(start, b):
        seg_of_body[k] = sid
      seg_lo.append(start); seg_hi.append(b); seg_tree.append(t)
      seg_hub.append(parent)
      seg_level.append(0 if parent == 0 else seg_level[seg_of_body[parent]] + 1)
  nlevel = max(seg_level) + 1 if seg_level else 1
  per = {}
  for sid, (t, lv) in enumerate(zip(seg_tree, seg_level)):
    per.setdefault((t, lv), []).append(sid)
  maxper = max([len(v) for v in per.values()] or [1])
  lvl_seg = []
  for t in range(len(tb_lo)):
    for lv in range(nlevel):
      ids = per.get((t, lv), [])
      lvl_seg += ids + [-1]*(maxper - len(ids))
  hubs = sorted(set(h for h in seg_hub if h > 0))
  body_hub = [hubs.index(b) if b in hubs else -1 for b in range(m.nbody)]
  def _jd(lo, hi, adr, num):
    ids = [int(adr[b]) + k for b in range(lo, hi) for k in range(int(num[b]))]
    return (ids[0], ids[-1] + 1) if ids else (0, 0)
  ci('NSEG', len(seg_lo)); ci('NSEGLEVEL', nlevel); ci('MAXSEGPERLEVEL', maxper)
  ci('NHUB', len(hubs))
  ti('seg_body_lo', seg_lo or [0]); ti('seg_body_hi', seg_hi or [0])
  ti('seg_jnt_lo', [_jd(a, b, m.body_jntadr, m.body_jntnum)[0] for a, b in zip(seg_lo, seg_hi)] or [0])
  ti('seg_jnt_hi', [_jd(a, b, m.body_jntadr, m.body_jntnum)[1] for a, b in zip(seg_lo, seg_hi)] or [0])
  ti('seg_dof_lo', [_jd(a, b, m.body_dofadr, m.body_dofnum)[0] for a, b in zip(seg_lo, seg_hi)] or [0])
  ti('seg_dof_hi', [_jd(a, b, m.body_dofadr, m.body_dofnum)[1] for a, b in zip(seg_lo, seg_hi)] or [0])
  ti('seg_hub_body', seg_hub or [0])
  ti('lvl_seg', lvl_seg or [-1])
  ti('body_hub', body_hub or [-1])
  # ... and of actuators (by the dof an actuator's transmission starts at); -1 / -1:
  # the model's actuators are not grouped by tree (then every tree's lane scans all)
  ta_lo, ta_hi, grouped = [], [], True
  first_dof = [int(w_dof[w_adr[i]]) if w_num[i] else -1 for i in range(m.nu)]
  for t in range(len(td_lo)):
    acts = [i for i in range(m.nu) if td_lo[t] <= first_dof[i] < td_hi[t]]
    if acts and acts != list(range(acts[0], acts[-1] + 1)):
      grouped = False
    ta_lo.append(acts[0] if acts else 0); ta_hi.append(acts[-1] + 1 if acts else 0)
  if not grouped or any(d < 0 for d in first_dof):
    ta_lo, ta_hi = [-1]*len(td_lo), [-1]*len(td_lo)
  ti('tree_act_lo', ta_lo or [0])
  ti('tree_act_hi', ta_hi or [0])
  ti('geom_tree', [tree_of_body[int(b)] for b in m.geom_bodyid] or [-1])
  ti('pair_tree1', [tree_of_body[int(m.geom_bodyid[p[0]])] for p in pairs] or [-1])
  ti('pair_tree2', [tree_of_body[int(m.geom_bodyid[p[1]])] for p in pairs] or [-1])
  # length of the run of consecutive pairs, from p on, between the same two
  # (different) trees: a far-apart pair of walkers skips the whole run at once
  keys = [(pair_run_key(m, tree_of_body, pr), float(mx['margin']))
          for pr, mx in zip(pairs, mixed)]
  run = [1]*len(pairs)
  for i in range(len(pairs) - 2, -1, -1):
    if keys[i] == keys[i + 1] and keys[i][0] is not None:
      run[i] = run[i + 1] + 1
  ti('pair_run', run or [1])
  # the list as runs: (first pair, length, keyed) -- keyed runs can be skipped by one
  # bounding test; the pairs in between (inside a tree, world against world) form
  # runs of their own.  Team builds test the runs one per lane before they walk them.
  run_first, run_len, run_keyed, p = [], [], [], 0
  while p < len(pairs):
    if keys[p][0] is not None:
      n = run[p]
      run_first.append(p); run_len.append(n); run_keyed.append(1)
    else:
      n = 1
      while p + n < len(pairs) and keys[p + n][0] is None:
        n += 1
      run_first.append(p); run_len.append(n); run_keyed.append(0)
    p += n
  ci('NRUN', len(run_first))
  ti('run_first', run_first or [0])
  ti('run_len', run_len or [0])
  ti('run_keyed', run_keyed or [0])
  # the world geom of a world-vs-tree pair (-1: none): its bound against the tree's
  ti('pair_wgeom', [k[0][1] if k[0] is not None and k[0][0] == 'wt' else -1 for k in keys] or [-1])
  ti('pair_b1', [int(m.geom_bodyid[p[0]]) for p in pairs])
  ti('pair_b2', [int(m.geom_bodyid[p[1]]) for p in pairs])
  ci('MAXCHAIN', maxchain)
  ti('body_chain_len', [len(c) for c in chains])
  ti('body_chain', [v for c in chains for v in (c + [0]*maxchain)[:maxchain]])
  ti('dof_anc_len', [len(a) for a in ancs])
  ti('dof_anc', [v for a in ancs for v in (a + [0]*maxchain)[:maxchain]])
  # last dof of the chain that moves the body (-1: welded to the world)
  ti('body_lastdof', [c[0] if c else -1 for c in chains])
  # tables of the several-lanes-per-env kernel (csrc/dmc_coop.hip): bodies
  # grouped by tree depth (one lane per body, one pass per level), subtree sizes
  # (bodies are in depth-first order, so a subtree is a contiguous index range
  # and "accumulate into the parent" becomes a race-free range sum), the joint
  # of every dof, contact rows per pair
  level = [0]*m.nbody
  for b in range(1, m.nbody):
    level[b] = level[int(m.body_parentid[b])] + 1
  nlevel = max(level) if m.nbody > 1 else 0
  order = sorted(range(1, m.nbody), key=lambda b: (level[b], b))
  level_adr = [sum(1 for b in order if level[b] <= lv) for lv in range(nlevel + 1)]
  subtree_n = [1]*m.nbody
  for b in range(m.nbody - 1, 0, -1):
    subtree_n[int(m.body_parentid[b])] += subtree_n[b]
  for b in range(1, m.nbody):
    p = int(m.body_parentid[b])
    if not p <= b < p + subtree_n[p] or p >= b:
      raise UnsupportedModelError('bodies are not in depth-first order')
  dof_jntid = [0]*m.nv
  for j in range(m.njnt):
    width = {mdl.JNT_FREE: 6, mdl.JNT_BALL: 3}.get(int(m.jnt_type[j]), 1)
    for k in range(width):
      dof_jntid[int(m.jnt_dofadr[j]) + k] = j
  ci('NLEVEL', nlevel)
  ti('level_adr', level_adr)
  ti('level_body', order)
  ti('body_subtree_n', subtree_n)
  ti('dof_jntid', dof_jntid)
  ti('pair_nrow', [1 if mx['dim'] == 1 else 2*(mx['dim'] - 1) for mx in mixed])
  w('}  // namespace dmc_model')
  return '\n'.join(out) + '\n'


def model_info(m, task=TASK_NONE, ncon_max=None):
  """Sizes the host needs without loading the code object."""
  pairs = collision_pairs(m)
  ncon_max, nefc_max = capacities(m, pairs, ncon_max)
  return dict(nq=m.nq, nv=m.nv, nu=m.nu, nbody=m.nbody,
              nobs=observation_size(m, task), nsensordata=m.nsensordata,
              ncon_max=ncon_max, nefc_max=nefc_max, npair=len(pairs),
              ws_per_env=nefc_max*(m.nv + 4))
