/* oracle/mjstep.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * fp64 CPU restatement of the physics step that the reference delegates to
 * libmujoco (MuJoCo Pro 2.00, pinned at
 * /root/reference/dm_control/mujoco/wrapper/util.py:45).  Call sites restated:
 *   Physics.step        /root/reference/dm_control/mujoco/engine.py:149-166
 *                       (Euler: mj_step2 then mj_step1; RK4: mj_step then mj_step1)
 *   Physics.reset       engine.py:268-289   (mj_resetData + mj_forward, actuation off)
 *   Physics.after_reset engine.py:291-295   (mj_forward, actuation off)
 *
 * libmujoco 2.00 is a closed third-party binary that is absent from the
 * reference tree and from this image, so the algorithm below is restated from
 * MuJoCo's published documentation (pipeline order, soft-constraint model,
 * Newton solver, collision primitives: SURVEY.md Appendix A) and is pinned by
 * the reference's own known-answer tests (see
 * tests/test_oracle_kat.py).  No golden qpos/qvel trajectory exists in the
 * reference, so trajectory-level parity is "parity unpinned" beyond those KATs.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file.  The product path (dm_control_amd/) never does.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MINVAL 1e-15
#define MAXVAL 1e10
#define MINIMP 1e-4
#define MAXIMP 0.9999

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_BOX = 6 };
enum { INT_EULER = 0, INT_RK4 = 1 };
enum { BIAS_NONE = 0, BIAS_AFFINE = 1 };
enum {
  DSBL_CONSTRAINT = 1 << 0, DSBL_LIMIT = 1 << 3, DSBL_CONTACT = 1 << 4,
  DSBL_PASSIVE = 1 << 5, DSBL_GRAVITY = 1 << 6, DSBL_CLAMPCTRL = 1 << 7,
  DSBL_WARMSTART = 1 << 8, DSBL_FILTERPARENT = 1 << 9,
  DSBL_ACTUATION = 1 << 10, DSBL_REFSAFE = 1 << 11
};
enum { WARN_INERTIA, WARN_CONTACTFULL, WARN_CNSTRFULL, WARN_VGEOMFULL,
       WARN_BADQPOS, WARN_BADQVEL, WARN_BADQACC, WARN_BADCTRL, NWARN };
enum { SENS_JOINTPOS = 8, SENS_JOINTVEL = 9, SENS_SUBTREECOM = 34,
       SENS_SUBTREELINVEL = 35 };
enum { TRN_JOINT = 0, TRN_TENDON = 3 };
enum { EFC_LIMIT = 0, EFC_CONTACT_FRICTIONLESS = 1, EFC_CONTACT_PYRAMIDAL = 2 };

/* field tables: identical names to dm_control_amd/mjcf/model.py:FIELDS */
#define INT_FIELDS(X) \
  X(nq) X(nv) X(nu) X(nbody) X(njnt) X(ngeom) X(nsensor) X(nsensordata) \
  X(nexclude) X(ntendon) X(nwrap) X(integrator) X(cone) X(solver) \
  X(iterations) X(disableflags) X(enableflags)
#define DBL_FIELDS(X) X(timestep) X(tolerance) X(impratio) X(meaninertia)
#define IARR_FIELDS(X) \
  X(body_parentid) X(body_rootid) X(body_weldid) X(body_jntnum) \
  X(body_jntadr) X(body_dofnum) X(body_dofadr) X(body_geomnum) \
  X(body_geomadr) X(jnt_type) X(jnt_qposadr) X(jnt_dofadr) X(jnt_bodyid) \
  X(jnt_limited) X(dof_bodyid) X(dof_jntid) X(dof_parentid) X(geom_type) \
  X(geom_contype) X(geom_conaffinity) X(geom_condim) X(geom_bodyid) \
  X(geom_priority) X(actuator_trntype) X(actuator_trnid) \
  X(actuator_ctrllimited) X(actuator_forcelimited) X(actuator_gaintype) \
  X(actuator_biastype) X(tendon_adr) X(tendon_num) X(wrap_objid) \
  X(sensor_type) X(sensor_objid) X(sensor_adr) \
  X(sensor_dim) X(exclude_signature)
#define DARR_FIELDS(X) \
  X(gravity) X(qpos0) X(qpos_spring) X(body_pos) X(body_quat) X(body_ipos) \
  X(body_iquat) X(body_mass) X(body_subtreemass) X(body_inertia) \
  X(body_invweight0) X(jnt_pos) X(jnt_axis) X(jnt_stiffness) X(jnt_range) \
  X(jnt_margin) X(jnt_solref) X(jnt_solimp) X(dof_armature) X(dof_damping) \
  X(dof_invweight0) X(geom_size) X(geom_pos) X(geom_quat) X(geom_friction) \
  X(geom_solmix) X(geom_solref) X(geom_solimp) X(geom_margin) X(geom_gap) \
  X(geom_rbound) X(actuator_gear) X(actuator_ctrlrange) \
  X(actuator_forcerange) X(actuator_gainprm) X(actuator_biasprm) X(wrap_prm)

typedef struct mjoModel {
#define X(n) int n;
  INT_FIELDS(X)
#undef X
#define X(n) double n;
  DBL_FIELDS(X)
#undef X
#define X(n) int* n;
  IARR_FIELDS(X)
#undef X
#define X(n) double* n;
  DARR_FIELDS(X)
#undef X
  int nconmax, nefcmax;
} mjoModel;

typedef struct mjoContact {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2],
      solimp[5];
  int dim, geom1, geom2, efc_address;
} mjoContact;

/* per-instance state; DATA_FIELDS lists every named buffer python may view */
#define DATA_FIELDS(X) \
  X(qpos, m->nq) X(qvel, m->nv) X(ctrl, m->nu) X(qacc, m->nv) \
  X(qacc_warmstart, m->nv) X(qfrc_applied, m->nv) \
  X(xpos, 3*m->nbody) X(xquat, 4*m->nbody) X(xmat, 9*m->nbody) \
  X(xipos, 3*m->nbody) X(ximat, 9*m->nbody) X(xanchor, 3*m->njnt) \
  X(xaxis, 3*m->njnt) X(geom_xpos, 3*m->ngeom) X(geom_xmat, 9*m->ngeom) \
  X(subtree_com, 3*m->nbody) X(cinert, 10*m->nbody) X(crb, 10*m->nbody) \
  X(cdof, 6*m->nv) X(cdof_dot, 6*m->nv) X(cvel, 6*m->nbody) \
  X(cacc, 6*m->nbody) X(cfrc, 6*m->nbody) X(qM, m->nv*m->nv) \
  X(qL, m->nv*m->nv) X(qfrc_bias, m->nv) X(qfrc_passive, m->nv) \
  X(qfrc_actuator, m->nv) X(qfrc_smooth, m->nv) X(qacc_smooth, m->nv) \
  X(qfrc_constraint, m->nv) X(actuator_force, m->nu) \
  X(subtree_linvel, 3*m->nbody) X(sensordata, m->nsensordata) \
  X(efc_J, m->nefcmax*m->nv) X(efc_pos, m->nefcmax) \
  X(efc_margin, m->nefcmax) X(efc_diagApprox, m->nefcmax) \
  X(efc_R, m->nefcmax) X(efc_D, m->nefcmax) X(efc_aref, m->nefcmax) \
  X(efc_vel, m->nefcmax) X(efc_force, m->nefcmax) X(efc_b, m->nefcmax)

typedef struct mjoData {
  const mjoModel* model;
  double time;
  int ncon, nefc, solver_iter;
  int warning[NWARN];
#define X(n, sz) double* n;
  DATA_FIELDS(X)
#undef X
  int* efc_type;
  int* efc_id;
  double* efc_solref; /* 2 per row */
  double* efc_solimp; /* 5 per row */
  mjoContact* contact;
  double* scratch; /* solver workspace */
} mjoData;

/* ------------------------------------------------------------------ */
/* small vector / quaternion helpers                                   */
/* ------------------------------------------------------------------ */
static double dot3(const double* a, const double* b) {
  return a[0]*b[0] + a[1]*b[1] + a[2]*b[2];
}
static void cross3(double* r, const double* a, const double* b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2],
         z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static double norm3(const double* a) { return sqrt(dot3(a, a)); }
static double normalize3(double* a) {
  double n = norm3(a);
  if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
  else { a[0] /= n; a[1] /= n; a[2] /= n; }
  return n;
}
static void normalize4(double* q) {
  double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n; }
}
static void mulquat(double* r, const double* a, const double* b) {
  double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static void quat2mat(double* m, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w*w + x*x - y*y - z*z; m[1] = 2*(x*y - w*z); m[2] = 2*(x*z + w*y);
  m[3] = 2*(x*y + w*z); m[4] = w*w - x*x + y*y - z*z; m[5] = 2*(y*z - w*x);
  m[6] = 2*(x*z - w*y); m[7] = 2*(y*z + w*x); m[8] = w*w - x*x - y*y + z*z;
}
/* r = R(q) v */
static void rotvecquat(double* r, const double* v, const double* q) {
  double m[9];
  quat2mat(m, q);
  double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulmatvec3(double* r, const double* m, const double* v) {
  double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  double s = sin(angle*0.5);
  q[0] = cos(angle*0.5); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
/* q <- q * exp(h*w/2), w in the local frame (free/ball joints) */
static void quat_integrate(double* q, const double* w, double h) {
  double ax[3] = {w[0], w[1], w[2]};
  double n = normalize3(ax);
  double angle = h*n;
  double dq[4], r[4];
  if (n < MINVAL) return;
  axisangle2quat(dq, ax, angle);
  mulquat(r, q, dq);
  normalize4(r);
  memcpy(q, r, sizeof r);
}

/* spatial vectors are [angular(3), linear(3)] about the subtree-root CoM */
static void cross_motion(double* r, const double* v, const double* s) {
  double a[3], b[3], c[3];
  cross3(a, v, s);         /* w x s_ang */
  cross3(b, v, s + 3);     /* w x s_lin */
  cross3(c, v + 3, s);     /* v x s_ang */
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
static void cross_force(double* r, const double* v, const double* f) {
  double a[3], b[3], c[3];
  cross3(a, v, f);         /* w x f_ang */
  cross3(b, v + 3, f + 3); /* v x f_lin */
  cross3(c, v, f + 3);     /* w x f_lin */
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
/* 10-number spatial inertia: [Ixx Iyy Izz Ixy Ixz Iyz, m*dx m*dy m*dz, m] */
static void inert_com(double* res, const double* inertia, const double* mat,
                      const double* dif, double mass) {
  double t[9];
  int i, j;
  for (i = 0; i < 3; i++)
    for (j = 0; j < 3; j++)
      t[3*i + j] = mat[3*i]*inertia[0]*mat[3*j] +
                   mat[3*i + 1]*inertia[1]*mat[3*j + 1] +
                   mat[3*i + 2]*inertia[2]*mat[3*j + 2];
  res[0] = t[0] + mass*(dif[1]*dif[1] + dif[2]*dif[2]);
  res[1] = t[4] + mass*(dif[0]*dif[0] + dif[2]*dif[2]);
  res[2] = t[8] + mass*(dif[0]*dif[0] + dif[1]*dif[1]);
  res[3] = t[1] - mass*dif[0]*dif[1];
  res[4] = t[2] - mass*dif[0]*dif[2];
  res[5] = t[5] - mass*dif[1]*dif[2];
  res[6] = mass*dif[0]; res[7] = mass*dif[1]; res[8] = mass*dif[2];
  res[9] = mass;
}
static void mul_inert_vec(double* r, const double* i, const double* v) {
  r[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  r[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  r[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  r[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  r[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  r[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
static double dotn(const double* a, const double* b, int n) {
  double s = 0;
  int i;
  for (i = 0; i < n; i++) s += a[i]*b[i];
  return s;
}

/* ------------------------------------------------------------------ */
/* model / data lifetime                                                */
/* ------------------------------------------------------------------ */
mjoModel* mjo_model_new(void) {
  mjoModel* m = (mjoModel*)calloc(1, sizeof(mjoModel));
  m->nconmax = 128;
  m->nefcmax = 600;
  return m;
}
void mjo_model_free(mjoModel* m) {
  if (!m) return;
#define X(n) free(m->n);
  IARR_FIELDS(X)
  DARR_FIELDS(X)
#undef X
  free(m);
}
int mjo_model_set_int(mjoModel* m, const char* name, int v) {
#define X(n) if (!strcmp(name, #n)) { m->n = v; return 0; }
  INT_FIELDS(X)
#undef X
  if (!strcmp(name, "nconmax")) { m->nconmax = v; return 0; }
  if (!strcmp(name, "nefcmax")) { m->nefcmax = v; return 0; }
  return -1;
}
int mjo_model_get_int(const mjoModel* m, const char* name) {
#define X(n) if (!strcmp(name, #n)) return m->n;
  INT_FIELDS(X)
#undef X
  return -1;
}
int mjo_model_set_double(mjoModel* m, const char* name, double v) {
#define X(n) if (!strcmp(name, #n)) { m->n = v; return 0; }
  DBL_FIELDS(X)
#undef X
  return -1;
}
int mjo_model_set_iarr(mjoModel* m, const char* name, const int* v, int n) {
#define X(f) if (!strcmp(name, #f)) { free(m->f); \
    m->f = (int*)malloc(sizeof(int)*(n > 0 ? n : 1)); \
    memcpy(m->f, v, sizeof(int)*n); return 0; }
  IARR_FIELDS(X)
#undef X
  return -1;
}
int mjo_model_set_darr(mjoModel* m, const char* name, const double* v, int n) {
#define X(f) if (!strcmp(name, #f)) { free(m->f); \
    m->f = (double*)malloc(sizeof(double)*(n > 0 ? n : 1)); \
    memcpy(m->f, v, sizeof(double)*n); return 0; }
  DARR_FIELDS(X)
#undef X
  return -1;
}
double* mjo_model_darr(mjoModel* m, const char* name) {
#define X(f) if (!strcmp(name, #f)) return m->f;
  DARR_FIELDS(X)
#undef X
  return NULL;
}
int* mjo_model_iarr(mjoModel* m, const char* name) {
#define X(f) if (!strcmp(name, #f)) return m->f;
  IARR_FIELDS(X)
#undef X
  return NULL;
}

void mjo_reset_data(const mjoModel* m, mjoData* d);

mjoData* mjo_data_new(const mjoModel* m) {
  mjoData* d = (mjoData*)calloc(1, sizeof(mjoData));
  int nv = m->nv;
  d->model = m;
#define X(n, sz) d->n = (double*)calloc((size_t)((sz) > 0 ? (sz) : 1), sizeof(double));
  DATA_FIELDS(X)
#undef X
  d->efc_type = (int*)calloc(m->nefcmax, sizeof(int));
  d->efc_id = (int*)calloc(m->nefcmax, sizeof(int));
  d->efc_solref = (double*)calloc(2*m->nefcmax, sizeof(double));
  d->efc_solimp = (double*)calloc(5*m->nefcmax, sizeof(double));
  d->contact = (mjoContact*)calloc(m->nconmax, sizeof(mjoContact));
  d->scratch = (double*)calloc((size_t)(2*nv*nv + 16*nv + 4*m->nefcmax + 64),
                               sizeof(double));
  mjo_reset_data(m, d);
  return d;
}
void mjo_data_free(mjoData* d) {
  if (!d) return;
#define X(n, sz) free(d->n);
  DATA_FIELDS(X)
#undef X
  free(d->efc_type); free(d->efc_id); free(d->efc_solref);
  free(d->efc_solimp); free(d->contact); free(d->scratch);
  free(d);
}
double* mjo_data_ptr(mjoData* d, const char* name, int* size) {
  const mjoModel* m = d->model;
#define X(n, sz) if (!strcmp(name, #n)) { if (size) *size = (sz); return d->n; }
  DATA_FIELDS(X)
#undef X
  if (!strcmp(name, "time")) { if (size) *size = 1; return &d->time; }
  return NULL;
}
int* mjo_data_warnings(mjoData* d) { return d->warning; }
int mjo_data_ncon(const mjoData* d) { return d->ncon; }
int mjo_data_nefc(const mjoData* d) { return d->nefc; }
int mjo_data_solver_iter(const mjoData* d) { return d->solver_iter; }
void mjo_data_set_time(mjoData* d, double t) { d->time = t; }
double mjo_data_time(const mjoData* d) { return d->time; }

void mjo_copy_data(mjoData* dst, const mjoData* src) {
  const mjoModel* m = src->model;
  dst->time = src->time;
  dst->ncon = src->ncon; dst->nefc = src->nefc;
  dst->solver_iter = src->solver_iter;
  memcpy(dst->warning, src->warning, sizeof src->warning);
#define X(n, sz) memcpy(dst->n, src->n, sizeof(double)*(size_t)(sz));
  DATA_FIELDS(X)
#undef X
  memcpy(dst->efc_type, src->efc_type, sizeof(int)*m->nefcmax);
  memcpy(dst->efc_id, src->efc_id, sizeof(int)*m->nefcmax);
  memcpy(dst->efc_solref, src->efc_solref, sizeof(double)*2*m->nefcmax);
  memcpy(dst->efc_solimp, src->efc_solimp, sizeof(double)*5*m->nefcmax);
  memcpy(dst->contact, src->contact, sizeof(mjoContact)*m->nconmax);
}

/* mj_resetData: qpos <- qpos0, everything else zero (warnings are kept by the
 * caller when the reset is triggered by a bad state). */
void mjo_reset_data(const mjoModel* m, mjoData* d) {
  int keep[NWARN];
  memcpy(keep, d->warning, sizeof keep);
#define X(n, sz) memset(d->n, 0, sizeof(double)*(size_t)((sz) > 0 ? (sz) : 1));
  DATA_FIELDS(X)
#undef X
  memcpy(d->qpos, m->qpos0, sizeof(double)*m->nq);
  d->time = 0;
  d->ncon = d->nefc = d->solver_iter = 0;
  memcpy(d->warning, keep, sizeof keep);
}
void mjo_clear_warnings(mjoData* d) { memset(d->warning, 0, sizeof d->warning); }

/* ------------------------------------------------------------------ */
/* position stage                                                       */
/* ------------------------------------------------------------------ */
static void mjo_kinematics(const mjoModel* m, mjoData* d) {
  int i, j;
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  quat2mat(d->xmat, d->xquat);
  memcpy(d->xipos, d->xpos, 3*sizeof(double));
  memcpy(d->ximat, d->xmat, 9*sizeof(double));
  for (i = 1; i < m->nbody; i++) {
    double xpos[3], xquat[4];
    int jadr = m->body_jntadr[i], jnum = m->body_jntnum[i];
    if (jnum == 1 && m->jnt_type[jadr] == JNT_FREE) {
      int qa = m->jnt_qposadr[jadr];
      memcpy(xpos, d->qpos + qa, 3*sizeof(double));
      memcpy(xquat, d->qpos + qa + 3, 4*sizeof(double));
      normalize4(xquat);
      memcpy(d->xanchor + 3*jadr, xpos, 3*sizeof(double));
      memcpy(d->xaxis + 3*jadr, m->jnt_axis + 3*jadr, 3*sizeof(double));
    } else {
      int pid = m->body_parentid[i];
      double v[3];
      mulmatvec3(v, d->xmat + 9*pid, m->body_pos + 3*i);
      xpos[0] = d->xpos[3*pid] + v[0];
      xpos[1] = d->xpos[3*pid + 1] + v[1];
      xpos[2] = d->xpos[3*pid + 2] + v[2];
      mulquat(xquat, d->xquat + 4*pid, m->body_quat + 4*i);
      for (j = 0; j < jnum; j++) {
        int jid = jadr + j, qa = m->jnt_qposadr[jid];
        double* anchor = d->xanchor + 3*jid;
        double* axis = d->xaxis + 3*jid;
        rotvecquat(axis, m->jnt_axis + 3*jid, xquat);
        rotvecquat(anchor, m->jnt_pos + 3*jid, xquat);
        anchor[0] += xpos[0]; anchor[1] += xpos[1]; anchor[2] += xpos[2];
        if (m->jnt_type[jid] == JNT_SLIDE) {
          double q = d->qpos[qa] - m->qpos0[qa];
          xpos[0] += axis[0]*q; xpos[1] += axis[1]*q; xpos[2] += axis[2]*q;
        } else if (m->jnt_type[jid] == JNT_HINGE ||
                   m->jnt_type[jid] == JNT_BALL) {
          double qloc[4], r[4], vec[3];
          if (m->jnt_type[jid] == JNT_BALL) {
            memcpy(qloc, d->qpos + qa, 4*sizeof(double));
            normalize4(qloc);
          } else {
            axisangle2quat(qloc, m->jnt_axis + 3*jid,
                           d->qpos[qa] - m->qpos0[qa]);
          }
          mulquat(r, xquat, qloc);
          memcpy(xquat, r, sizeof r);
          rotvecquat(vec, m->jnt_pos + 3*jid, xquat);
          xpos[0] = anchor[0] - vec[0];
          xpos[1] = anchor[1] - vec[1];
          xpos[2] = anchor[2] - vec[2];
        }
      }
    }
    normalize4(xquat);
    memcpy(d->xpos + 3*i, xpos, sizeof xpos);
    memcpy(d->xquat + 4*i, xquat, sizeof xquat);
    quat2mat(d->xmat + 9*i, xquat);
  }
  for (i = 1; i < m->nbody; i++) {
    double v[3], q[4];
    mulmatvec3(v, d->xmat + 9*i, m->body_ipos + 3*i);
    d->xipos[3*i] = d->xpos[3*i] + v[0];
    d->xipos[3*i + 1] = d->xpos[3*i + 1] + v[1];
    d->xipos[3*i + 2] = d->xpos[3*i + 2] + v[2];
    mulquat(q, d->xquat + 4*i, m->body_iquat + 4*i);
    quat2mat(d->ximat + 9*i, q);
  }
  for (i = 0; i < m->ngeom; i++) {
    int b = m->geom_bodyid[i];
    double v[3], q[4];
    mulmatvec3(v, d->xmat + 9*b, m->geom_pos + 3*i);
    d->geom_xpos[3*i] = d->xpos[3*b] + v[0];
    d->geom_xpos[3*i + 1] = d->xpos[3*b + 1] + v[1];
    d->geom_xpos[3*i + 2] = d->xpos[3*b + 2] + v[2];
    mulquat(q, d->xquat + 4*b, m->geom_quat + 4*i);
    normalize4(q);
    quat2mat(d->geom_xmat + 9*i, q);
  }
}

static void mjo_com_pos(const mjoModel* m, mjoData* d) {
  int i, j, k;
  for (i = 0; i < m->nbody; i++)
    for (k = 0; k < 3; k++)
      d->subtree_com[3*i + k] = m->body_mass[i]*d->xipos[3*i + k];
  for (i = m->nbody - 1; i > 0; i--)
    for (k = 0; k < 3; k++)
      d->subtree_com[3*m->body_parentid[i] + k] += d->subtree_com[3*i + k];
  for (i = 0; i < m->nbody; i++) {
    if (m->body_subtreemass[i] < MINVAL)
      memcpy(d->subtree_com + 3*i, d->xipos + 3*i, 3*sizeof(double));
    else
      for (k = 0; k < 3; k++) d->subtree_com[3*i + k] /= m->body_subtreemass[i];
  }
  memset(d->cinert, 0, 10*sizeof(double));
  for (i = 1; i < m->nbody; i++) {
    double off[3];
    const double* com = d->subtree_com + 3*m->body_rootid[i];
    for (k = 0; k < 3; k++) off[k] = d->xipos[3*i + k] - com[k];
    inert_com(d->cinert + 10*i, m->body_inertia + 3*i, d->ximat + 9*i, off,
              m->body_mass[i]);
  }
  for (j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const double* com = d->subtree_com + 3*m->body_rootid[b];
    double off[3];
    double* cd = d->cdof + 6*da;
    for (k = 0; k < 3; k++) off[k] = com[k] - d->xanchor[3*j + k];
    switch (m->jnt_type[j]) {
      case JNT_FREE:
        memset(cd, 0, 18*sizeof(double));
        cd[3] = 1; cd[6 + 4] = 1; cd[12 + 5] = 1;
        cd += 18;
        /* fall through: rotational part like a ball joint */
      case JNT_BALL:
        for (k = 0; k < 3; k++) {
          double ax[3] = {d->xmat[9*b + k], d->xmat[9*b + 3 + k],
                          d->xmat[9*b + 6 + k]};
          memcpy(cd + 6*k, ax, sizeof ax);
          cross3(cd + 6*k + 3, ax, off);
        }
        break;
      case JNT_SLIDE:
        cd[0] = cd[1] = cd[2] = 0;
        memcpy(cd + 3, d->xaxis + 3*j, 3*sizeof(double));
        break;
      default: /* hinge */
        memcpy(cd, d->xaxis + 3*j, 3*sizeof(double));
        cross3(cd + 3, d->xaxis + 3*j, off);
    }
  }
}

/* composite rigid body -> dense symmetric M (+ armature) */
static void mjo_crb(const mjoModel* m, mjoData* d) {
  int i, j, k, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(double)*10*m->nbody);
  for (i = m->nbody - 1; i > 0; i--)
    if (m->body_parentid[i] > 0)
      for (k = 0; k < 10; k++)
        d->crb[10*m->body_parentid[i] + k] += d->crb[10*i + k];
  memset(d->qM, 0, sizeof(double)*nv*nv);
  for (i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10*m->dof_bodyid[i], d->cdof + 6*i);
    d->qM[i*nv + i] = dotn(d->cdof + 6*i, buf, 6) + m->dof_armature[i];
    for (j = m->dof_parentid[i]; j >= 0; j = m->dof_parentid[j]) {
      double v = dotn(d->cdof + 6*j, buf, 6);
      d->qM[i*nv + j] = v;
      d->qM[j*nv + i] = v;
    }
  }
}

/* dense Cholesky A = L L^T (lower); returns rank deficiency count */
static int chol_factor(double* L, const double* A, int n) {
  int i, j, k, bad = 0;
  memcpy(L, A, sizeof(double)*n*n);
  for (j = 0; j < n; j++) {
    double s = L[j*n + j];
    for (k = 0; k < j; k++) s -= L[j*n + k]*L[j*n + k];
    if (s < MINVAL) { s = MINVAL; bad++; }
    s = sqrt(s);
    L[j*n + j] = s;
    for (i = j + 1; i < n; i++) {
      double t = L[i*n + j];
      for (k = 0; k < j; k++) t -= L[i*n + k]*L[j*n + k];
      L[i*n + j] = t/s;
    }
  }
  return bad;
}
static void chol_solve(double* x, const double* L, const double* b, int n) {
  int i, k;
  if (x != b) memcpy(x, b, sizeof(double)*n);
  for (i = 0; i < n; i++) {
    for (k = 0; k < i; k++) x[i] -= L[i*n + k]*x[k];
    x[i] /= L[i*n + i];
  }
  for (i = n - 1; i >= 0; i--) {
    for (k = i + 1; k < n; k++) x[i] -= L[k*n + i]*x[k];
    x[i] /= L[i*n + i];
  }
}

/* translational / rotational Jacobian of a world point attached to `body` */
static void mjo_jac(const mjoModel* m, const mjoData* d, double* jacp,
                    double* jacr, const double* point, int body) {
  int nv = m->nv, i, k;
  double off[3];
  if (jacp) memset(jacp, 0, sizeof(double)*3*nv);
  if (jacr) memset(jacr, 0, sizeof(double)*3*nv);
  for (k = 0; k < 3; k++)
    off[k] = point[k] - d->subtree_com[3*m->body_rootid[body] + k];
  while (body && !m->body_dofnum[body]) body = m->body_parentid[body];
  if (!body) return;
  i = m->body_dofadr[body] + m->body_dofnum[body] - 1;
  for (; i >= 0; i = m->dof_parentid[i]) {
    const double* cd = d->cdof + 6*i;
    if (jacr) for (k = 0; k < 3; k++) jacr[k*nv + i] = cd[k];
    if (jacp) {
      double t[3];
      cross3(t, cd, off);
      for (k = 0; k < 3; k++) jacp[k*nv + i] = cd[3 + k] + t[k];
    }
  }
}

/* ---- collision ---------------------------------------------------- */
static void make_frame(double* frame) {
  double t;
  normalize3(frame);
  if (norm3(frame + 3) < 0.5) {
    frame[3] = frame[4] = frame[5] = 0;
    if (frame[1] < 0.5 && frame[1] > -0.5) frame[4] = 1; else frame[5] = 1;
  }
  t = dot3(frame, frame + 3);
  frame[3] -= t*frame[0]; frame[4] -= t*frame[1]; frame[5] -= t*frame[2];
  normalize3(frame + 3);
  cross3(frame + 6, frame, frame + 3);
}

typedef struct { double dist, pos[3], frame[9]; } RawCon;

static int plane_sphere(RawCon* c, double margin, const double* ppos,
                        const double* pmat, const double* spos, double r) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3], dist;
  int k;
  for (k = 0; k < 3; k++) dif[k] = spos[k] - ppos[k];
  dist = dot3(dif, n) - r;
  if (dist > margin) return 0;
  c->dist = dist;
  for (k = 0; k < 3; k++) c->pos[k] = spos[k] - n[k]*(r + 0.5*dist);
  memset(c->frame, 0, sizeof c->frame);
  memcpy(c->frame, n, sizeof n);
  return 1;
}
static int sphere_sphere(RawCon* c, double margin, const double* p1,
                         const double* p2, double r1, double r2) {
  double dif[3], len, dist;
  int k;
  for (k = 0; k < 3; k++) dif[k] = p2[k] - p1[k];
  len = norm3(dif);
  dist = len - r1 - r2;
  if (dist > margin) return 0;
  c->dist = dist;
  memset(c->frame, 0, sizeof c->frame);
  if (len < MINVAL) { c->frame[0] = 1; }
  else for (k = 0; k < 3; k++) c->frame[k] = dif[k]/len;
  for (k = 0; k < 3; k++) c->pos[k] = p1[k] + c->frame[k]*(r1 + 0.5*dist);
  return 1;
}
static int plane_capsule(RawCon* c, double margin, const double* ppos,
                         const double* pmat, const double* cpos,
                         const double* cmat, const double* size) {
  double axis[3] = {cmat[2], cmat[5], cmat[8]}, p[3];
  int k, n = 0, n1;
  for (k = 0; k < 3; k++) p[k] = cpos[k] + axis[k]*size[1];
  n1 = plane_sphere(c, margin, ppos, pmat, p, size[0]);
  if (n1) memcpy(c[0].frame + 3, axis, sizeof axis);
  n += n1;
  for (k = 0; k < 3; k++) p[k] = cpos[k] - axis[k]*size[1];
  n1 = plane_sphere(c + n, margin, ppos, pmat, p, size[0]);
  if (n1) memcpy(c[n].frame + 3, axis, sizeof axis);
  return n + n1;
}
static int plane_box(RawCon* c, double margin, const double* ppos,
                     const double* pmat, const double* bpos,
                     const double* bmat, const double* size) {
  double n[3] = {pmat[2], pmat[5], pmat[8]}, dif[3], dist;
  int i, k, cnt = 0;
  for (k = 0; k < 3; k++) dif[k] = bpos[k] - ppos[k];
  dist = dot3(dif, n);
  for (i = 0; i < 8; i++) {
    double v[3], corner[3], ld;
    v[0] = (i & 1 ? size[0] : -size[0]);
    v[1] = (i & 2 ? size[1] : -size[1]);
    v[2] = (i & 4 ? size[2] : -size[2]);
    mulmatvec3(corner, bmat, v);
    ld = dot3(n, corner);
    if (dist + ld > margin || ld > 0) continue;
    c[cnt].dist = dist + ld;
    for (k = 0; k < 3; k++)
      c[cnt].pos[k] = corner[k] + bpos[k] - n[k]*0.5*(dist + ld);
    memset(c[cnt].frame, 0, sizeof c[cnt].frame);
    memcpy(c[cnt].frame, n, sizeof n);
    if (++cnt >= 4) return 4;
  }
  return cnt;
}
/* sphere - box (MuJoCo's mjc_SphereBox, restated from its documentation of the
 * box primitives): the sphere centre is taken into the box frame and clamped
 * to the box; outside, the contact joins the clamped point and the sphere;
 * with the centre inside the box the nearest face decides.  Normal: from the
 * sphere (geom 1) towards the box (geom 2). */
static int sphere_box(RawCon* c, double margin, const double* spos, double r,
                      const double* bpos, const double* bmat, const double* size) {
  double dif[3], loc[3], clamped[3], delta[3], nl[3], pl[3], dist, len = 0;
  int k, inside = 1;
  for (k = 0; k < 3; k++) dif[k] = spos[k] - bpos[k];
  for (k = 0; k < 3; k++)   /* loc = R^T dif */
    loc[k] = bmat[k]*dif[0] + bmat[3 + k]*dif[1] + bmat[6 + k]*dif[2];
  for (k = 0; k < 3; k++) {
    clamped[k] = loc[k] < -size[k] ? -size[k] : (loc[k] > size[k] ? size[k] : loc[k]);
    delta[k] = loc[k] - clamped[k];
    if (delta[k] != 0) inside = 0;
    len += delta[k]*delta[k];
  }
  len = sqrt(len);
  if (!inside && len >= MINVAL) {
    dist = len - r;
    if (dist > margin) return 0;
    for (k = 0; k < 3; k++) { nl[k] = delta[k]/len; pl[k] = clamped[k] + nl[k]*0.5*dist; }
  } else {
    int best = 0;
    double depth = size[0] - fabs(loc[0]);
    for (k = 1; k < 3; k++)
      if (size[k] - fabs(loc[k]) < depth) { depth = size[k] - fabs(loc[k]); best = k; }
    nl[0] = nl[1] = nl[2] = 0;
    nl[best] = loc[best] < 0 ? -1 : 1;
    dist = -depth - r;
    for (k = 0; k < 3; k++) pl[k] = loc[k] + nl[k]*0.5*(depth - r);
  }
  c->dist = dist;
  memset(c->frame, 0, sizeof c->frame);
  for (k = 0; k < 3; k++) {   /* world: R pl + bpos; normal sphere -> box = -R nl */
    c->pos[k] = bpos[k] + bmat[3*k]*pl[0] + bmat[3*k + 1]*pl[1] + bmat[3*k + 2]*pl[2];
    c->frame[k] = -(bmat[3*k]*nl[0] + bmat[3*k + 1]*nl[1] + bmat[3*k + 2]*nl[2]);
  }
  return 1;
}
static int sphere_capsule(RawCon* c, double margin, const double* spos,
                          double r, const double* cpos, const double* cmat,
                          const double* size) {
  double axis[3] = {cmat[2], cmat[5], cmat[8]}, v[3], x, p[3];
  int k;
  for (k = 0; k < 3; k++) v[k] = spos[k] - cpos[k];
  x = dot3(axis, v);
  if (x > size[1]) x = size[1];
  if (x < -size[1]) x = -size[1];
  for (k = 0; k < 3; k++) p[k] = cpos[k] + axis[k]*x;
  return sphere_sphere(c, margin, spos, p, r, size[0]);
}
static double clampd(double x, double lo, double hi) {
  return x < lo ? lo : (x > hi ? hi : x);
}
static int capsule_capsule(RawCon* c, double margin, const double* pos1,
                           const double* mat1, const double* size1,
                           const double* pos2, const double* mat2,
                           const double* size2) {
  double a1[3], a2[3], dif[3], ma, mb, mc, u, v, det, x1, x2, v1[3], v2[3];
  int k, n = 0;
  for (k = 0; k < 3; k++) {
    a1[k] = mat1[3*k + 2]*size1[1];
    a2[k] = mat2[3*k + 2]*size2[1];
    dif[k] = pos1[k] - pos2[k];
  }
  ma = dot3(a1, a1); mb = -dot3(a1, a2); mc = dot3(a2, a2);
  u = -dot3(a1, dif); v = dot3(a2, dif);
  det = ma*mc - mb*mb;
  if (fabs(det) >= MINVAL) {
    x1 = (mc*u - mb*v)/det;
    x2 = (ma*v - mb*u)/det;
    if (x1 > 1) { x1 = 1; x2 = (v - mb)/mc; }
    else if (x1 < -1) { x1 = -1; x2 = (v + mb)/mc; }
    if (x2 > 1) { x2 = 1; x1 = clampd((u - mb)/ma, -1, 1); }
    else if (x2 < -1) { x2 = -1; x1 = clampd((u + mb)/ma, -1, 1); }
    for (k = 0; k < 3; k++) {
      v1[k] = pos1[k] + a1[k]*x1;
      v2[k] = pos2[k] + a2[k]*x2;
    }
    return sphere_sphere(c, margin, v1, v2, size1[0], size2[0]);
  }
  /* parallel axes: test both ends of each segment, keep at most two */
  for (k = 0; k < 3; k++) v1[k] = pos1[k] + a1[k];
  x2 = clampd((v - mb)/mc, -1, 1);
  for (k = 0; k < 3; k++) v2[k] = pos2[k] + a2[k]*x2;
  n += sphere_sphere(c + n, margin, v1, v2, size1[0], size2[0]);
  for (k = 0; k < 3; k++) v1[k] = pos1[k] - a1[k];
  x2 = clampd((v + mb)/mc, -1, 1);
  for (k = 0; k < 3; k++) v2[k] = pos2[k] + a2[k]*x2;
  n += sphere_sphere(c + n, margin, v1, v2, size1[0], size2[0]);
  if (n == 2) return n;
  for (k = 0; k < 3; k++) v2[k] = pos2[k] + a2[k];
  x1 = clampd((u - mb)/ma, -1, 1);
  for (k = 0; k < 3; k++) v1[k] = pos1[k] + a1[k]*x1;
  n += sphere_sphere(c + n, margin, v1, v2, size1[0], size2[0]);
  if (n == 2) return n;
  for (k = 0; k < 3; k++) v2[k] = pos2[k] - a2[k];
  x1 = clampd((u + mb)/ma, -1, 1);
  for (k = 0; k < 3; k++) v1[k] = pos1[k] + a1[k]*x1;
  n += sphere_sphere(c + n, margin, v1, v2, size1[0], size2[0]);
  return n;
}

/* capsule - box and box - box.  MuJoCo 2.0's own routines for these pairs
 * (mjc_CapsuleBox, mjc_BoxBox) are part of the closed binary and the reference
 * holds no value that pins their contact manifolds, so these two are OUR
 * construction -- "parity unpinned": geometry checked against brute-force
 * sampling of the surfaces and a force-balance KAT (tests/test_closed_form.py),
 * not against libmujoco.  Conventions as for the other pairs: normal from
 * geom 1 to geom 2, dist < 0 = penetration, pos = midpoint. */

/* d/dt of f(t) = sum_k max(0, |c0_k + t a_k| - s_k)^2: squared distance from the
 * point c0 + t a (box frame) to the box |x_k| <= s_k; nondecreasing in t */
static double seg_box_slope(const double* c0, const double* a, const double* s,
                            double t) {
  double g = 0;
  int k;
  for (k = 0; k < 3; k++) {
    double x = c0[k] + t*a[k];
    if (x > s[k]) g += 2*a[k]*(x - s[k]);
    else if (x < -s[k]) g += 2*a[k]*(x + s[k]);
  }
  return g;
}
/* parameters [tA, tB] (tA == tB unless a whole stretch is equally near) of the
 * points of the segment c0 + t a, |t| <= h, nearest to the box.  f is convex and
 * piecewise quadratic with breaks where the segment crosses a slab plane, so
 * its slope is piecewise linear: evaluate it at the <= 8 break points. */
static void seg_box_nearest(const double* c0, const double* a, double h,
                            const double* s, double* tA, double* tB) {
  double cand[8], g[8], tlo = -h, glo = 0, thi = h, ghi = 0, zlo = 0, zhi = 0;
  int n = 0, k, i, sgn, haveneg = 0, havepos = 0, havezero = 0;
  cand[n++] = -h; cand[n++] = h;
  for (k = 0; k < 3; k++)
    if (fabs(a[k]) > MINVAL)
      for (sgn = -1; sgn <= 1; sgn += 2) {
        double t = (sgn*s[k] - c0[k])/a[k];
        if (t > -h && t < h) cand[n++] = t;
      }
  for (i = 0; i < n; i++) {
    g[i] = seg_box_slope(c0, a, s, cand[i]);
    if (g[i] < 0) {
      if (!haveneg || cand[i] > tlo) { tlo = cand[i]; glo = g[i]; }
      haveneg = 1;
    } else if (g[i] > 0) {
      if (!havepos || cand[i] < thi) { thi = cand[i]; ghi = g[i]; }
      havepos = 1;
    } else {
      if (!havezero || cand[i] < zlo) zlo = cand[i];
      if (!havezero || cand[i] > zhi) zhi = cand[i];
      havezero = 1;
    }
  }
  if (havezero) { *tA = zlo; *tB = zhi; }
  else if (!haveneg) *tA = *tB = -h;            /* rising everywhere */
  else if (!havepos) *tA = *tB = h;             /* falling everywhere */
  else *tA = *tB = tlo + (thi - tlo)*(-glo)/(ghi - glo);
}
static int capsule_box(RawCon* c, double margin, const double* cpos,
                       const double* cmat, const double* csize,
                       const double* bpos, const double* bmat,
                       const double* bsize) {
  double axis[3] = {cmat[2], cmat[5], cmat[8]}, dif[3], c0[3], a[3], tA, tB, ts[4], p[3];
  double h = csize[1];
  int k, n = 0, nt = 0, i, j;
  for (k = 0; k < 3; k++) dif[k] = cpos[k] - bpos[k];
  for (k = 0; k < 3; k++) {   /* into the box frame */
    c0[k] = bmat[k]*dif[0] + bmat[3 + k]*dif[1] + bmat[6 + k]*dif[2];
    a[k] = bmat[k]*axis[0] + bmat[3 + k]*axis[1] + bmat[6 + k]*axis[2];
  }
  seg_box_nearest(c0, a, h, bsize, &tA, &tB);
  ts[nt++] = tA;
  if (tB - tA > 1e-9) ts[nt++] = tB;
  else { ts[nt++] = -h; ts[nt++] = h; }   /* the ends, if they touch as well */
  for (i = 0; i < nt && n < 2; i++) {
    int dup = 0;
    for (j = 0; j < i; j++) if (fabs(ts[i] - ts[j]) <= 1e-9) dup = 1;
    if (dup) continue;
    for (k = 0; k < 3; k++) p[k] = cpos[k] + axis[k]*ts[i];
    if (sphere_box(c + n, margin, p, csize[0], bpos, bmat, bsize)) {
      memcpy(c[n].frame + 3, axis, sizeof axis);
      n++;
    } else if (i == 0) {
      return 0;             /* the nearest point is out of reach: so is the rest */
    }
  }
  return n;
}

/* clips the polygon (u, v, d)[n] against u*sgn <= lim (axis 0) or v*sgn <= lim
 * (axis 1); d (depth) is interpolated.  Sutherland-Hodgman, <= 8 vertices. */
static int clip_poly(double (*poly)[3], int n, int axis, double sgn, double lim) {
  double out[8][3];
  int i, m = 0, k;
  for (i = 0; i < n; i++) {
    const double* A = poly[i];
    const double* B = poly[(i + 1) % n];
    double da = sgn*A[axis] - lim, db = sgn*B[axis] - lim;
    if (da <= 0 && m < 8) { for (k = 0; k < 3; k++) out[m][k] = A[k]; m++; }
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      double t = da/(da - db);
      if (m < 8) { for (k = 0; k < 3; k++) out[m][k] = A[k] + t*(B[k] - A[k]); m++; }
    }
  }
  memcpy(poly, out, sizeof(double)*3*(size_t)m);
  return m;
}
static int box_box(RawCon* c, double margin, const double* p1, const double* m1,
                   const double* s1, const double* p2, const double* m2,
                   const double* s2) {
  double R[3][3], AR[3][3], t[3], dw[3], best = -1e30, bestedge = -1e30, sep;
  int i, j, k, code = -1, ecode = -1, n = 0;
  double bsign = 1, esign = 1;
  for (k = 0; k < 3; k++) dw[k] = p2[k] - p1[k];
  for (i = 0; i < 3; i++) {
    t[i] = m1[i]*dw[0] + m1[3 + i]*dw[1] + m1[6 + i]*dw[2];
    for (j = 0; j < 3; j++) {
      R[i][j] = m1[i]*m2[j] + m1[3 + i]*m2[3 + j] + m1[6 + i]*m2[6 + j];
      AR[i][j] = fabs(R[i][j]) + 1e-12;
    }
  }
  /* face axes of box 1, then of box 2: separation along +-axis */
  for (i = 0; i < 3; i++) {
    double rb = s2[0]*AR[i][0] + s2[1]*AR[i][1] + s2[2]*AR[i][2];
    sep = fabs(t[i]) - (s1[i] + rb);
    if (sep > best) { best = sep; code = i; bsign = t[i] < 0 ? -1 : 1; }
  }
  for (j = 0; j < 3; j++) {
    double ra = s1[0]*AR[0][j] + s1[1]*AR[1][j] + s1[2]*AR[2][j];
    double tj = t[0]*R[0][j] + t[1]*R[1][j] + t[2]*R[2][j];
    sep = fabs(tj) - (ra + s2[j]);
    if (sep > best) { best = sep; code = 3 + j; bsign = tj < 0 ? -1 : 1; }
  }
  /* edge x edge axes (skipped when the edges are parallel) */
  for (i = 0; i < 3; i++)
    for (j = 0; j < 3; j++) {
      int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double len = sqrt(1 - R[i][j]*R[i][j] > 0 ? 1 - R[i][j]*R[i][j] : 0), ra, rb, tl;
      if (len < 1e-6) continue;
      ra = s1[i1]*AR[i2][j] + s1[i2]*AR[i1][j];
      rb = s2[j1]*AR[i][j2] + s2[j2]*AR[i][j1];
      tl = t[i2]*R[i1][j] - t[i1]*R[i2][j];
      sep = (fabs(tl) - (ra + rb))/len;
      if (sep > bestedge) { bestedge = sep; ecode = 3*i + j; esign = tl < 0 ? -1 : 1; }
    }
  /* an edge pair wins only when clearly less penetrating than every face */
  if (ecode >= 0 && bestedge > best + 1e-6 + 1e-3*fabs(best)) {
    double nrm[3], e1[3], e2[3], c1[3], c2[3], w[3], b, d1, d2, den, u, v, q1[3], q2[3];
    i = ecode/3; j = ecode % 3;
    if (bestedge > margin) return 0;
    for (k = 0; k < 3; k++) { e1[k] = m1[3*k + i]; e2[k] = m2[3*k + j]; }
    cross3(nrm, e1, e2);
    normalize3(nrm);
    if (dot3(nrm, dw) < 0) for (k = 0; k < 3; k++) nrm[k] = -nrm[k];
    (void)esign;
    for (k = 0; k < 3; k++) { c1[k] = p1[k]; c2[k] = p2[k]; }
    for (k = 0; k < 3; k++) {       /* supporting edges: towards / away from n */
      double ax1[3] = {m1[k], m1[3 + k], m1[6 + k]}, ax2[3] = {m2[k], m2[3 + k], m2[6 + k]};
      int q;
      if (k != i) {
        double sg = dot3(nrm, ax1) >= 0 ? 1 : -1;
        for (q = 0; q < 3; q++) c1[q] += sg*s1[k]*ax1[q];
      }
      if (k != j) {
        double sg = dot3(nrm, ax2) >= 0 ? -1 : 1;
        for (q = 0; q < 3; q++) c2[q] += sg*s2[k]*ax2[q];
      }
    }
    for (k = 0; k < 3; k++) w[k] = c1[k] - c2[k];
    b = dot3(e1, e2); d1 = dot3(e1, w); d2 = dot3(e2, w);
    den = 1 - b*b;
    u = clampd((b*d2 - d1)/den, -s1[i], s1[i]);
    v = clampd((d2 - b*d1)/den, -s2[j], s2[j]);
    for (k = 0; k < 3; k++) { q1[k] = c1[k] + u*e1[k]; q2[k] = c2[k] + v*e2[k]; }
    for (k = 0; k < 3; k++) w[k] = q2[k] - q1[k];
    c->dist = dot3(w, nrm);
    if (c->dist > margin) return 0;
    memset(c->frame, 0, sizeof c->frame);
    for (k = 0; k < 3; k++) { c->pos[k] = 0.5*(q1[k] + q2[k]); c->frame[k] = nrm[k]; }
    return 1;
  }
  if (best > margin) return 0;
  {
    /* reference face on box `ref`, incident face on the other box */
    const double *pr, *mr, *sr, *pi, *mi, *si;
    double nout[3], poly[8][3], fc[3], ua[3], va[3], su, sv;
    int ax = code % 3, inc = 0, iu, iv, cnt;
    double most = 1e30;
    if (code < 3) { pr = p1; mr = m1; sr = s1; pi = p2; mi = m2; si = s2; }
    else { pr = p2; mr = m2; sr = s2; pi = p1; mi = m1; si = s1; bsign = -bsign; }
    /* outward normal of the reference face (towards the other box) */
    for (k = 0; k < 3; k++) nout[k] = bsign*mr[3*k + ax];
    for (k = 0; k < 3; k++) fc[k] = pr[k] + nout[k]*sr[ax];
    iu = (ax + 1) % 3; iv = (ax + 2) % 3;
    for (k = 0; k < 3; k++) { ua[k] = mr[3*k + iu]; va[k] = mr[3*k + iv]; }
    su = sr[iu]; sv = sr[iv];
    /* incident face: the face of the other box whose normal opposes nout most */
    for (k = 0; k < 3; k++) {
      double axk[3] = {mi[k], mi[3 + k], mi[6 + k]}, d = dot3(axk, nout);
      if (-fabs(d) < most) { most = -fabs(d); inc = k; bsign = d > 0 ? -1 : 1; }
    }
    {
      int ju = (inc + 1) % 3, jv = (inc + 2) % 3, q, vtx;
      static const int su4[4] = {1, -1, -1, 1}, sv4[4] = {1, 1, -1, -1};
      for (vtx = 0; vtx < 4; vtx++) {
        double wpt[3], rel[3];
        for (q = 0; q < 3; q++)
          wpt[q] = pi[q] + bsign*si[inc]*mi[3*q + inc] + su4[vtx]*si[ju]*mi[3*q + ju] +
                   sv4[vtx]*si[jv]*mi[3*q + jv];
        for (q = 0; q < 3; q++) rel[q] = wpt[q] - fc[q];
        poly[vtx][0] = dot3(rel, ua); poly[vtx][1] = dot3(rel, va); poly[vtx][2] = dot3(rel, nout);
      }
    }
    cnt = clip_poly(poly, 4, 0, 1, su);
    cnt = clip_poly(poly, cnt, 0, -1, su);
    cnt = clip_poly(poly, cnt, 1, 1, sv);
    cnt = clip_poly(poly, cnt, 1, -1, sv);
    {
      int keep[8], nk = 0, take;
      for (k = 0; k < cnt; k++) if (poly[k][2] <= margin) keep[nk++] = k;
      for (take = 0; take < (nk < 4 ? nk : 4); take++) {
        int idx = keep[nk <= 4 ? take : (take*nk)/4], q;
        const double* v3 = poly[idx];
        double dir = code < 3 ? 1 : -1;        /* normal from box 1 to box 2 */
        c[n].dist = v3[2];
        memset(c[n].frame, 0, sizeof c[n].frame);
        for (q = 0; q < 3; q++) {
          c[n].pos[q] = fc[q] + v3[0]*ua[q] + v3[1]*va[q] + 0.5*v3[2]*nout[q];
          c[n].frame[q] = dir*nout[q];
        }
        n++;
      }
    }
  }
  return n;
}

static int pair_filtered(const mjoModel* m, int g1, int g2) {
  int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2], w1, w2, i;
  if (m->geom_type[g1] == GEOM_PLANE && m->geom_type[g2] == GEOM_PLANE)
    return 1;
  if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) ||
        (m->geom_contype[g2] & m->geom_conaffinity[g1])))
    return 1;
  w1 = m->body_weldid[b1]; w2 = m->body_weldid[b2];
  if (w1 == w2) return 1;
  if (!(m->disableflags & DSBL_FILTERPARENT) && w1 && w2) {
    int pw1 = m->body_weldid[m->body_parentid[w1]];
    int pw2 = m->body_weldid[m->body_parentid[w2]];
    if (w1 == pw2 || w2 == pw1) return 1;
  }
  for (i = 0; i < m->nexclude; i++) {
    int lo = b1 < b2 ? b1 : b2, hi = b1 < b2 ? b2 : b1;
    if (m->exclude_signature[i] == (lo << 16) + hi) return 1;
  }
  return 0;
}

static void mjo_collision(const mjoModel* m, mjoData* d) {
  int ga, gb, k, i;
  d->ncon = 0;
  if (m->disableflags & (DSBL_CONTACT | DSBL_CONSTRAINT)) return;
  for (ga = 0; ga < m->ngeom; ga++)
    for (gb = ga + 1; gb < m->ngeom; gb++) {
      int g1 = ga, g2 = gb, t1, t2, n = 0;
      double margin, gap;
      const double *p1, *p2, *m1, *m2, *s1, *s2;
      RawCon rc[4];
      if (pair_filtered(m, g1, g2)) continue;
      if (m->geom_type[g1] > m->geom_type[g2]) { g1 = gb; g2 = ga; }
      t1 = m->geom_type[g1]; t2 = m->geom_type[g2];
      margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
      gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
      p1 = d->geom_xpos + 3*g1; p2 = d->geom_xpos + 3*g2;
      m1 = d->geom_xmat + 9*g1; m2 = d->geom_xmat + 9*g2;
      s1 = m->geom_size + 3*g1; s2 = m->geom_size + 3*g2;
      /* bounding-sphere rejection */
      if (t1 == GEOM_PLANE) {
        double nrm[3] = {m1[2], m1[5], m1[8]}, dif[3];
        for (k = 0; k < 3; k++) dif[k] = p2[k] - p1[k];
        if (dot3(dif, nrm) > m->geom_rbound[g2] + margin) continue;
      } else {
        double dif[3], bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
        for (k = 0; k < 3; k++) dif[k] = p2[k] - p1[k];
        if (dot3(dif, dif) > bound*bound) continue;
      }
      if (t1 == GEOM_PLANE && t2 == GEOM_SPHERE)
        n = plane_sphere(rc, margin, p1, m1, p2, s2[0]);
      else if (t1 == GEOM_PLANE && t2 == GEOM_CAPSULE)
        n = plane_capsule(rc, margin, p1, m1, p2, m2, s2);
      else if (t1 == GEOM_PLANE && t2 == GEOM_BOX)
        n = plane_box(rc, margin, p1, m1, p2, m2, s2);
      else if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE)
        n = sphere_sphere(rc, margin, p1, p2, s1[0], s2[0]);
      else if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE)
        n = sphere_capsule(rc, margin, p1, s1[0], p2, m2, s2);
      else if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE)
        n = capsule_capsule(rc, margin, p1, m1, s1, p2, m2, s2);
      else if (t1 == GEOM_SPHERE && t2 == GEOM_BOX)
        n = sphere_box(rc, margin, p1, s1[0], p2, m2, s2);
      else if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX)
        n = capsule_box(rc, margin, p1, m1, s1, p2, m2, s2);
      else if (t1 == GEOM_BOX && t2 == GEOM_BOX)
        n = box_box(rc, margin, p1, m1, s1, p2, m2, s2);
      else
        continue; /* pair types rejected at compile time by the host */
      for (i = 0; i < n; i++) {
        mjoContact* c;
        double mix, sm1 = m->geom_solmix[g1], sm2 = m->geom_solmix[g2];
        if (d->ncon >= m->nconmax) { d->warning[WARN_CONTACTFULL]++; return; }
        c = d->contact + d->ncon++;
        c->dist = rc[i].dist;
        memcpy(c->pos, rc[i].pos, sizeof c->pos);
        memcpy(c->frame, rc[i].frame, sizeof c->frame);
        make_frame(c->frame);
        c->includemargin = margin - gap;
        c->geom1 = g1; c->geom2 = g2;
        c->efc_address = -1;
        /* parameter mixing, equal priority */
        if (m->geom_priority[g1] != m->geom_priority[g2]) {
          int gp = m->geom_priority[g1] > m->geom_priority[g2] ? g1 : g2;
          c->dim = m->geom_condim[gp];
          c->friction[0] = c->friction[1] = m->geom_friction[3*gp];
          c->friction[2] = m->geom_friction[3*gp + 1];
          c->friction[3] = c->friction[4] = m->geom_friction[3*gp + 2];
          memcpy(c->solref, m->geom_solref + 2*gp, sizeof c->solref);
          memcpy(c->solimp, m->geom_solimp + 5*gp, sizeof c->solimp);
        } else {
          double f[3];
          c->dim = m->geom_condim[g1] > m->geom_condim[g2]
                       ? m->geom_condim[g1] : m->geom_condim[g2];
          for (k = 0; k < 3; k++)
            f[k] = fmax(m->geom_friction[3*g1 + k], m->geom_friction[3*g2 + k]);
          c->friction[0] = c->friction[1] = f[0];
          c->friction[2] = f[1];
          c->friction[3] = c->friction[4] = f[2];
          if (sm1 >= MINVAL && sm2 >= MINVAL) mix = sm1/(sm1 + sm2);
          else if (sm1 < MINVAL && sm2 < MINVAL) mix = 0.5;
          else mix = sm1 < MINVAL ? 0.0 : 1.0;
          if (m->geom_solref[2*g1] > 0 && m->geom_solref[2*g2] > 0)
            for (k = 0; k < 2; k++)
              c->solref[k] = mix*m->geom_solref[2*g1 + k] +
                             (1 - mix)*m->geom_solref[2*g2 + k];
          else
            for (k = 0; k < 2; k++)
              c->solref[k] = fmin(m->geom_solref[2*g1 + k],
                                  m->geom_solref[2*g2 + k]);
          for (k = 0; k < 5; k++)
            c->solimp[k] = mix*m->geom_solimp[5*g1 + k] +
                           (1 - mix)*m->geom_solimp[5*g2 + k];
        }
      }
    }
}

/* ---- constraint construction -------------------------------------- */
static double* add_row(const mjoModel* m, mjoData* d, int type, int id,
                       double pos, double margin, const double* solref,
                       const double* solimp, double diagApprox) {
  int r = d->nefc;
  if (r >= m->nefcmax) { d->warning[WARN_CNSTRFULL]++; return NULL; }
  d->nefc++;
  d->efc_type[r] = type; d->efc_id[r] = id;
  d->efc_pos[r] = pos; d->efc_margin[r] = margin;
  d->efc_diagApprox[r] = diagApprox;
  memcpy(d->efc_solref + 2*r, solref, 2*sizeof(double));
  memcpy(d->efc_solimp + 5*r, solimp, 5*sizeof(double));
  memset(d->efc_J + (size_t)r*m->nv, 0, sizeof(double)*m->nv);
  return d->efc_J + (size_t)r*m->nv;
}

static void mjo_make_constraint(const mjoModel* m, mjoData* d) {
  int nv = m->nv, i, j, k;
  double* jac = d->scratch;            /* 6*nv: body2 minus body1 */
  double* jac1 = d->scratch + 6*nv;    /* 6*nv */
  d->nefc = 0;
  if (m->disableflags & DSBL_CONSTRAINT) return;
  /* joint limits */
  if (!(m->disableflags & DSBL_LIMIT))
    for (j = 0; j < m->njnt; j++) {
      int side;
      if (!m->jnt_limited[j]) continue;
      if (m->jnt_type[j] != JNT_SLIDE && m->jnt_type[j] != JNT_HINGE) continue;
      for (side = -1; side <= 1; side += 2) {
        double value = d->qpos[m->jnt_qposadr[j]];
        double dist = side < 0 ? value - m->jnt_range[2*j]
                               : m->jnt_range[2*j + 1] - value;
        if (dist < m->jnt_margin[j]) {
          double* row = add_row(m, d, EFC_LIMIT, j, dist, m->jnt_margin[j],
                                m->jnt_solref + 2*j, m->jnt_solimp + 5*j,
                                m->dof_invweight0[m->jnt_dofadr[j]]);
          if (!row) return;
          row[m->jnt_dofadr[j]] = -(double)side;
        }
      }
    }
  /* contacts */
  for (i = 0; i < d->ncon; i++) {
    mjoContact* c = d->contact + i;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    int dim = c->dim, nrot = dim > 3 ? dim - 3 : 0;
    double tran, rot;
    double* jc = d->scratch + 12*nv; /* 6*nv contact-frame rows (scratch holds 2 nv^2 + 16 nv) */
    if (c->dist >= c->includemargin) continue; /* inside margin-gap band only */
    mjo_jac(m, d, jac, jac + 3*nv, c->pos, b2);
    mjo_jac(m, d, jac1, jac1 + 3*nv, c->pos, b1);
    for (k = 0; k < 6*nv; k++) jac[k] -= jac1[k];
    for (k = 0; k < 3; k++)
      for (j = 0; j < nv; j++)
        jc[k*nv + j] = c->frame[3*k]*jac[j] + c->frame[3*k + 1]*jac[nv + j] +
                       c->frame[3*k + 2]*jac[2*nv + j];
    for (k = 0; k < nrot; k++)
      for (j = 0; j < nv; j++)
        jc[(3 + k)*nv + j] = c->frame[3*k]*jac[3*nv + j] +
                             c->frame[3*k + 1]*jac[4*nv + j] +
                             c->frame[3*k + 2]*jac[5*nv + j];
    tran = m->body_invweight0[2*b1] + m->body_invweight0[2*b2];
    rot = m->body_invweight0[2*b1 + 1] + m->body_invweight0[2*b2 + 1];
    c->efc_address = d->nefc;
    if (dim == 1) {
      double* row = add_row(m, d, EFC_CONTACT_FRICTIONLESS, i, c->dist,
                            c->includemargin, c->solref, c->solimp, tran);
      if (!row) return;
      memcpy(row, jc, sizeof(double)*nv);
    } else {
      for (k = 1; k < dim; k++) {
        double fri = c->friction[k - 1];
        double da = tran + fri*fri*(k < 3 ? tran : rot);
        int sgn;
        for (sgn = 1; sgn >= -1; sgn -= 2) {
          double* row = add_row(m, d, EFC_CONTACT_PYRAMIDAL, i, c->dist,
                                c->includemargin, c->solref, c->solimp, da);
          if (!row) return;
          for (j = 0; j < nv; j++) row[j] = jc[j] + sgn*fri*jc[k*nv + j];
        }
      }
    }
  }
}

static void impedance(const double* solimp_in, double x, double* imp) {
  double s[5];
  double y;
  s[0] = clampd(solimp_in[0], MINIMP, MAXIMP);
  s[1] = clampd(solimp_in[1], MINIMP, MAXIMP);
  s[2] = fmax(0.0, solimp_in[2]);
  s[3] = clampd(solimp_in[3], MINIMP, MAXIMP);
  s[4] = fmax(1.0, solimp_in[4]);
  if (s[0] == s[1] || s[2] <= MINVAL) { *imp = 0.5*(s[0] + s[1]); return; }
  x = fabs(x)/s[2];
  if (x >= 1) { *imp = s[1]; return; }
  if (x <= 0) { *imp = s[0]; return; }
  if (s[4] == 1) y = x;
  else if (x <= s[3]) y = pow(x, s[4])/pow(s[3], s[4] - 1);
  else y = 1 - pow(1 - x, s[4])/pow(1 - s[3], s[4] - 1);
  *imp = s[0] + y*(s[1] - s[0]);
}

/* velocity-dependent part: efc_vel, aref, R, D (needs qvel) */
static void mjo_reference_constraint(const mjoModel* m, mjoData* d) {
  int i, nv = m->nv;
  for (i = 0; i < d->nefc; i++) {
    double tc = d->efc_solref[2*i], dr = d->efc_solref[2*i + 1];
    double dmax = clampd(d->efc_solimp[5*i + 1], MINIMP, MAXIMP);
    double K, B, imp, pm = d->efc_pos[i] - d->efc_margin[i];
    if (tc > 0) {
      if (!(m->disableflags & DSBL_REFSAFE)) tc = fmax(tc, 2*m->timestep);
      K = 1/fmax(MINVAL, dmax*dmax*tc*tc*dr*dr);
      B = 2/fmax(MINVAL, dmax*tc);
    } else {
      K = -tc/fmax(MINVAL, dmax*dmax);
      B = -dr/fmax(MINVAL, dmax);
    }
    impedance(d->efc_solimp + 5*i, pm, &imp);
    d->efc_vel[i] = dotn(d->efc_J + (size_t)i*nv, d->qvel, nv);
    d->efc_aref[i] = -B*d->efc_vel[i] - K*imp*pm;
    d->efc_R[i] = fmax(MINVAL, (1 - imp)*d->efc_diagApprox[i]/imp);
  }
  /* pyramidal contacts: every edge gets 2*mu^2*R(first edge) */
  for (i = 0; i < d->nefc; i++)
    if (d->efc_type[i] == EFC_CONTACT_PYRAMIDAL) {
      const mjoContact* c = d->contact + d->efc_id[i];
      int rows = 2*(c->dim - 1), j;
      double rpy = 2*c->friction[0]*c->friction[0]*d->efc_R[i];
      rpy = fmax(MINVAL, rpy);
      /* (a contact cut short by the row capacity has fewer rows than `rows`) */
      for (j = 0; j < rows && i + j < d->nefc; j++) d->efc_R[i + j] = rpy;
      i += rows - 1;
    }
  for (i = 0; i < d->nefc; i++) d->efc_D[i] = 1/d->efc_R[i];
}

static void mjo_fwd_position(const mjoModel* m, mjoData* d) {
  mjo_kinematics(m, d);
  mjo_com_pos(m, d);
  mjo_crb(m, d);
  if (chol_factor(d->qL, d->qM, m->nv)) d->warning[WARN_INERTIA]++;
  mjo_collision(m, d);
  mjo_make_constraint(m, d);
}

/* ------------------------------------------------------------------ */
/* velocity stage                                                       */
/* ------------------------------------------------------------------ */
static void mjo_com_vel(const mjoModel* m, mjoData* d) {
  int i, j, k;
  memset(d->cvel, 0, 6*sizeof(double));
  for (i = 1; i < m->nbody; i++) {
    double cvel[6];
    int jadr = m->body_jntadr[i];
    memcpy(cvel, d->cvel + 6*m->body_parentid[i], sizeof cvel);
    for (j = 0; j < m->body_jntnum[i]; j++) {
      int jid = jadr + j, da = m->jnt_dofadr[jid], nd, first = 0;
      switch (m->jnt_type[jid]) {
        case JNT_FREE:
          memset(d->cdof_dot + 6*da, 0, 18*sizeof(double));
          for (k = 0; k < 3; k++) {
            int c;
            for (c = 0; c < 6; c++)
              cvel[c] += d->cdof[6*(da + k) + c]*d->qvel[da + k];
          }
          first = 3;
          /* fall through */
        case JNT_BALL:
          for (k = 0; k < 3; k++)
            cross_motion(d->cdof_dot + 6*(da + first + k), cvel,
                         d->cdof + 6*(da + first + k));
          for (k = 0; k < 3; k++) {
            int c;
            for (c = 0; c < 6; c++)
              cvel[c] += d->cdof[6*(da + first + k) + c]*d->qvel[da + first + k];
          }
          break;
        default:
          nd = 1; (void)nd;
          cross_motion(d->cdof_dot + 6*da, cvel, d->cdof + 6*da);
          for (k = 0; k < 6; k++) cvel[k] += d->cdof[6*da + k]*d->qvel[da];
      }
    }
    memcpy(d->cvel + 6*i, cvel, sizeof cvel);
  }
}

static void mjo_passive(const mjoModel* m, mjoData* d) {
  int j, i;
  memset(d->qfrc_passive, 0, sizeof(double)*m->nv);
  if (m->disableflags & DSBL_PASSIVE) return;
  for (j = 0; j < m->njnt; j++) {
    if (m->jnt_stiffness[j] == 0) continue;
    if (m->jnt_type[j] == JNT_SLIDE || m->jnt_type[j] == JNT_HINGE) {
      int qa = m->jnt_qposadr[j];
      d->qfrc_passive[m->jnt_dofadr[j]] =
          -m->jnt_stiffness[j]*(d->qpos[qa] - m->qpos_spring[qa]);
    }
  }
  for (i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i]*d->qvel[i];
}

/* RNE with zero joint acceleration -> qfrc_bias */
static void mjo_rne(const mjoModel* m, mjoData* d, double* result) {
  int i, j, k, nv = m->nv;
  memset(d->cacc, 0, 6*sizeof(double));
  if (!(m->disableflags & DSBL_GRAVITY))
    for (k = 0; k < 3; k++) d->cacc[3 + k] = -m->gravity[k];
  memset(d->cfrc, 0, 6*sizeof(double));
  for (i = 1; i < m->nbody; i++) {
    double tmp[6], tmp1[6];
    int da = m->body_dofadr[i];
    memcpy(d->cacc + 6*i, d->cacc + 6*m->body_parentid[i], 6*sizeof(double));
    for (j = 0; j < m->body_dofnum[i]; j++)
      for (k = 0; k < 6; k++)
        d->cacc[6*i + k] += d->cdof_dot[6*(da + j) + k]*d->qvel[da + j];
    mul_inert_vec(d->cfrc + 6*i, d->cinert + 10*i, d->cacc + 6*i);
    mul_inert_vec(tmp, d->cinert + 10*i, d->cvel + 6*i);
    cross_force(tmp1, d->cvel + 6*i, tmp);
    for (k = 0; k < 6; k++) d->cfrc[6*i + k] += tmp1[k];
  }
  for (i = m->nbody - 1; i > 0; i--)
    if (m->body_parentid[i])
      for (k = 0; k < 6; k++)
        d->cfrc[6*m->body_parentid[i] + k] += d->cfrc[6*i + k];
  for (i = 0; i < nv; i++)
    result[i] = dotn(d->cdof + 6*i, d->cfrc + 6*m->dof_bodyid[i], 6);
}

static void mjo_subtree_vel(const mjoModel* m, mjoData* d) {
  int i, k;
  for (i = 0; i < m->nbody; i++) {
    double dif[3], t[3];
    const double* com = d->subtree_com + 3*m->body_rootid[i];
    for (k = 0; k < 3; k++) dif[k] = d->xipos[3*i + k] - com[k];
    cross3(t, d->cvel + 6*i, dif); /* w x dif */
    for (k = 0; k < 3; k++)
      d->subtree_linvel[3*i + k] = m->body_mass[i]*(d->cvel[6*i + 3 + k] + t[k]);
  }
  for (i = m->nbody - 1; i > 0; i--)
    for (k = 0; k < 3; k++)
      d->subtree_linvel[3*m->body_parentid[i] + k] += d->subtree_linvel[3*i + k];
  for (i = 0; i < m->nbody; i++)
    for (k = 0; k < 3; k++)
      d->subtree_linvel[3*i + k] /= fmax(MINVAL, m->body_subtreemass[i]);
}

static void mjo_sensor_pos(const mjoModel* m, mjoData* d) {
  int i;
  for (i = 0; i < m->nsensor; i++) {
    int a = m->sensor_adr[i], o = m->sensor_objid[i];
    if (m->sensor_type[i] == SENS_JOINTPOS)
      d->sensordata[a] = d->qpos[m->jnt_qposadr[o]];
    else if (m->sensor_type[i] == SENS_SUBTREECOM)
      memcpy(d->sensordata + a, d->subtree_com + 3*o, 3*sizeof(double));
  }
}
static void mjo_sensor_vel(const mjoModel* m, mjoData* d) {
  int i, need = 0;
  for (i = 0; i < m->nsensor; i++)
    if (m->sensor_type[i] == SENS_SUBTREELINVEL) need = 1;
  if (need) mjo_subtree_vel(m, d);
  for (i = 0; i < m->nsensor; i++) {
    int a = m->sensor_adr[i], o = m->sensor_objid[i];
    if (m->sensor_type[i] == SENS_JOINTVEL)
      d->sensordata[a] = d->qvel[m->jnt_dofadr[o]];
    else if (m->sensor_type[i] == SENS_SUBTREELINVEL)
      memcpy(d->sensordata + a, d->subtree_linvel + 3*o, 3*sizeof(double));
  }
}

static void mjo_fwd_velocity(const mjoModel* m, mjoData* d) {
  mjo_com_vel(m, d);
  mjo_passive(m, d);
  mjo_rne(m, d, d->qfrc_bias);
  mjo_reference_constraint(m, d);
}

/* ------------------------------------------------------------------ */
/* actuation, acceleration, constraint solve                            */
/* ------------------------------------------------------------------ */
static void mjo_fwd_actuation(const mjoModel* m, mjoData* d) {
  int i;
  memset(d->qfrc_actuator, 0, sizeof(double)*m->nv);
  memset(d->actuator_force, 0, sizeof(double)*m->nu);
  if (m->disableflags & DSBL_ACTUATION) return;
  for (i = 0; i < m->nu; i++) {
    /* transmission: a joint, or a fixed tendon = sum_k coef_k * joint_k
       (actuator length = gear * tendon length, moment on dof_k = gear * coef_k) */
    const int tendon = m->actuator_trntype[i] == TRN_TENDON;
    const int t = m->actuator_trnid[i];
    const int nw = tendon ? m->tendon_num[t] : 1;
    const int w0 = tendon ? m->tendon_adr[t] : 0;
    int k;
    double ctrl = d->ctrl[i], gear = m->actuator_gear[i], force;
    double length = 0, velocity = 0;
    for (k = 0; k < nw; k++) {
      const int j = tendon ? m->wrap_objid[w0 + k] : t;
      const double coef = tendon ? m->wrap_prm[w0 + k] : 1.0;
      length += gear*coef*d->qpos[m->jnt_qposadr[j]];
      velocity += gear*coef*d->qvel[m->jnt_dofadr[j]];
    }
    if (m->actuator_ctrllimited[i] && !(m->disableflags & DSBL_CLAMPCTRL))
      ctrl = clampd(ctrl, m->actuator_ctrlrange[2*i],
                    m->actuator_ctrlrange[2*i + 1]);
    force = m->actuator_gainprm[3*i]*ctrl;
    if (m->actuator_biastype[i] == BIAS_AFFINE)
      force += m->actuator_biasprm[3*i] + m->actuator_biasprm[3*i + 1]*length +
               m->actuator_biasprm[3*i + 2]*velocity;
    if (m->actuator_forcelimited[i])
      force = clampd(force, m->actuator_forcerange[2*i],
                     m->actuator_forcerange[2*i + 1]);
    d->actuator_force[i] = force;
    for (k = 0; k < nw; k++) {
      const int j = tendon ? m->wrap_objid[w0 + k] : t;
      const double coef = tendon ? m->wrap_prm[w0 + k] : 1.0;
      d->qfrc_actuator[m->jnt_dofadr[j]] += gear*coef*force;
    }
  }
}

static void mjo_fwd_acceleration(const mjoModel* m, mjoData* d) {
  int i;
  for (i = 0; i < m->nv; i++)
    d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] +
                        d->qfrc_applied[i] + d->qfrc_actuator[i];
  chol_solve(d->qacc_smooth, d->qL, d->qfrc_smooth, m->nv);
}

static void mat_vec(double* r, const double* A, const double* x, int nr, int nc) {
  int i;
  for (i = 0; i < nr; i++) r[i] = dotn(A + (size_t)i*nc, x, nc);
}

/* constraint cost s(jar), forces, qfrc_constraint; all rows are unilateral
 * quadratics here (limits, frictionless and pyramidal contacts) */
static double constraint_update(const mjoModel* m, mjoData* d,
                                const double* jar, int want_force) {
  int i, j, nv = m->nv;
  double cost = 0;
  if (want_force) memset(d->qfrc_constraint, 0, sizeof(double)*nv);
  for (i = 0; i < d->nefc; i++) {
    if (jar[i] < 0) {
      cost += 0.5*d->efc_D[i]*jar[i]*jar[i];
      if (want_force) {
        double f = -d->efc_D[i]*jar[i];
        const double* row = d->efc_J + (size_t)i*nv;
        d->efc_force[i] = f;
        for (j = 0; j < nv; j++) d->qfrc_constraint[j] += row[j]*f;
      }
    } else if (want_force) {
      d->efc_force[i] = 0;
    }
  }
  return cost;
}

typedef struct { double alpha, cost, d0, d1; } LsPoint;

static void ls_eval(LsPoint* p, double alpha, const mjoData* d,
                    const double* jaref, const double* jv, const double* qg) {
  int i;
  double cost = alpha*alpha*qg[2] + alpha*qg[1] + qg[0];
  double d0 = 2*alpha*qg[2] + qg[1], d1 = 2*qg[2];
  for (i = 0; i < d->nefc; i++) {
    double x = jaref[i] + alpha*jv[i];
    if (x < 0) {
      double D = d->efc_D[i];
      cost += 0.5*D*x*x; d0 += D*x*jv[i]; d1 += D*jv[i]*jv[i];
    }
  }
  p->alpha = alpha; p->cost = cost; p->d0 = d0;
  p->d1 = d1 > MINVAL ? d1 : MINVAL;
}

/* exact line search on the convex piecewise-quadratic cost along `search`:
 * safeguarded Newton on the directional derivative */
static double line_search(const mjoModel* m, const mjoData* d,
                          const double* jaref, const double* jv,
                          const double* qg, double gtol) {
  LsPoint p0, p, best;
  double lo = 0, hi = 0, a;
  int have_hi = 0, it;
  (void)m;
  ls_eval(&p0, 0.0, d, jaref, jv, qg);
  if (p0.d0 >= 0) return 0;
  best = p0;
  a = -p0.d0/p0.d1;
  for (it = 0; it < 50; it++) {
    double an;
    ls_eval(&p, a, d, jaref, jv, qg);
    if (p.cost < best.cost) best = p;
    if (fabs(p.d0) < gtol) break;
    if (p.d0 < 0) lo = a; else { hi = a; have_hi = 1; }
    an = a - p.d0/p.d1;
    if (have_hi) {
      if (!(an > lo && an < hi)) an = 0.5*(lo + hi);
      if (hi - lo < 1e-15*fmax(1.0, fabs(hi))) break;
    } else if (an <= lo) {
      an = 2*a;
    }
    a = an;
  }
  return best.alpha;
}

static void mjo_solve_newton(const mjoModel* m, mjoData* d) {
  int nv = m->nv, nefc = d->nefc, i, j, k, iter;
  double* w = d->scratch;
  double* H = w;               w += nv*nv;
  double* HL = w;              w += nv*nv;
  double* Ma = w;              w += nv;
  double* Mv = w;              w += nv;
  double* grad = w;            w += nv;
  double* search = w;          w += nv;
  double* tmp = w;             w += nv;
  double* jaref = w;           w += m->nefcmax;
  double* jv = w;              w += m->nefcmax;
  double scale = 1/(m->meaninertia*(nv > 1 ? nv : 1));
  double cost, gauss, oldcost = 0;
  d->solver_iter = 0;

  mat_vec(Ma, d->qM, d->qacc, nv, nv);
  mat_vec(jaref, d->efc_J, d->qacc, nefc, nv);
  for (i = 0; i < nefc; i++) jaref[i] -= d->efc_aref[i];

  for (iter = 0;; iter++) {
    double snorm, gtol, qg[3], alpha, improvement, gradnorm;
    /* update constraint state at the current qacc */
    cost = constraint_update(m, d, jaref, 1);
    gauss = 0;
    for (i = 0; i < nv; i++)
      gauss += 0.5*(Ma[i] - d->qfrc_smooth[i])*(d->qacc[i] - d->qacc_smooth[i]);
    cost += gauss;
    if (iter > 0) {
      improvement = scale*(oldcost - cost);
      for (i = 0; i < nv; i++)
        grad[i] = Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
      gradnorm = scale*sqrt(dotn(grad, grad, nv));
      d->solver_iter = iter;
      if (improvement < m->tolerance || gradnorm < m->tolerance) break;
    }
    if (iter >= m->iterations) break;
    /* Hessian of the active set and Newton direction */
    memcpy(H, d->qM, sizeof(double)*nv*nv);
    for (i = 0; i < nefc; i++)
      if (jaref[i] < 0) {
        const double* row = d->efc_J + (size_t)i*nv;
        double D = d->efc_D[i];
        for (j = 0; j < nv; j++) {
          double s = D*row[j];
          if (s == 0) continue;
          for (k = 0; k <= j; k++) H[j*nv + k] += s*row[k];
        }
      }
    for (j = 0; j < nv; j++)
      for (k = j + 1; k < nv; k++) H[j*nv + k] = H[k*nv + j];
    chol_factor(HL, H, nv);
    for (i = 0; i < nv; i++)
      grad[i] = Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
    chol_solve(tmp, HL, grad, nv);
    for (i = 0; i < nv; i++) search[i] = -tmp[i];
    /* line search */
    snorm = sqrt(dotn(search, search, nv));
    if (snorm < MINVAL) break;
    gtol = m->tolerance*0.01*snorm/scale;
    mat_vec(Mv, d->qM, search, nv, nv);
    mat_vec(jv, d->efc_J, search, nefc, nv);
    qg[0] = gauss;
    qg[1] = dotn(search, Ma, nv) - dotn(search, d->qfrc_smooth, nv);
    qg[2] = 0.5*dotn(search, Mv, nv);
    alpha = line_search(m, d, jaref, jv, qg, gtol);
    if (alpha == 0) break;
    for (i = 0; i < nv; i++) { d->qacc[i] += alpha*search[i]; Ma[i] += alpha*Mv[i]; }
    for (i = 0; i < nefc; i++) jaref[i] += alpha*jv[i];
    oldcost = cost;
  }
}

static void mjo_fwd_constraint(const mjoModel* m, mjoData* d) {
  int nv = m->nv, nefc = d->nefc, i;
  if (!nefc) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double)*nv);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double)*nv);
    memset(d->qfrc_constraint, 0, sizeof(double)*nv);
    d->solver_iter = 0;
    return;
  }
  mat_vec(d->efc_b, d->efc_J, d->qacc_smooth, nefc, nv);
  for (i = 0; i < nefc; i++) d->efc_b[i] -= d->efc_aref[i];
  /* warmstart: better of previous qacc and the unconstrained qacc */
  if (!(m->disableflags & DSBL_WARMSTART)) {
    double* jar = d->scratch + 2*nv*nv + 8*nv;
    double* Ma = d->scratch;
    double cw, cs;
    mat_vec(jar, d->efc_J, d->qacc_warmstart, nefc, nv);
    for (i = 0; i < nefc; i++) jar[i] -= d->efc_aref[i];
    cw = constraint_update(m, d, jar, 0);
    mat_vec(Ma, d->qM, d->qacc_warmstart, nv, nv);
    for (i = 0; i < nv; i++)
      cw += 0.5*(Ma[i] - d->qfrc_smooth[i])*
            (d->qacc_warmstart[i] - d->qacc_smooth[i]);
    cs = constraint_update(m, d, d->efc_b, 0);
    memcpy(d->qacc, cw > cs ? d->qacc_smooth : d->qacc_warmstart,
           sizeof(double)*nv);
  } else {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double)*nv);
  }
  mjo_solve_newton(m, d);
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double)*nv);
}

/* ------------------------------------------------------------------ */
/* checks, integrators, top-level pipeline                              */
/* ------------------------------------------------------------------ */
static int bad(double x) { return isnan(x) || x > MAXVAL || x < -MAXVAL; }
static void check_array(const mjoModel* m, mjoData* d, const double* a, int n,
                        int warn) {
  int i;
  for (i = 0; i < n; i++)
    if (bad(a[i])) {
      d->warning[warn]++;
      mjo_reset_data(m, d);
      return;
    }
}

static void integrate_pos(const mjoModel* m, double* qpos, const double* qvel,
                          double h) {
  int j, k;
  for (j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    switch (m->jnt_type[j]) {
      case JNT_FREE:
        for (k = 0; k < 3; k++) qpos[qa + k] += h*qvel[da + k];
        quat_integrate(qpos + qa + 3, qvel + da + 3, h);
        break;
      case JNT_BALL:
        quat_integrate(qpos + qa, qvel + da, h);
        break;
      default:
        qpos[qa] += h*qvel[da];
    }
  }
}

static void mjo_advance(const mjoModel* m, mjoData* d, const double* qacc,
                        const double* qvel_for_pos) {
  int i;
  for (i = 0; i < m->nv; i++) d->qvel[i] += m->timestep*qacc[i];
  integrate_pos(m, d->qpos, qvel_for_pos ? qvel_for_pos : d->qvel, m->timestep);
  d->time += m->timestep;
}

static void mjo_euler(const mjoModel* m, mjoData* d) {
  int nv = m->nv, i, damped = 0;
  double* qacc = d->scratch + 2*nv*nv;
  for (i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
  if (!damped) {
    memcpy(qacc, d->qacc, sizeof(double)*nv);
  } else {
    double* A = d->scratch;
    double* L = d->scratch + nv*nv;
    memcpy(A, d->qM, sizeof(double)*nv*nv);
    for (i = 0; i < nv; i++) {
      A[i*nv + i] += m->timestep*m->dof_damping[i];
      qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    }
    chol_factor(L, A, nv);
    chol_solve(qacc, L, qacc, nv);
  }
  mjo_advance(m, d, qacc, NULL);
}

static void forward_skip(const mjoModel* m, mjoData* d, int skipsensor) {
  mjo_fwd_position(m, d);
  if (!skipsensor) mjo_sensor_pos(m, d);
  mjo_fwd_velocity(m, d);
  if (!skipsensor) mjo_sensor_vel(m, d);
  mjo_fwd_actuation(m, d);
  mjo_fwd_acceleration(m, d);
  mjo_fwd_constraint(m, d);
}
void mjo_forward(const mjoModel* m, mjoData* d) { forward_skip(m, d, 0); }

static void mjo_rk4(const mjoModel* m, mjoData* d) {
  static const double A[3][3] = {{0.5, 0, 0}, {0, 0.5, 0}, {0, 0, 1}};
  static const double B[4] = {1.0/6, 1.0/3, 1.0/3, 1.0/6};
  int nq = m->nq, nv = m->nv, i, j, k;
  double h = m->timestep, time0 = d->time;
  double* buf = (double*)malloc(sizeof(double)*(size_t)(nq + 10*nv + 8));
  double* q0 = buf; double* v0 = q0 + nq;
  double* Fv = v0 + nv;       /* 4*nv stage velocities */
  double* Fa = Fv + 4*nv;     /* 4*nv stage accelerations */
  double* dv = Fa + 4*nv;
  memcpy(q0, d->qpos, sizeof(double)*nq);
  memcpy(v0, d->qvel, sizeof(double)*nv);
  memcpy(Fv, d->qvel, sizeof(double)*nv);
  memcpy(Fa, d->qacc, sizeof(double)*nv);
  for (i = 1; i < 4; i++) {
    double tc = 0;
    for (k = 0; k < nv; k++) {
      double sv = 0, sa = 0;
      for (j = 0; j < i; j++) { sv += A[i-1][j]*Fv[j*nv + k]; sa += A[i-1][j]*Fa[j*nv + k]; }
      dv[k] = sv;
      d->qvel[k] = v0[k] + h*sa;
    }
    for (j = 0; j < i; j++) tc += A[i-1][j];
    memcpy(d->qpos, q0, sizeof(double)*nq);
    integrate_pos(m, d->qpos, dv, h);
    d->time = time0 + tc*h;
    forward_skip(m, d, 1);
    memcpy(Fv + i*nv, d->qvel, sizeof(double)*nv);
    memcpy(Fa + i*nv, d->qacc, sizeof(double)*nv);
  }
  {
    double* acc = dv;
    double* vel = (double*)malloc(sizeof(double)*(nv > 0 ? nv : 1));
    for (k = 0; k < nv; k++) {
      double sv = 0, sa = 0;
      for (j = 0; j < 4; j++) { sv += B[j]*Fv[j*nv + k]; sa += B[j]*Fa[j*nv + k]; }
      vel[k] = sv; acc[k] = sa;
    }
    memcpy(d->qpos, q0, sizeof(double)*nq);
    memcpy(d->qvel, v0, sizeof(double)*nv);
    d->time = time0;
    mjo_advance(m, d, acc, vel);
    free(vel);
  }
  free(buf);
}

static void check_pos(const mjoModel* m, mjoData* d) { check_array(m, d, d->qpos, m->nq, WARN_BADQPOS); }
static void check_vel(const mjoModel* m, mjoData* d) { check_array(m, d, d->qvel, m->nv, WARN_BADQVEL); }
static void check_acc(const mjoModel* m, mjoData* d) {
  int before = d->warning[WARN_BADQACC];
  check_array(m, d, d->qacc, m->nv, WARN_BADQACC);
  if (d->warning[WARN_BADQACC] != before) mjo_forward(m, d);
}

void mjo_step(const mjoModel* m, mjoData* d) {
  check_pos(m, d); check_vel(m, d);
  mjo_forward(m, d);
  check_acc(m, d);
  if (m->integrator == INT_RK4) mjo_rk4(m, d); else mjo_euler(m, d);
}
void mjo_step1(const mjoModel* m, mjoData* d) {
  check_pos(m, d); check_vel(m, d);
  mjo_fwd_position(m, d);
  mjo_sensor_pos(m, d);
  mjo_fwd_velocity(m, d);
  mjo_sensor_vel(m, d);
}
void mjo_step2(const mjoModel* m, mjoData* d) {
  mjo_fwd_actuation(m, d);
  mjo_fwd_acceleration(m, d);
  mjo_fwd_constraint(m, d);
  check_acc(m, d);
  mjo_euler(m, d);
}
/* Physics.step of the reference (engine.py:149-166) */
void mjo_physics_step(const mjoModel* m, mjoData* d) {
  if (m->integrator == INT_EULER) mjo_step2(m, d); else mjo_step(m, d);
  mjo_step1(m, d);
}

/* contact force in the contact frame (normal, tangents, torsion, rolling) */
void mjo_contact_force(const mjoModel* m, const mjoData* d, int id, double* out) {
  const mjoContact* c = d->contact + id;
  int k;
  (void)m;
  for (k = 0; k < 6; k++) out[k] = 0;
  if (c->efc_address < 0) return;
  if (c->dim == 1) { out[0] = d->efc_force[c->efc_address]; return; }
  for (k = 1; k < c->dim; k++) {
    double f1 = d->efc_force[c->efc_address + 2*(k - 1)];
    double f2 = d->efc_force[c->efc_address + 2*(k - 1) + 1];
    out[0] += f1 + f2;
    out[k] = c->friction[k - 1]*(f1 - f2);
  }
}
void mjo_contact_get(const mjoData* d, int id, double* dist, double* pos,
                     double* frame, int* geoms) {
  const mjoContact* c = d->contact + id;
  *dist = c->dist;
  memcpy(pos, c->pos, sizeof c->pos);
  memcpy(frame, c->frame, sizeof c->frame);
  geoms[0] = c->geom1; geoms[1] = c->geom2; geoms[2] = c->dim;
}

/* batched stepping over independent instances (cpu_baseline leg of bench.py;
 * mirrors one process per env in scripts/vec_env.py:334-459, as threads).
 * ctrl is [nenv][nu] row-major. */
int mjo_batch_step(const mjoModel* m, mjoData** ds, int nenv,
                   const double* ctrl, int nsub, int nthreads) {
  int e;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
  for (e = 0; e < nenv; e++) {
    int s;
    if (ctrl) memcpy(ds[e]->ctrl, ctrl + (size_t)e*m->nu, sizeof(double)*m->nu);
    for (s = 0; s < nsub; s++) mjo_physics_step(m, ds[e]);
  }
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  (void)nthreads;
  return 1;
#endif
}

/* linear / angular velocity of a world point rigidly attached to `body`
 * (world frame): v = J(point) qvel.  Used by the known-answer test that mirrors
 * wrapper/core_test.py:407-459 (mj_objectVelocity). */
void mjo_point_velocity(const mjoModel* m, mjoData* d, int body,
                        const double* point, double* linvel, double* angvel) {
  double* jacp = d->scratch;
  double* jacr = d->scratch + 3*m->nv;
  int k;
  mjo_jac(m, d, jacp, jacr, point, body);
  for (k = 0; k < 3; k++) {
    linvel[k] = dotn(jacp + k*m->nv, d->qvel, m->nv);
    angvel[k] = dotn(jacr + k*m->nv, d->qvel, m->nv);
  }
}
