// dmc_kernels.hip -- batched MuJoCo-style physics step for gfx950 (MI355X).
//
// One code object is built per compiled model: the model constants arrive as
// `static __device__ constexpr` tables in a generated header (-include'd by
// the build, see dm_control_amd/codegen.py), so tree topology, geom pairs and
// solver parameters are compile-time data for this translation unit.
//
// Replaces, for B independent environment instances per launch, the per-env
// native calls of the reference:
//   Physics.step        /root/reference/dm_control/mujoco/engine.py:149-166
//   mj_step2 / mj_step / mj_step1 inside libmujoco
//   task.get_observation / get_reward   suite/<domain>.py (TASK_* below)
//
// Layout: struct-of-arrays state field[k][env] in HBM; one environment per
// lane, 64-lane workgroups (one wavefront each) so consecutive lanes read
// consecutive addresses.  The per-lane working set lives in registers (the
// unrolled build indexes everything statically); constraint rows and the
// contact list are per-lane records in LDS ([word][lane]) with an overflow
// tier of the same layout in an HBM workspace.  No MFMA: the largest dense
// object is the NV x NV Hessian (45 packed entries for cheetah).
//
// csrc/dmc_coop.hip includes this file for its helpers (math, narrowphase,
// impedance, touch sensors, task layer, episode initialisation) and supplies
// the several-lanes-per-env versions of dmc_step / dmc_observe.
//
// `real` is float (default) or double (-DDMC_REAL_IS_DOUBLE) -- the fp64 build
// is the tight-parity mode, the fp32 build is the throughput mode.

#if defined(__HIPCC_RTC__)
// in-process build (dmc_model_compile, HIP runtime compilation): the device
// runtime is built in, system headers are not available
typedef unsigned int uint32_t;
typedef unsigned long long uint64_t;
#else
#ifndef DMC_HOST_SHIM
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#endif

#ifdef DMC_REAL_IS_DOUBLE
typedef double real;   // must match dmc_real in the generated header
#define DMC_TOL_FLOOR 0.0
#define DMC_MINVAL 1e-15
#define DMC_F32_RULES 0
#else
typedef float real;
#define DMC_MINVAL 1e-15f
#ifdef DMC_F64_RULES
// experiment (tools/gpu_precision_study.py): fp32 arithmetic with the fp64
// build's unmodified stopping rules, to separate rounding from the rules below
#define DMC_TOL_FLOOR 0.0
#define DMC_F32_RULES 0
#else
// fp32 cannot resolve cost changes below ~1e-7 relative; see DESIGN.md
#define DMC_TOL_FLOOR 1e-6
// fp32-only stopping rules: the absolute 1e-8-style thresholds of
// the fp64 algorithm sit below fp32 rounding noise of the cost, so the solver
// additionally stops (a) the line search once the directional derivative has
// dropped by 1e-5 relative to its value at alpha = 0 and (b) the Newton loop
// after a full step (alpha ~ 1) that left the active set unchanged -- at that
// point the iterate is the exact minimiser of the current quadratic piece.
#define DMC_F32_RULES 1
#endif
#endif

#ifndef DMC_MODEL_HEADER
#error "build with -DDMC_MODEL_HEADER=\"<generated model header>\" (dm_control_amd/codegen.py)"
#endif
#include DMC_MODEL_HEADER

using namespace dmc_model;

// line-search evaluations per Newton iteration (experiments may lower it)
#ifndef DMC_LS_MAXIT
#define DMC_LS_MAXIT (DMC_F32_RULES ? 20 : 50)
#endif

#define R(x) ((real)(x))
#define DEV static __device__ __forceinline__
#define DEVN static __device__ __forceinline__

// ---------------------------------------------------------------------------
// Team mode (-DDMC_TEAM=64, big scenes only): the TEAM lanes of a wavefront
// advance ONE env together.  Every lane runs the whole step and keeps its own
// copy of the per-env vectors (positions, velocities, forces: identical in all
// lanes, so control flow stays uniform); the work on what is shared -- the
// matrices, constraint rows and contact records in the HBM workspace -- is
// split over the lanes: each shared word has one writer per phase, phases end
// with tsync().  Sums over lanes are butterflies, so every lane ends with the
// bitwise-identical total.  (One env per lane leaves a 1024-pitch batch with 16
// wavefronts on a 1024-SIMD chip; a wavefront per pitch fills it.)
// ---------------------------------------------------------------------------
#ifdef DMC_TEAM
constexpr int TEAM = DMC_TEAM;
static_assert(TEAM >= 2 && TEAM <= 64 && (TEAM & (TEAM - 1)) == 0, "team = power of two <= 64");
#ifdef DMC_COOP_BUILD
#error "team mode belongs to dmc_kernels.hip"
#endif
#else
constexpr int TEAM = 1;
#endif
constexpr bool TEAMED = TEAM > 1;
// the lanes of a team in groups: a group per kinematic tree at a time (its first
// lane runs the tree's recursions).  Group g = the lanes g, g + NGROUPS, ...: the
// first lanes of the groups are neighbours, so what they keep in scratch (dword
// interleaved over the lanes of a wavefront) shares memory sectors.
constexpr int NGROUPS = TEAM >= 8 ? 4 : (TEAM >= 2 ? TEAM/2 : 1);
constexpr int LANES_PER_GROUP = TEAM/NGROUPS;

#if defined(DMC_TEAM) && !defined(DMC_HOST_SHIM)
DEV int tlane() { return (int)(threadIdx.x & (TEAM - 1)); }
// phase boundary: the lanes of a wavefront run in lock step and the memory
// pipeline keeps one wavefront's accesses in order, so what has to be stopped
// is the compiler moving accesses across the boundary
DEV void tsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <class T> DEV T txor(T x, int m) { return __shfl_xor(x, m, TEAM); }
template <class T> DEV T tup(T x, int d) { return __shfl_up(x, d, TEAM); }
DEV int tget_bits(int x, int src) {      // `src` uniform
  if (TEAM == 64) return __builtin_amdgcn_readlane(x, src);
  return __shfl(x, src, TEAM);
}
DEV int tget(int x, int src) { return tget_bits(x, src); }
DEV unsigned tget(unsigned x, int src) { return (unsigned)tget_bits((int)x, src); }
DEV float tget(float x, int src) { return __int_as_float(tget_bits(__float_as_int(x), src)); }
DEV double tget(double x, int src) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)tget_bits((int)(unsigned)(b & 0xffffffffLL), src);
  const unsigned hi = (unsigned)tget_bits((int)(unsigned)((unsigned long long)b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int CTRL>
DEV int tdpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
template <int CTRL> DEV int tdpp(int x) { return tdpp_i<CTRL>(x); }
template <int CTRL> DEV float tdpp(float x) { return __int_as_float(tdpp_i<CTRL>(__float_as_int(x))); }
template <int CTRL> DEV double tdpp(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)tdpp_i<CTRL>((int)(unsigned)(b & 0xffffffffLL));
  const unsigned hi = (unsigned)tdpp_i<CTRL>((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <class T>
DEV T tsum(T x) {
  if (TEAM == 64) {     // inside a 16-lane row: DPP moves; the four row totals: readlane
    x += tdpp<0xB1>(x);      // xor 1
    x += tdpp<0x4E>(x);      // xor 2
    x += tdpp<0x141>(x);     // half-row mirror
    x += tdpp<0x140>(x);     // row mirror
    const T r0 = tget(x, 0), r1 = tget(x, 16), r2 = tget(x, 32), r3 = tget(x, 48);
    return (r0 + r1) + (r2 + r3);
  }
  for (int m = TEAM/2; m > 0; m >>= 1) x += txor(x, m);
  return x;
}
#elif defined(DMC_TEAM)      // host shim: one OS thread per lane (tests/host_shim/shim_coop.h)
DEV int tlane() { return shim_lane(); }
DEV void tsync() { gsync(); }
template <class T> DEV T txor(T x, int m) { return gxor(x, m); }
template <class T> DEV T tup(T x, int d) { return gup(x, d); }
template <class T> DEV T tget(T x, int src) { return gget(x, src); }
template <class T>
DEV T tsum(T x) {
  for (int m = TEAM/2; m > 0; m >>= 1) x += txor(x, m);
  return x;
}
#else
DEV int tlane() { return 0; }
DEV void tsync() {}
template <class T> DEV T txor(T x, int) { return x; }
template <class T> DEV T tup(T x, int) { return x; }
template <class T> DEV T tget(T x, int) { return x; }
template <class T> DEV T tsum(T x) { return x; }
#endif
DEV int tmin(int x) {
  for (int m = TEAM/2; m > 0; m >>= 1) { const int y = txor(x, m); x = y < x ? y : x; }
  return x;
}
DEV unsigned tor(unsigned x) {
  for (int m = TEAM/2; m > 0; m >>= 1) x |= txor(x, m);
  return x;
}
DEV bool tany(bool b) { return tor(b ? 1u : 0u) != 0; }
// exclusive prefix sum in lane order + team total
DEV int tscan(int x, int& total) {
  int v = x;
  for (int d = 1; d < TEAM; d <<= 1) {
    const int t = tup(v, d);
    if (tlane() >= d) v += t;
  }
  total = tget(v, TEAM - 1);
  return v - x;
}

constexpr int NM = NV*(NV + 1)/2;      // packed lower triangle
constexpr int LANES = 64;               // envs per workgroup = one wavefront
// Packed nv x nv matrices (M, its factor, the Newton Hessian, Euler's M + h D)
// are per-lane arrays -- registers in the unrolled build, private memory in the
// rolled one -- while they fit: a lane's private segment ends at 128 KB on
// gfx9.  Scenes with several walkers (a 2v2 soccer pitch: nv 254, 32385
// entries per matrix) keep the four of them in the HBM workspace instead,
// [entry][env] like the overflow rows (one-env-per-lane kernel only).
#ifndef DMC_MAT_PRIVATE_BYTES
#define DMC_MAT_PRIVATE_BYTES 16384
#endif
#ifdef DMC_COOP_BUILD
constexpr bool MAT_IN_WS = false;
#else
constexpr bool MAT_IN_WS = (long long)NM*(long long)sizeof(real) > DMC_MAT_PRIVATE_BYTES;
#endif
// The two 18-element clears of a free joint's translational cdof / cdof_dot must
// stay loops in the generic builds: with the backend allowed to unroll them FULLY
// (trip count 18; `-unroll-full-max-count=16` is enough to avoid it) the 2v2
// pitch build returned a wrong mass matrix on gfx950 -- no spills involved, host
// build clean under ASan/UBSan (profiles/r03_unroll_18_*).
// -DDMC_UNROLL_18=1 restores the faulty form (tools/debug/pitch_tiers.py).
#if DMC_GENERIC_BUILD && !defined(DMC_UNROLL_18) && !defined(DMC_HOST_SHIM)
#define DMC_KEEP_ROLLED _Pragma("nounroll")
#else
#define DMC_KEEP_ROLLED
#endif
constexpr int MAT_REGS = MAT_IN_WS ? 1 : (NM > 0 ? NM : 1);
constexpr int NVX = NV > 0 ? NV : 1;
constexpr int NUX = NU > 0 ? NU : 1;
constexpr int NQX = NQ > 0 ? NQ : 1;
constexpr int NTDX = NTASKDATA > 0 ? NTASKDATA : 1;
constexpr real MAXVAL = R(1e10);

enum { JNT_FREE = 0, JNT_BALL = 1, JNT_SLIDE = 2, JNT_HINGE = 3 };
enum { GEOM_PLANE = 0, GEOM_SPHERE = 2, GEOM_CAPSULE = 3, GEOM_BOX = 6 };
enum { DSBL_CONSTRAINT = 1 << 0, DSBL_LIMIT = 1 << 3, DSBL_CONTACT = 1 << 4,
       DSBL_PASSIVE = 1 << 5, DSBL_GRAVITY = 1 << 6, DSBL_CLAMPCTRL = 1 << 7,
       DSBL_WARMSTART = 1 << 8, DSBL_ACTUATION = 1 << 10 };
// warn_mask bits follow mjtWarning order (engine.py:322-330)
enum { WARN_INERTIA = 1, WARN_CONTACTFULL = 2, WARN_CNSTRFULL = 4,
       WARN_BADQPOS = 16, WARN_BADQVEL = 32, WARN_BADQACC = 64,
       WARN_BADCTRL = 128 };
enum { TASK_NONE = 0, TASK_CARTPOLE = 1, TASK_CHEETAH = 2, TASK_HUMANOID = 3,
       TASK_WALKER = 4, TASK_PENDULUM = 5, TASK_ACROBOT = 6, TASK_HOPPER = 7,
       TASK_REACHER = 8, TASK_POINTMASS = 9 };

#define DMC_REALPTR real*
#define DMC_CREALPTR const real*
#include "dmc_args.h"

// Layout of the 2-D state fields in HBM.  One env per lane: [k][env], so a
// wave's loads are unit-stride over envs.  One env per group of lanes
// (dmc_coop.hip): [env][k], so a group's loads are unit-stride over k -- with
// [k][env] it would touch one 32-byte sector per word (measured: 47 MB moved
// per humanoid launch for 8 MB of state).  The C ABI presents [k][env] either
// way (dmc_api.cpp transposes on the host side of read / set_state).
#ifdef DMC_COOP_BUILD
#define DMC_ENV_MAJOR 1
#else
#define DMC_ENV_MAJOR 0
#endif
static __device__ __forceinline__ long long sidx(int k, long long e, long long n, int K) {
  return DMC_ENV_MAJOR ? e*K + k : k*n + e;
}

// ---------------------------------------------------------------------------
// math helpers
// ---------------------------------------------------------------------------
DEV real rsqrt_(real x) { return R(1)/sqrt(x); }
DEV real dot3(const real* a, const real* b) {
  return a[0]*b[0] + a[1]*b[1] + a[2]*b[2];
}
DEV void cross3(real* r, const real* a, const real* b) {
  real x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2],
       z = a[0]*b[1] - a[1]*b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
DEV real normalize3(real* a) {
  real n = sqrt(dot3(a, a));
  if (n < DMC_MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; }
  else { real s = R(1)/n; a[0] *= s; a[1] *= s; a[2] *= s; }
  return n;
}
DEV void normalize4(real* q) {
  real n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < DMC_MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else { real s = R(1)/n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
}
DEV void mulquat(real* r, const real* a, const real* b) {
  real w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  real x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  real y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  real z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
DEV void quat2mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w*w + x*x - y*y - z*z; m[1] = 2*(x*y - w*z); m[2] = 2*(x*z + w*y);
  m[3] = 2*(x*y + w*z); m[4] = w*w - x*x + y*y - z*z; m[5] = 2*(y*z - w*x);
  m[6] = 2*(x*z - w*y); m[7] = 2*(y*z + w*x); m[8] = w*w - x*x - y*y + z*z;
}
DEV void mulmatvec3(real* r, const real* m, const real* v) {
  real x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  real y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  real z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
DEV void rotvecquat(real* r, const real* v, const real* q) {
  real m[9];
  quat2mat(m, q);
  mulmatvec3(r, m, v);
}
DEV void axisangle2quat(real* q, const real* axis, real angle) {
  real s, c;
#ifdef DMC_REAL_IS_DOUBLE
  sincos(angle*R(0.5), &s, &c);
#else
  sincosf(angle*R(0.5), &s, &c);
#endif
  q[0] = c; q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
DEV void quat_integrate(real* q, const real* w, real h) {
  real ax[3] = {w[0], w[1], w[2]};
  real n = normalize3(ax);
  if (n < DMC_MINVAL) return;
  real dq[4], r[4];
  axisangle2quat(dq, ax, h*n);
  mulquat(r, q, dq);
  normalize4(r);
  q[0] = r[0]; q[1] = r[1]; q[2] = r[2]; q[3] = r[3];
}
DEV real clampr(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
// |x| > MAXVAL, +-inf or NaN, decided on the bit pattern: the build uses
// -ffinite-math-only (so that the structural zeros of planar models fold away),
// under which a floating-point comparison may not be relied on to catch NaN.
// For IEEE numbers the magnitude bits order like the magnitudes, and every
// inf / NaN pattern lies above every finite one.
#ifdef DMC_HOST_SHIM
DEV bool bad(real x) { return !(x <= MAXVAL && x >= -MAXVAL); }
#elif defined(DMC_REAL_IS_DOUBLE)
DEV bool bad(double x) {
  return ((unsigned long long)__double_as_longlong(x) & 0x7fffffffffffffffULL) >
         (unsigned long long)__double_as_longlong(1e10);
}
#else
DEV bool bad(float x) {
  return ((unsigned)__float_as_int(x) & 0x7fffffffu) > (unsigned)__float_as_int(1e10f);
}
#endif

// spatial vectors [angular, linear] about the subtree-root centre of mass
DEV void cross_motion(real* r, const real* v, const real* s) {
  real a[3], b[3], c[3];
  cross3(a, v, s); cross3(b, v, s + 3); cross3(c, v + 3, s);
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2];
  r[3] = b[0] + c[0]; r[4] = b[1] + c[1]; r[5] = b[2] + c[2];
}
DEV void cross_force(real* r, const real* v, const real* f) {
  real a[3], b[3], c[3];
  cross3(a, v, f); cross3(b, v + 3, f + 3); cross3(c, v, f + 3);
  r[0] = a[0] + b[0]; r[1] = a[1] + b[1]; r[2] = a[2] + b[2];
  r[3] = c[0]; r[4] = c[1]; r[5] = c[2];
}
DEV void mul_inert_vec(real* r, const real* i, const real* v) {
  r[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  r[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  r[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  r[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  r[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  r[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
DEV real dot6(const real* a, const real* b) {
  return a[0]*b[0] + a[1]*b[1] + a[2]*b[2] + a[3]*b[3] + a[4]*b[4] + a[5]*b[5];
}
DEV int tri(int i, int j) { return i*(i + 1)/2 + j; }   // i >= j

// packed symmetric nv x nv matrix in a per-lane register array
struct RegMat {
  real* v;
  __device__ __forceinline__ real get(int i) const { return v[i]; }
  __device__ __forceinline__ void set(int i, real x) const { v[i] = x; }
};

// y = A x for packed symmetric A
template <class Mat>
DEV void symv(real* y, const Mat& A, const real* x) {
  DMC_UNROLL
  for (int i = 0; i < NV; i++) y[i] = 0;
  // one pass over the stored triangle: entry (i, j) feeds y[i] and y[j]
  DMC_UNROLL
  for (int i = 0; i < NV; i++) {
    DMC_UNROLL
    for (int j = 0; j <= i; j++) {
      const real a = A.get(tri(i, j));
      y[i] += a*x[j];
      if (j != i) y[j] += a*x[i];
    }
  }
}
// in-place packed Cholesky A = L L^T; the diagonal slots hold 1/L_jj so the
// triangular solves multiply instead of divide; returns clamped pivots
template <class Mat>
DEV int chol_factor_t(const Mat& A) {
  int nbad = 0;
  DMC_UNROLL
  for (int j = 0; j < NV; j++) {
    real rowj[NVX];               // row j of L, columns < j, read once
    real s = A.get(tri(j, j));
    DMC_UNROLL
    for (int k = 0; k < j; k++) { rowj[k] = A.get(tri(j, k)); s -= rowj[k]*rowj[k]; }
    if (!(s >= DMC_MINVAL)) { s = DMC_MINVAL; nbad++; }
    const real inv = rsqrt_(s);
    A.set(tri(j, j), inv);
    DMC_UNROLL
    for (int i = j + 1; i < NV; i++) {
      real t = A.get(tri(i, j));
      DMC_UNROLL
      for (int k = 0; k < j; k++) t -= A.get(tri(i, k))*rowj[k];
      A.set(tri(i, j), t*inv);
    }
  }
  return nbad;
}
template <class Mat>
DEV int chol_factor(const Mat& A) { return chol_factor_t(A); }

template <class Mat>
DEV void chol_solve(real* x, const Mat& L) {
  DMC_UNROLL
  for (int i = 0; i < NV; i++) {
    real s = x[i];
    DMC_UNROLL
    for (int k = 0; k < i; k++) s -= L.get(tri(i, k))*x[k];
    x[i] = s*L.get(tri(i, i));
  }
  DMC_UNROLL
  for (int i = NV - 1; i >= 0; i--) {
    real s = x[i];
    DMC_UNROLL
    for (int k = i + 1; k < NV; k++) s -= L.get(tri(k, i))*x[k];
    x[i] = s*L.get(tri(i, i));
  }
}

// Envelope ("skyline") forms for big scenes (MAT_IN_WS): row i of the matrix is
// zero left of column lo(i) -- M is block diagonal over the kinematic trees, the
// Hessian couples two trees only where a contact does -- and a Cholesky factor
// has no fill outside that envelope, so every loop over a row starts at lo(i).
// The skipped terms are exact zeros: same values as the dense loops.
struct LoTree {      // the envelope of M: the first dof of the row's kinematic tree
  __device__ __forceinline__ int operator()(int i) const { return dof_treeroot[i]; }
};
struct LoArr {
  const int* p;
  __device__ __forceinline__ int operator()(int i) const { return p[i]; }
};
// The matrices of a big scene live in HBM: a lone wave hides that latency only
// if several loads are in flight, and the generic builds compile with the loop
// unroller off -- so the inner loops are blocked by eight by hand: eight loads
// first, then the arithmetic IN THE ORIGINAL ORDER (same rounding as the plain
// loop).
template <class Mat>
DEV real env_dot_sub(real t, const Mat& A, int base, const real* v, int k0, int k1) {
  int k = k0;
  for (; k + 32 <= k1; k += 32) {        // 32 HBM loads in flight
    real a[32];
    _Pragma("unroll") for (int u = 0; u < 32; u++) a[u] = A.get(base + k + u);
    _Pragma("unroll") for (int u = 0; u < 32; u++) t -= a[u]*v[k + u];
  }
  for (; k + 8 <= k1; k += 8) {
    real a[8];
    _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = A.get(base + k + u);
    _Pragma("unroll") for (int u = 0; u < 8; u++) t -= a[u]*v[k + u];
  }
  for (; k < k1; k++) t -= A.get(base + k)*v[k];
  return t;
}
template <class Mat, class Lo>
DEV void symv_env(real* y, const Mat& A, const real* x, const Lo& lo) {
  for (int i = 0; i < NV; i++) y[i] = 0;
  for (int i = 0; i < NV; i++) {
    const int base = tri(i, 0);
    const real xi = x[i];
    real yi = y[i];
    int j = lo(i);
    for (; j + 8 <= i; j += 8) {           // strictly below the diagonal
      real a[8];
      _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = A.get(base + j + u);
      _Pragma("unroll") for (int u = 0; u < 8; u++) { yi += a[u]*x[j + u]; y[j + u] += a[u]*xi; }
    }
    for (; j < i; j++) {
      const real a = A.get(base + j);
      yi += a*x[j];
      y[j] += a*xi;
    }
    y[i] = yi + A.get(base + i)*xi;
  }
}
// hi[j]: last row whose envelope reaches column j (>= j), from lo[]: the rows of
// the factorisation's column sweep -- scanning all NV rows per column with a
// table look-up each was most of its time
DEV void env_last_rows(int* hi, const int* lo) {
  for (int j = 0; j < NV; j++) hi[j] = j;
  for (int i = 0; i < NV; i++) { const int l = lo[i]; if (hi[l] < i) hi[l] = i; }
  for (int j = 1; j < NV; j++) if (hi[j] < hi[j - 1]) hi[j] = hi[j - 1];
}
template <class Mat>
DEV int chol_factor_env(const Mat& A, const int* lo, const int* hi) {
  int nbad = 0;
  for (int j = 0; j < NV; j++) {
    real rowj[NVX];
    const int lj = lo[j], bj = tri(j, 0);
    real s = A.get(bj + j);
    int k = lj;
    for (; k + 8 <= j; k += 8) {
      real a[8];
      _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = A.get(bj + k + u);
      _Pragma("unroll") for (int u = 0; u < 8; u++) { rowj[k + u] = a[u]; s -= a[u]*a[u]; }
    }
    for (; k < j; k++) { rowj[k] = A.get(bj + k); s -= rowj[k]*rowj[k]; }
    if (!(s >= DMC_MINVAL)) { s = DMC_MINVAL; nbad++; }
    const real inv = rsqrt_(s);
    A.set(bj + j, inv);
    const int last = hi[j];
    for (int i = j + 1; i <= last; i++) {
      const int li = lo[i];
      if (li > j) continue;
      const int bi = tri(i, 0);
      const real t = env_dot_sub(A.get(bi + j), A, bi, rowj, li > lj ? li : lj, j);
      A.set(bi + j, t*inv);
    }
  }
  return nbad;
}
template <class Mat, class Lo>
DEV void chol_solve_env(real* x, const Mat& L, const Lo& lo) {
  for (int i = 0; i < NV; i++) {
    const int bi = tri(i, 0);
    x[i] = env_dot_sub(x[i], L, bi, x, lo(i), i)*L.get(bi + i);
  }
  for (int i = NV - 1; i >= 0; i--) {     // row i of L pushes x[i] into the earlier entries
    const int bi = tri(i, 0);
    const real xi = x[i]*L.get(bi + i);
    x[i] = xi;
    int k = lo(i);
    for (; k + 8 <= i; k += 8) {
      real a[8];
      _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = L.get(bi + k + u);
      _Pragma("unroll") for (int u = 0; u < 8; u++) x[k + u] -= a[u]*xi;
    }
    for (; k < i; k++) x[k] -= L.get(bi + k)*xi;
  }
}
// dst = src inside dst's envelope `lo`; src is zero left of `slo`
template <class Mat, class Lo, class SLo>
DEV void copy_env(const Mat& dst, const Mat& src, const Lo& lo, const SLo& slo) {
  for (int i = 0; i < NV; i++) {
    const int si = slo(i), bi = tri(i, 0);
    int j = lo(i);
    for (; j < si && j <= i; j++) dst.set(bi + j, R(0));
    for (; j + 8 <= i + 1; j += 8) {
      real a[8];
      _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = src.get(bi + j + u);
      _Pragma("unroll") for (int u = 0; u < 8; u++) dst.set(bi + j + u, a[u]);
    }
    for (; j <= i; j++) dst.set(bi + j, src.get(bi + j));
  }
}

// ---------------------------------------------------------------------------
// per-lane working set of one environment
// ---------------------------------------------------------------------------
// Team mode: what one tree's recursions produce and another stage reads across
// trees (body frames, dof axes, the force and acceleration vectors) is SHARED --
// one copy in the team's LDS, the members below are pointers into it -- and the
// recursions of a tree run on one lane (`rb0` ...: that tree's ranges), the four
// walkers of a pitch on four lanes at once.
#ifdef DMC_TEAM
#define DMC_SHARED(name, n) real* name
#define BODY_LO(E) ((E).rb0)
#define BODY_LO0(E) ((E).rb0)
#define BODY_HI(E) ((E).rb1)
#define JNT_LO(E) ((E).rj0)
#define JNT_HI(E) ((E).rj1)
#define DOF_LO(E) ((E).rd0)
#define DOF_HI(E) ((E).rd1)
#define ACT_LO(E) ((E).ra0)
#define ACT_HI(E) ((E).ra1)
#else
#define DMC_SHARED(name, n) real name[n]
#define BODY_LO(E) 1
#define BODY_LO0(E) 0
#define BODY_HI(E) NBODY
#define JNT_LO(E) 0
#define JNT_HI(E) NJNT
#define DOF_LO(E) 0
#define DOF_HI(E) NV
#define ACT_LO(E) 0
#define ACT_HI(E) NU
#endif
struct Env {
  DMC_SHARED(qpos, NQ > 0 ? NQ : 1); DMC_SHARED(qvel, NVX); DMC_SHARED(warm, NVX);
  real ctrl[NUX];
#ifdef DMC_STATE_COMP
  real qpos_lo[NQ > 0 ? NQ : 1], qvel_lo[NVX];   // low words of the fp64 state
#endif
#ifdef DMC_TEAM
  int rb0, rb1, rj0, rj1, rd0, rd1, ra0, ra1;   // the bodies / joints / dofs / actuators this lane's recursions cover
#endif
  DMC_SHARED(xpos, NBODY*3); DMC_SHARED(xquat, NBODY*4); DMC_SHARED(xmat, NBODY*9);
  // (big scenes, MAT_IN_WS: the inertial frames are recomputed where com_pos
  // needs them instead of being kept -- the fp64 frame must stay below 128 KB)
  real xipos[NBODY*3], ximat[MAT_IN_WS ? 9 : NBODY*9];
  // (xanchor ... cvel: tree-local, but kept in LDS words that are idle while the
  // recursions run -- scratch latency was most of that phase)
  DMC_SHARED(xanchor, (NJNT > 0 ? NJNT : 1)*3); DMC_SHARED(xaxis, (NJNT > 0 ? NJNT : 1)*3);
  DMC_SHARED(subtree_com, NBODY*3);
  real cinert[NBODY*10];
  DMC_SHARED(cdof, NVX*6);
  DMC_SHARED(cdof_dot, NVX*6); DMC_SHARED(cvel, NBODY*6);
  real qM[MAT_REGS], qL[MAT_REGS];
  DMC_SHARED(qfrc_smooth, NVX); DMC_SHARED(qfrc_constraint, NVX);
  DMC_SHARED(qacc_smooth, NVX); DMC_SHARED(qacc, NVX);
  real subtree_linvel[NBODY*3];
  real touch[NTOUCH > 0 ? NTOUCH : 1];   // touch sensor readings (mj_sensorAcc)
  real taskdata[NTDX];                   // per-instance task parameters
#if defined(DMC_SOLVER_PROFILE) || defined(DMC_STEP_PROFILE)
  real prof[8];
#endif
  int ncon, nefc, nefc_limit, iters;
  int nmerged;     // pyramid edge pairs stored as one row (PLANAR_MERGE)
  // big scenes: envelope of the Newton Hessian (first column per row, last row per
  // column) and of M (the kinematic trees; filled once per launch)
  int hlo[MAT_IN_WS ? NVX : 1], hhi[MAT_IN_WS ? NVX : 1];
  int mlo[MAT_IN_WS ? NVX : 1], mhi[MAT_IN_WS ? NVX : 1];
  unsigned warn;
};

// Constraint rows: record r = [J(0..NV-1), D, aref, Jaref, Jv].  The first
// LDS_ROWS records of each lane live in LDS ([record word][lane]: every lane
// hits its own bank, conflict-free for any per-lane row index); records beyond
// that spill to the HBM workspace with the same [word][env] layout.
// Contact records ([pos3 n3 tangent-hint3 dist pair]) use the same two tiers.
// (big scenes, MAT_IN_WS: + the span [lo, hi] of the row's non-zero dofs; the
// passes over a row then touch only that span -- a joint-limit row is one dof, a
// foot-ground contact the 62 dofs of one walker, not the 254 of the pitch)
// (team mode: + the row's pending change of the Hessian, ROW_FLIP = +-D or 0)
constexpr int RW = NV + 4 + (MAT_IN_WS ? 2 : 0) + (TEAMED ? 1 : 0);
constexpr int CW = 11;
#ifndef DMC_LDS_BUDGET
#define DMC_LDS_BUDGET (128*1024)
#endif
#ifndef DMC_CON_LDS
#define DMC_CON_LDS 12
#endif
constexpr int REC_BYTES = LANES*(int)sizeof(real);    // one record word, all lanes
// (team mode: every row and contact record lives in the workspace, [record][word]
// contiguous per env; the team's LDS holds the matrix tile being worked on)
constexpr int REC_BUDGET = TEAMED ? 0 : DMC_LDS_BUDGET;
constexpr int LDS_CONS_WANT = DMC_CON_LDS < NCON_MAX ? DMC_CON_LDS : NCON_MAX;
constexpr int LDS_CONS_FIT = (REC_BUDGET/2)/(CW*REC_BYTES);   // <= half the budget
constexpr int LDS_CONS = LDS_CONS_WANT < LDS_CONS_FIT ? LDS_CONS_WANT : LDS_CONS_FIT;
constexpr int LDS_ROWS_FIT = (REC_BUDGET - LDS_CONS*CW*REC_BYTES)/(RW*REC_BYTES);
constexpr int LDS_ROWS = LDS_ROWS_FIT < NEFC_MAX ? LDS_ROWS_FIT : NEFC_MAX;
static_assert(LDS_CONS >= 0 && LDS_ROWS >= 0, "LDS budget arithmetic");
constexpr int GLB_ROWS = NEFC_MAX - LDS_ROWS > 0 ? NEFC_MAX - LDS_ROWS : 0;
constexpr int GLB_CONS = NCON_MAX - LDS_CONS > 0 ? NCON_MAX - LDS_CONS : 0;
enum { ROW_D = NV, ROW_AREF = NV + 1, ROW_JAR = NV + 2, ROW_JV = NV + 3,
       ROW_LO = NV + 4, ROW_HI = NV + 5, ROW_FLIP = NV + 6 };
// span of a row record: compile-time [0, NV) unless the record carries one
template <class Rec> DEV int row_lo(const Rec& rec) { return MAT_IN_WS ? (int)rec.get(ROW_LO) : 0; }
template <class Rec> DEV int row_hi(const Rec& rec) { return MAT_IN_WS ? (int)rec.get(ROW_HI) : NV - 1; }

struct LdsRow {
  real* p;
  __device__ __forceinline__ real get(int k) const { return p[k*LANES]; }
  __device__ __forceinline__ void set(int k, real v) const { p[k*LANES] = v; }
};
struct GlbRow {
  real* p; long long n;
  __device__ __forceinline__ real get(int k) const { return p[k*n]; }
  __device__ __forceinline__ void set(int k, real v) const { p[k*n] = v; }
};
// team mode: the Jacobian part of a row is contiguous ([row][dof]: a team reads a
// row with unit-stride loads), the scalar words are kept [word][row], so the
// passes that run one row per lane (line search) read them with unit stride too
struct TeamRow {
  real* j; real* sc;
  __device__ __forceinline__ real get(int k) const { return k < NV ? j[k] : sc[(k - NV)*NEFC_MAX]; }
  __device__ __forceinline__ void set(int k, real v) const {
    if (k < NV) j[k] = v; else sc[(k - NV)*NEFC_MAX] = v;
  }
};
// workspace words per env: overflow rows, overflow contacts, then
// (-DDMC_STATE_COMP) the low words of the fp64 state: qpos/qvel are then
// carried between steps as fp64 values split into the public fp32 field (high
// word) and a low word kept here together with the high word it belongs to (a
// field overwritten from outside -- set_state, reset -- no longer matches its
// tag and the low word is dropped)
constexpr int WS_COMP = GLB_ROWS*RW + GLB_CONS*CW;
#ifdef DMC_STATE_COMP
constexpr int WS_MAT = WS_COMP + 2*(NQ + NV);
#else
constexpr int WS_MAT = WS_COMP;
#endif
constexpr long long WS_GEOM = (long long)WS_MAT + (MAT_IN_WS ? 4LL*NM : 0LL);   // geom-pose mirror
constexpr long long WS_WORDS_LL = WS_GEOM + (MAT_IN_WS ? 12LL*NGEOM : 0LL);
static_assert(WS_WORDS_LL < (1LL << 31), "workspace words per env");
constexpr int WS_WORDS = (int)WS_WORDS_LL;
// a packed matrix in the workspace: entry i of env e at p[i*n]
struct GlbMat {
  real* p; long long n;
  __device__ __forceinline__ real get(int i) const { return p[(long long)i*n]; }
  __device__ __forceinline__ void set(int i, real x) const { p[(long long)i*n] = x; }
};
template <bool B, class T, class F> struct pick_ { typedef T type; };
template <class T, class F> struct pick_<false, T, F> { typedef F type; };
typedef pick_<MAT_IN_WS, GlbMat, RegMat>::type LaneMat;   // where a lane's matrices live
enum { MAT_M = 0, MAT_L = 1, MAT_H = 2, MAT_A = 3 };
// team LDS: two vectors for hand-overs between lanes, one row segment, one
// matrix tile (a kinematic tree's diagonal block, row stride odd)
constexpr int TB = 64;                  // largest tree (dofs) a team build takes
constexpr int TSTR = TB + 1;
// (and the vectors of the dynamics: one copy per env instead of one per lane in scratch)
constexpr int TL_X = 0, TL_Q = TL_X + NVX, TL_MA = TL_Q + NVX,
              TL_MV = TL_MA + NVX, TL_FS = TL_MV + NVX, TL_FC = TL_FS + NVX, TL_QAS = TL_FC + NVX,
              TL_ROW = TL_QAS + NVX, TL_HLO = TL_ROW + (NVX > 4*TB ? NVX : 4*TB),
              TL_QPOS = TL_HLO + NVX, TL_QVEL = TL_QPOS + (NQ > 0 ? NQ : 1), TL_WARM = TL_QVEL + NVX,
              TL_PHASE = TL_WARM + NVX;
// What follows is used in two phases of a step that do not overlap.  Phase 2
// (linear algebra): the tile and the list of pending Hessian changes.
constexpr int TL_TILE = TL_PHASE, TL_FLIPS = TL_TILE + TB*TSTR, TL_FLIP1 = TL_FLIPS + 4*256,
              TL_END2 = TL_FLIP1 + 2*512;
// Phase 1 (frames, forces, constraint rows): the geom-pose mirror the narrowphase
// reads and the shared members of Env.
constexpr int NGX_ = NGEOM > 0 ? NGEOM : 1;
constexpr int TL_GEOM = TL_PHASE, TL_XPOS = TL_GEOM + 12*NGX_, TL_XQUAT = TL_XPOS + 3*NBODY,
              TL_XMAT = TL_XQUAT + 4*NBODY, TL_SCOM = TL_XMAT + 9*NBODY,
              TL_CDOF = TL_SCOM + 3*NBODY, TL_END1 = TL_CDOF + 6*NVX;
// While the tree recursions run, the solver's vectors, the row segment, the
// envelope and the geom mirror are idle: their words hold the recursions' temporaries.
// (where a model's sizes do not allow it, they get words of their own at the end)
constexpr int NJX_ = NJNT > 0 ? NJNT : 1;
constexpr int TL_EXTRA = TL_END1 > TL_END2 ? TL_END1 : TL_END2;
constexpr bool FIT_G = 6*NVX + 3*NJX_ <= 12*NGX_;        // cdof_dot, xaxis: the geom mirror's words
constexpr bool FIT_V = 6*NBODY <= TL_FS - TL_X;          // cvel: x, qacc, Ma, Mv
constexpr bool FIT_R = 3*NJX_ <= TL_QPOS - TL_FC;        // xanchor: fc, qacc_smooth, row segment, envelope
constexpr int TL_CDOFDOT = FIT_G ? TL_GEOM : TL_EXTRA, TL_XAXIS = TL_CDOFDOT + 6*NVX;
constexpr int TL_EX1 = FIT_G ? TL_EXTRA : TL_XAXIS + 3*NJX_;
constexpr int TL_CVEL = FIT_V ? TL_X : TL_EX1, TL_EX2 = FIT_V ? TL_EX1 : TL_EX1 + 6*NBODY;
constexpr int TL_XANCHOR = FIT_R ? TL_FC : TL_EX2, TL_EX3 = FIT_R ? TL_EX2 : TL_EX2 + 3*NJX_;
// what the segments of a tree hand to each other at their hubs (the body a segment
// hangs from): the hub's acceleration, and what the hanging segments add to its
// force and composite inertia
constexpr int NHUBX = NHUB > 0 ? NHUB : 1;
constexpr int HUB_WORDS = 22, HUB_CACC = 0, HUB_CFRC = 6, HUB_CRB = 12;
constexpr int TL_HUB = TL_EX3;
constexpr int TEAM_LDS_WORDS = TEAMED ? TL_HUB + HUB_WORDS*NHUBX : 1;
static_assert(!TEAMED || (long long)TEAM_LDS_WORDS*sizeof(real) <= 160*1024, "team LDS beyond a CU's");

template <bool T> struct WsRowT { typedef GlbRow type; };
template <> struct WsRowT<true> { typedef TeamRow type; };
typedef WsRowT<TEAMED>::type WsRow;

struct Work {
  real* lds;   // LDS base + lane (rows, then contact records); team mode: the team's LDS
  real* glb;   // workspace base + env (team mode: base of the env's contiguous block)
  long long nenv;   // stride between the words of a record (team mode: 1)
  __device__ __forceinline__ LdsRow lrow(int r) const { return LdsRow{lds + r*RW*LANES}; }
  template <bool T = TEAMED>
  __device__ __forceinline__ typename WsRowT<T>::type grow(int r) const {
    if constexpr (T) {
      return TeamRow{glb + (long long)r*NV, glb + (long long)NEFC_MAX*NV + r};
    } else {
      return GlbRow{glb + (long long)(r - LDS_ROWS)*RW*nenv, nenv};
    }
  }
  __device__ __forceinline__ GlbMat mat(int k) const {
    return GlbMat{glb + ((long long)WS_MAT + (long long)k*NM)*nenv, nenv};
  }
  __device__ __forceinline__ LdsRow lcon(int k) const {
    return LdsRow{lds + (LDS_ROWS*RW + k*CW)*LANES};
  }
  __device__ __forceinline__ GlbRow gcon(int k) const {
    return GlbRow{glb + ((long long)GLB_ROWS*RW + (long long)(k - LDS_CONS)*CW)*nenv, nenv};
  }
};
constexpr int LDS_WORDS = (LDS_ROWS*RW + LDS_CONS*CW > 0 ? LDS_ROWS*RW + LDS_CONS*CW : 1)*LANES;
// f(row handle) for rows [0, nefc): LDS tier first, then the HBM tier
template <class F>
static __device__ __forceinline__ void for_rows(const Work& W, int nefc, F&& f) {
  const int n1 = nefc < LDS_ROWS ? nefc : LDS_ROWS;
  for (int r = 0; r < n1; r++) f(W.lrow(r));
  if (LDS_ROWS < NEFC_MAX)
    for (int r = LDS_ROWS; r < nefc; r++) f(W.grow(r));
}
// The same in two stages: `load(row handle)` returns the words a row needs,
// `use(words)` consumes them; the words of row r + 1 are requested before row r
// is consumed, so that a lone wave overlaps the LDS round trip of the cheap
// passes (line search: 3 words and a dozen instructions per row) with work.
template <class L, class U>
static __device__ __forceinline__ void for_rows_ahead(const Work& W, int nefc, L&& load, U&& use) {
  const int n1 = nefc < LDS_ROWS ? nefc : LDS_ROWS;
  if (n1 > 0) {
    auto cur = load(W.lrow(0));
    for (int r = 1; r < n1; r++) {
      const auto nxt = load(W.lrow(r));
      use(cur);
      cur = nxt;
    }
    use(cur);
  }
  if (LDS_ROWS < NEFC_MAX)
    for (int r = LDS_ROWS; r < nefc; r++) use(load(W.grow(r)));
}

template <bool WS> struct MatsT {     // per-lane arrays
  static __device__ __forceinline__ RegMat M(Env& E, const Work&) { return RegMat{E.qM}; }
  static __device__ __forceinline__ RegMat L(Env& E, const Work&) { return RegMat{E.qL}; }
  static __device__ __forceinline__ RegMat local(real* a, const Work&, int) { return RegMat{a}; }
};
template <> struct MatsT<true> {       // HBM workspace
  static __device__ __forceinline__ GlbMat M(Env&, const Work& W) { return W.mat(MAT_M); }
  static __device__ __forceinline__ GlbMat L(Env&, const Work& W) { return W.mat(MAT_L); }
  static __device__ __forceinline__ GlbMat local(real*, const Work& W, int k) { return W.mat(k); }
};
typedef MatsT<MAT_IN_WS> Mats;

// ===========================================================================
// Team algebra (team mode, big scenes).  Packed matrices live in the env's
// workspace block with unit stride; a kinematic tree's diagonal block (<= TB
// dofs) is worked on as a tile in the team's LDS.  Vectors are SHARED (one
// copy in the team's LDS): a lane reads any entry, entry k is written by lane
// k mod TEAM (or by lane 0 where every lane holds the value), with a phase
// boundary before it is read.  An envelope is `hlo` (LDS ints: first column of
// every row) or, hlo == nullptr, the kinematic trees' (dof_treeroot); bit t of
// `coupled`: tree t has rows that start left of it (a contact between two trees).
// ===========================================================================
constexpr int KPL = (TB + TEAM - 1)/TEAM;       // tile columns (or rows) per lane
static_assert(!TEAMED || MAT_IN_WS, "team mode is for scenes whose matrices live in the workspace");
static_assert(!TEAMED || MAXTREEDOF <= TB, "team mode: a kinematic tree has at most TB dofs");
static_assert(!TEAMED || NDTREE <= 32, "team mode: one bit per tree");

#if defined(DMC_TEAM) && !defined(DMC_HOST_SHIM)
DEV unsigned long long tballot(bool b) {
  const unsigned long long w = __ballot(b);
  if (TEAM == 64) return w;
  return (w >> ((threadIdx.x/TEAM)*TEAM)) & ((1ull << TEAM) - 1);
}
DEV void tatomic_min(int* p, int v) { atomicMin(p, v); }
#elif defined(DMC_TEAM)
DEV unsigned long long tballot(bool b) {
  unsigned long long x = b ? 1ull << tlane() : 0ull;
  for (int m = TEAM/2; m > 0; m >>= 1) x |= txor(x, m);
  return x;
}
DEV void tatomic_min(int* p, int v) {
  int cur = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (v < cur && !__atomic_compare_exchange_n(p, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
}
#else
DEV unsigned long long tballot(bool b) { return b ? 1ull : 0ull; }
DEV void tatomic_min(int* p, int v) { if (v < *p) *p = v; }
#endif
DEV void tatomic_add(real* p, real v) {       // LDS; the order of the additions is not fixed
#if defined(DMC_TEAM) && !defined(DMC_HOST_SHIM)
  atomicAdd(p, v);
#elif defined(DMC_TEAM)
  typedef typename pick_<sizeof(real) == 8, unsigned long long, unsigned>::type bits;
  bits* q = reinterpret_cast<bits*>(p);
  bits cur = __atomic_load_n(q, __ATOMIC_RELAXED), nxt;
  do {
    real c; memcpy(&c, &cur, sizeof c);
    c += v; memcpy(&nxt, &c, sizeof c);
  } while (!__atomic_compare_exchange_n(q, &cur, nxt, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
#else
  *p += v;
#endif
}
DEV int tfirst_bit(unsigned long long m) { return __builtin_ctzll(m); }
DEV int tpopc(unsigned long long m) { return __builtin_popcountll(m); }
// first k >= lo that lane `tl` owns (k = tl mod TEAM)
DEV int towned_from(int lo, int tl) { return lo + ((tl - lo) & (TEAM - 1)); }

// tile <- A[s .. s+n-1][s .. row]; 32 rows in flight (the rows live in HBM; a lone
// wavefront hides that latency only with loads in flight)
template <class Mat>
DEV void tile_load(real* T, const Mat& A, int s, int n) {
  const int tl = tlane();
  constexpr int RB = KPL == 1 ? 32 : 16;        // rows per group (64 at once measured slower)
  for (int i0 = 0; i0 < n; i0 += RB) {
    real a[RB][KPL];
    _Pragma("unroll")
    for (int u = 0; u < RB; u++) {
      const int ii = i0 + u;
      _Pragma("unroll")
      for (int m = 0; m < KPL; m++) {
        const int k = tl + m*TEAM;
        a[u][m] = (ii < n && k <= ii) ? A.get(tri(s + ii, s + k)) : R(0);
      }
    }
    _Pragma("unroll")
    for (int u = 0; u < RB; u++) {
      const int ii = i0 + u;
      _Pragma("unroll")
      for (int m = 0; m < KPL; m++) {
        const int k = tl + m*TEAM;
        if (ii < n && k <= ii) T[ii*TSTR + k] = a[u][m];
      }
    }
  }
  tsync();
}
template <class Mat>
DEV void tile_store(const Mat& A, const real* T, int s, int n) {
  const int tl = tlane();
  for (int i0 = 0; i0 < n; i0 += 16) {          // sixteen rows' LDS reads in flight
    real v[16][KPL];
    _Pragma("unroll")
    for (int u = 0; u < 16; u++)
      _Pragma("unroll")
      for (int m = 0; m < KPL; m++) {
        const int ii = i0 + u, k = tl + m*TEAM;
        v[u][m] = (ii < n && k <= ii) ? T[ii*TSTR + k] : R(0);
      }
    _Pragma("unroll")
    for (int u = 0; u < 16; u++)
      _Pragma("unroll")
      for (int m = 0; m < KPL; m++) {
        const int ii = i0 + u, k = tl + m*TEAM;
        if (ii < n && k <= ii) A.set(tri(s + ii, s + k), v[u][m]);
      }
  }
  tsync();
}
// in place T = L L^T (lower part), the diagonal keeps the inverse pivots.  One
// row per lane; the update of a row runs eight columns at a time so that the LDS
// round trips of a block overlap.
// With a whole wavefront per env (TEAM = 64) the factorisation never touches
// memory: lane i keeps row i of the tile in registers, the pivot and the
// multipliers of a column travel by lane broadcast (straight-line code: register
// indices must be constants).
static __device__ __noinline__ int tile_factor_rows(real* T, int n) {     // (one copy: three call sites)
  const int i = tlane();
  real r[TB];
  _Pragma("unroll")
  for (int k = 0; k < TB; k++) r[k] = (i < n && k <= i) ? T[i*TSTR + k] : R(0);
  int nbad = 0;
  _Pragma("unroll")
  for (int j = 0; j < TB; j++) {
    if (j < n) {
      real d = tget(r[j], j);
      if (!(d >= DMC_MINVAL)) { d = DMC_MINVAL; nbad++; }
      const real inv = rsqrt_(d);
      const real lij = r[j]*inv;            // (meaningful in the lanes below the pivot)
      _Pragma("unroll")
      for (int k = j + 1; k < TB; k++) r[k] -= lij*tget(lij, k);   // (rows >= n are zero)
      r[j] = i == j ? inv : lij;
    }
  }
  _Pragma("unroll")
  for (int k = 0; k < TB; k++) if (i < n && k <= i) T[i*TSTR + k] = r[k];
  tsync();
  return nbad;
}
DEV int tile_factor(real* T, int n) {
  if (TEAM == TB) return tile_factor_rows(T, n);
  const int tl = tlane();
  int nbad = 0;
  for (int j = 0; j < n; j++) {
    real d = T[j*TSTR + j];
    if (!(d >= DMC_MINVAL)) { d = DMC_MINVAL; nbad++; }
    const real inv = rsqrt_(d);
    tsync();                                  // every lane has read the pivot
    for (int ii = j + 1 + tl; ii < n; ii += TEAM) T[ii*TSTR + j] *= inv;
    if (tl == 0) T[j*TSTR + j] = inv;
    tsync();
    for (int ii = j + 1 + tl; ii < n; ii += TEAM) {
      real* row = T + ii*TSTR;
      const real lij = row[j];
      int k = j + 1;
      for (; k + 8 <= ii + 1; k += 8) {
        real c[8], r[8];
        _Pragma("unroll") for (int u = 0; u < 8; u++) { c[u] = T[(k + u)*TSTR + j]; r[u] = row[k + u]; }
        _Pragma("unroll") for (int u = 0; u < 8; u++) row[k + u] = r[u] - lij*c[u];
      }
      for (; k <= ii; k++) row[k] -= lij*T[k*TSTR + j];
    }
    tsync();
  }
  return nbad;
}
// sum_k A[bi + k] B[bj + k], k in [k0, k1): 32 (then 8) loads of each in flight --
// the rows live in HBM and a lone wavefront only hides that latency with loads in flight
template <class Mat>
DEV real env_dot2(const Mat& A, int bi, int bj, int k0, int k1) {
  real t = 0;
  int k = k0;
  for (; k + 32 <= k1; k += 32) {
    real a[32], b[32];
    _Pragma("unroll") for (int u = 0; u < 32; u++) { a[u] = A.get(bi + k + u); b[u] = A.get(bj + k + u); }
    _Pragma("unroll") for (int u = 0; u < 32; u++) t += a[u]*b[u];
  }
  for (; k + 8 <= k1; k += 8) {
    real a[8], b[8];
    _Pragma("unroll") for (int u = 0; u < 8; u++) { a[u] = A.get(bi + k + u); b[u] = A.get(bj + k + u); }
    _Pragma("unroll") for (int u = 0; u < 8; u++) t += a[u]*b[u];
  }
  for (; k < k1; k++) t += A.get(bi + k)*A.get(bj + k);
  return t;
}
// rows of the tree [s, e] whose envelope starts left of it: their entries left of
// the tile, one row per lane
template <class Mat>
DEV void team_coupled_left(const Mat& F, const Mat& H, const int* hlo, int s, int e) {
  for (int i = s + tlane(); i <= e; i += TEAM) {
    const int li = hlo[i];
    if (li >= s) continue;
    const int bi = tri(i, 0);
    for (int j = li; j < s; j++) {
      const int lj = hlo[j], bj = tri(j, 0);
      const real t = H.get(bi + j) - env_dot2(F, bi, bj, li > lj ? li : lj, j);
      F.set(bi + j, t*F.get(bj + j));
    }
  }
  tsync();
}
// ... and what they take out of the tile
template <class Mat>
DEV void team_coupled_tile(real* T, const Mat& F, const int* hlo, int s, int e) {
  for (int i = s + tlane(); i <= e; i += TEAM) {
    const int li = hlo[i];
    if (li >= s) continue;
    for (int k = s; k <= i; k++) {
      const int lk = hlo[k];
      if (lk >= s) continue;
      T[(i - s)*TSTR + (k - s)] -= env_dot2(F, tri(i, 0), tri(k, 0), li > lk ? li : lk, s);
    }
  }
  tsync();
}
// The same two steps for a whole wavefront per env (TEAM = TB: one row of the
// tree per lane), without walking HBM rows: for every earlier tree the coupled
// rows reach, that tree's factor tile is staged in LDS, each coupled row takes
// its right-hand side from H into registers, is solved against the tile there
// (x L^T = h, straight-line code) and stored as its part of F; what the rows take
// out of the tree's own tile, G[i][c] = sum_j x_i[j] x_c[j], is summed in
// over the earlier trees (the other row's x by lane broadcast) in the words of F
// that the tree's tile will overwrite.  team_factor() subtracts it once the tile
// is in LDS.
template <class Mat>
DEV void team_coupled_rows(real* T, const Mat& F, const Mat& H, const int* hlo, int tu) {
  const int s = dtree_lo[tu], e = dtree_hi[tu];
  const int tl = tlane(), i = s + tl;
  const bool mine = i <= e && hlo[i <= e ? i : e] < s;
  const int li = mine ? hlo[i] : s;
  const unsigned long long cmask = tballot(mine);
  const int lmin = tmin(li);
  const int bi = tri(i <= e ? i : e, 0);
  bool first_tile = true;
  for (int tp = 0; tp < tu; tp++) {
    const int sp = dtree_lo[tp], ep = dtree_hi[tp], np = ep - sp + 1;
    if (ep < lmin) continue;
    tile_load(T, F, sp, np);
    real x[TB];
    _Pragma("unroll")
    for (int j = 0; j < TB; j++)
      x[j] = (mine && j < np && sp + j >= li) ? H.get(bi + sp + j) : R(0);
    _Pragma("unroll")
    for (int j = 0; j < TB; j++)
      if (j < np) {
        real a = x[j];
        _Pragma("unroll")
        for (int k = 0; k < j; k++) a -= x[k]*T[j*TSTR + k];
        x[j] = a*T[j*TSTR + j];
      }
    _Pragma("unroll")
    for (int j = 0; j < TB; j++)
      if (mine && j < np && sp + j >= li) F.set(bi + sp + j, x[j]);
    // G[il][c] += x_il . x_c: lane c has its own x, row il's comes by broadcast; the
    // sum is kept in F's words of the tree's tile (the broadcasts hide the load)
    unsigned long long m = cmask;
    while (m) {
      const int il = tfirst_bit(m);
      m &= m - 1;
      const int at = tri(s + il, s + tl);
      const bool keep = tl <= il;
      const real g = (keep && !first_tile) ? F.get(at) : R(0);
      real acc = 0;
      _Pragma("unroll") for (int j = 0; j < TB; j++) acc += x[j]*tget(x[j], il);
      if (keep) F.set(at, g + acc);
    }
    first_tile = false;
    tsync();                      // (the tile is reloaded in the next round)
  }
}
// ... and the subtraction, with the tree's tile in LDS
template <class Mat>
DEV void team_coupled_take(real* T, const Mat& F, const int* hlo, int s, int e) {
  const int tl = tlane(), i = s + tl;
  if (i <= e && hlo[i] < s)
    for (int c0 = 0; c0 <= tl; c0 += 16) {         // sixteen loads in flight
      real g[16];
      _Pragma("unroll")
      for (int u = 0; u < 16; u++) {
        const int c = c0 + u;
        g[u] = (c <= tl && hlo[s + (c <= tl ? c : 0)] < s) ? F.get(tri(i, s + c)) : R(0);
      }
      _Pragma("unroll")
      for (int u = 0; u < 16; u++) if (c0 + u <= tl) T[tl*TSTR + c0 + u] -= g[u];
    }
  tsync();
}
// One pending change of the Hessian: row `r` (dofs lo..hi) enters or leaves with
// weight w = +-D; a one-dof row (joint limit) carries w = +-D J^2 and is one
// diagonal entry.  Pass A of the solver lists them in LDS (in row order).
constexpr int NFLIP = 256, NFLIP1 = 512;    // beyond them: the rows are scanned (ROW_FLIP)
static_assert(TL_FLIP1 - TL_FLIPS == 4*NFLIP && TL_END2 - TL_FLIP1 == 2*NFLIP1, "team LDS layout");
// p: (row, lo, hi, w) of the rows with several dofs; p1: (dof, w) of the one-dof rows
struct FlipList { real* p; int n; real* p1; int n1; };        // n < 0: not listed
DEV void flip_apply_row(real* T, const real* seg, real w, int c0, int c1, int s) {
  const int tl = tlane();
  for (int j = c0; j <= c1; j++) {
    const real vj = seg[j - s];
    if (vj == 0) continue;
    const real sj = w*vj;
    for (int k = c0 + tl; k <= j; k += TEAM) {
      const real vk = seg[k - s];
      if (vk != 0) T[(j - s)*TSTR + (k - s)] += sj*vk;
    }
  }
}
// the listed changes that touch the tile [s, e]; four row segments in flight
DEV bool team_tile_flips_listed(real* T, const Work& W, const FlipList& L, int s, int e) {
  const int tl = tlane();
  real* seg = W.lds + TL_ROW;
  bool any = false;
  for (int f = tl; f < L.n1; f += TEAM) {          // the one-dof rows: one diagonal entry each
    const int d = (int)L.p1[2*f];
    if (d >= s && d <= e) { tatomic_add(T + (d - s)*TSTR + (d - s), L.p1[2*f + 1]); any = true; }
  }
  any = tany(any);
  tsync();
  int f = 0;
  while (f < L.n) {
    int ids[4], c0s[4], c1s[4], cnt = 0;
    real ws[4];
    for (; f < L.n && cnt < 4; f++) {
      const real* q = L.p + 4*f;
      const int lo = (int)q[1], hi = (int)q[2];
      if (hi < s || lo > e) continue;
      any = true;
      ids[cnt] = (int)q[0]; ws[cnt] = q[3];
      c0s[cnt] = lo > s ? lo : s; c1s[cnt] = hi < e ? hi : e;
      cnt++;
    }
    real v[4][KPL];
    _Pragma("unroll")
    for (int u = 0; u < 4; u++)
      _Pragma("unroll")
      for (int m = 0; m < KPL; m++) {
        const int k = s + tl + m*TEAM;
        v[u][m] = (u < cnt && k >= c0s[u] && k <= c1s[u]) ? W.grow(ids[u]).get(k) : R(0);
      }
    tsync();                       // (the one-dof updates above; the previous group's reads of seg)
    _Pragma("unroll")
    for (int u = 0; u < 4; u++)
      _Pragma("unroll")
      for (int m = 0; m < KPL; m++) {
        const int k = tl + m*TEAM;
        if (u < cnt && k < TB) seg[u*TB + k] = v[u][m];
      }
    tsync();
    for (int u = 0; u < cnt; u++) {
      flip_apply_row(T, seg + u*TB, ws[u], c0s[u], c1s[u], s);
      tsync();
    }
  }
  tsync();
  return any;
}
// the same from the rows' ROW_FLIP words (more changes than the list holds)
DEV bool team_tile_flips_scanned(real* T, const Work& W, int nefc, int s, int e) {
  const int tl = tlane();
  real* seg = W.lds + TL_ROW;
  bool any = false;
  for (int r0 = 0; r0 < nefc; r0 += TEAM) {
    const int r = r0 + tl;
    bool mine = false;
    if (r < nefc) {
      const auto rec = W.grow(r);
      mine = rec.get(ROW_FLIP) != 0 && (int)rec.get(ROW_HI) >= s && (int)rec.get(ROW_LO) <= e;
    }
    unsigned long long m = tballot(mine);
    while (m) {
      const int b = tfirst_bit(m);
      m &= m - 1;
      any = true;
      const auto rec = W.grow(r0 + b);
      const real Ds = rec.get(ROW_FLIP);
      const int lo = (int)rec.get(ROW_LO), hi = (int)rec.get(ROW_HI);
      if (lo == hi) {
        if (tl == 0) { const real v = rec.get(lo); T[(lo - s)*TSTR + (lo - s)] += Ds*v*v; }
        tsync();
        continue;
      }
      const int c0 = lo > s ? lo : s, c1 = hi < e ? hi : e;
      for (int k = c0 + tl; k <= c1; k += TEAM) seg[k - s] = rec.get(k);
      tsync();
      flip_apply_row(T, seg, Ds, c0, c1, s);
      tsync();
    }
  }
  return any;
}
// the changes LEFT of the tiles: rows whose dofs lie in two trees.  `first`: the
// coupled rows' left parts start from zero (M has nothing there).
template <class Mat>
DEV void team_cross_flips(const Mat& H, const Work& W, int nefc, const int* hlo, bool first) {
  const int tl = tlane();
  if (first) {
    for (int i = 0; i < NV; i++) {
      const int s = dof_treeroot[i];
      for (int k = hlo[i] + tl; k < s; k += TEAM) H.set(tri(i, k), R(0));
    }
    tsync();
  }
  real* seg = W.lds + TL_ROW;
  for (int r0 = 0; r0 < nefc; r0 += TEAM) {
    const int r = r0 + tl;
    bool mine = false;
    if (r < nefc) {
      const auto rec = W.grow(r);
      mine = rec.get(ROW_FLIP) != 0 &&
             (int)rec.get(ROW_LO) < dof_treeroot[(int)rec.get(ROW_HI)];
    }
    unsigned long long m = tballot(mine);
    while (m) {
      const int b = tfirst_bit(m);
      m &= m - 1;
      const auto rec = W.grow(r0 + b);
      const real Ds = rec.get(ROW_FLIP);
      const int lo = (int)rec.get(ROW_LO), hi = (int)rec.get(ROW_HI);
      for (int k = lo + tl; k <= hi; k += TEAM) seg[k] = rec.get(k);
      tsync();
      for (int j = lo; j <= hi; j++) {
        const real vj = seg[j];
        const int sj = dof_treeroot[j];
        if (vj == 0 || sj <= lo) continue;
        const real dj = Ds*vj;
        for (int k = lo + tl; k < sj; k += TEAM) {
          const real vk = seg[k];
          if (vk != 0) H.set(tri(j, k), H.get(tri(j, k)) + dj*vk);
        }
      }
      tsync();
    }
  }
}
// forward substitution, what the earlier trees contribute to the coupled rows of [s, e]
template <class Mat>
DEV void team_forward_coupled(real* x, const Mat& F, const int* hlo, int s, int e) {
  for (int i = s + tlane(); i <= e; i += TEAM) {
    const int li = hlo[i];
    if (li >= s) continue;
    const int bi = tri(i, 0);
    real acc = 0;
    int k = li;
    for (; k + 8 <= s; k += 8) {
      real a[8];
      _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = F.get(bi + k + u);
      _Pragma("unroll") for (int u = 0; u < 8; u++) acc += a[u]*x[k + u];
    }
    for (; k < s; k++) acc += F.get(bi + k)*x[k];
    x[i] -= acc;
  }
  tsync();
}
DEV void tile_forward(const real* T, real* x, int n);
// dst <- Cholesky factor of (src [+ h*damping on the diagonal] [+ pending row
// changes]), tree by tree.  `flips`: the Newton Hessian -- the changed tiles
// also go back to `H`, rows coupled to an earlier tree take their left parts from H.
// `xfwd`: a right-hand side (shared vector) that takes the forward substitution
// on the way (team_solve(..., forward_done = true) finishes it).
template <class Mat>
DEV int team_factor(const Work& W, const Mat& dst, const Mat& src, const int* hlo,
                    unsigned coupled, real damping_h, bool flips, const FlipList& L,
                    bool first, int nefc, const Mat& H, real* xfwd = nullptr) {
  real* T = W.lds + TL_TILE;
  const int tl = tlane();
  int nbad = 0;
  if (flips && coupled) team_cross_flips(H, W, nefc, hlo, first);
  for (int t = 0; t < NDTREE; t++) {
    const int s = dtree_lo[t], e = dtree_hi[t], n = e - s + 1;
    if (flips && !first && !coupled && L.n >= 0 && xfwd) {
      // a block that no listed change touches keeps its factor: only the
      // right-hand side passes through it
      bool touched = false;
      for (int f = tl; f < L.n1; f += TEAM) { const int d = (int)L.p1[2*f]; touched |= d >= s && d <= e; }
      for (int f = tl; f < L.n; f += TEAM) touched |= (int)L.p[4*f + 2] >= s && (int)L.p[4*f + 1] <= e;
      if (!tany(touched)) {
        tile_load(T, dst, s, n);
        tile_forward(T, xfwd + s, n);
        continue;
      }
    }
    const bool coupled_here = flips && ((coupled >> t) & 1u);
    // (a chain -- this tree coupled to one that is itself coupled further left -- is
    // left to the rows-in-HBM path: the staged solve assumes block-diagonal earlier factors)
    bool staged = coupled_here && TEAM == TB;
    if (staged)
      for (int tp = 0; tp < t; tp++) staged = staged && !((coupled >> tp) & 1u);
    if (staged) team_coupled_rows(T, dst, H, hlo, t);
    tile_load(T, src, s, n);
    if (damping_h != 0) {
      for (int ii = tl; ii < n; ii += TEAM) T[ii*TSTR + ii] += damping_h*R(dof_damping[s + ii]);
      tsync();
    }
    if (flips) {
      const bool changed = L.n >= 0 ? team_tile_flips_listed(T, W, L, s, e)
                                    : team_tile_flips_scanned(T, W, nefc, s, e);
      if (changed || first) tile_store(H, T, s, n);
      if (coupled_here) {
        if (staged) team_coupled_take(T, dst, hlo, s, e);
        else {                       // (a chain, or fewer lanes than rows: the rows are walked in HBM)
          team_coupled_left(dst, H, hlo, s, e);
          team_coupled_tile(T, dst, hlo, s, e);
        }
      }
    }
    nbad += tile_factor(T, n);
    if (xfwd) {       // the forward substitution of a right-hand side while the block is here
      if ((coupled >> t) & 1u) team_forward_coupled(xfwd, dst, hlo, s, e);
      tile_forward(T, xfwd + s, n);
    }
    tile_store(dst, T, s, n);
  }
  return nbad;
}
// Triangular solves with the tile in LDS: row ii = tl + m*TEAM of the right-hand
// side lives in a register of lane tl (one row per lane on the GPU), the pivot
// row's value reaches the others through a lane broadcast, and the column of the
// tile that the next step needs is requested one step ahead -- the dependence
// chain per column is one broadcast and one multiply-add.
DEV void tile_forward(const real* T, real* x, int n) {
  const int tl = tlane();
  real xv[KPL], dg[KPL], col[KPL];
  _Pragma("unroll")
  for (int m = 0; m < KPL; m++) {
    const int ii = tl + m*TEAM;
    xv[m] = ii < n ? x[ii] : R(0);
    dg[m] = ii < n ? T[ii*TSTR + ii] : R(1);
    col[m] = (ii < n && ii > 0) ? T[ii*TSTR] : R(0);
  }
  for (int j = 0; j < n; j++) {
    const int oj = j & (TEAM - 1), sj = j/TEAM;
    const real xj = tget(KPL == 1 ? xv[0] : xv[sj], oj)*tget(KPL == 1 ? dg[0] : dg[sj], oj);
    real nxt[KPL];
    _Pragma("unroll")
    for (int m = 0; m < KPL; m++) {
      const int ii = tl + m*TEAM;
      nxt[m] = (ii < n && ii > j + 1) ? T[ii*TSTR + j + 1] : R(0);
    }
    _Pragma("unroll")
    for (int m = 0; m < KPL; m++) {
      const int ii = tl + m*TEAM;
      if (ii == j) xv[m] = xj;
      else if (ii > j) xv[m] -= col[m]*xj;
      col[m] = nxt[m];
    }
  }
  _Pragma("unroll")
  for (int m = 0; m < KPL; m++) { const int ii = tl + m*TEAM; if (ii < n) x[ii] = xv[m]; }
  tsync();
}
DEV void tile_backward(const real* T, real* x, int n) {
  const int tl = tlane();
  real xv[KPL], dg[KPL], row[KPL];
  _Pragma("unroll")
  for (int m = 0; m < KPL; m++) {
    const int kk = tl + m*TEAM;
    xv[m] = kk < n ? x[kk] : R(0);
    dg[m] = kk < n ? T[kk*TSTR + kk] : R(1);
    row[m] = kk < n - 1 ? T[(n - 1)*TSTR + kk] : R(0);
  }
  for (int i = n - 1; i >= 0; i--) {
    const int oi = i & (TEAM - 1), si = i/TEAM;
    const real xi = tget(KPL == 1 ? xv[0] : xv[si], oi)*tget(KPL == 1 ? dg[0] : dg[si], oi);
    real nxt[KPL];
    _Pragma("unroll")
    for (int m = 0; m < KPL; m++) {
      const int kk = tl + m*TEAM;
      nxt[m] = (i > 0 && kk < i - 1) ? T[(i - 1)*TSTR + kk] : R(0);
    }
    _Pragma("unroll")
    for (int m = 0; m < KPL; m++) {
      const int kk = tl + m*TEAM;
      if (kk == i) xv[m] = xi;
      else if (kk < i) xv[m] -= row[m]*xi;
      row[m] = nxt[m];
    }
  }
  _Pragma("unroll")
  for (int m = 0; m < KPL; m++) { const int kk = tl + m*TEAM; if (kk < n) x[kk] = xv[m]; }
  tsync();
}
// x <- F^-T F^-1 x, tile by tile: the tree's block of the factor in LDS
template <class Mat>
DEV void team_solve(const Work& W, real* x, const Mat& F, const int* hlo, unsigned coupled,
                    bool forward_done = false) {
  real* T = W.lds + TL_TILE;
  const int tl = tlane();
  if (!forward_done)
  for (int t = 0; t < NDTREE; t++) {
    const int s = dtree_lo[t], e = dtree_hi[t], n = e - s + 1;
    tile_load(T, F, s, n);
    if ((coupled >> t) & 1u) team_forward_coupled(x, F, hlo, s, e);
    tile_forward(T, x + s, n);
  }
  for (int t = NDTREE - 1; t >= 0; t--) {
    const int s = dtree_lo[t], e = dtree_hi[t], n = e - s + 1;
    tile_load(T, F, s, n);
    tile_backward(T, x + s, n);
    if ((coupled >> t) & 1u) {            // the coupled rows push their values into the earlier trees
      for (int i = s; i <= e; i++) {
        const int li = hlo[i];
        if (li >= s) continue;
        const real xi = x[i];
        const int bi = tri(i, 0);
        for (int k = towned_from(li, tl); k < s; k += TEAM) x[k] -= F.get(bi + k)*xi;
      }
      tsync();
    }
  }
}
// y = M x for a matrix with the trees' envelope (block diagonal), tile by tile
template <class Mat>
DEV void team_symv(const Work& W, real* y, const Mat& A, const real* x) {
  real* T = W.lds + TL_TILE;
  const int tl = tlane();
  for (int t = 0; t < NDTREE; t++) {
    const int s = dtree_lo[t], n = dtree_hi[t] - s + 1;
    tile_load(T, A, s, n);
    for (int ii = tl; ii < n; ii += TEAM) {
      real acc = 0;
      int k = 0;
      for (; k + 8 <= n; k += 8) {
        real a[8], b[8];
        _Pragma("unroll") for (int u = 0; u < 8; u++) {
          const int kk = k + u;
          a[u] = kk <= ii ? T[ii*TSTR + kk] : T[kk*TSTR + ii];
          b[u] = x[s + kk];
        }
        _Pragma("unroll") for (int u = 0; u < 8; u++) acc += a[u]*b[u];
      }
      for (; k < n; k++) acc += (k <= ii ? T[ii*TSTR + k] : T[k*TSTR + ii])*x[s + k];
      y[s + ii] = acc;
    }
    tsync();
  }
}
// shared <- every lane's copy (all equal) / every lane's copy <- shared
DEV void team_put(real* shared, const real* mine) {
  for (int k = tlane(); k < NV; k += TEAM) shared[k] = mine[k];
  tsync();
}
DEV void team_take(real* mine, const real* shared) {
  for (int k = 0; k < NV; k++) mine[k] = shared[k];
  tsync();
}

// ---------------------------------------------------------------------------
// position stage: kinematics, centre-of-mass frame, composite inertia
// ---------------------------------------------------------------------------
// the world body's frame and velocity (team mode: shared words, written once
// per pass by lane 0 before the trees' lanes start)
DEV void world_frames(Env& E) {
  E.xpos[0] = E.xpos[1] = E.xpos[2] = 0;
  E.xquat[0] = 1; E.xquat[1] = E.xquat[2] = E.xquat[3] = 0;
  quat2mat(E.xmat, E.xquat);
  for (int k = 0; k < 6; k++) E.cvel[k] = 0;
}
DEV void kinematics(Env& E) {
  if (!TEAMED) {
    E.xpos[0] = E.xpos[1] = E.xpos[2] = 0;
    E.xquat[0] = 1; E.xquat[1] = E.xquat[2] = E.xquat[3] = 0;
    quat2mat(E.xmat, E.xquat);
  }
  DMC_UNROLL
  for (int k = 0; k < 3; k++) E.xipos[k] = 0;
  DMC_UNROLL
  for (int k = 0; k < 9; k++) E.ximat[k] = E.xmat[k];
  DMC_UNROLL
  for (int i = BODY_LO(E); i < BODY_HI(E); i++) {
    real xpos[3], xquat[4];
    const int jadr = body_jntadr[i], jnum = body_jntnum[i];
    if (jnum == 1 && jnt_type[jadr < 0 ? 0 : jadr] == JNT_FREE) {
      const int qa = jnt_qposadr[jadr];
      DMC_UNROLL
      for (int k = 0; k < 3; k++) xpos[k] = E.qpos[qa + k];
      DMC_UNROLL
      for (int k = 0; k < 4; k++) xquat[k] = E.qpos[qa + 3 + k];
      normalize4(xquat);
      DMC_UNROLL
      for (int k = 0; k < 3; k++) {
        E.xanchor[3*jadr + k] = xpos[k];
        E.xaxis[3*jadr + k] = R(jnt_axis[3*jadr + k]);
      }
    } else {
      const int pid = body_parentid[i];
      real bp[3] = {R(body_pos[3*i]), R(body_pos[3*i + 1]), R(body_pos[3*i + 2])};
      real bq[4] = {R(body_quat[4*i]), R(body_quat[4*i + 1]),
                    R(body_quat[4*i + 2]), R(body_quat[4*i + 3])};
      real v[3];
      mulmatvec3(v, E.xmat + 9*pid, bp);
      DMC_UNROLL
      for (int k = 0; k < 3; k++) xpos[k] = E.xpos[3*pid + k] + v[k];
      mulquat(xquat, E.xquat + 4*pid, bq);
      DMC_UNROLL
      for (int j = 0; j < jnum; j++) {
        const int jid = jadr + j, qa = jnt_qposadr[jid];
        real jax[3] = {R(jnt_axis[3*jid]), R(jnt_axis[3*jid + 1]), R(jnt_axis[3*jid + 2])};
        real jp[3] = {R(jnt_pos[3*jid]), R(jnt_pos[3*jid + 1]), R(jnt_pos[3*jid + 2])};
        real* anchor = E.xanchor + 3*jid;
        real* axis = E.xaxis + 3*jid;
        rotvecquat(axis, jax, xquat);
        rotvecquat(anchor, jp, xquat);
        DMC_UNROLL
        for (int k = 0; k < 3; k++) anchor[k] += xpos[k];
        if (jnt_type[jid] == JNT_SLIDE) {
          real q = E.qpos[qa] - R(qpos0[qa]);
          DMC_UNROLL
          for (int k = 0; k < 3; k++) xpos[k] += axis[k]*q;
        } else if (jnt_type[jid] == JNT_HINGE || jnt_type[jid] == JNT_BALL) {
          real qloc[4], r[4], vec[3];
          if (jnt_type[jid] == JNT_BALL) {
            DMC_UNROLL
            for (int k = 0; k < 4; k++) qloc[k] = E.qpos[qa + k];
            normalize4(qloc);
          } else {
            axisangle2quat(qloc, jax, E.qpos[qa] - R(qpos0[qa]));
          }
          mulquat(r, xquat, qloc);
          DMC_UNROLL
          for (int k = 0; k < 4; k++) xquat[k] = r[k];
          rotvecquat(vec, jp, xquat);
          DMC_UNROLL
          for (int k = 0; k < 3; k++) xpos[k] = anchor[k] - vec[k];
        }
      }
    }
    normalize4(xquat);
    DMC_UNROLL
    for (int k = 0; k < 3; k++) E.xpos[3*i + k] = xpos[k];
    DMC_UNROLL
    for (int k = 0; k < 4; k++) E.xquat[4*i + k] = xquat[k];
    quat2mat(E.xmat + 9*i, xquat);
    // inertial frame
    real ip[3] = {R(body_ipos[3*i]), R(body_ipos[3*i + 1]), R(body_ipos[3*i + 2])};
    real iq[4] = {R(body_iquat[4*i]), R(body_iquat[4*i + 1]),
                  R(body_iquat[4*i + 2]), R(body_iquat[4*i + 3])};
    real v[3], q[4];
    mulmatvec3(v, E.xmat + 9*i, ip);
    DMC_UNROLL
    for (int k = 0; k < 3; k++) E.xipos[3*i + k] = xpos[k] + v[k];
    mulquat(q, xquat, iq);
    if (!MAT_IN_WS) quat2mat(E.ximat + 9*i, q);
  }
}

DEV void com_pos_b(Env& E);
DEV void com_pos(Env& E) {
  DMC_UNROLL
  for (int i = BODY_LO0(E); i < BODY_HI(E); i++)
    DMC_UNROLL
    for (int k = 0; k < 3; k++)
      E.subtree_com[3*i + k] = R(body_mass[i])*E.xipos[3*i + k];
  DMC_UNROLL
  for (int i = BODY_HI(E) - 1; i >= BODY_LO(E); i--) {
    if (TEAMED && body_parentid[i] == 0) continue;    // (the world's entry is shared and unused)
    DMC_UNROLL
    for (int k = 0; k < 3; k++)
      E.subtree_com[3*body_parentid[i] + k] += E.subtree_com[3*i + k];
  }
  DMC_UNROLL
  for (int i = BODY_LO0(E); i < BODY_HI(E); i++) {
    if (body_subtreemass[i] < 1e-15) {
      DMC_UNROLL
      for (int k = 0; k < 3; k++) E.subtree_com[3*i + k] = E.xipos[3*i + k];
    } else {
      real inv = R(1.0/(body_subtreemass[i] < 1e-15 ? 1.0 : body_subtreemass[i]));
      DMC_UNROLL
      for (int k = 0; k < 3; k++) E.subtree_com[3*i + k] *= inv;
    }
  }
  com_pos_b(E);
}
// the second half of com_pos: inertias and dof axes about the tree's centre of mass
DEV void com_pos_b(Env& E) {
  DMC_UNROLL
  for (int k = 0; k < 10; k++) E.cinert[k] = 0;
  DMC_UNROLL
  for (int i = BODY_LO(E); i < BODY_HI(E); i++) {
    const real* com = E.subtree_com + 3*body_rootid[i];
    real mat_[9];
    if (MAT_IN_WS) {     // same two operations as in kinematics(), on the stored xquat
      const real iq[4] = {R(body_iquat[4*i]), R(body_iquat[4*i + 1]),
                          R(body_iquat[4*i + 2]), R(body_iquat[4*i + 3])};
      real q[4];
      mulquat(q, E.xquat + 4*i, iq);
      quat2mat(mat_, q);
    }
    const real* mat = MAT_IN_WS ? mat_ : E.ximat + 9*i;
    real dif[3], t[9];
    const real mass = R(body_mass[i]);
    const real in0 = R(body_inertia[3*i]), in1 = R(body_inertia[3*i + 1]),
               in2 = R(body_inertia[3*i + 2]);
    DMC_UNROLL
    for (int k = 0; k < 3; k++) dif[k] = E.xipos[3*i + k] - com[k];
    DMC_UNROLL
    for (int a = 0; a < 3; a++)
      DMC_UNROLL
      for (int b = 0; b < 3; b++)
        t[3*a + b] = mat[3*a]*in0*mat[3*b] + mat[3*a + 1]*in1*mat[3*b + 1] +
                     mat[3*a + 2]*in2*mat[3*b + 2];
    real* res = E.cinert + 10*i;
    res[0] = t[0] + mass*(dif[1]*dif[1] + dif[2]*dif[2]);
    res[1] = t[4] + mass*(dif[0]*dif[0] + dif[2]*dif[2]);
    res[2] = t[8] + mass*(dif[0]*dif[0] + dif[1]*dif[1]);
    res[3] = t[1] - mass*dif[0]*dif[1];
    res[4] = t[2] - mass*dif[0]*dif[2];
    res[5] = t[5] - mass*dif[1]*dif[2];
    res[6] = mass*dif[0]; res[7] = mass*dif[1]; res[8] = mass*dif[2];
    res[9] = mass;
  }
  DMC_UNROLL
  for (int j = JNT_LO(E); j < JNT_HI(E); j++) {
    const int b = jnt_bodyid[j], da = jnt_dofadr[j];
    const real* com = E.subtree_com + 3*body_rootid[b];
    real off[3];
    DMC_UNROLL
    for (int k = 0; k < 3; k++) off[k] = com[k] - E.xanchor[3*j + k];
    real* cd = E.cdof + 6*da;
    if (jnt_type[j] == JNT_FREE || jnt_type[j] == JNT_BALL) {
      if (jnt_type[j] == JNT_FREE) {
        DMC_UNROLL DMC_KEEP_ROLLED
        for (int k = 0; k < 18; k++) cd[k] = 0;
        cd[3] = 1; cd[6 + 4] = 1; cd[12 + 5] = 1;
        cd += 18;
      }
      DMC_UNROLL
      for (int k = 0; k < 3; k++) {
        real ax[3] = {E.xmat[9*b + k], E.xmat[9*b + 3 + k], E.xmat[9*b + 6 + k]};
        DMC_UNROLL
        for (int c = 0; c < 3; c++) cd[6*k + c] = ax[c];
        cross3(cd + 6*k + 3, ax, off);
      }
    } else if (jnt_type[j] == JNT_SLIDE) {
      cd[0] = cd[1] = cd[2] = 0;
      DMC_UNROLL
      for (int k = 0; k < 3; k++) cd[3 + k] = E.xaxis[3*j + k];
    } else {
      DMC_UNROLL
      for (int k = 0; k < 3; k++) cd[k] = E.xaxis[3*j + k];
      cross3(cd + 3, E.xaxis + 3*j, off);
    }
  }
}

// team mode: composite inertias of this lane's tree and, per dof, the composite
// inertia times the dof's axis (6 words, parked in the workspace words of Euler's
// matrix, which is idle here); crb_rows_team() turns them into the rows of M
DEV void crb_tree(Env& E, const Work& W) {
  const auto P = W.mat(MAT_A);
  real crb[NBODY*10];
  for (int i = BODY_LO(E); i < BODY_HI(E); i++)
    for (int k = 0; k < 10; k++) crb[10*i + k] = E.cinert[10*i + k];
  for (int i = BODY_HI(E) - 1; i >= BODY_LO(E); i--)
    if (body_parentid[i] > 0)
      for (int k = 0; k < 10; k++) crb[10*body_parentid[i] + k] += crb[10*i + k];
  for (int i = DOF_LO(E); i < DOF_HI(E); i++) {
    real buf[6];
    mul_inert_vec(buf, crb + 10*dof_bodyid[i], E.cdof + 6*i);
    for (int k = 0; k < 6; k++) P.set(6*i + k, buf[k]);
  }
}
// the rows of M (cleared beforehand): the lanes of a tree's group split its dofs,
// a dof's lane writes the diagonal and the entries of the dof's ancestors
DEV void crb_rows_team(Env& E, const Work& W) {
  const auto M = Mats::M(E, W);
  const auto P = W.mat(MAT_A);
  const int tl = tlane();
  for (int t = tl % NGROUPS; t < NTREE; t += NGROUPS)
    for (int i = tree_dof_lo[t] + tl/NGROUPS; i < tree_dof_hi[t]; i += LANES_PER_GROUP) {
      real buf[6];
      for (int k = 0; k < 6; k++) buf[k] = P.get(6*i + k);
      M.set(tri(i, i), dot6(E.cdof + 6*i, buf) + R(dof_armature[i]));
      for (int a = 0; a < dof_anc_len[i]; a++) {
        const int j = dof_anc[i*MAXCHAIN + a];
        M.set(tri(i, j), dot6(E.cdof + 6*j, buf));
      }
    }
  tsync();
}

// composite rigid body algorithm -> packed M, then M = L L^T
DEV void crb_factor(Env& E, const Work& W) {
  const auto M = Mats::M(E, W);
  const auto L = Mats::L(E, W);
  real crb[NBODY*10];
  DMC_UNROLL
  for (int i = 0; i < NBODY*10; i++) crb[i] = E.cinert[i];
  DMC_UNROLL
  for (int i = NBODY - 1; i > 0; i--)
    if (body_parentid[i] > 0)
      DMC_UNROLL
      for (int k = 0; k < 10; k++) crb[10*body_parentid[i] + k] += crb[10*i + k];
  if (MAT_IN_WS) {
    for (int i = 0; i < NV; i++)
      for (int j = dof_treeroot[i]; j <= i; j++) M.set(tri(i, j), 0);
  } else {
    DMC_UNROLL
    for (int i = 0; i < NM; i++) M.set(i, 0);
  }
  DMC_UNROLL
  for (int i = 0; i < NV; i++) {
    real buf[6];
    mul_inert_vec(buf, crb + 10*dof_bodyid[i], E.cdof + 6*i);
    M.set(tri(i, i), dot6(E.cdof + 6*i, buf) + R(dof_armature[i]));
    DMC_UNROLL
    for (int a = 0; a < MAXCHAIN; a++)
      if (a < dof_anc_len[i]) {
        const int j = dof_anc[i*MAXCHAIN + a];
        M.set(tri(i, j), dot6(E.cdof + 6*j, buf));
      }
  }
  if (MAT_IN_WS) {
    copy_env(L, M, LoTree{}, LoTree{});
    if (chol_factor_env(L, E.mlo, E.mhi)) E.warn |= WARN_INERTIA;
    return;
  }
  DMC_UNROLL
  for (int i = 0; i < NM; i++) L.set(i, M.get(i));
  if (chol_factor(L)) E.warn |= WARN_INERTIA;
}

// ---------------------------------------------------------------------------
// velocity stage: body velocities, passive forces, RNE bias
// ---------------------------------------------------------------------------
DEV void com_vel(Env& E) {
  if (!TEAMED)
  DMC_UNROLL
  for (int k = 0; k < 6; k++) E.cvel[k] = 0;
  DMC_UNROLL
  for (int i = BODY_LO(E); i < BODY_HI(E); i++) {
    real cvel[6];
    DMC_UNROLL
    for (int k = 0; k < 6; k++) cvel[k] = E.cvel[6*body_parentid[i] + k];
    const int jadr = body_jntadr[i];
    DMC_UNROLL
    for (int j = 0; j < body_jntnum[i]; j++) {
      const int jid = jadr + j;
      int da = jnt_dofadr[jid];
      if (jnt_type[jid] == JNT_FREE || jnt_type[jid] == JNT_BALL) {
        if (jnt_type[jid] == JNT_FREE) {
          DMC_UNROLL DMC_KEEP_ROLLED
          for (int k = 0; k < 18; k++) E.cdof_dot[6*da + k] = 0;
          DMC_UNROLL
          for (int k = 0; k < 3; k++)
            DMC_UNROLL
            for (int c = 0; c < 6; c++)
              cvel[c] += E.cdof[6*(da + k) + c]*E.qvel[da + k];
          da += 3;
        }
        DMC_UNROLL
        for (int k = 0; k < 3; k++)
          cross_motion(E.cdof_dot + 6*(da + k), cvel, E.cdof + 6*(da + k));
        DMC_UNROLL
        for (int k = 0; k < 3; k++)
          DMC_UNROLL
          for (int c = 0; c < 6; c++)
            cvel[c] += E.cdof[6*(da + k) + c]*E.qvel[da + k];
      } else {
        cross_motion(E.cdof_dot + 6*da, cvel, E.cdof + 6*da);
        DMC_UNROLL
        for (int c = 0; c < 6; c++) cvel[c] += E.cdof[6*da + c]*E.qvel[da];
      }
    }
    DMC_UNROLL
    for (int k = 0; k < 6; k++) E.cvel[6*i + k] = cvel[k];
  }
}

// coefficient of transmission entry w (compile-time, or per-instance task data)
template <class EnvT>
DEV real wrap_coef(const EnvT& E, int w) {
  if (TASK == TASK_POINTMASS) return E.taskdata[w];
  return R(act_wrap_coef[w]);
}

// qfrc_smooth = passive - bias + actuator ; qacc_smooth = M^-1 qfrc_smooth
// passive joint forces (springs, dampers) of the joints j0, j0 + js, ... < j1 and
// the dofs d0, d0 + ds, ... < d1 (team mode: strided over the lanes)
DEV void passive_forces(Env& E, int j0, int j1, int js, int d0, int d1, int ds) {
  if (DISABLEFLAGS & DSBL_PASSIVE) return;
  DMC_UNROLL
  for (int j = j0; j < j1; j += js)
    if (jnt_stiffness[j] != 0 &&
        (jnt_type[j] == JNT_SLIDE || jnt_type[j] == JNT_HINGE)) {
      const int qa = jnt_qposadr[j];
      E.qfrc_smooth[jnt_dofadr[j]] -=
          R(jnt_stiffness[j])*(E.qpos[qa] - R(qpos_spring[qa]));
    }
  if (TEAMED) tsync();       // (a joint's lane and its dof's lane differ)
  DMC_UNROLL
  for (int i = d0; i < d1; i += ds)
    E.qfrc_smooth[i] -= R(dof_damping[i])*E.qvel[i];
}
// actuator forces of the actuators a0, a0 + as, ... < a1.
// transmission = list of (dof, coefficient): a joint, or the joints a fixed
// tendon wraps; the point-mass task varies the coefficients per instance
DEV void actuator_forces(Env& E, int a0, int a1, int as) {
  if (DISABLEFLAGS & DSBL_ACTUATION) return;
  DMC_UNROLL
  for (int i = a0; i < a1; i += as) {
    const real gear = R(actuator_gear[i]);
    real c = E.ctrl[i];
    if (actuator_ctrllimited[i] && !(DISABLEFLAGS & DSBL_CLAMPCTRL))
      c = clampr(c, R(actuator_ctrlrange[2*i]), R(actuator_ctrlrange[2*i + 1]));
    real force = R(actuator_gainprm[3*i])*c;
    if (actuator_biastype[i] == 1) {
      real length = 0, velocity = 0;
      DMC_UNROLL
      for (int k = 0; k < act_wrap_num[i]; k++) {
        const int w = act_wrap_adr[i] + k;
        const real coef = wrap_coef(E, w);
        length += coef*E.qpos[act_wrap_qadr[w]];
        velocity += coef*E.qvel[act_wrap_dof[w]];
      }
      force += R(actuator_biasprm[3*i]) + R(actuator_biasprm[3*i + 1])*gear*length +
               R(actuator_biasprm[3*i + 2])*gear*velocity;
    }
    if (actuator_forcelimited[i])
      force = clampr(force, R(actuator_forcerange[2*i]), R(actuator_forcerange[2*i + 1]));
    DMC_UNROLL
    for (int k = 0; k < act_wrap_num[i]; k++) {
      const int w = act_wrap_adr[i] + k;
      if (TEAMED) tatomic_add(E.qfrc_smooth + act_wrap_dof[w], gear*wrap_coef(E, w)*force);
      else E.qfrc_smooth[act_wrap_dof[w]] += gear*wrap_coef(E, w)*force;
    }
  }
}
DEV void smooth_forces(Env& E, const Work& W, bool actuation) {
  real cacc[NBODY*6], cfrc[NBODY*6];
  DMC_UNROLL
  for (int k = 0; k < 6; k++) { cacc[k] = 0; cfrc[k] = 0; }
  if (!(DISABLEFLAGS & DSBL_GRAVITY))
    DMC_UNROLL
    for (int k = 0; k < 3; k++) cacc[3 + k] = -R(gravity[k]);
  DMC_UNROLL
  for (int i = BODY_LO(E); i < BODY_HI(E); i++) {
    real tmp[6], tmp1[6];
    const int da = body_dofadr[i];
    DMC_UNROLL
    for (int k = 0; k < 6; k++) cacc[6*i + k] = cacc[6*body_parentid[i] + k];
    DMC_UNROLL
    for (int j = 0; j < body_dofnum[i]; j++)
      DMC_UNROLL
      for (int k = 0; k < 6; k++)
        cacc[6*i + k] += E.cdof_dot[6*(da + j) + k]*E.qvel[da + j];
    mul_inert_vec(cfrc + 6*i, E.cinert + 10*i, cacc + 6*i);
    mul_inert_vec(tmp, E.cinert + 10*i, E.cvel + 6*i);
    cross_force(tmp1, E.cvel + 6*i, tmp);
    DMC_UNROLL
    for (int k = 0; k < 6; k++) cfrc[6*i + k] += tmp1[k];
  }
  DMC_UNROLL
  for (int i = BODY_HI(E) - 1; i >= BODY_LO(E); i--)
    if (body_parentid[i] > 0)
      DMC_UNROLL
      for (int k = 0; k < 6; k++) cfrc[6*body_parentid[i] + k] += cfrc[6*i + k];
  DMC_UNROLL
  for (int i = DOF_LO(E); i < DOF_HI(E); i++)
    E.qfrc_smooth[i] = -dot6(E.cdof + 6*i, cfrc + 6*dof_bodyid[i]);
  if (TEAMED) return;        // (forward_team(): passive and actuator forces by the whole team,
                             //  qacc_smooth once M is factored)
  // (one env per lane: the loops stay inline here -- through the helpers above the
  // generic fp64 cheetah build spilled 437 VGPRs instead of 12)
  if (!(DISABLEFLAGS & DSBL_PASSIVE)) {
    DMC_UNROLL
    for (int j = JNT_LO(E); j < JNT_HI(E); j++)
      if (jnt_stiffness[j] != 0 &&
          (jnt_type[j] == JNT_SLIDE || jnt_type[j] == JNT_HINGE)) {
        const int qa = jnt_qposadr[j];
        E.qfrc_smooth[jnt_dofadr[j]] -=
            R(jnt_stiffness[j])*(E.qpos[qa] - R(qpos_spring[qa]));
      }
    DMC_UNROLL
    for (int i = DOF_LO(E); i < DOF_HI(E); i++)
      E.qfrc_smooth[i] -= R(dof_damping[i])*E.qvel[i];
  }
  if (actuation && !(DISABLEFLAGS & DSBL_ACTUATION)) {
    // transmission = list of (dof, coefficient): a joint, or the joints a fixed
    // tendon wraps; the point-mass task varies the coefficients per instance
    DMC_UNROLL
    for (int i = ACT_LO(E); i < ACT_HI(E); i++) {
      const real gear = R(actuator_gear[i]);
      real c = E.ctrl[i];
      if (actuator_ctrllimited[i] && !(DISABLEFLAGS & DSBL_CLAMPCTRL))
        c = clampr(c, R(actuator_ctrlrange[2*i]), R(actuator_ctrlrange[2*i + 1]));
      real force = R(actuator_gainprm[3*i])*c;
      if (actuator_biastype[i] == 1) {
        real length = 0, velocity = 0;
        DMC_UNROLL
        for (int k = 0; k < act_wrap_num[i]; k++) {
          const int w = act_wrap_adr[i] + k;
          const real coef = wrap_coef(E, w);
          length += coef*E.qpos[act_wrap_qadr[w]];
          velocity += coef*E.qvel[act_wrap_dof[w]];
        }
        force += R(actuator_biasprm[3*i]) + R(actuator_biasprm[3*i + 1])*gear*length +
                 R(actuator_biasprm[3*i + 2])*gear*velocity;
      }
      if (actuator_forcelimited[i])
        force = clampr(force, R(actuator_forcerange[2*i]), R(actuator_forcerange[2*i + 1]));
      DMC_UNROLL
      for (int k = 0; k < act_wrap_num[i]; k++) {
        const int w = act_wrap_adr[i] + k;
        E.qfrc_smooth[act_wrap_dof[w]] += gear*wrap_coef(E, w)*force;
      }
    }
  }
  DMC_UNROLL
  for (int i = 0; i < NV; i++) E.qacc_smooth[i] = E.qfrc_smooth[i];
  if (MAT_IN_WS) chol_solve_env(E.qacc_smooth, Mats::L(E, W), LoTree{});
  else chol_solve(E.qacc_smooth, Mats::L(E, W));
}

DEV void subtree_vel(Env& E) {
  DMC_UNROLL
  for (int i = BODY_LO0(E); i < BODY_HI(E); i++) {
    real dif[3], t[3];
    const real* com = E.subtree_com + 3*body_rootid[i];
    DMC_UNROLL
    for (int k = 0; k < 3; k++) dif[k] = E.xipos[3*i + k] - com[k];
    cross3(t, E.cvel + 6*i, dif);
    DMC_UNROLL
    for (int k = 0; k < 3; k++)
      E.subtree_linvel[3*i + k] = R(body_mass[i])*(E.cvel[6*i + 3 + k] + t[k]);
  }
  DMC_UNROLL
  for (int i = BODY_HI(E) - 1; i >= BODY_LO(E); i--)
    DMC_UNROLL
    for (int k = 0; k < 3; k++)
      E.subtree_linvel[3*body_parentid[i] + k] += E.subtree_linvel[3*i + k];
  DMC_UNROLL
  for (int i = BODY_LO0(E); i < BODY_HI(E); i++) {
    real inv = R(1.0/(body_subtreemass[i] < 1e-15 ? 1e-15 : body_subtreemass[i]));
    DMC_UNROLL
    for (int k = 0; k < 3; k++) E.subtree_linvel[3*i + k] *= inv;
  }
}

// ---------------------------------------------------------------------------
// constraints: joint limits + contacts -> rows in the HBM workspace
// ---------------------------------------------------------------------------
DEV real impedance(const real* s, real x) {
  // s = sanitised (d0, dmax, width, midpoint, power), generated by the host
  const real d0 = R(s[0]), d1 = R(s[1]), width = R(s[2]), mid = R(s[3]),
             power = R(s[4]);
  if (s[0] == s[1] || s[2] <= R(1e-15)) return R(0.5)*(d0 + d1);
  x = fabs(x)/width;
  if (x >= 1) return d1;
  if (x <= 0) return d0;
  real y;
  if (s[4] == R(1)) y = x;
  else if (s[4] == R(2)) y = x <= mid ? x*x/mid : 1 - (1 - x)*(1 - x)/(1 - mid);
  else if (x <= mid) y = pow(x, power)/pow(mid, power - 1);
  else y = 1 - pow(1 - x, power)/pow(1 - mid, power - 1);
  return d0 + y*(d1 - d0);
}

// row[j] += sign * dir . d(point velocity)/d(qvel_j) for a point on `body`
DEV void add_jac_dir(real* row, const Env& E, int body, const real* point,
                     const real* dir, real sign) {
  if (body_chain_len[body] == 0) return;
  real off[3], w[3];
  const int root = body_rootid[body];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) off[k] = point[k] - E.subtree_com[3*root + k];
  cross3(w, off, dir);   // dir.(ang x off) = ang.(off x dir)
  DMC_UNROLL
  for (int c = 0; c < MAXCHAIN; c++)
    if (c < body_chain_len[body]) {
      const int i = body_chain[body*MAXCHAIN + c];
      const real* cd = E.cdof + 6*i;
      row[i] += sign*(dot3(dir, cd + 3) + dot3(w, cd));
    }
}
DEV void add_jac_rot(real* row, const Env& E, int body, const real* dir, real sign) {
  DMC_UNROLL
  for (int c = 0; c < MAXCHAIN; c++)
    if (c < body_chain_len[body]) {
      const int i = body_chain[body*MAXCHAIN + c];
      row[i] += sign*dot3(dir, E.cdof + 6*i);
    }
}

template <class Row>
DEV void write_row(const Row& rec, const Env& E, const real* row, real pm,
                   real K, real B, real imp, real Rrow) {
  real vel = 0;
  if (MAT_IN_WS) {
    int lo = NV, hi = -1;
    for (int j = 0; j < NV; j++)
      if (row[j] != 0) { if (lo == NV) lo = j; hi = j; }
    if (hi < 0) lo = hi = 0;
    for (int j = lo; j <= hi; j++) { rec.set(j, row[j]); vel += row[j]*E.qvel[j]; }
    rec.set(ROW_LO, (real)lo); rec.set(ROW_HI, (real)hi);
  } else {
    DMC_UNROLL
    for (int j = 0; j < NV; j++) { rec.set(j, row[j]); vel += row[j]*E.qvel[j]; }
  }
  rec.set(ROW_AREF, -B*vel - K*imp*pm);
  rec.set(ROW_D, R(1)/(Rrow < DMC_MINVAL ? DMC_MINVAL : Rrow));
}
DEV bool push_row(Env& E, const Work& W, const real* row, real pos_minus_margin,
                  real K, real B, real imp, real Rrow) {
  if (E.nefc >= NEFC_MAX) { E.warn |= WARN_CNSTRFULL; return false; }
  const int r = E.nefc++;
  if (LDS_ROWS >= NEFC_MAX || r < LDS_ROWS)
    write_row(W.lrow(r), E, row, pos_minus_margin, K, B, imp, Rrow);
  else
    write_row(W.grow(r), E, row, pos_minus_margin, K, B, imp, Rrow);
  return true;
}

// a row with the single entry `value` at `dof` (joint limits of big scenes)
template <class Row>
DEV void write_row_1(const Row& rec, const Env& E, int dof, real value, real pm,
                     real K, real B, real imp, real Rrow) {
  rec.set(dof, value);
  rec.set(ROW_LO, (real)dof); rec.set(ROW_HI, (real)dof);
  rec.set(ROW_AREF, -B*(value*E.qvel[dof]) - K*imp*pm);
  rec.set(ROW_D, R(1)/(Rrow < DMC_MINVAL ? DMC_MINVAL : Rrow));
}
DEV bool push_row_1(Env& E, const Work& W, int dof, real value, real pm,
                    real K, real B, real imp, real Rrow) {
  if (E.nefc >= NEFC_MAX) { E.warn |= WARN_CNSTRFULL; return false; }
  const int r = E.nefc++;
  if (LDS_ROWS >= NEFC_MAX || r < LDS_ROWS) write_row_1(W.lrow(r), E, dof, value, pm, K, B, imp, Rrow);
  else write_row_1(W.grow(r), E, dof, value, pm, K, B, imp, Rrow);
  return true;
}
DEV void limit_rows(Env& E, const Work& W) {
  if (DISABLEFLAGS & (DSBL_LIMIT | DSBL_CONSTRAINT)) return;
  DMC_UNROLL
  for (int l = 0; l < NLIMIT; l++) {
    const int j = limit_jnt[l], qa = jnt_qposadr[j], dof = jnt_dofadr[j];
    const real margin = R(jnt_margin[j]);
    const real q = E.qpos[qa];
    DMC_UNROLL
    for (int side = -1; side <= 1; side += 2) {
      const real dist = side < 0 ? q - R(jnt_range[2*j]) : R(jnt_range[2*j + 1]) - q;
      if (dist < margin) {
        const real pm = dist - margin;
        const real imp = impedance(limit_solimp + 5*l, pm);
        const real Rr = (1 - imp)*R(dof_invweight0[dof])/imp;
        if (MAT_IN_WS) {      // the record of a one-dof row, written directly
          push_row_1(E, W, dof, -(real)side, pm, R(limit_K[l]), R(limit_B[l]), imp, Rr);
        } else {
          real row[NVX];
          DMC_UNROLL
          for (int k = 0; k < NV; k++) row[k] = 0;
          row[dof] = -(real)side;
          push_row(E, W, row, pm, R(limit_K[l]), R(limit_B[l]), imp, Rr);
        }
      }
    }
  }
}

struct RawCon { real dist, pos[3], frame[6]; };   // frame: normal, tangent hint

DEV void make_frame(const real* fin, real* f) {   // f[9]
  DMC_UNROLL
  for (int k = 0; k < 6; k++) f[k] = fin[k];
  normalize3(f);
  if (sqrt(dot3(f + 3, f + 3)) < R(0.5)) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < R(0.5) && f[1] > R(-0.5)) f[4] = 1; else f[5] = 1;
  }
  real t = dot3(f, f + 3);
  DMC_UNROLL
  for (int k = 0; k < 3; k++) f[3 + k] -= t*f[k];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}

DEV int plane_sphere(RawCon* c, real margin, const real* ppos, const real* pn,
                     const real* spos, real r) {
  real dif[3];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) dif[k] = spos[k] - ppos[k];
  const real dist = dot3(dif, pn) - r;
  if (dist > margin) return 0;
  c->dist = dist;
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    c->pos[k] = spos[k] - pn[k]*(r + R(0.5)*dist);
    c->frame[k] = pn[k]; c->frame[3 + k] = 0;
  }
  return 1;
}
DEV int sphere_sphere(RawCon* c, real margin, const real* p1, const real* p2,
                      real r1, real r2) {
  real dif[3];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) dif[k] = p2[k] - p1[k];
  const real len = sqrt(dot3(dif, dif));
  const real dist = len - r1 - r2;
  if (dist > margin) return 0;
  c->dist = dist;
#ifdef DMC_COOP_BUILD
  // selects per component: as an if / else over a loop LLVM turned the stores
  // into a dynamically indexed stack slot (scratch traffic from 2048 waves).
  // Only in the several-lanes build, for the reason given at put_slot().
  const bool apart = !(len < DMC_MINVAL);
  c->frame[0] = apart ? dif[0]/len : R(1);
  c->frame[1] = apart ? dif[1]/len : R(0);
  c->frame[2] = apart ? dif[2]/len : R(0);
#else
  if (len < DMC_MINVAL) { c->frame[0] = 1; c->frame[1] = c->frame[2] = 0; }
  else for (int k = 0; k < 3; k++) c->frame[k] = dif[k]/len;
#endif
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    c->pos[k] = p1[k] + c->frame[k]*(r1 + R(0.5)*dist);
    c->frame[3 + k] = 0;
  }
  return 1;
}

// world poses of all geoms, computed once per step: G[12*g] = pos(3), mat(9).
// The narrowphase reads only this mirror, so a rolled pair loop (large models)
// indexes G dynamically while Env itself stays statically indexed.
// where the mirror lives: a per-lane array, or -- big scenes -- the workspace
struct ArrPoses {
  real* p;
  __device__ __forceinline__ real get(int i) const { return p[i]; }
  __device__ __forceinline__ void set(int i, real x) const { p[i] = x; }
};
struct WsPoses {
  real* p; long long n;
  __device__ __forceinline__ real get(int i) const { return p[(long long)i*n]; }
  __device__ __forceinline__ void set(int i, real x) const { p[(long long)i*n] = x; }
};
template <bool WS> struct PoseSrc {
  static __device__ __forceinline__ ArrPoses make(const Work&, real* local) { return ArrPoses{local}; }
};
template <> struct PoseSrc<true> {
  static __device__ __forceinline__ WsPoses make(const Work& W, real*) {
    return WsPoses{W.glb + WS_GEOM*W.nenv, W.nenv};
  }
};
template <class Poses>
DEV void geom_poses(const Env& E, const Poses& G) {
  DMC_UNROLL
  for (int g = TEAMED ? tlane() : 0; g < NGEOM; g += TEAM) {     // (team mode: a geom per lane)
    const int b = geom_bodyid[g];
    real gp[3] = {R(geom_pos[3*g]), R(geom_pos[3*g + 1]), R(geom_pos[3*g + 2])};
    real gq[4] = {R(geom_quat[4*g]), R(geom_quat[4*g + 1]), R(geom_quat[4*g + 2]),
                  R(geom_quat[4*g + 3])};
    real v[3], q[4], mat[9];
    mulmatvec3(v, E.xmat + 9*b, gp);
    DMC_UNROLL
    for (int k = 0; k < 3; k++) G.set(12*g + k, E.xpos[3*b + k] + v[k]);
    mulquat(q, E.xquat + 4*b, gq);
    normalize4(q);
    quat2mat(mat, q);
    DMC_UNROLL
    for (int k = 0; k < 9; k++) G.set(12*g + 3 + k, mat[k]);
  }
}
template <class Poses>
DEV void geom_pose(const Poses& G, int g, real* pos, real* mat) {
  DMC_UNROLL
  for (int k = 0; k < 3; k++) pos[k] = G.get(12*g + k);
  DMC_UNROLL
  for (int k = 0; k < 9; k++) mat[k] = G.get(12*g + 3 + k);
}

// rc[cnt] = c (cnt <= 3).  The several-lanes-per-env build writes it as
// per-field selects over all four slots: the if-chain below is folded back into
// a dynamically indexed store by the optimiser, which puts the whole array into
// scratch memory (2048 waves x 10 KB: more than the L2s hold).  The one-lane
// build keeps the if-chain: its working set is in registers already, and with
// forty more live registers the over-budget unrolled fp64 build of the
// 20-dof known-answer model (2508 spilled VGPRs, never selected by mode
// "auto") again computed a wrong trajectory on the GPU.
#if !defined(DMC_COOP_BUILD) && !defined(DMC_SELECT_SLOTS)   // -DDMC_SELECT_SLOTS: tools/spill_hazard/
DEV void put_slot(RawCon* rc, int cnt, const RawCon& c) {
  if (cnt == 0) rc[0] = c;
  else if (cnt == 1) rc[1] = c;
  else if (cnt == 2) rc[2] = c;
  else rc[3] = c;
}
#else
DEV void put_slot(RawCon* rc, int cnt, const RawCon& c) {
  DMC_UNROLL
  for (int s = 0; s < 4; s++) {
    const bool hit = cnt == s;
    rc[s].dist = hit ? c.dist : rc[s].dist;
    DMC_UNROLL
    for (int k = 0; k < 3; k++) rc[s].pos[k] = hit ? c.pos[k] : rc[s].pos[k];
    DMC_UNROLL
    for (int k = 0; k < 6; k++) rc[s].frame[k] = hit ? c.frame[k] : rc[s].frame[k];
  }
}
#endif

// sphere (centre `p1`, radius) against the box (p2, m2, half sizes s2): sphere
// centre clamped to the box in the box frame; with the centre inside the box
// the nearest face decides; normal from the sphere towards the box
DEV int sphere_box(RawCon* rc, real margin, const real* p1, real radius,
                   const real* p2, const real* m2, const real* s2) {
  real dif[3];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) dif[k] = p2[k] - p1[k];
    // sphere centre clamped to the box in the box frame; with the centre inside
  // the box the nearest face decides; normal from the sphere towards the box
  real loc[3], nl[3], pl[3], len = 0, dist;
  bool inside = true;
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    loc[k] = -(m2[k]*dif[0] + m2[3 + k]*dif[1] + m2[6 + k]*dif[2]);   // R2^T (centre - p2)
    const real cl = clampr(loc[k], -s2[k], s2[k]);
    nl[k] = loc[k] - cl;
    inside = inside && nl[k] == 0;
    len += nl[k]*nl[k];
    pl[k] = cl;
  }
  len = sqrt(len);
  if (!inside && len >= DMC_MINVAL) {
    dist = len - radius;
    if (dist > margin) return 0;
    DMC_UNROLL
    for (int k = 0; k < 3; k++) { nl[k] /= len; pl[k] += nl[k]*R(0.5)*dist; }
  } else {
    int best = 0;
    real depth = s2[0] - fabs(loc[0]);
    DMC_UNROLL
    for (int k = 1; k < 3; k++)
      if (s2[k] - fabs(loc[k]) < depth) { depth = s2[k] - fabs(loc[k]); best = k; }
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      nl[k] = k == best ? (loc[k] < 0 ? R(-1) : R(1)) : R(0);
      pl[k] = loc[k] + nl[k]*R(0.5)*(depth - radius);
    }
    dist = -depth - radius;
  }
  rc->dist = dist;
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    rc->pos[k] = p2[k] + m2[3*k]*pl[0] + m2[3*k + 1]*pl[1] + m2[3*k + 2]*pl[2];
    rc->frame[k] = -(m2[3*k]*nl[0] + m2[3*k + 1]*nl[1] + m2[3*k + 2]*nl[2]);
    rc->frame[3 + k] = 0;
  }
  return 1;
}

// capsule - box and box - box: our own construction (MuJoCo's routines for
// these pairs are in the closed binary and nothing in the reference pins their
// manifolds; see oracle/mjstep.c, which this follows step for step, and
// tests/test_closed_form.py).
// slope of f(t) = squared distance from c0 + t a (box frame) to the box
DEV real seg_box_slope(const real* c0, const real* a, const real* s, real t) {
  real g = 0;
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    const real x = c0[k] + t*a[k];
    if (x > s[k]) g += 2*a[k]*(x - s[k]);
    else if (x < -s[k]) g += 2*a[k]*(x + s[k]);
  }
  return g;
}
// [tA, tB]: parameters of the points of the segment c0 + t a, |t| <= h, nearest
// to the box (a stretch when a whole piece is equally near)
DEV void seg_box_nearest(const real* c0, const real* a, real h, const real* s,
                         real& tA, real& tB) {
  real tlo = -h, glo = 0, thi = h, ghi = 0, zlo = 0, zhi = 0;
  bool haveneg = false, havepos = false, havezero = false;
  DMC_UNROLL
  for (int i = 0; i < 8; i++) {
    real t;
    if (i == 0) t = -h;
    else if (i == 1) t = h;
    else {
      const int k = (i - 2) >> 1;
      const real sgn = (i & 1) ? R(1) : R(-1);
      if (!(fabs(a[k]) > DMC_MINVAL)) continue;
      t = (sgn*s[k] - c0[k])/a[k];
      if (!(t > -h && t < h)) continue;
    }
    const real g = seg_box_slope(c0, a, s, t);
    if (g < 0) {
      if (!haveneg || t > tlo) { tlo = t; glo = g; }
      haveneg = true;
    } else if (g > 0) {
      if (!havepos || t < thi) { thi = t; ghi = g; }
      havepos = true;
    } else {
      if (!havezero || t < zlo) zlo = t;
      if (!havezero || t > zhi) zhi = t;
      havezero = true;
    }
  }
  if (havezero) { tA = zlo; tB = zhi; }
  else if (!haveneg) tA = tB = -h;
  else if (!havepos) tA = tB = h;
  else tA = tB = tlo + (thi - tlo)*(-glo)/(ghi - glo);
}
DEV int capsule_box(RawCon* rc, real margin, const real* cpos, const real* cmat,
                    const real* csize, const real* bpos, const real* bmat,
                    const real* bsize) {
  const real axis[3] = {cmat[2], cmat[5], cmat[8]};
  const real h = csize[1];
  real dif[3], c0[3], a[3], tA, tB;
  DMC_UNROLL
  for (int k = 0; k < 3; k++) dif[k] = cpos[k] - bpos[k];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    c0[k] = bmat[k]*dif[0] + bmat[3 + k]*dif[1] + bmat[6 + k]*dif[2];
    a[k] = bmat[k]*axis[0] + bmat[3 + k]*axis[1] + bmat[6 + k]*axis[2];
  }
  seg_box_nearest(c0, a, h, bsize, tA, tB);
  const bool stretch = tB - tA > R(1e-9);
  // oracle order: nearest point, then the far end of the stretch or both ends
  const real ts[3] = {tA, stretch ? tB : -h, h};
  int n = 0;
  DMC_UNROLL
  for (int i = 0; i < 3; i++) {
    if (n >= 2 || (stretch && i == 2)) continue;
    bool dup = false;
    DMC_UNROLL
    for (int j = 0; j < 3; j++) if (j < i && fabs(ts[i] - ts[j]) <= R(1e-9)) dup = true;
    if (dup) continue;
    real p[3];
    DMC_UNROLL
    for (int k = 0; k < 3; k++) p[k] = cpos[k] + axis[k]*ts[i];
    RawCon c;
    if (sphere_box(&c, margin, p, csize[0], bpos, bmat, bsize)) {
      DMC_UNROLL
      for (int k = 0; k < 3; k++) c.frame[3 + k] = axis[k];
      put_slot(rc, n, c);
      n++;
    } else if (i == 0) {
      return 0;
    }
  }
  return (1 << n) - 1;
}

// Sutherland-Hodgman against sgn*poly[.][axis] <= lim; (u, v, depth) vertices
DEV int clip_poly(real (*poly)[3], int n, int axis, real sgn, real lim) {
  real out[8][3];
  int m = 0;
  for (int i = 0; i < n; i++) {
    const real* A = poly[i];
    const real* B = poly[i + 1 < n ? i + 1 : 0];
    const real da = sgn*A[axis] - lim, db = sgn*B[axis] - lim;
    if (da <= 0 && m < 8) { for (int k = 0; k < 3; k++) out[m][k] = A[k]; m++; }
    if ((da < 0 && db > 0) || (da > 0 && db < 0)) {
      const real t = da/(da - db);
      if (m < 8) { for (int k = 0; k < 3; k++) out[m][k] = A[k] + t*(B[k] - A[k]); m++; }
    }
  }
  for (int i = 0; i < m; i++)
    for (int k = 0; k < 3; k++) poly[i][k] = out[i][k];
  return m;
}
DEV int box_box(RawCon* rc, real margin, const real* p1, const real* m1, const real* s1,
                const real* p2, const real* m2, const real* s2) {
  real Rm[3][3], AR[3][3], t[3], dw[3], best = R(-1e30), bestedge = R(-1e30);
  int code = -1, ecode = -1;
  real bsign = 1;
  for (int k = 0; k < 3; k++) dw[k] = p2[k] - p1[k];
  for (int i = 0; i < 3; i++) {
    t[i] = m1[i]*dw[0] + m1[3 + i]*dw[1] + m1[6 + i]*dw[2];
    for (int j = 0; j < 3; j++) {
      Rm[i][j] = m1[i]*m2[j] + m1[3 + i]*m2[3 + j] + m1[6 + i]*m2[6 + j];
      AR[i][j] = fabs(Rm[i][j]) + R(1e-12);
    }
  }
  for (int i = 0; i < 3; i++) {
    const real rb = s2[0]*AR[i][0] + s2[1]*AR[i][1] + s2[2]*AR[i][2];
    const real sep = fabs(t[i]) - (s1[i] + rb);
    if (sep > best) { best = sep; code = i; bsign = t[i] < 0 ? R(-1) : R(1); }
  }
  for (int j = 0; j < 3; j++) {
    const real ra = s1[0]*AR[0][j] + s1[1]*AR[1][j] + s1[2]*AR[2][j];
    const real tj = t[0]*Rm[0][j] + t[1]*Rm[1][j] + t[2]*Rm[2][j];
    const real sep = fabs(tj) - (ra + s2[j]);
    if (sep > best) { best = sep; code = 3 + j; bsign = tj < 0 ? R(-1) : R(1); }
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const real l2 = 1 - Rm[i][j]*Rm[i][j];
      const real len = sqrt(l2 > 0 ? l2 : R(0));
      if (len < R(1e-6)) continue;
      const real ra = s1[i1]*AR[i2][j] + s1[i2]*AR[i1][j];
      const real rb = s2[j1]*AR[i][j2] + s2[j2]*AR[i][j1];
      const real tl = t[i2]*Rm[i1][j] - t[i1]*Rm[i2][j];
      const real sep = (fabs(tl) - (ra + rb))/len;
      if (sep > bestedge) { bestedge = sep; ecode = 3*i + j; }
    }
  if (ecode >= 0 && bestedge > best + R(1e-6) + R(1e-3)*fabs(best)) {
    if (bestedge > margin) return 0;
    const int i = ecode/3, j = ecode % 3;
    real nrm[3], e1[3], e2[3], c1[3], c2[3], w[3];
    for (int k = 0; k < 3; k++) { e1[k] = m1[3*k + i]; e2[k] = m2[3*k + j]; }
    cross3(nrm, e1, e2);
    normalize3(nrm);
    if (dot3(nrm, dw) < 0) for (int k = 0; k < 3; k++) nrm[k] = -nrm[k];
    for (int k = 0; k < 3; k++) { c1[k] = p1[k]; c2[k] = p2[k]; }
    for (int k = 0; k < 3; k++) {
      const real ax1[3] = {m1[k], m1[3 + k], m1[6 + k]}, ax2[3] = {m2[k], m2[3 + k], m2[6 + k]};
      if (k != i) {
        const real sg = dot3(nrm, ax1) >= 0 ? R(1) : R(-1);
        for (int q = 0; q < 3; q++) c1[q] += sg*s1[k]*ax1[q];
      }
      if (k != j) {
        const real sg = dot3(nrm, ax2) >= 0 ? R(-1) : R(1);
        for (int q = 0; q < 3; q++) c2[q] += sg*s2[k]*ax2[q];
      }
    }
    for (int k = 0; k < 3; k++) w[k] = c1[k] - c2[k];
    const real b = dot3(e1, e2), d1 = dot3(e1, w), d2 = dot3(e2, w), den = 1 - b*b;
    const real u = clampr((b*d2 - d1)/den, -s1[i], s1[i]);
    const real v = clampr((d2 - b*d1)/den, -s2[j], s2[j]);
    real q1[3], q2[3];
    for (int k = 0; k < 3; k++) { q1[k] = c1[k] + u*e1[k]; q2[k] = c2[k] + v*e2[k]; w[k] = q2[k] - q1[k]; }
    const real dist = dot3(w, nrm);
    if (dist > margin) return 0;
    rc->dist = dist;
    for (int k = 0; k < 3; k++) {
      rc->pos[k] = R(0.5)*(q1[k] + q2[k]); rc->frame[k] = nrm[k]; rc->frame[3 + k] = 0;
    }
    return 1;
  }
  if (best > margin) return 0;
  const bool first = code < 3;
  const real *pr = first ? p1 : p2, *mr = first ? m1 : m2, *sr = first ? s1 : s2;
  const real *pi = first ? p2 : p1, *mi = first ? m2 : m1, *si = first ? s2 : s1;
  if (!first) bsign = -bsign;
  const int ax = code % 3, iu = (ax + 1) % 3, iv = (ax + 2) % 3;
  real nout[3], fc[3], ua[3], va[3], poly[8][3];
  for (int k = 0; k < 3; k++) {
    nout[k] = bsign*mr[3*k + ax];
    fc[k] = pr[k] + nout[k]*sr[ax];
    ua[k] = mr[3*k + iu]; va[k] = mr[3*k + iv];
  }
  int inc = 0;
  real most = R(1e30), isign = 1;
  for (int k = 0; k < 3; k++) {
    const real axk[3] = {mi[k], mi[3 + k], mi[6 + k]};
    const real d = dot3(axk, nout);
    if (-fabs(d) < most) { most = -fabs(d); inc = k; isign = d > 0 ? R(-1) : R(1); }
  }
  const int ju = (inc + 1) % 3, jv = (inc + 2) % 3;
  for (int vtx = 0; vtx < 4; vtx++) {
    const real su = (vtx == 0 || vtx == 3) ? R(1) : R(-1), sv = vtx < 2 ? R(1) : R(-1);
    real rel[3];
    for (int q = 0; q < 3; q++)
      rel[q] = pi[q] + isign*si[inc]*mi[3*q + inc] + su*si[ju]*mi[3*q + ju] +
               sv*si[jv]*mi[3*q + jv] - fc[q];
    poly[vtx][0] = dot3(rel, ua); poly[vtx][1] = dot3(rel, va); poly[vtx][2] = dot3(rel, nout);
  }
  int cnt = clip_poly(poly, 4, 0, R(1), sr[iu]);
  cnt = clip_poly(poly, cnt, 0, R(-1), sr[iu]);
  cnt = clip_poly(poly, cnt, 1, R(1), sr[iv]);
  cnt = clip_poly(poly, cnt, 1, R(-1), sr[iv]);
  int keep[8], nk = 0;
  for (int k = 0; k < cnt; k++) if (poly[k][2] <= margin) keep[nk++] = k;
  const int ntake = nk < 4 ? nk : 4;
  const real dir = first ? R(1) : R(-1);
  for (int take = 0; take < ntake; take++) {
    const real* v3 = poly[keep[nk <= 4 ? take : (take*nk)/4]];
    RawCon c;
    c.dist = v3[2];
    for (int q = 0; q < 3; q++) {
      c.pos[q] = fc[q] + v3[0]*ua[q] + v3[1]*va[q] + R(0.5)*v3[2]*nout[q];
      c.frame[q] = dir*nout[q]; c.frame[3 + q] = 0;
    }
    put_slot(rc, take, c);
  }
  return (1 << ntake) - 1;
}

// narrowphase of static pair p; returns a bit mask of valid contact slots
template <class Poses>
DEV int collide_pair(const Poses& G, int p, RawCon* rc) {
  const int g1 = pair_g1[p], g2 = pair_g2[p];
  const int t1 = geom_type[g1], t2 = geom_type[g2];
  const real margin = R(pair_margin[p]);
  // positions (and a plane's normal) first: most pairs end at the bounding-sphere
  // test, and where the pose mirror lives in HBM (big scenes) the 18 words of the
  // two rotation matrices are not worth fetching for those
  real p1[3], m1[9], p2[3], m2[9];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) { p1[k] = G.get(12*g1 + k); p2[k] = G.get(12*g2 + k); }
  real dif[3];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) dif[k] = p2[k] - p1[k];
  if (t1 == GEOM_PLANE) {
    const real nn[3] = {G.get(12*g1 + 3 + 2), G.get(12*g1 + 3 + 5), G.get(12*g1 + 3 + 8)};
    if (dot3(dif, nn) > R(geom_rbound[g2]) + margin) return 0;
  } else {
    const real bound = R(geom_rbound[g1]) + R(geom_rbound[g2]) + margin;
    if (dot3(dif, dif) > bound*bound) return 0;
  }
  DMC_UNROLL
  for (int k = 0; k < 9; k++) { m1[k] = G.get(12*g1 + 3 + k); m2[k] = G.get(12*g2 + 3 + k); }
  const real s1[3] = {R(geom_size[3*g1]), R(geom_size[3*g1 + 1]), R(geom_size[3*g1 + 2])};
  const real s2[3] = {R(geom_size[3*g2]), R(geom_size[3*g2 + 1]), R(geom_size[3*g2 + 2])};
  if (t1 == GEOM_PLANE) {
    const real n[3] = {m1[2], m1[5], m1[8]};
    if (t2 == GEOM_SPHERE) return plane_sphere(rc, margin, p1, n, p2, s2[0]);
    if (t2 == GEOM_CAPSULE) {
      const real ax[3] = {m2[2], m2[5], m2[8]};
      real q[3];
      DMC_UNROLL
      for (int k = 0; k < 3; k++) q[k] = p2[k] + ax[k]*s2[1];
      const int c1 = plane_sphere(rc, margin, p1, n, q, s2[0]);
      DMC_UNROLL
      for (int k = 0; k < 3; k++) q[k] = p2[k] - ax[k]*s2[1];
      const int c2 = plane_sphere(rc + 1, margin, p1, n, q, s2[0]);
      DMC_UNROLL
      for (int k = 0; k < 3; k++) { rc[0].frame[3 + k] = ax[k]; rc[1].frame[3 + k] = ax[k]; }
      return c1 | (c2 << 1);
    }
    if (t2 == GEOM_BOX) {
      const real dist = dot3(dif, n);
      int cnt = 0;
      DMC_UNROLL
      for (int i = 0; i < 8; i++) {
        real v[3] = {(i & 1) ? s2[0] : -s2[0], (i & 2) ? s2[1] : -s2[1],
                     (i & 4) ? s2[2] : -s2[2]};
        real corner[3];
        mulmatvec3(corner, m2, v);
        const real ld = dot3(n, corner);
        if (dist + ld > margin || ld > 0 || cnt >= 4) continue;
        RawCon c;
        c.dist = dist + ld;
        DMC_UNROLL
        for (int k = 0; k < 3; k++) {
          c.pos[k] = corner[k] + p2[k] - n[k]*R(0.5)*(dist + ld);
          c.frame[k] = n[k]; c.frame[3 + k] = 0;
        }
        put_slot(rc, cnt, c);
        cnt++;
      }
      return (1 << cnt) - 1;
    }
    return 0;
  }
  if (t1 == GEOM_SPHERE && t2 == GEOM_SPHERE)
    return sphere_sphere(rc, margin, p1, p2, s1[0], s2[0]);
  if (t1 == GEOM_SPHERE && t2 == GEOM_BOX)
    return sphere_box(rc, margin, p1, s1[0], p2, m2, s2);
  if (t1 == GEOM_CAPSULE && t2 == GEOM_BOX) return capsule_box(rc, margin, p1, m1, s1, p2, m2, s2);
  if (t1 == GEOM_BOX && t2 == GEOM_BOX) return box_box(rc, margin, p1, m1, s1, p2, m2, s2);
  if (t1 == GEOM_SPHERE && t2 == GEOM_CAPSULE) {
    const real ax[3] = {m2[2], m2[5], m2[8]};
    real v[3], q[3];
    DMC_UNROLL
    for (int k = 0; k < 3; k++) v[k] = p1[k] - p2[k];
    const real x = clampr(dot3(ax, v), -s2[1], s2[1]);
    DMC_UNROLL
    for (int k = 0; k < 3; k++) q[k] = p2[k] + ax[k]*x;
    return sphere_sphere(rc, margin, p1, q, s1[0], s2[0]);
  }
  if (t1 == GEOM_CAPSULE && t2 == GEOM_CAPSULE) {
    real a1[3], a2[3], d[3], v1[3], v2[3];
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      a1[k] = m1[3*k + 2]*s1[1]; a2[k] = m2[3*k + 2]*s2[1]; d[k] = p1[k] - p2[k];
    }
    const real ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
    const real u = -dot3(a1, d), v = dot3(a2, d);
    const real det = ma*mc - mb*mb;
    if (fabs(det) >= DMC_MINVAL) {
      real x1 = (mc*u - mb*v)/det, x2 = (ma*v - mb*u)/det;
      if (x1 > 1) { x1 = 1; x2 = (v - mb)/mc; }
      else if (x1 < -1) { x1 = -1; x2 = (v + mb)/mc; }
      if (x2 > 1) { x2 = 1; x1 = clampr((u - mb)/ma, -1, 1); }
      else if (x2 < -1) { x2 = -1; x1 = clampr((u + mb)/ma, -1, 1); }
      DMC_UNROLL
      for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k]*x1; v2[k] = p2[k] + a2[k]*x2; }
      return sphere_sphere(rc, margin, v1, v2, s1[0], s2[0]);
    }
    // parallel axes: both ends of each segment, at most two contacts
    int cnt = 0;
    DMC_UNROLL
    for (int t = 0; t < 4; t++) {
      if (cnt >= 2) continue;
      const real sgn = (t & 1) ? R(-1) : R(1);
      if (t < 2) {
        const real x = clampr((v - sgn*mb)/mc, -1, 1);
        DMC_UNROLL
        for (int k = 0; k < 3; k++) { v1[k] = p1[k] + sgn*a1[k]; v2[k] = p2[k] + a2[k]*x; }
      } else {
        const real x = clampr((u - sgn*mb)/ma, -1, 1);
        DMC_UNROLL
        for (int k = 0; k < 3; k++) { v2[k] = p2[k] + sgn*a2[k]; v1[k] = p1[k] + a1[k]*x; }
      }
      RawCon c;
      if (sphere_sphere(&c, margin, v1, v2, s1[0], s2[0])) { put_slot(rc, cnt, c); cnt++; }
    }
    return (1 << cnt) - 1;
  }
  return 0;
}

template <class Rec>
DEV void write_contact(const Rec& rec, int p, const RawCon& c) {
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    rec.set(k, c.pos[k]); rec.set(3 + k, c.frame[k]); rec.set(6 + k, c.frame[3 + k]);
  }
  rec.set(9, c.dist);
  rec.set(10, (real)p);
}

// Team mode: the narrowphase runs one pair per lane, a chunk of TEAM pairs at a
// time; the contacts of a chunk are appended in pair order (exclusive scan of the
// per-lane counts), so the contact list equals the serial loop's.
DEV real tmaxr(real x) {
  for (int m = TEAM/2; m > 0; m >>= 1) { const real y = txor(x, m); x = y > x ? y : x; }
  return x;
}
DEV void detect_contacts_team(Env& E, const Work& W) {
  const int tl = tlane();
  const ArrPoses G = {W.lds + TL_GEOM};      // the mirror lives in the team's LDS
  geom_poses(E, G);
  tsync();
  constexpr int NTREEX = NTREE > 0 ? NTREE : 1;
  real tcen[NTREEX*3], trad[NTREEX];
  for (int t = 0; t < NTREE; t++) {
    trad[t] = 0;
    for (int k = 0; k < 3; k++) tcen[3*t + k] = E.subtree_com[3*root_body[t] + k];
  }
  for (int g = tl; g < NGEOM; g += TEAM) {
    const int t = geom_tree[g];
    if (t < 0) continue;
    real d2 = 0;
    for (int k = 0; k < 3; k++) {
      const real d = G.get(12*g + k) - tcen[3*t + k];
      d2 += d*d;
    }
    const real reach = sqrt(d2) + R(geom_rbound[g]);
    if (reach > trad[t]) trad[t] = reach;
  }
  for (int t = 0; t < NTREE; t++) trad[t] = tmaxr(trad[t]);
  // can no geom of the pair's tree(s) reach the other side?  Two trees: their
  // bounding spheres; a world geom and a tree: the geom's bound (a plane: its
  // half space) against the tree's sphere.
  auto far_apart = [&](int p) {
    const int t1 = pair_tree1[p], t2 = pair_tree2[p];
    if (t1 >= 0 && t2 >= 0) {
      if (t1 == t2) return false;
      real d2 = 0;
      for (int k = 0; k < 3; k++) {
        const real d = tcen[3*t1 + k] - tcen[3*t2 + k];
        d2 += d*d;
      }
      const real reach = trad[t1] + trad[t2] + R(pair_margin[p]);
      return d2 > reach*reach;
    }
    const int wg = pair_wgeom[p];
    if (wg < 0) return false;
    const int t = t1 >= 0 ? t1 : t2;
    real dif[3];
    for (int k = 0; k < 3; k++) dif[k] = tcen[3*t + k] - G.get(12*wg + k);
    if (geom_type[wg] == GEOM_PLANE) {
      const real d = dif[0]*G.get(12*wg + 5) + dif[1]*G.get(12*wg + 8) + dif[2]*G.get(12*wg + 11);
      return d > trad[t] + R(pair_margin[p]);
    }
    const real reach = trad[t] + R(geom_rbound[wg]) + R(pair_margin[p]);
    return dot3(dif, dif) > reach*reach;
  };
  // the runs of the pair list (all pairs between the same two trees, or a world geom
  // and a tree) are tested one per lane; the survivors are walked in chunks of TEAM pairs
  for (int r0 = 0; r0 < NRUN; r0 += TEAM) {
    const int r = r0 + tl;
    bool walk = false;
    if (r < NRUN) {
      const int pf = run_first[r];
      walk = !(run_keyed[r] && NTREE > 1 && pair_margin[pf] == 0 && far_apart(pf));
    }
    unsigned long long live = tballot(walk);
    while (live) {
      const int rr = r0 + tfirst_bit(live);
      live &= live - 1;
      const int pend = run_first[rr] + run_len[rr];
      for (int p0 = run_first[rr]; p0 < pend; p0 += TEAM) {
        const int p = p0 + tl;
        RawCon rc[4];
        int mask = 0;
        if (p < pend && !(NTREE > 1 && far_apart(p))) mask = collide_pair(G, p, rc);
        if (!tany(mask != 0)) continue;
        int cnt = 0;
        for (int c = 0; c < 4; c++) cnt += (mask >> c) & 1;
        int total;
        int k = E.ncon + tscan(cnt, total);
        for (int c = 0; c < 4; c++) {
          if (!((mask >> c) & 1)) continue;
          if (k < NCON_MAX) write_contact(W.gcon(k), p, rc[c]);
          k++;
        }
        if (E.ncon + total > NCON_MAX) { E.warn |= WARN_CONTACTFULL; E.ncon = NCON_MAX; }
        else E.ncon += total;
      }
    }
  }
  tsync();
}

// phase 1: narrowphase over the static pair list -> compact contact list
DEV void detect_contacts(Env& E, const Work& W) {
  if (DISABLEFLAGS & (DSBL_CONTACT | DSBL_CONSTRAINT)) return;
  if (TEAMED) { detect_contacts_team(E, W); return; }
  real Garr[MAT_IN_WS ? 1 : NGEOM*12];
  const auto G = PoseSrc<MAT_IN_WS>::make(W, Garr);
  geom_poses(E, G);
  // Big scenes: one bounding sphere per kinematic tree (centre: the tree's
  // centre of mass), so the blocks of pairs between two walkers that are nowhere
  // near each other are skipped without touching their geoms.  Conservative:
  // a skipped pair's own bounding-sphere test would have rejected it.
  constexpr int NTREEX = NTREE > 0 ? NTREE : 1;
  real tcen[MAT_IN_WS ? NTREEX*3 : 1], trad[MAT_IN_WS ? NTREEX : 1];
  if (MAT_IN_WS && NTREE > 1) {
    for (int t = 0; t < NTREE; t++) {
      trad[t] = 0;
      for (int k = 0; k < 3; k++) tcen[3*t + k] = E.subtree_com[3*root_body[t] + k];
    }
    for (int g = 0; g < NGEOM; g++) {
      const int t = geom_tree[g];
      if (t < 0) continue;
      real d2 = 0;
      for (int k = 0; k < 3; k++) {
        const real d = G.get(12*g + k) - tcen[3*t + k];
        d2 += d*d;
      }
      const real reach = sqrt(d2) + R(geom_rbound[g]);
      if (reach > trad[t]) trad[t] = reach;
    }
  }
  DMC_UNROLL_PAIRS
  for (int p = 0; p < NPAIR; p++) {
    if (MAT_IN_WS && NTREE > 1) {
      const int t1 = pair_tree1[p], t2 = pair_tree2[p];
      if (t1 >= 0 && t2 >= 0 && t1 != t2) {
        real d2 = 0;
        for (int k = 0; k < 3; k++) {
          const real d = tcen[3*t1 + k] - tcen[3*t2 + k];
          d2 += d*d;
        }
        const real reach = trad[t1] + trad[t2] + R(pair_margin[p]);
        // (margins are per pair; the run is skipped only if its first pair is
        // out of reach by more than any margin could bridge: margin 0 here, a
        // positive one re-tests pair by pair)
        if (d2 > reach*reach) {
          if (pair_margin[p] == 0) p += pair_run[p] - 1;
          continue;
        }
      }
    }
    RawCon rc[4];
    const int mask = collide_pair(G, p, rc);
    if (mask == 0) continue;
    _Pragma("unroll")
    for (int c = 0; c < 4; c++) {
      if (!((mask >> c) & 1)) continue;
      if (E.ncon >= NCON_MAX) { E.warn |= WARN_CONTACTFULL; continue; }
      const int k = E.ncon++;
      if (LDS_CONS >= NCON_MAX || k < LDS_CONS) write_contact(W.lcon(k), p, rc[c]);
      else write_contact(W.gcon(k), p, rc[c]);
    }
  }
}

// phase 2: one generic row builder, run once per contact of the busiest lane.
// The pair index is per-lane data, so pair parameters come from the constant
// tables through vector loads, and the body chains are applied as dof bit
// masks over the statically indexed cdof registers (no dynamic indexing of
// per-lane state).
// pyramid edge pairs of planar models stored as one row: word 10 of a contact
// record then carries pair + MERGE_STRIDE * (merged pairs of this contact)
constexpr int MERGE_STRIDE = 4096;
#ifndef DMC_NO_PLANAR_MERGE
constexpr bool PLANAR_MERGE = PLANAR_XZ != 0 && NPAIR < MERGE_STRIDE;
#else
constexpr bool PLANAR_MERGE = false;     // ablation builds (tools/gpu_ablate.py)
#endif
template <class Rec>
DEV void rows_of_contact(Env& E, const Work& W, const Rec& rec) {
  const int p = (int)rec.get(10);
  const real dist = rec.get(9);
  const real includemargin = pair_includemargin[p];
  if (dist >= includemargin) return;
  real pos[3], fin[6], f[9];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) { pos[k] = rec.get(k); fin[k] = rec.get(3 + k); fin[3 + k] = rec.get(6 + k); }
  make_frame(fin, f);
  const int b1 = pair_b1[p], b2 = pair_b2[p];
  const int dim = pair_dim[p];
  unsigned m1[NMASKW], m2[NMASKW];     // which dofs move body 1 / body 2
  DMC_UNROLL
  for (int k = 0; k < NMASKW; k++) {
    m1[k] = body_dofmask[NMASKW*b1 + k];
    m2[k] = body_dofmask[NMASKW*b2 + k];
  }
  // offsets of the contact point from the subtree-root CoM of each body
  real off1[3] = {0, 0, 0}, off2[3] = {0, 0, 0};
  const int r1 = body_rootidx[b1], r2 = body_rootidx[b2];
  DMC_UNROLL
  for (int r = 0; r < NROOT; r++) {
    const real* com = E.subtree_com + 3*root_body[r];
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      if (r1 == r) off1[k] = pos[k] - com[k];
      if (r2 == r) off2[k] = pos[k] - com[k];
    }
  }
  const real pm = dist - includemargin;
  const real imp = impedance(pair_solimp + 5*p, pm);
  const real K = pair_K[p], B = pair_B[p];
  // translational basis rows (normal, tangent 1, tangent 2)
  real jb[3][NVX];
  DMC_UNROLL
  for (int d = 0; d < 3; d++) {
    const real* dir = f + 3*d;
    real w1[3], w2[3];
    cross3(w1, off1, dir);
    cross3(w2, off2, dir);
    DMC_UNROLL
    for (int j = 0; j < NV; j++) {
      const bool in1 = ((m1[j >> 5] >> (j & 31)) & 1u) != 0;
      const bool in2 = ((m2[j >> 5] >> (j & 31)) & 1u) != 0;
      const real* cd = E.cdof + 6*j;
      const real dl = dot3(dir, cd + 3);
      const real v2 = dl + dot3(w2, cd), v1 = dl + dot3(w1, cd);
      jb[d][j] = (in2 ? v2 : R(0)) - (in1 ? v1 : R(0));
    }
  }
  if (dim == 1) {
    const real Rr = (1 - imp)*pair_diag[6*p]/imp;
    push_row(E, W, jb[0], pm, K, B, imp, Rr);
    return;
  }
  // pyramidal: every edge gets 2 mu0^2 R(first edge)
  const real mu0 = pair_friction[5*p];
  real R0 = (1 - imp)*pair_diag[6*p + 1]/imp;
  if (R0 < DMC_MINVAL) R0 = DMC_MINVAL;
  const real Rpy = 2*mu0*mu0*R0;
  int merged = 0;
  DMC_UNROLL
  for (int k = 1; k < 3; k++) {
    const real mu = pair_friction[5*p + k - 1];
    real row[NVX];
    // A model that moves inside the x-z plane (codegen.planar_in_xz) has a zero
    // Jacobian along world y, so for a tangent that is exactly +-y the two
    // pyramid edges J_n +- mu*0 are the same row: one row of twice the weight
    // (D = 2/R) has the same cost, force and Hessian term as the pair.
    if (PLANAR_MERGE && f[3*k] == 0 && f[3*k + 2] == 0) {
      if (push_row(E, W, jb[0], pm, K, B, imp, Rpy*R(0.5))) { merged++; E.nmerged++; }
      continue;
    }
    DMC_UNROLL
    for (int j = 0; j < NV; j++) row[j] = jb[0][j] + mu*jb[k][j];
    push_row(E, W, row, pm, K, B, imp, Rpy);
    DMC_UNROLL
    for (int j = 0; j < NV; j++) row[j] = jb[0][j] - mu*jb[k][j];
    push_row(E, W, row, pm, K, B, imp, Rpy);
  }
  if (dim > 3) {   // torsional / rolling edges: rotational Jacobian rows
    DMC_UNROLL
    for (int k = 3; k < 6; k++) {
      if (k >= dim) continue;
      const real* dir = f + 3*(k - 3);
      const real mu = pair_friction[5*p + k - 1];
      real jt[NVX], row[NVX];
      DMC_UNROLL
      for (int j = 0; j < NV; j++) {
        const bool in1 = ((m1[j >> 5] >> (j & 31)) & 1u) != 0;
        const bool in2 = ((m2[j >> 5] >> (j & 31)) & 1u) != 0;
        const real v = dot3(dir, E.cdof + 6*j);
        jt[j] = (in2 ? v : R(0)) - (in1 ? v : R(0));
      }
      DMC_UNROLL
      for (int j = 0; j < NV; j++) row[j] = jb[0][j] + mu*jt[j];
      push_row(E, W, row, pm, K, B, imp, Rpy);
      DMC_UNROLL
      for (int j = 0; j < NV; j++) row[j] = jb[0][j] - mu*jt[j];
      push_row(E, W, row, pm, K, B, imp, Rpy);
    }
  }
  // the touch sensors walk the rows contact by contact: note how many this one has
  if (PLANAR_MERGE && merged) rec.set(10, (real)(p + MERGE_STRIDE*merged));
}

// Team mode: the rows of one contact, built over the span of the two bodies'
// dof chains only (a foot on the ground: the <= 26 dofs between the foot and its
// walker's root, not the 254 of the pitch).  Same rows as rows_of_contact().
template <class Row>
DEV void write_row_span(const Row& rec, const Env& E, const real* row, int lo, int hi,
                        real pm, real K, real B, real imp, real Rrow) {
  int first = -1, last = -1;
  real vel = 0;
  for (int j = lo; j <= hi; j++) {
    const real v = row[j];
    rec.set(j, v);
    if (v != 0) { if (first < 0) first = j; last = j; vel += v*E.qvel[j]; }
  }
  if (first < 0) first = last = lo;
  rec.set(ROW_LO, (real)first); rec.set(ROW_HI, (real)last);
  rec.set(ROW_AREF, -B*vel - K*imp*pm);
  rec.set(ROW_D, R(1)/(Rrow < DMC_MINVAL ? DMC_MINVAL : Rrow));
}
DEV bool push_row_span(Env& E, const Work& W, const real* row, int lo, int hi, real pm,
                       real K, real B, real imp, real Rrow) {
  if (E.nefc >= NEFC_MAX) { E.warn |= WARN_CNSTRFULL; return false; }
  write_row_span(W.grow(E.nefc++), E, row, lo, hi, pm, K, B, imp, Rrow);
  return true;
}
template <class Rec>
DEV void rows_of_contact_team(Env& E, const Work& W, const Rec& rec) {
  const int p = (int)rec.get(10);
  const real dist = rec.get(9);
  const real includemargin = pair_includemargin[p];
  if (dist >= includemargin) return;
  real pos[3], fin[6], f[9];
  for (int k = 0; k < 3; k++) { pos[k] = rec.get(k); fin[k] = rec.get(3 + k); fin[3 + k] = rec.get(6 + k); }
  make_frame(fin, f);
  const int b1 = pair_b1[p], b2 = pair_b2[p];
  const int dim = pair_dim[p];
  const int n1 = body_chain_len[b1], n2 = body_chain_len[b2];
  int lo = NV, hi = -1;
  for (int c = 0; c < n1; c++) { const int i = body_chain[b1*MAXCHAIN + c]; lo = i < lo ? i : lo; hi = i > hi ? i : hi; }
  for (int c = 0; c < n2; c++) { const int i = body_chain[b2*MAXCHAIN + c]; lo = i < lo ? i : lo; hi = i > hi ? i : hi; }
  if (hi < 0) lo = hi = 0;
  real off1[3], off2[3];
  for (int k = 0; k < 3; k++) {
    off1[k] = n1 ? pos[k] - E.subtree_com[3*body_rootid[b1] + k] : R(0);
    off2[k] = n2 ? pos[k] - E.subtree_com[3*body_rootid[b2] + k] : R(0);
  }
  const real pm = dist - includemargin;
  const real imp = impedance(pair_solimp + 5*p, pm);
  const real K = pair_K[p], B = pair_B[p];
  real jb[3][NVX], jt[NVX], row[NVX];
  for (int d = 0; d < 3; d++) {
    const real* dir = f + 3*d;
    real w1[3], w2[3];
    cross3(w1, off1, dir);
    cross3(w2, off2, dir);
    for (int j = lo; j <= hi; j++) jb[d][j] = 0;
    for (int c = 0; c < n2; c++) {
      const int i = body_chain[b2*MAXCHAIN + c];
      const real* cd = E.cdof + 6*i;
      jb[d][i] += dot3(dir, cd + 3) + dot3(w2, cd);
    }
    for (int c = 0; c < n1; c++) {
      const int i = body_chain[b1*MAXCHAIN + c];
      const real* cd = E.cdof + 6*i;
      jb[d][i] -= dot3(dir, cd + 3) + dot3(w1, cd);
    }
  }
  if (dim == 1) {
    const real Rr = (1 - imp)*pair_diag[6*p]/imp;
    push_row_span(E, W, jb[0], lo, hi, pm, K, B, imp, Rr);
    return;
  }
  const real mu0 = pair_friction[5*p];
  real R0 = (1 - imp)*pair_diag[6*p + 1]/imp;
  if (R0 < DMC_MINVAL) R0 = DMC_MINVAL;
  const real Rpy = 2*mu0*mu0*R0;
  for (int k = 1; k < 3; k++) {
    const real mu = pair_friction[5*p + k - 1];
    for (int j = lo; j <= hi; j++) row[j] = jb[0][j] + mu*jb[k][j];
    push_row_span(E, W, row, lo, hi, pm, K, B, imp, Rpy);
    for (int j = lo; j <= hi; j++) row[j] = jb[0][j] - mu*jb[k][j];
    push_row_span(E, W, row, lo, hi, pm, K, B, imp, Rpy);
  }
  for (int k = 3; k < 6; k++) {      // torsional / rolling edges
    if (k >= dim) continue;
    const real* dir = f + 3*(k - 3);
    const real mu = pair_friction[5*p + k - 1];
    for (int j = lo; j <= hi; j++) jt[j] = 0;
    for (int c = 0; c < n2; c++) { const int i = body_chain[b2*MAXCHAIN + c]; jt[i] += dot3(dir, E.cdof + 6*i); }
    for (int c = 0; c < n1; c++) { const int i = body_chain[b1*MAXCHAIN + c]; jt[i] -= dot3(dir, E.cdof + 6*i); }
    for (int j = lo; j <= hi; j++) row[j] = jb[0][j] + mu*jt[j];
    push_row_span(E, W, row, lo, hi, pm, K, B, imp, Rpy);
    for (int j = lo; j <= hi; j++) row[j] = jb[0][j] - mu*jt[j];
    push_row_span(E, W, row, lo, hi, pm, K, B, imp, Rpy);
  }
}
static_assert(!(TEAMED && PLANAR_MERGE), "team mode: no merged pyramid rows");
// rows a contact record will push (team mode: known before they are written)
template <class Rec>
DEV int rows_of_contact_count(const Rec& rec) {
  const int p = (int)rec.get(10);
  if (rec.get(9) >= pair_includemargin[p]) return 0;
  const int dim = pair_dim[p];
  return dim == 1 ? 1 : 2*(dim - 1);
}
DEV void contact_rows(Env& E, const Work& W) {
  detect_contacts(E, W);
  if (TEAMED) {
    // one contact per lane; the rows keep the serial order (exclusive scan of the counts)
    const int tl = tlane();
    for (int k0 = 0; k0 < E.ncon; k0 += TEAM) {
      const int k = k0 + tl;
      const int mine = k < E.ncon ? rows_of_contact_count(W.gcon(k)) : 0;
      int total;
      const int first = E.nefc, at = tscan(mine, total);
      if (mine > 0) { E.nefc = first + at; rows_of_contact_team(E, W, W.gcon(k)); }
      E.warn = tor(E.warn);
      E.nefc = first + total < NEFC_MAX ? first + total : NEFC_MAX;
    }
    tsync();
    return;
  }
  const int n1 = E.ncon < LDS_CONS ? E.ncon : LDS_CONS;
  for (int k = 0; k < n1; k++) rows_of_contact(E, W, W.lcon(k));
  if (LDS_CONS < NCON_MAX)
    for (int k = LDS_CONS; k < E.ncon; k++) rows_of_contact(E, W, W.gcon(k));
}

// ---------------------------------------------------------------------------
// Newton solver on the primal problem (mj_fwdConstraint)
// ---------------------------------------------------------------------------
struct LsPoint { real alpha, dcost, d0, d1; };
struct LsWords { real x0, v, D; };

// cost relative to alpha = 0 (no cancellation), first and second derivative
DEV void ls_eval(LsPoint& P, real alpha, const Env& E, const Work& W, real q1, real q2) {
  real dcost = alpha*alpha*q2 + alpha*q1;
  real d0 = 2*alpha*q2 + q1, d1 = 2*q2;
  for_rows_ahead(W, E.nefc,
      [&](auto rec) { return LsWords{rec.get(ROW_JAR), rec.get(ROW_JV), rec.get(ROW_D)}; },
      [&](const LsWords& w) {
        const real x = w.x0 + alpha*w.v;
        const real a = x < 0 ? x : R(0), a0 = w.x0 < 0 ? w.x0 : R(0);
        dcost += R(0.5)*w.D*(a*a - a0*a0);
        if (x < 0) { d0 += w.D*x*w.v; d1 += w.D*w.v*w.v; }
      });
  P.alpha = alpha; P.dcost = dcost; P.d0 = d0;
  P.d1 = d1 > DMC_MINVAL ? d1 : DMC_MINVAL;
}

// -DDMC_SOLVER_PROFILE (experiments): 100 MHz time stamps per solver phase,
// summed per lane, returned through Env::prof and written over the observation
#ifdef DMC_SOLVER_PROFILE
#define SPROF(k) do { const long long t_ = wall_clock64(); E.prof[k] += (real)(t_ - tl_); tl_ = t_; } while (0)
#define SCOUNT(k) do { E.prof[k] += 1; } while (0)
#else
#define SPROF(k) do {} while (0)
#define SCOUNT(k) do {} while (0)
#endif
// A wave runs every pass over the rows as long as its busiest lane needs, and
// a lone wave cannot hide the LDS round trip of a row, so the solver is built
// around few, fused passes:
//   (A) applies the previous step to Jaref, notes active-set changes and
//       accumulates the constraint force; the Hessian M + J^T D_active J is
//       kept across iterations and only the rows that changed sides are added
//       or removed (most iterations: none), instead of rebuilding 45 products
//       per row and iteration;
//   (B) computes Jv and the line-search derivatives at alpha = 0 and, in the
//       fp32 build, already evaluates alpha = 1 -- the exact Newton step, which
//       is what -d0/d1 evaluates to when the active set does not change -- so
//       the usual iteration needs no separate line-search pass.
// `start_smooth`: Jaref of the start point is the candidate parked in ROW_JV
// (the warm start lost against qacc_smooth); folded into the first pass A.
DEV void solve_newton(Env& E, const Work& W, real tol, bool start_smooth) {
#ifdef DMC_SOLVER_PROFILE
  long long tl_ = wall_clock64();
#endif
  real Ma[NVX], Mv[NVX], grad[NVX], search[NVX], Hreg[MAT_REGS];
  const auto M = Mats::M(E, W);
  const LaneMat H = Mats::local(Hreg, W, MAT_H);
  // the factor of M is dead once qacc_smooth has been solved: its registers
  // take the factor of the Hessian
  const LaneMat F = Mats::L(E, W);
  const real scale = R(1.0/(meaninertia*(NV > 1 ? NV : 1)));
  const int nefc = E.nefc;
  if (MAT_IN_WS) {
    // envelope of the Hessian: M's, widened for the dofs of every constraint row
    // to the row's first dof (a contact between two trees couples their blocks)
    for (int i = 0; i < NV; i++) E.hlo[i] = dof_treeroot[i];
    for_rows(W, nefc, [&](auto rec) {
      const int lo = row_lo(rec), hi = row_hi(rec);
      if (hi == lo) return;                    // one dof (a joint limit): diagonal only
      for (int j = lo; j <= hi; j++)
        if (rec.get(j) != 0 && lo < E.hlo[j]) E.hlo[j] = lo;
    });
    env_last_rows(E.hhi, E.hlo);
    symv_env(Ma, M, E.qacc, LoTree{});
    copy_env(H, M, LoArr{E.hlo}, LoTree{});
  } else {
    symv(Ma, M, E.qacc);
    DMC_UNROLL
    for (int i = 0; i < NM; i++) H.set(i, M.get(i));
  }
  real improvement = 0, alpha_prev = 0;
  int iter = 0;
  for (;; iter++) {
    DMC_UNROLL
    for (int i = 0; i < NV; i++) E.qfrc_constraint[i] = 0;
    bool changed = false;
    for_rows(W, nefc, [&](auto rec) {
      real jar = rec.get(ROW_JAR);
      const real jv = rec.get(ROW_JV), D = rec.get(ROW_D);
      real row[NVX];
      const int jlo = row_lo(rec), jhi = row_hi(rec);
      DMC_UNROLL
      for (int j = jlo; j <= jhi; j++) row[j] = rec.get(j);
      bool was = false;                 // is this row's term in H?
      if (iter > 0) {
        was = jar < 0;
        jar += alpha_prev*jv;
        rec.set(ROW_JAR, jar);
      } else if (start_smooth) {
        jar = jv;
        rec.set(ROW_JAR, jar);
      }
      const bool now = jar < 0;
      if (now) {
        const real f = -D*jar;
        DMC_UNROLL
        for (int j = jlo; j <= jhi; j++) E.qfrc_constraint[j] += row[j]*f;
      }
      if (now != was) {
        changed = true;
        const real Ds = now ? D : -D;
        if (MAT_IN_WS) {               // only the row's non-zero dofs touch H
          for (int j = jlo; j <= jhi; j++) {
            if (row[j] == 0) continue;
            const real s = Ds*row[j];
            for (int k = jlo; k <= j; k++)
              if (row[k] != 0) H.set(tri(j, k), H.get(tri(j, k)) + s*row[k]);
          }
        } else {
          DMC_UNROLL
          for (int j = 0; j < NV; j++) {
            const real s = Ds*row[j];
            DMC_UNROLL
            for (int k = 0; k <= j; k++) H.set(tri(j, k), H.get(tri(j, k)) + s*row[k]);
          }
        }
      }
    });
    // fp32: after a full step that left the active set unchanged the iterate is
    // the exact minimiser of the current quadratic piece
    const bool converged = DMC_F32_RULES && iter > 0 && !changed &&
                           fabs(alpha_prev - 1) < R(1e-3);
    real gn = 0;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) {
      grad[i] = Ma[i] - E.qfrc_smooth[i] - E.qfrc_constraint[i];
      gn += grad[i]*grad[i];
    }
    SPROF(0);
    if (iter > 0 && (converged || scale*improvement < tol || scale*sqrt(gn) < tol)) break;
    if (iter >= ITERATIONS) break;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) search[i] = -grad[i];
    if (MAT_IN_WS) {
      copy_env(F, H, LoArr{E.hlo}, LoArr{E.hlo});
      chol_factor_env(F, E.hlo, E.hhi);
      chol_solve_env(search, F, LoArr{E.hlo});
    } else {
      DMC_UNROLL
      for (int i = 0; i < NM; i++) F.set(i, H.get(i));
      chol_factor(F);
      chol_solve(search, F);
    }
    SPROF(1);
    real sn = 0;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) sn += search[i]*search[i];
    sn = sqrt(sn);
    alpha_prev = 0;               // nothing pending if one of the exits below is taken
    if (sn < DMC_MINVAL) break;
    const real gtol = tol*R(0.01)*sn/scale;
    if (MAT_IN_WS) symv_env(Mv, M, search, LoTree{});
    else symv(Mv, M, search);
    real q1 = 0, q2 = 0;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) {
      q1 += search[i]*(Ma[i] - E.qfrc_smooth[i]);
      q2 += R(0.5)*search[i]*Mv[i];
    }
    // pass B: Jv, the derivatives of the cost along `search` at alpha = 0 and
    // (fp32) the line-search point alpha = 1
    LsPoint p0, p, best;
    p.alpha = 1; p.dcost = q2 + q1; p.d0 = 2*q2 + q1; p.d1 = 2*q2;
    {
      real d0 = q1, d1 = 2*q2;
      for_rows(W, nefc, [&](auto rec) {
        real row[NVX];
        const int jlo = row_lo(rec), jhi = row_hi(rec);
        DMC_UNROLL
        for (int j = jlo; j <= jhi; j++) row[j] = rec.get(j);
        const real x0 = rec.get(ROW_JAR), D = rec.get(ROW_D);
        real sacc = 0;
        DMC_UNROLL
        for (int j = jlo; j <= jhi; j++) sacc += row[j]*search[j];
        rec.set(ROW_JV, sacc);
        const real Dv = D*sacc;
        if (x0 < 0) { d0 += Dv*x0; d1 += Dv*sacc; }
        if (DMC_F32_RULES) {
          const real x = x0 + sacc;
          const real a = x < 0 ? x : R(0), a0 = x0 < 0 ? x0 : R(0);
          p.dcost += R(0.5)*D*(a*a - a0*a0);
          if (x < 0) { p.d0 += Dv*x; p.d1 += Dv*sacc; }
        }
      });
      p0.alpha = 0; p0.dcost = 0; p0.d0 = d0;
      p0.d1 = d1 > DMC_MINVAL ? d1 : DMC_MINVAL;
      p.d1 = p.d1 > DMC_MINVAL ? p.d1 : DMC_MINVAL;
    }
    SPROF(2);
    // exact line search: safeguarded Newton on the directional derivative
    if (!(p0.d0 < 0)) break;
    best = p0;
    real lo = 0, hi = 0, a = DMC_F32_RULES ? R(1) : -p0.d0/p0.d1;
    bool have_hi = false;
    const real dtol = DMC_F32_RULES ? fmax(gtol, R(1e-5)*fabs(p0.d0)) : gtol;
    for (int it = 0; it < DMC_LS_MAXIT; it++) {
      if (!(DMC_F32_RULES && it == 0)) { ls_eval(p, a, E, W, q1, q2); SCOUNT(5); }
      if (p.dcost < best.dcost) best = p;
      if (fabs(p.d0) < dtol) break;
      if (p.d0 < 0) lo = a; else { hi = a; have_hi = true; }
      real an = a - p.d0/p.d1;
      if (have_hi) {
        if (!(an > lo && an < hi)) an = R(0.5)*(lo + hi);
        if (hi - lo < R(1e-6)*hi) break;
      } else if (an <= lo) {
        an = 2*a;
      }
      a = an;
    }
    const real alpha = best.alpha;
    SPROF(3);
    if (alpha == 0) break;
    improvement = -best.dcost;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) { E.qacc[i] += alpha*search[i]; Ma[i] += alpha*Mv[i]; }
    alpha_prev = alpha;           // applied to Jaref by the next pass A
    SPROF(4);
  }
  E.iters = iter;
}

// Team-mode Newton solver: the algorithm of solve_newton() with the vectors in
// the team's LDS (one copy per env), the passes over the rows run one row per
// lane, the Hessian kept and factored tile by tile (team_factor) and the sums
// over rows and dofs taken as team sums.
DEV void solve_newton_team(Env& E, const Work& W, real tol, bool start_smooth) {
  real* const q = W.lds + TL_Q;        // qacc
  real* const Ma = W.lds + TL_MA;
  real* const Mv = W.lds + TL_MV;
  real* const fs = W.lds + TL_FS;      // qfrc_smooth
  real* const fc = W.lds + TL_FC;      // qfrc_constraint
  real* const search = W.lds + TL_X;
  const auto M = Mats::M(E, W);
  const auto H = Mats::local(nullptr, W, MAT_H);
  const auto F = Mats::L(E, W);
  const real scale = R(1.0/(meaninertia*(NV > 1 ? NV : 1)));
  const int nefc = E.nefc;
  const int tl = tlane();
#ifdef DMC_SOLVER_PROFILE
  long long tl_ = wall_clock64();
#endif
  // envelope of the Hessian: M's, widened by the rows whose dofs lie in two trees
  // (LDS ints); bit t of `coupled`: tree t has rows that start left of it
  int* const hlo = reinterpret_cast<int*>(W.lds + TL_HLO);
  unsigned coupled = 0;
  {
    bool cross = false;
    for (int r = tl; r < nefc; r += TEAM) {
      const auto rec = W.grow(r);
      cross |= (int)rec.get(ROW_LO) < dof_treeroot[(int)rec.get(ROW_HI)];
    }
    if (tany(cross)) {
      for (int i = tl; i < NV; i += TEAM) hlo[i] = dof_treeroot[i];
      tsync();
      for (int r = tl; r < nefc; r += TEAM) {
        const auto rec = W.grow(r);
        const int lo = (int)rec.get(ROW_LO), hi = (int)rec.get(ROW_HI);
        if (!(lo < dof_treeroot[hi])) continue;
        for (int j = lo + 1; j <= hi; j++)
          if (rec.get(j) != 0) tatomic_min(hlo + j, lo);
      }
      tsync();
      for (int t = 0; t < NDTREE; t++) {
        bool c = false;
        for (int i = dtree_lo[t] + tl; i <= dtree_hi[t]; i += TEAM) c |= hlo[i] < dtree_lo[t];
        if (tany(c)) coupled |= 1u << t;
      }
    }
    tsync();
  }
  const int* const henv = coupled ? hlo : nullptr;
  FlipList flips = {W.lds + TL_FLIPS, 0, W.lds + TL_FLIP1, 0};
  team_symv(W, Ma, M, q);
  real improvement = 0, alpha_prev = 0;
  int iter = 0;
  for (;; iter++) {
    // pass A: one row per lane
    for (int i = tl; i < NV; i += TEAM) fc[i] = 0;
    tsync();
    bool changed = false;
    flips.n = 0; flips.n1 = 0;
    for (int r0 = 0; r0 < nefc; r0 += TEAM) {
      const int r = r0 + tl;
      bool flipped = false;
      real fw = 0;
      int flo = 0, fhi = 0;
      if (r < nefc) {
      const auto rec = W.grow(r);
      real jar = rec.get(ROW_JAR);
      const real jv = rec.get(ROW_JV), D = rec.get(ROW_D);
      bool was = false;
      if (iter > 0) {
        was = jar < 0;
        jar += alpha_prev*jv;
        rec.set(ROW_JAR, jar);
      } else if (start_smooth) {
        jar = jv;
        rec.set(ROW_JAR, jar);
      }
      const bool now = jar < 0;
      if (now) {
        const real f = -D*jar;
        const int jlo = (int)rec.get(ROW_LO), jhi = (int)rec.get(ROW_HI);
        int j = jlo;
        for (; j + 8 <= jhi + 1; j += 8) {
          real a[8];
          _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = rec.get(j + u);
          _Pragma("unroll") for (int u = 0; u < 8; u++) if (a[u] != 0) tatomic_add(fc + j + u, a[u]*f);
        }
        for (; j <= jhi; j++) { const real a = rec.get(j); if (a != 0) tatomic_add(fc + j, a*f); }
      }
      rec.set(ROW_FLIP, now != was ? (now ? D : -D) : R(0));
      if (now != was) {
        flipped = true;
        flo = (int)rec.get(ROW_LO); fhi = (int)rec.get(ROW_HI);
        fw = now ? D : -D;
        if (flo == fhi) { const real v = rec.get(flo); fw *= v*v; }
      }
      }
      // the changes of this round, listed in row order (one-dof rows apart)
      const bool one = flipped && flo == fhi;
      const unsigned long long fm = tballot(flipped && !one), fm1 = tballot(one);
      const unsigned long long below = (1ull << tl) - 1;
      if (one) {
        const int at = flips.n1 + tpopc(fm1 & below);
        if (at < NFLIP1) { flips.p1[2*at] = (real)flo; flips.p1[2*at + 1] = fw; }
      } else if (flipped) {
        const int at = flips.n + tpopc(fm & below);
        if (at < NFLIP) {
          real* e = flips.p + 4*at;
          e[0] = (real)r; e[1] = (real)flo; e[2] = (real)fhi; e[3] = fw;
        }
      }
      flips.n += tpopc(fm); flips.n1 += tpopc(fm1);
      changed |= (fm | fm1) != 0;
    }
    if (flips.n > NFLIP || flips.n1 > NFLIP1) flips.n = -1;   // too many to list: team_factor scans the rows
    tsync();                      // fc, ROW_FLIP and ROW_JAR are read by other lanes from here on
    real gn = 0;
    for (int i = tl; i < NV; i += TEAM) {
      const real g = Ma[i] - fs[i] - fc[i];
      search[i] = -g;
      gn += g*g;
    }
    gn = tsum(gn);
    SPROF(0);
    const bool converged = DMC_F32_RULES && iter > 0 && !changed &&
                           fabs(alpha_prev - 1) < R(1e-3);
    if (iter > 0 && (converged || scale*improvement < tol || scale*sqrt(gn) < tol)) break;
    if (iter >= ITERATIONS) break;
    tsync();
    team_factor(W, F, iter == 0 ? M : H, henv, coupled, R(0), true, flips, iter == 0, nefc, H, search);
    SPROF(1);
    team_solve(W, search, F, henv, coupled, true);
    SPROF(2);
    real sn = 0;
    for (int i = tl; i < NV; i += TEAM) sn += search[i]*search[i];
    sn = sqrt(tsum(sn));
    alpha_prev = 0;
    if (sn < DMC_MINVAL) break;
    const real gtol = tol*R(0.01)*sn/scale;
    team_symv(W, Mv, M, search);
    real q1 = 0, q2 = 0;
    for (int i = tl; i < NV; i += TEAM) {
      q1 += search[i]*(Ma[i] - fs[i]);
      q2 += R(0.5)*search[i]*Mv[i];
    }
    q1 = tsum(q1); q2 = tsum(q2);
    SPROF(3);
    // pass B: Jv and the line-search sums at alpha = 0 (and, fp32, at alpha = 1)
    LsPoint p0, p, best;
    {
      real d0 = 0, d1 = 0, c1 = 0, e0 = 0, e1 = 0;
      for (int r = tl; r < nefc; r += TEAM) {
        const auto rec = W.grow(r);
        const int jlo = (int)rec.get(ROW_LO), jhi = (int)rec.get(ROW_HI);
        real sacc = 0;
        int j = jlo;
        for (; j + 8 <= jhi + 1; j += 8) {
          real a[8];
          _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = rec.get(j + u);
          _Pragma("unroll") for (int u = 0; u < 8; u++) sacc += a[u]*search[j + u];
        }
        for (; j <= jhi; j++) sacc += rec.get(j)*search[j];
        rec.set(ROW_JV, sacc);
        const real x0 = rec.get(ROW_JAR), D = rec.get(ROW_D);
        const real Dv = D*sacc;
        if (x0 < 0) { d0 += Dv*x0; d1 += Dv*sacc; }
        if (DMC_F32_RULES) {
          const real x = x0 + sacc;
          const real a = x < 0 ? x : R(0), a0 = x0 < 0 ? x0 : R(0);
          c1 += R(0.5)*D*(a*a - a0*a0);
          if (x < 0) { e0 += Dv*x; e1 += Dv*sacc; }
        }
      }
      d0 = tsum(d0) + q1; d1 = tsum(d1) + 2*q2;
      p0.alpha = 0; p0.dcost = 0; p0.d0 = d0;
      p0.d1 = d1 > DMC_MINVAL ? d1 : DMC_MINVAL;
      p.alpha = 1; p.dcost = q2 + q1; p.d0 = 2*q2 + q1; p.d1 = 2*q2;
      if (DMC_F32_RULES) {
        p.dcost += tsum(c1); p.d0 += tsum(e0); p.d1 += tsum(e1);
      }
      p.d1 = p.d1 > DMC_MINVAL ? p.d1 : DMC_MINVAL;
    }
    tsync();
    SPROF(4);
    if (!(p0.d0 < 0)) break;
    best = p0;
    real lo = 0, hi = 0, a = DMC_F32_RULES ? R(1) : -p0.d0/p0.d1;
    bool have_hi = false;
    const real dtol = DMC_F32_RULES ? fmax(gtol, R(1e-5)*fabs(p0.d0)) : gtol;
    for (int it = 0; it < DMC_LS_MAXIT; it++) {
      if (!(DMC_F32_RULES && it == 0)) {
        SCOUNT(7);
        real dcost = 0, d0 = 0, d1 = 0;
        for (int r = tl; r < nefc; r += TEAM) {
          const auto rec = W.grow(r);
          const real x0 = rec.get(ROW_JAR), v = rec.get(ROW_JV), D = rec.get(ROW_D);
          const real x = x0 + a*v;
          const real xa = x < 0 ? x : R(0), xa0 = x0 < 0 ? x0 : R(0);
          dcost += R(0.5)*D*(xa*xa - xa0*xa0);
          if (x < 0) { d0 += D*x*v; d1 += D*v*v; }
        }
        p.alpha = a;
        p.dcost = tsum(dcost) + a*a*q2 + a*q1;
        p.d0 = tsum(d0) + 2*a*q2 + q1;
        d1 = tsum(d1) + 2*q2;
        p.d1 = d1 > DMC_MINVAL ? d1 : DMC_MINVAL;
      }
      if (p.dcost < best.dcost) best = p;
      if (fabs(p.d0) < dtol) break;
      if (p.d0 < 0) lo = a; else { hi = a; have_hi = true; }
      real an = a - p.d0/p.d1;
      if (have_hi) {
        if (!(an > lo && an < hi)) an = R(0.5)*(lo + hi);
        if (hi - lo < R(1e-6)*hi) break;
      } else if (an <= lo) {
        an = 2*a;
      }
      a = an;
    }
    const real alpha = best.alpha;
    SPROF(5);
    if (alpha == 0) break;
    improvement = -best.dcost;
    SCOUNT(6);
    for (int i = tl; i < NV; i += TEAM) { q[i] += alpha*search[i]; Ma[i] += alpha*Mv[i]; }
    alpha_prev = alpha;
    tsync();
  }
  tsync();        // (E.qacc and E.qfrc_constraint ARE q and fc in team mode)
  E.iters = iter;
}

// ---------------------------------------------------------------------------
// touch sensors (mjSENS_TOUCH in mj_sensorAcc): sum of the normal forces of the
// contacts that involve the sensor site's body and whose force ray, cast from
// the contact point, meets the site's spherical zone
// ---------------------------------------------------------------------------
DEV real row_force(const Work& W, int r) {
  real jar, D;
  if (LDS_ROWS >= NEFC_MAX || r < LDS_ROWS) { jar = W.lrow(r).get(ROW_JAR); D = W.lrow(r).get(ROW_D); }
  else { jar = W.grow(r).get(ROW_JAR); D = W.grow(r).get(ROW_D); }
  return jar < 0 ? -D*jar : R(0);
}
// smallest t >= 0 with |o + t d| = radius (d unit), or -1
DEV real ray_sphere(const real* o, const real* d, real radius) {
  const real b = dot3(o, d), c = dot3(o, o) - radius*radius;
  const real disc = b*b - c;
  if (disc < 0) return -1;
  const real sq = sqrt(disc);
  if (-b - sq >= 0) return -b - sq;
  if (-b + sq >= 0) return -b + sq;
  return -1;
}
// smallest t >= 0 at which o + t d meets a face of the box |x_i| <= size_i, or -1
// (a ray that runs along a face or an edge counts, as in mju_rayGeom: the
// contact points of a box resting on a plane sit on the edges of a box site)
DEV real ray_box(const real* o, const real* d, const real* size) {
  real best = -1;
  DMC_UNROLL
  for (int i = 0; i < 3; i++) {
    if (fabs(d[i]) < DMC_MINVAL) continue;
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    DMC_UNROLL
    for (int side = -1; side <= 1; side += 2) {
      const real t = (side*size[i] - o[i])/d[i];
      if (t < 0) continue;
      const real pj = o[j] + t*d[j], pk = o[k] + t*d[k];
      if (fabs(pj) <= size[j] && fabs(pk) <= size[k] && (best < 0 || t < best)) best = t;
    }
  }
  return best;
}
template <class EnvT>
DEV real touch_hit(const EnvT& E, int s, const real* pos, const real* normal,
                   int b1, int b2) {
  const int body = touch_body[s];
  if (body != b1 && body != b2) return 0;
  real o[3], d[3];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) {
    const real site = E.xpos[3*body + k] + E.xmat[9*body + 3*k]*R(touch_pos[3*s]) +
                      E.xmat[9*body + 3*k + 1]*R(touch_pos[3*s + 1]) +
                      E.xmat[9*body + 3*k + 2]*R(touch_pos[3*s + 2]);
    o[k] = pos[k] - site;
    d[k] = body == b2 ? -normal[k] : normal[k];   // ray flips if the sensor is on body 2
  }
  if (touch_type[s] == GEOM_BOX) {
    // into the site frame: world <- body (xmat) <- site (touch_mat)
    real ob[3], db[3], ol[3], dl[3];
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      ob[k] = E.xmat[9*body + k]*o[0] + E.xmat[9*body + 3 + k]*o[1] + E.xmat[9*body + 6 + k]*o[2];
      db[k] = E.xmat[9*body + k]*d[0] + E.xmat[9*body + 3 + k]*d[1] + E.xmat[9*body + 6 + k]*d[2];
    }
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      ol[k] = R(touch_mat[9*s + k])*ob[0] + R(touch_mat[9*s + 3 + k])*ob[1] + R(touch_mat[9*s + 6 + k])*ob[2];
      dl[k] = R(touch_mat[9*s + k])*db[0] + R(touch_mat[9*s + 3 + k])*db[1] + R(touch_mat[9*s + 6 + k])*db[2];
    }
    const real size[3] = {R(touch_size[3*s]), R(touch_size[3*s + 1]), R(touch_size[3*s + 2])};
    return ray_box(ol, dl, size) >= 0 ? R(1) : R(0);
  }
  return ray_sphere(o, d, R(touch_size[3*s])) >= 0 ? R(1) : R(0);
}
template <class Rec>
DEV void touch_of_contact(Env& E, const Work& W, const Rec& rec, int& r) {
  const int word = (int)rec.get(10);
  const int p = PLANAR_MERGE ? word % MERGE_STRIDE : word;
  if (rec.get(9) >= pair_includemargin[p]) return;   // contact without rows
  const int nrow = pair_nrow[p] - (PLANAR_MERGE ? word/MERGE_STRIDE : 0);
  real fn = 0;
  for (int j = 0; j < nrow; j++)
    if (r + j < E.nefc) fn += row_force(W, r + j);
  r += nrow;
  if (!(fn > 0)) return;
  real pos[3], normal[3];
  DMC_UNROLL
  for (int k = 0; k < 3; k++) { pos[k] = rec.get(k); normal[k] = rec.get(3 + k); }
  normalize3(normal);
  DMC_UNROLL
  for (int s = 0; s < NTOUCH; s++)
    E.touch[s] += fn*touch_hit(E, s, pos, normal, pair_b1[p], pair_b2[p]);
}
DEV void touch_sensors(Env& E, const Work& W) {
  DMC_UNROLL
  for (int s = 0; s < NTOUCH; s++) E.touch[s] = 0;
  if (E.nefc == 0) return;
  int r = E.nefc_limit;
  const int n1 = E.ncon < LDS_CONS ? E.ncon : LDS_CONS;
  for (int k = 0; k < n1; k++) touch_of_contact(E, W, W.lcon(k), r);
  if (LDS_CONS < NCON_MAX)
    for (int k = LDS_CONS; k < E.ncon; k++) touch_of_contact(E, W, W.gcon(k), r);
}

// forward dynamics at (qpos, qvel, ctrl): fills qacc and the force terms
// -DDMC_STEP_PROFILE (experiments, tools/debug/pitch_profile.py): 100 MHz stamps
// per stage of forward(), summed per lane into E.prof[0..6], written over the
// first words of the aux xpos output
#if defined(DMC_STEP_PROFILE) && defined(DMC_TREE_PROFILE)
#define FPROF(k) do { (void)tf_; } while (0)
#elif defined(DMC_STEP_PROFILE)
#define FPROF(k) do { const long long t_ = wall_clock64(); E.prof[k] += (real)(t_ - tf_); tf_ = t_; } while (0)
#else
#define FPROF(k) do {} while (0)
#endif
// ---------------------------------------------------------------------------
// Team mode: one forward pass.  Phase 1 -- the tree recursions on one lane per
// tree (NGROUPS trees at a time), then limits and contacts with the whole team;
// phase 2 -- the linear algebra (factor of M, qacc_smooth, warm start, Newton).
// ---------------------------------------------------------------------------
#ifdef DMC_TEAM
DEV void team_bind(Env& E, const Work& W) {
  E.xpos = W.lds + TL_XPOS; E.xquat = W.lds + TL_XQUAT; E.xmat = W.lds + TL_XMAT;
  E.subtree_com = W.lds + TL_SCOM; E.cdof = W.lds + TL_CDOF;
  E.qfrc_smooth = W.lds + TL_FS; E.qfrc_constraint = W.lds + TL_FC;
  E.qacc_smooth = W.lds + TL_QAS; E.qacc = W.lds + TL_Q;
  E.qpos = W.lds + TL_QPOS; E.qvel = W.lds + TL_QVEL; E.warm = W.lds + TL_WARM;
  E.cdof_dot = W.lds + TL_CDOFDOT; E.xaxis = W.lds + TL_XAXIS;
  E.cvel = W.lds + TL_CVEL; E.xanchor = W.lds + TL_XANCHOR;
  E.rb0 = 1; E.rb1 = NBODY; E.rj0 = 0; E.rj1 = NJNT; E.rd0 = 0; E.rd1 = NV; E.ra0 = 0; E.ra1 = NU;
}
DEV void team_range(Env& E, int t) {
  E.rb0 = tree_body_lo[t]; E.rb1 = tree_body_hi[t];
  E.rj0 = tree_jnt_lo[t]; E.rj1 = tree_jnt_hi[t];
  E.rd0 = tree_dof_lo[t]; E.rd1 = tree_dof_hi[t];
  // (actuators grouped by tree: the tree's; else all of them, filtered by dof)
  E.ra0 = tree_act_lo[t] < 0 ? 0 : tree_act_lo[t]; E.ra1 = tree_act_lo[t] < 0 ? NU : tree_act_hi[t];
}
DEV void team_range_all(Env& E) {
  E.rb0 = 1; E.rb1 = NBODY; E.rj0 = 0; E.rj1 = NJNT; E.rd0 = 0; E.rd1 = NV; E.ra0 = 0; E.ra1 = NU;
}
#else
DEV void team_bind(Env&, const Work&) {}
DEV void team_range(Env&, int) {}
DEV void team_range_all(Env&) {}
#endif
static_assert(!TEAMED || NTOUCH == 0, "team mode: touch sensors read frames after the solver");


// ---------------------------------------------------------------------------
// Team mode: the recursions of ONE SEGMENT of a tree (a chain of bodies without
// branches: a leg, the spine, an arm).  The segments of one level of a tree run
// on different lanes; a segment reads its hub's (the body it hangs from) frames
// and velocity from the shared arrays, its acceleration from the hub words in LDS,
// and adds what it passes up (centre-of-mass sums, forces, composite inertia)
// there.  E's range members name the segment.
// ---------------------------------------------------------------------------
DEV void seg_com_init(Env& E) {
  for (int i = BODY_LO(E); i < BODY_HI(E); i++)
    for (int k = 0; k < 3; k++) E.subtree_com[3*i + k] = R(body_mass[i])*E.xipos[3*i + k];
}
DEV void seg_com_up(Env& E) {
  for (int i = BODY_HI(E) - 1; i >= BODY_LO(E); i--) {
    const int pid = body_parentid[i];
    if (pid == 0) continue;
    for (int k = 0; k < 3; k++) {
      if (pid >= BODY_LO(E)) E.subtree_com[3*pid + k] += E.subtree_com[3*i + k];
      else tatomic_add(E.subtree_com + 3*pid + k, E.subtree_com[3*i + k]);
    }
  }
}
DEV void seg_com_div(Env& E) {
  for (int i = BODY_LO(E); i < BODY_HI(E); i++) {
    if (body_subtreemass[i] < 1e-15) {
      for (int k = 0; k < 3; k++) E.subtree_com[3*i + k] = E.xipos[3*i + k];
    } else {
      const real inv = R(1.0/(body_subtreemass[i] < 1e-15 ? 1.0 : body_subtreemass[i]));
      for (int k = 0; k < 3; k++) E.subtree_com[3*i + k] *= inv;
    }
  }
}
// RNE, outward pass: accelerations and the bodies' own forces
DEV void seg_rne_out(Env& E, real* cacc, real* cfrc, real* hubs) {
  for (int i = BODY_LO(E); i < BODY_HI(E); i++) {
    real tmp[6], tmp1[6];
    const int da = body_dofadr[i], pid = body_parentid[i];
    for (int k = 0; k < 6; k++) {
      if (pid >= BODY_LO(E)) cacc[6*i + k] = cacc[6*pid + k];
      else if (pid == 0) cacc[6*i + k] = (k >= 3 && !(DISABLEFLAGS & DSBL_GRAVITY)) ? -R(gravity[k - 3]) : R(0);
      else cacc[6*i + k] = hubs[HUB_WORDS*body_hub[pid] + HUB_CACC + k];
    }
    for (int j = 0; j < body_dofnum[i]; j++)
      for (int k = 0; k < 6; k++)
        cacc[6*i + k] += E.cdof_dot[6*(da + j) + k]*E.qvel[da + j];
    if (body_hub[i] >= 0)
      for (int k = 0; k < 6; k++) hubs[HUB_WORDS*body_hub[i] + HUB_CACC + k] = cacc[6*i + k];
    mul_inert_vec(cfrc + 6*i, E.cinert + 10*i, cacc + 6*i);
    mul_inert_vec(tmp, E.cinert + 10*i, E.cvel + 6*i);
    cross_force(tmp1, E.cvel + 6*i, tmp);
    for (int k = 0; k < 6; k++) cfrc[6*i + k] += tmp1[k];
  }
}
// RNE, inward pass, and the composite inertias (accumulated in place of cinert);
// then the segment's dofs: qfrc_smooth and composite inertia x axis (see crb_tree)
DEV void seg_rne_crb_in(Env& E, const Work& W, real* cfrc, real* hubs) {
  const auto P = W.mat(MAT_A);
  for (int i = BODY_HI(E) - 1; i >= BODY_LO(E); i--) {
    const int pid = body_parentid[i], h = body_hub[i];
    if (h >= 0) {           // what the segments hanging from this body have added
      for (int k = 0; k < 6; k++) cfrc[6*i + k] += hubs[HUB_WORDS*h + HUB_CFRC + k];
      for (int k = 0; k < 10; k++) E.cinert[10*i + k] += hubs[HUB_WORDS*h + HUB_CRB + k];
    }
    if (pid == 0) continue;
    if (pid >= BODY_LO(E)) {
      for (int k = 0; k < 6; k++) cfrc[6*pid + k] += cfrc[6*i + k];
      for (int k = 0; k < 10; k++) E.cinert[10*pid + k] += E.cinert[10*i + k];
    } else {
      real* hp = hubs + HUB_WORDS*body_hub[pid];
      for (int k = 0; k < 6; k++) tatomic_add(hp + HUB_CFRC + k, cfrc[6*i + k]);
      for (int k = 0; k < 10; k++) tatomic_add(hp + HUB_CRB + k, E.cinert[10*i + k]);
    }
  }
  for (int i = DOF_LO(E); i < DOF_HI(E); i++) {
    real buf[6];
    E.qfrc_smooth[i] = -dot6(E.cdof + 6*i, cfrc + 6*dof_bodyid[i]);
    mul_inert_vec(buf, E.cinert + 10*dof_bodyid[i], E.cdof + 6*i);
    for (int k = 0; k < 6; k++) P.set(6*i + k, buf[k]);
  }
}
#ifdef DMC_TEAM
DEV void team_seg(Env& E, int sid) {
  E.rb0 = seg_body_lo[sid]; E.rb1 = seg_body_hi[sid];
  E.rj0 = seg_jnt_lo[sid]; E.rj1 = seg_jnt_hi[sid];
  E.rd0 = seg_dof_lo[sid]; E.rd1 = seg_dof_hi[sid];
}
#else
DEV void team_seg(Env&, int) {}
#endif
// f() for the segments of `level` that this lane runs (group g = lane mod NGROUPS
// takes the trees g, g + NGROUPS, ...; the lanes of a group split a level's segments)
template <class F>
DEV void for_my_segments(Env& E, int level, F&& f) {
  const int tl = tlane();
  for (int t = tl % NGROUPS; t < NTREE; t += NGROUPS)
    for (int q = tl/NGROUPS; q < MAXSEGPERLEVEL; q += LANES_PER_GROUP) {
      const int sid = lvl_seg[(t*NSEGLEVEL + level)*MAXSEGPERLEVEL + q];
      if (sid >= 0) { team_seg(E, sid); f(); }
    }
}
// joint limits, a chunk of TEAM limits at a time; rows in the serial order
DEV void limit_rows_team(Env& E, const Work& W) {
  if (DISABLEFLAGS & (DSBL_LIMIT | DSBL_CONSTRAINT)) return;
  const int tl = tlane();
  for (int l0 = 0; l0 < NLIMIT; l0 += TEAM) {
    const int l = l0 + tl;
    int cnt = 0;
    real pm[2] = {0, 0};
    bool on[2] = {false, false};
    int j = 0, dof = 0;
    if (l < NLIMIT) {
      j = limit_jnt[l]; dof = jnt_dofadr[j];
      const real margin = R(jnt_margin[j]);
      const real q = E.qpos[jnt_qposadr[j]];
      const real d0 = q - R(jnt_range[2*j]), d1 = R(jnt_range[2*j + 1]) - q;
      if (d0 < margin) { on[0] = true; pm[0] = d0 - margin; cnt++; }
      if (d1 < margin) { on[1] = true; pm[1] = d1 - margin; cnt++; }
    }
    int total;
    const int first = E.nefc, at = tscan(cnt, total);
    if (cnt > 0) {
      E.nefc = first + at;
      for (int sd = 0; sd < 2; sd++) {
        if (!on[sd]) continue;
        const real imp = impedance(limit_solimp + 5*l, pm[sd]);
        const real Rr = (1 - imp)*R(dof_invweight0[dof])/imp;
        push_row_1(E, W, dof, sd == 0 ? R(1) : R(-1), pm[sd], R(limit_K[l]), R(limit_B[l]), imp, Rr);
      }
    }
    E.warn = tor(E.warn);
    E.nefc = first + total < NEFC_MAX ? first + total : NEFC_MAX;
  }
  tsync();
}

DEV void solve_newton_team(Env& E, const Work& W, real tol, bool start_smooth);
DEV void forward_team(Env& E, const Work& W, bool actuation, real tol) {
#ifdef DMC_STEP_PROFILE
  long long tf_ = wall_clock64();
#endif
  const int tl = tlane();
  const auto M = Mats::M(E, W);
  const auto L = Mats::L(E, W);
  real* const x = W.lds + TL_X;
  real* const q = W.lds + TL_Q;
  real* const Ma = W.lds + TL_MA;
  real* const fs = W.lds + TL_FS;
  real* const fc = W.lds + TL_FC;
  real* const qas = W.lds + TL_QAS;
  // the rows of M inside the trees' envelope start from zero
  for (int i = 0; i < NV; i++)
    for (int j = dof_treeroot[i] + tl; j <= i; j += TEAM) M.set(tri(i, j), 0);
  if (tl == 0) world_frames(E);
  tsync();
  // the recursions, segment by segment (see seg_*): outward passes level 0, 1, ...;
  // inward passes from the deepest level; a phase boundary after every level
  real* const hubs = W.lds + TL_HUB;
  real cacc[NBODY*6], cfrc[NBODY*6];          // (the entries of this lane's segments)
  for (int k = tl; k < HUB_WORDS*NHUB; k += TEAM) hubs[k] = 0;
  for (int lv = 0; lv < NSEGLEVEL; lv++) {
    for_my_segments(E, lv, [&] { kinematics(E); seg_com_init(E); });
    tsync();
  }
  for (int lv = NSEGLEVEL - 1; lv >= 0; lv--) {
    for_my_segments(E, lv, [&] { seg_com_up(E); });
    tsync();
  }
  for (int lv = 0; lv < NSEGLEVEL; lv++) for_my_segments(E, lv, [&] { seg_com_div(E); });
  tsync();
  for (int lv = 0; lv < NSEGLEVEL; lv++) for_my_segments(E, lv, [&] { com_pos_b(E); });
  tsync();
  for (int lv = 0; lv < NSEGLEVEL; lv++) {
    for_my_segments(E, lv, [&] { com_vel(E); seg_rne_out(E, cacc, cfrc, hubs); });
    tsync();
  }
  for (int lv = NSEGLEVEL - 1; lv >= 0; lv--) {
    for_my_segments(E, lv, [&] { seg_rne_crb_in(E, W, cfrc, hubs); });
    tsync();
  }
  team_range_all(E);
  tsync();
  crb_rows_team(E, W);
  passive_forces(E, tl, NJNT, TEAM, tl, NV, TEAM);
  tsync();
  if (actuation) actuator_forces(E, tl, NU, TEAM);
  tsync();
  FPROF(0);
  E.ncon = 0; E.nefc = 0; E.iters = 0; E.nmerged = 0;
  limit_rows_team(E, W);
  E.nefc_limit = E.nefc;
  FPROF(3);
  if (NPAIR > 0) contact_rows(E, W);
  FPROF(4);
  // ---- phase 2 (the frames in LDS are dead from here on)
  for (int i = tl; i < NV; i += TEAM) { qas[i] = fs[i]; fc[i] = 0; }
  tsync();
  if (team_factor(W, L, M, nullptr, 0u, R(0), false, FlipList{nullptr, -1, nullptr, 0}, false, 0, L, qas))
    E.warn |= WARN_INERTIA;
  FPROF(1);
  team_solve(W, qas, L, nullptr, 0u, true);
  FPROF(2);
  if (E.nefc == 0) {
    for (int i = tl; i < NV; i += TEAM) q[i] = qas[i];
    tsync();
  } else {
    // warmstart: better of previous qacc and the unconstrained acceleration;
    // Jaref of both candidates in one pass (warm -> ROW_JAR, smooth -> ROW_JV)
    const bool try_warm = !(DISABLEFLAGS & DSBL_WARMSTART);
    real cw = 0, cs = 0;
    team_put(x, E.warm);
    if (try_warm) {
      team_symv(W, Ma, M, x);
      real c = 0;
      for (int i = tl; i < NV; i += TEAM) c += R(0.5)*(Ma[i] - fs[i])*(x[i] - qas[i]);
      cw = tsum(c);
    }
    real rw = 0, rs = 0;
    for (int r = tl; r < E.nefc; r += TEAM) {
      const auto rec = W.grow(r);
      const int jlo = (int)rec.get(ROW_LO), jhi = (int)rec.get(ROW_HI);
      const real aref = rec.get(ROW_AREF), D = rec.get(ROW_D);
      real jw = 0, js = 0;
      int j = jlo;
      for (; j + 8 <= jhi + 1; j += 8) {
        real a[8];
        _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = rec.get(j + u);
        _Pragma("unroll") for (int u = 0; u < 8; u++) { jw += a[u]*x[j + u]; js += a[u]*qas[j + u]; }
      }
      for (; j <= jhi; j++) { const real a = rec.get(j); jw += a*x[j]; js += a*qas[j]; }
      jw -= aref; js -= aref;
      if (jw < 0) rw += R(0.5)*D*jw*jw;
      if (js < 0) rs += R(0.5)*D*js*js;
      rec.set(ROW_JAR, jw); rec.set(ROW_JV, js);
    }
    cw += tsum(rw); cs += tsum(rs);
    tsync();
    const bool use_warm = try_warm && !(cw > cs);
    for (int i = tl; i < NV; i += TEAM) q[i] = use_warm ? x[i] : qas[i];
    tsync();
    solve_newton_team(E, W, tol, !use_warm);
  }
  FPROF(5);
  for (int i = tl; i < NV; i += TEAM) E.warm[i] = q[i];
  tsync();
}

DEV void forward(Env& E, const Work& W, bool actuation, real tol) {
  if (TEAMED) { forward_team(E, W, actuation, tol); return; }
#ifdef DMC_STEP_PROFILE
  long long tf_ = wall_clock64();
#endif
  kinematics(E);
  com_pos(E);
  FPROF(0);
  crb_factor(E, W);
  FPROF(1);
  com_vel(E);
  smooth_forces(E, W, actuation);
  FPROF(2);
  E.ncon = 0; E.nefc = 0; E.iters = 0; E.nmerged = 0;
  limit_rows(E, W);
  E.nefc_limit = E.nefc;
  FPROF(3);
#ifndef DMC_ABLATE_CONTACT
  if (NPAIR > 0) contact_rows(E, W);
#endif
  FPROF(4);
  DMC_UNROLL
  for (int i = 0; i < NV; i++) E.qfrc_constraint[i] = 0;
#ifdef DMC_ABLATE_SOLVER
  if (true) {
#else
  if (E.nefc == 0) {
#endif
    DMC_UNROLL
    for (int i = 0; i < NV; i++) E.qacc[i] = E.qacc_smooth[i];
  } else {
    // warmstart: better of previous qacc and the unconstrained acceleration;
    // Jaref of both candidates in one pass (warm -> ROW_JAR, smooth -> ROW_JV)
    const bool try_warm = !(DISABLEFLAGS & DSBL_WARMSTART);
    real Ma[NVX], cw = 0, cs = 0;
    if (try_warm) {
      if (TEAMED) {
        real* x = W.lds + TL_X;
        real* y = W.lds + TL_MA;
        team_put(x, E.warm);
        team_symv(W, y, Mats::M(E, W), x);
        team_take(Ma, y);
      }
      else if (MAT_IN_WS) symv_env(Ma, Mats::M(E, W), E.warm, LoTree{});
      else symv(Ma, Mats::M(E, W), E.warm);
      DMC_UNROLL
      for (int i = 0; i < NV; i++)
        cw += R(0.5)*(Ma[i] - E.qfrc_smooth[i])*(E.warm[i] - E.qacc_smooth[i]);
    }
    if (TEAMED) {                       // one row per lane
      real rw = 0, rs = 0;
      for (int r = tlane(); r < E.nefc; r += TEAM) {
        const auto rec = W.grow(r);
        const int jlo = (int)rec.get(ROW_LO), jhi = (int)rec.get(ROW_HI);
        const real aref = rec.get(ROW_AREF), D = rec.get(ROW_D);
        real jw = 0, js = 0;
        int j = jlo;
        for (; j + 8 <= jhi + 1; j += 8) {
          real a[8];
          _Pragma("unroll") for (int u = 0; u < 8; u++) a[u] = rec.get(j + u);
          _Pragma("unroll") for (int u = 0; u < 8; u++) { jw += a[u]*E.warm[j + u]; js += a[u]*E.qacc_smooth[j + u]; }
        }
        for (; j <= jhi; j++) { const real a = rec.get(j); jw += a*E.warm[j]; js += a*E.qacc_smooth[j]; }
        jw -= aref; js -= aref;
        if (jw < 0) rw += R(0.5)*D*jw*jw;
        if (js < 0) rs += R(0.5)*D*js*js;
        rec.set(ROW_JAR, jw); rec.set(ROW_JV, js);
      }
      cw += tsum(rw); cs += tsum(rs);
      tsync();
    } else
    for_rows(W, E.nefc, [&](auto rec) {
      real row[NVX];
      const int jlo = row_lo(rec), jhi = row_hi(rec);
      DMC_UNROLL
      for (int j = jlo; j <= jhi; j++) row[j] = rec.get(j);
      const real aref = rec.get(ROW_AREF), D = rec.get(ROW_D);
      real jw = 0, js = 0;
      DMC_UNROLL
      for (int j = jlo; j <= jhi; j++) { jw += row[j]*E.warm[j]; js += row[j]*E.qacc_smooth[j]; }
      jw -= aref; js -= aref;
      if (jw < 0) cw += R(0.5)*D*jw*jw;
      if (js < 0) cs += R(0.5)*D*js*js;
      rec.set(ROW_JAR, jw); rec.set(ROW_JV, js);
    });
    const bool use_warm = try_warm && !(cw > cs);
    DMC_UNROLL
    for (int i = 0; i < NV; i++) E.qacc[i] = use_warm ? E.warm[i] : E.qacc_smooth[i];
    if (TEAMED) solve_newton_team(E, W, tol, !use_warm);
    else solve_newton(E, W, tol, !use_warm);
  }
  FPROF(5);
  if (NTOUCH > 0) touch_sensors(E, W);
  DMC_UNROLL
  for (int i = 0; i < NV; i++) E.warm[i] = E.qacc[i];
}

DEV void integrate_pos(real* qpos, const real* qvel, real h) {
  DMC_UNROLL
  for (int j = TEAMED ? tlane() : 0; j < NJNT; j += TEAM) {     // (team mode: a joint per lane)
    const int qa = jnt_qposadr[j], da = jnt_dofadr[j];
    if (jnt_type[j] == JNT_FREE) {
      DMC_UNROLL
      for (int k = 0; k < 3; k++) qpos[qa + k] += h*qvel[da + k];
      quat_integrate(qpos + qa + 3, qvel + da + 3, h);
    } else if (jnt_type[j] == JNT_BALL) {
      quat_integrate(qpos + qa, qvel + da, h);
    } else {
      qpos[qa] += h*qvel[da];
    }
  }
}

#ifdef DMC_STATE_COMP
// State update in fp64 on (high, low) fp32 pairs: qvel = v0 + h*acc, then
// qpos = q0 + h*(dq or the new qvel); quaternions keep the fp32 path.  Writes
// E.qvel/E.qpos and their low words.
DEV void integrate_comp(Env& E, const real* v0, const real* v0_lo, const real* q0,
                        const real* q0_lo, const real* acc, const real* dq) {
  const double h = timestep;
  double v[NVX];
  DMC_UNROLL
  for (int i = 0; i < NV; i++) {
    v[i] = (double)v0[i] + (double)v0_lo[i] + h*(double)acc[i];
    const real hi = (real)v[i];
    E.qvel[i] = hi; E.qvel_lo[i] = (real)(v[i] - (double)hi);
  }
  real qn[NQ > 0 ? NQ : 1], rate[NVX];
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) qn[i] = q0[i];
  DMC_UNROLL
  for (int i = 0; i < NV; i++) rate[i] = dq ? dq[i] : E.qvel[i];
  integrate_pos(qn, rate, R(timestep));          // quaternion parts
  DMC_UNROLL
  for (int j = 0; j < NJNT; j++) {
    const int qa = jnt_qposadr[j], da = jnt_dofadr[j];
    const int n = jnt_type[j] == JNT_FREE ? 3 : (jnt_type[j] == JNT_BALL ? 0 : 1);
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      if (k >= n) continue;
      const double r = dq ? (double)dq[da + k] : v[da + k];
      const double q = (double)q0[qa + k] + (double)q0_lo[qa + k] + h*r;
      const real hi = (real)q;
      qn[qa + k] = hi; E.qpos_lo[qa + k] = (real)(q - (double)hi);
    }
  }
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) E.qpos[i] = qn[i];
}
#endif

DEV void reset_state(Env& E, real& time) {   // mj_resetData
  if (TEAMED) {
    tsync();
    for (int i = tlane(); i < NQ; i += TEAM) E.qpos[i] = R(qpos0[i]);
    for (int i = tlane(); i < NV; i += TEAM) { E.qvel[i] = 0; E.warm[i] = 0; }
    tsync();
  } else {
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) E.qpos[i] = R(qpos0[i]);
  DMC_UNROLL
  for (int i = 0; i < NV; i++) { E.qvel[i] = 0; E.warm[i] = 0; }
  }
  DMC_UNROLL
  for (int i = 0; i < NU; i++) E.ctrl[i] = 0;
#ifdef DMC_STATE_COMP
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) E.qpos_lo[i] = 0;
  DMC_UNROLL
  for (int i = 0; i < NV; i++) E.qvel_lo[i] = 0;
#endif
  time = 0;
}
DEV bool check_state(Env& E, real& time) {   // mj_checkPos / mj_checkVel
  bool bp = false, bv = false;
  if (TEAMED) {
    for (int i = tlane(); i < NQ; i += TEAM) bp |= bad(E.qpos[i]);
    for (int i = tlane(); i < NV; i += TEAM) bv |= bad(E.qvel[i]);
    bp = tany(bp); bv = tany(bv);
  } else {
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) bp |= bad(E.qpos[i]);
  DMC_UNROLL
  for (int i = 0; i < NV; i++) bv |= bad(E.qvel[i]);
  }
  if (bp) { E.warn |= WARN_BADQPOS; reset_state(E, time); }
  else if (bv) { E.warn |= WARN_BADQVEL; reset_state(E, time); }
  return bp || bv;
}

// one `Physics.step()`: finish the step from the current state
// `stale`: the acceleration is computed from the position/velocity stage of the
// reset state (qpos0, zero velocity) and applied to the current state.  That is
// what the reference's cheetah does in the first of its 200 settle steps:
// reset_context runs mj_forward at qpos0, initialize_episode overwrites qpos
// without a forward pass and calls physics.step(), whose mj_step2 still sees
// the mass matrix, bias forces and contacts of qpos0 (suite/cheetah.py:63-77,
// engine.py:149-166).
DEV void physics_step(Env& E, const Work& W, real& time, real tol, bool stale = false) {
  const real h = R(timestep);
  check_state(E, time);
  static_assert(!TEAMED || INTEGRATOR == 0, "team mode: semi-implicit Euler");
  if (TEAMED) {
    // every vector is shared (LDS): the lanes split the dofs and the joints
    const int tl = tlane();
    forward(E, W, true, tol);
#ifdef DMC_STEP_PROFILE
    const long long tp_ = wall_clock64();
#endif
    bool ba = false;
    for (int i = tl; i < NV; i += TEAM) ba |= bad(E.qacc[i]);
    if (tany(ba)) { E.warn |= WARN_BADQACC; reset_state(E, time); return; }
    bool damped = false;
    for (int i = 0; i < NV; i++) damped |= dof_damping[i] > 0;
    real* x = W.lds + TL_X;
    if (damped) {
      const auto M = Mats::M(E, W);
      const auto A = Mats::local(nullptr, W, MAT_A);
      for (int i = tl; i < NV; i += TEAM) x[i] = E.qfrc_smooth[i] + E.qfrc_constraint[i];
      tsync();
      team_factor(W, A, M, nullptr, 0u, h, false, FlipList{nullptr, -1, nullptr, 0}, false, 0, A, x);
      team_solve(W, x, A, nullptr, 0u, true);
    } else {
      for (int i = tl; i < NV; i += TEAM) x[i] = E.qacc[i];
      tsync();
    }
    for (int i = tl; i < NV; i += TEAM) E.qvel[i] += h*x[i];
    tsync();
    integrate_pos(E.qpos, E.qvel, h);
    tsync();
    time += h;
#ifdef DMC_STEP_PROFILE
    E.prof[6] += (real)(wall_clock64() - tp_);
#endif
    return;
  }
  if (INTEGRATOR == 0) {
    real qkeep[NQ > 0 ? NQ : 1], vkeep[NVX];
    if (stale) {
      DMC_UNROLL
      for (int i = 0; i < NQ; i++) { qkeep[i] = E.qpos[i]; E.qpos[i] = R(qpos0[i]); }
      DMC_UNROLL
      for (int i = 0; i < NV; i++) { vkeep[i] = E.qvel[i]; E.qvel[i] = 0; }
    }
    forward(E, W, true, tol);
    bool ba = false;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) ba |= bad(E.qacc[i]);
    if (ba) { E.warn |= WARN_BADQACC; reset_state(E, time); return; }
    bool damped = false;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) damped |= dof_damping[i] > 0;
    real qacc[NVX];
    if (damped) {
      real Areg[MAT_REGS];
      const auto M = Mats::M(E, W);
      const LaneMat A = Mats::local(Areg, W, MAT_A);
      {
      if (MAT_IN_WS) copy_env(A, M, LoTree{}, LoTree{});
      else {
        DMC_UNROLL
        for (int i = 0; i < NM; i++) A.set(i, M.get(i));
      }
      DMC_UNROLL
      for (int i = 0; i < NV; i++) {
        A.set(tri(i, i), A.get(tri(i, i)) + h*R(dof_damping[i]));
        qacc[i] = E.qfrc_smooth[i] + E.qfrc_constraint[i];
      }
      if (MAT_IN_WS) {
        chol_factor_env(A, E.mlo, E.mhi);
        chol_solve_env(qacc, A, LoTree{});
      } else {
        chol_factor(A);
        chol_solve(qacc, A);
      }
      }
    } else {
      DMC_UNROLL
      for (int i = 0; i < NV; i++) qacc[i] = E.qacc[i];
    }
    if (stale) {
      DMC_UNROLL
      for (int i = 0; i < NQ; i++) E.qpos[i] = qkeep[i];
      DMC_UNROLL
      for (int i = 0; i < NV; i++) E.qvel[i] = vkeep[i];
    }
#ifdef DMC_STATE_COMP
    integrate_comp(E, E.qvel, E.qvel_lo, E.qpos, E.qpos_lo, qacc, nullptr);
#else
    DMC_UNROLL
    for (int i = 0; i < NV; i++) E.qvel[i] += h*qacc[i];
    integrate_pos(E.qpos, E.qvel, h);
#endif
    time += h;
  } else {
    // RK4 (tableau and stage handling as SURVEY.md Appendix A)
    real q0[NQ > 0 ? NQ : 1], v0[NVX], Fv[4*NVX], Fa[4*NVX], dv[NVX];
    const real t0 = time;
    DMC_UNROLL
    for (int i = 0; i < NQ; i++) q0[i] = E.qpos[i];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) v0[i] = E.qvel[i];
    forward(E, W, true, tol);
    bool ba = false;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) ba |= bad(E.qacc[i]);
    if (ba) { E.warn |= WARN_BADQACC; reset_state(E, time); return; }
    DMC_UNROLL
    for (int i = 0; i < NV; i++) { Fv[i] = E.qvel[i]; Fa[i] = E.qacc[i]; }
    const real Acoef[3] = {R(0.5), R(0.5), R(1)};
    DMC_UNROLL
    for (int s = 1; s < 4; s++) {
      const real a = Acoef[s - 1];
      DMC_UNROLL
      for (int i = 0; i < NV; i++) {
        dv[i] = a*Fv[(s - 1)*NV + i];
        E.qvel[i] = v0[i] + h*a*Fa[(s - 1)*NV + i];
      }
      DMC_UNROLL
      for (int i = 0; i < NQ; i++) E.qpos[i] = q0[i];
      integrate_pos(E.qpos, dv, h);
      forward(E, W, true, tol);
      DMC_UNROLL
      for (int i = 0; i < NV; i++) { Fv[s*NV + i] = E.qvel[i]; Fa[s*NV + i] = E.qacc[i]; }
    }
#ifdef DMC_STATE_COMP
    real acc[NVX];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) {
      dv[i] = (Fv[i] + 2*Fv[NV + i] + 2*Fv[2*NV + i] + Fv[3*NV + i])*R(1.0/6.0);
      acc[i] = (Fa[i] + 2*Fa[NV + i] + 2*Fa[2*NV + i] + Fa[3*NV + i])*R(1.0/6.0);
    }
    integrate_comp(E, v0, E.qvel_lo, q0, E.qpos_lo, acc, dv);
#else
    DMC_UNROLL
    for (int i = 0; i < NV; i++) {
      dv[i] = (Fv[i] + 2*Fv[NV + i] + 2*Fv[2*NV + i] + Fv[3*NV + i])*R(1.0/6.0);
      const real acc = (Fa[i] + 2*Fa[NV + i] + 2*Fa[2*NV + i] + Fa[3*NV + i])*R(1.0/6.0);
      E.qvel[i] = v0[i] + h*acc;
    }
    DMC_UNROLL
    for (int i = 0; i < NQ; i++) E.qpos[i] = q0[i];
    integrate_pos(E.qpos, dv, h);
#endif
    time = t0 + h;
  }
}

// position/velocity quantities the tasks and sensors read (the part of
// mj_step1 that the observation needs)
DEV void observe_stage(Env& E, real& time) {
  check_state(E, time);
  if (TEAMED) {                 // the recursions of all trees on lane 0 (frames land in LDS)
    if (tlane() == 0) { world_frames(E); kinematics(E); com_pos(E); com_vel(E); subtree_vel(E); }
    tsync();
    return;
  }
  kinematics(E);
  com_pos(E);
  com_vel(E);
  subtree_vel(E);
}

// ---------------------------------------------------------------------------
// task layer: rewards.tolerance + per-domain observation/reward
// (/root/reference/dm_control/utils/rewards.py:93-135 and suite/*.py)
// ---------------------------------------------------------------------------
enum { SIG_GAUSSIAN = 0, SIG_LINEAR = 1, SIG_QUADRATIC = 2 };
DEV real tolerance(real x, real lower, real upper, real margin, int sigmoid,
                   real value_at_margin) {
  const bool in_bounds = lower <= x && x <= upper;
  if (in_bounds) return 1;
  if (margin == 0) return 0;
  const real d = (x < lower ? lower - x : x - upper)/margin;
  if (sigmoid == SIG_GAUSSIAN) {
    const real scale = sqrt(-2*log(value_at_margin));
    return exp(R(-0.5)*(d*scale)*(d*scale));
  } else if (sigmoid == SIG_LINEAR) {
    const real sx = d*(1 - value_at_margin);
    return fabs(sx) < 1 ? 1 - sx : R(0);
  } else {
    const real sx = d*sqrt(1 - value_at_margin);
    return fabs(sx) < 1 ? 1 - sx*sx : R(0);
  }
}

#define OBS(k) obs[k]

template <class EnvT>
DEV real task_outputs(const EnvT& E, const DmcArgs& a, real* obs) {
  real reward = 0;
  const real inf = R(1e30);
  if (TASK == TASK_CARTPOLE) {
    // cartpole.py:145-148,197-225: position = [x, cos, sin per pole], velocity
    const int npole = NBODY - 2;
    OBS(0) = E.qpos[0];
    real upright = 0, sparse_angle = 1, minvel = 1;
    DMC_UNROLL
    for (int p = 0; p < npole; p++) {
      const real czz = E.xmat[9*(2 + p) + 8], sxz = E.xmat[9*(2 + p) + 2];
      OBS(1 + 2*p) = czz; OBS(2 + 2*p) = sxz;
      upright += (czz + 1)*R(0.5);
      sparse_angle *= tolerance(czz, R(0.995), R(1), 0, SIG_GAUSSIAN, R(0.1));
    }
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(1 + 2*npole + i) = E.qvel[i];
    if (a.task_param_i & 1) {   // sparse
      reward = tolerance(E.qpos[0], R(-0.25), R(0.25), 0, SIG_GAUSSIAN, R(0.1))*sparse_angle;
    } else {
      upright /= npole;
      real centered = tolerance(E.qpos[0], 0, 0, 2, SIG_GAUSSIAN, R(0.1));
      centered = (1 + centered)*R(0.5);
      real small_control = tolerance(E.ctrl[0], 0, 0, 1, SIG_QUADRATIC, 0);
      small_control = (4 + small_control)/5;
      DMC_UNROLL
      for (int i = 1; i < NV; i++) {
        const real t = tolerance(E.qvel[i], 0, 0, 5, SIG_GAUSSIAN, R(0.1));
        minvel = t < minvel ? t : minvel;
      }
      const real small_velocity = (1 + minvel)*R(0.5);
      reward = upright*small_control*small_velocity*centered;
    }
  } else if (TASK == TASK_CHEETAH) {
    // cheetah.py:79-93
    DMC_UNROLL
    for (int i = 1; i < NQ; i++) OBS(i - 1) = E.qpos[i];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(NQ - 1 + i) = E.qvel[i];
    const real speed = E.subtree_linvel[3*task_body[0]];
    reward = tolerance(speed, 10, inf, 10, SIG_LINEAR, 0);
  } else if (TASK == TASK_HUMANOID) {
    // humanoid.py:96-129,168-207; task_body = torso, head, l_hand, l_foot, r_hand, r_foot
    const int torso = task_body[0], head = task_body[1];
    int o = 0;
    DMC_UNROLL
    for (int i = 7; i < NQ; i++) OBS(o++) = E.qpos[i];
    const real head_height = E.xpos[3*head + 2];
    OBS(o++) = head_height;
    DMC_UNROLL
    for (int l = 0; l < 4; l++) {
      const int b = task_body[2 + l];
      real d[3];
      DMC_UNROLL
      for (int k = 0; k < 3; k++) d[k] = E.xpos[3*b + k] - E.xpos[3*torso + k];
      DMC_UNROLL
      for (int c = 0; c < 3; c++)
        OBS(o++) = d[0]*E.xmat[9*torso + c] + d[1]*E.xmat[9*torso + 3 + c] +
                   d[2]*E.xmat[9*torso + 6 + c];
    }
    DMC_UNROLL
    for (int c = 0; c < 3; c++) OBS(o++) = E.xmat[9*torso + 6 + c];
    const real* cv = E.subtree_linvel + 3*torso;
    DMC_UNROLL
    for (int c = 0; c < 3; c++) OBS(o++) = cv[c];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(o++) = E.qvel[i];
    const real standing = tolerance(head_height, R(1.4), inf, R(0.35), SIG_GAUSSIAN, R(0.1));
    const real upright = tolerance(E.xmat[9*torso + 8], R(0.9), inf, R(1.9), SIG_LINEAR, 0);
    real sc = 0;
    DMC_UNROLL
    for (int i = 0; i < NU; i++) sc += tolerance(E.ctrl[i], 0, 0, 1, SIG_QUADRATIC, 0);
    const real small_control = (4 + sc/NU)/5;
    const real move_speed = R(a.task_param_r[0]);
    if (move_speed == 0) {
      const real dont_move = R(0.5)*(tolerance(cv[0], 0, 0, 2, SIG_GAUSSIAN, R(0.1)) +
                                     tolerance(cv[1], 0, 0, 2, SIG_GAUSSIAN, R(0.1)));
      reward = small_control*standing*upright*dont_move;
    } else {
      const real speed = sqrt(cv[0]*cv[0] + cv[1]*cv[1]);
      real move = tolerance(speed, move_speed, inf, move_speed, SIG_LINEAR, 0);
      move = (5*move + 1)/6;
      reward = small_control*standing*upright*move;
    }
  } else if (TASK == TASK_WALKER) {
    // walker.py:86-160; task_body[0] = torso
    const int torso = task_body[0];
    int o = 0;
    DMC_UNROLL
    for (int b = 1; b < NBODY; b++) { OBS(o++) = E.xmat[9*b]; OBS(o++) = E.xmat[9*b + 2]; }
    const real height = E.xpos[3*torso + 2];
    OBS(o++) = height;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(o++) = E.qvel[i];
    const real standing = tolerance(height, R(1.2), inf, R(0.6), SIG_GAUSSIAN, R(0.1));
    const real upright = (1 + E.xmat[9*torso + 8])*R(0.5);
    const real stand_reward = (3*standing + upright)*R(0.25);
    const real move_speed = R(a.task_param_r[0]);
    if (move_speed == 0) {
      reward = stand_reward;
    } else {
      const real move = tolerance(E.subtree_linvel[3*torso], move_speed, inf,
                                  move_speed*R(0.5), SIG_LINEAR, R(0.5));
      reward = stand_reward*(5*move + 1)/6;
    }
  } else if (TASK == TASK_PENDULUM) {
    // pendulum.py:52-120; task_body[0] = pole; cos(8 deg) bound, margin 0
    const int pole = task_body[0];
    OBS(0) = E.xmat[9*pole + 8];
    OBS(1) = E.xmat[9*pole + 2];
    OBS(2) = E.qvel[0];
    reward = tolerance(E.xmat[9*pole + 8], R(0.9902680687415704), R(1), 0,
                       SIG_GAUSSIAN, R(0.1));
  } else if (TASK == TASK_HOPPER) {
    // hopper.py:74-140; task_body = torso, foot; touch sensors toe, heel
    const int torso = task_body[0], foot = task_body[1];
    int o = 0;
    DMC_UNROLL
    for (int i = 1; i < NQ; i++) OBS(o++) = E.qpos[i];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(o++) = E.qvel[i];
    DMC_UNROLL
    for (int s = 0; s < NTOUCH; s++) OBS(o++) = log1p(E.touch[s]);
    const real height = E.xipos[3*torso + 2] - E.xipos[3*foot + 2];
    const real standing = tolerance(height, R(0.6), R(2), 0, SIG_GAUSSIAN, R(0.1));
    if (a.task_param_i & 1) {   // hop
      const real speed = E.subtree_linvel[3*torso];
      reward = standing*tolerance(speed, 2, inf, 1, SIG_LINEAR, R(0.5));
    } else {
      real sc = 0;
      DMC_UNROLL
      for (int i = 0; i < NU; i++) sc += tolerance(E.ctrl[i], 0, 0, 1, SIG_QUADRATIC, 0);
      reward = standing*(sc/NU + 4)/5;
    }
  } else if (TASK == TASK_POINTMASS) {
    // point_mass.py:59-130; task_site = pointmass geom, target geom
    int o = 0;
    DMC_UNROLL
    for (int i = 0; i < NQ; i++) OBS(o++) = E.qpos[i];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(o++) = E.qvel[i];
    real d2 = 0;
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      real p[2];
      DMC_UNROLL
      for (int s = 0; s < 2; s++) {
        const int b = task_site_body[s];
        p[s] = E.xpos[3*b + k] + E.xmat[9*b + 3*k]*R(task_site_pos[3*s]) +
               E.xmat[9*b + 3*k + 1]*R(task_site_pos[3*s + 1]) +
               E.xmat[9*b + 3*k + 2]*R(task_site_pos[3*s + 2]);
      }
      d2 += (p[1] - p[0])*(p[1] - p[0]);
    }
    const real size = R(task_site_size[1]);
    const real near = tolerance(sqrt(d2), 0, size, size, SIG_GAUSSIAN, R(0.1));
    real sc = 0;
    DMC_UNROLL
    for (int i = 0; i < NU; i++) sc += tolerance(E.ctrl[i], 0, 0, 1, SIG_QUADRATIC, 0);
    reward = near*(sc/NU + 4)/5;
  } else if (TASK == TASK_REACHER) {
    // reacher.py:64-122; task_body = finger; task_site = finger geom, target
    // geom; the target's x, y are per-instance data (the reference rewrites
    // model.geom_pos each episode), task_param_r[0] = target + finger radius
    const int finger = task_site_body[0];
    real to[2];
    DMC_UNROLL
    for (int k = 0; k < 2; k++) {
      const real f = E.xpos[3*finger + k] + E.xmat[9*finger + 3*k]*R(task_site_pos[0]) +
                     E.xmat[9*finger + 3*k + 1]*R(task_site_pos[1]) +
                     E.xmat[9*finger + 3*k + 2]*R(task_site_pos[2]);
      to[k] = E.taskdata[k] - f;
    }
    int o = 0;
    DMC_UNROLL
    for (int i = 0; i < NQ; i++) OBS(o++) = E.qpos[i];
    OBS(o++) = to[0]; OBS(o++) = to[1];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(o++) = E.qvel[i];
    reward = tolerance(sqrt(to[0]*to[0] + to[1]*to[1]), 0, R(a.task_param_r[0]), 0,
                       SIG_GAUSSIAN, R(0.1));
  } else if (TASK == TASK_ACROBOT) {
    // acrobot.py:62-81,109-126; task_body = upper_arm, lower_arm;
    // task_site = tip (on lower_arm), target (world)
    const int upper = task_body[0], lower = task_body[1];
    OBS(0) = E.xmat[9*upper + 2]; OBS(1) = E.xmat[9*lower + 2];
    OBS(2) = E.xmat[9*upper + 8]; OBS(3) = E.xmat[9*lower + 8];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(4 + i) = E.qvel[i];
    real d2 = 0;
    DMC_UNROLL
    for (int k = 0; k < 3; k++) {
      real p[2];
      DMC_UNROLL
      for (int s = 0; s < 2; s++) {
        const int b = task_site_body[s];
        p[s] = E.xpos[3*b + k] + E.xmat[9*b + 3*k]*R(task_site_pos[3*s]) +
               E.xmat[9*b + 3*k + 1]*R(task_site_pos[3*s + 1]) +
               E.xmat[9*b + 3*k + 2]*R(task_site_pos[3*s + 2]);
      }
      d2 += (p[1] - p[0])*(p[1] - p[0]);
    }
    reward = tolerance(sqrt(d2), 0, R(task_site_size[1]),
                       (a.task_param_i & 1) ? R(0) : R(1), SIG_GAUSSIAN, R(0.1));
  } else {
    DMC_UNROLL
    for (int i = 0; i < NQ; i++) OBS(i) = E.qpos[i];
    DMC_UNROLL
    for (int i = 0; i < NV; i++) OBS(NQ + i) = E.qvel[i];
  }
  return reward;
}

// ---------------------------------------------------------------------------
// kernels (one env per lane; csrc/dmc_coop.hip includes this file for the
// helpers above and supplies its own dmc_step / dmc_observe)
// ---------------------------------------------------------------------------
#ifndef DMC_COOP_BUILD
DEV void load_env(Env& E, const DmcArgs& a, int e, real& time) {
  const long long n = a.nenv;
  if (TEAMED) {       // the shared state: every lane fetches its share
    for (int i = tlane(); i < NQ; i += TEAM) E.qpos[i] = a.qpos[i*n + e];
    for (int i = tlane(); i < NV; i += TEAM) { E.qvel[i] = a.qvel[i*n + e]; E.warm[i] = a.warm[i*n + e]; }
    tsync();
  } else {
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) E.qpos[i] = a.qpos[i*n + e];
  DMC_UNROLL
  for (int i = 0; i < NV; i++) { E.qvel[i] = a.qvel[i*n + e]; E.warm[i] = a.warm[i*n + e]; }
  }
  time = a.time[e];
#ifdef DMC_STATE_COMP
  {
    const long long np = n;
    const real* c = a.ws + (long long)WS_COMP*np + e;
    DMC_UNROLL
    for (int i = 0; i < NQ; i++) {
      const real tag = c[i*np], lo = c[(NQ + i)*np];
      E.qpos_lo[i] = tag == E.qpos[i] ? lo : R(0);
    }
    c += 2LL*NQ*np;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) {
      const real tag = c[i*np], lo = c[(NV + i)*np];
      E.qvel_lo[i] = tag == E.qvel[i] ? lo : R(0);
    }
  }
#endif
  DMC_UNROLL
  for (int i = 0; i < NTASKDATA; i++) E.taskdata[i] = a.taskdata[sidx(i, e, n, NTDX)];
#if defined(DMC_SOLVER_PROFILE) || defined(DMC_STEP_PROFILE)
  for (int k = 0; k < 8; k++) E.prof[k] = 0;
#endif
  if (MAT_IN_WS) {
    for (int i = 0; i < NV; i++) E.mlo[i] = dof_treeroot[i];
    env_last_rows(E.mhi, E.mlo);
  }
  E.warn = 0; E.ncon = 0; E.nefc = 0; E.nefc_limit = 0; E.iters = 0; E.nmerged = 0;
  DMC_UNROLL
  for (int s = 0; s < (NTOUCH > 0 ? NTOUCH : 1); s++) E.touch[s] = 0;
}
DEV void store_env(const Env& E, const DmcArgs& a, int e, real time) {
  const long long n = a.nenv;
  if (TEAMED) {
    tsync();
    for (int i = tlane(); i < NQ; i += TEAM) a.qpos[i*n + e] = E.qpos[i];
    for (int i = tlane(); i < NV; i += TEAM) { a.qvel[i*n + e] = E.qvel[i]; a.warm[i*n + e] = E.warm[i]; }
    if (tlane() == 0) {
      a.time[e] = time;
      if (E.warn) a.warn[e] |= E.warn;
    }
    return;
  }
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) a.qpos[i*n + e] = E.qpos[i];
  DMC_UNROLL
  for (int i = 0; i < NV; i++) { a.qvel[i*n + e] = E.qvel[i]; a.warm[i*n + e] = E.warm[i]; }
  a.time[e] = time;
#ifdef DMC_STATE_COMP
  {
    const long long np = n;
    real* c = a.ws + (long long)WS_COMP*np + e;
    DMC_UNROLL
    for (int i = 0; i < NQ; i++) { c[i*np] = E.qpos[i]; c[(NQ + i)*np] = E.qpos_lo[i]; }
    c += 2LL*NQ*np;
    DMC_UNROLL
    for (int i = 0; i < NV; i++) { c[i*np] = E.qvel[i]; c[(NV + i)*np] = E.qvel_lo[i]; }
  }
#endif
  if (E.warn) a.warn[e] |= E.warn;
}
// The observation is handed over in the agent layout [env][NOBS].  One lane
// owns one env, so writing it directly is a 4-byte store every NOBS words
// (measured: 3x write amplification at the memory side).  The rows of the 64
// envs of a workgroup are one contiguous chunk, so they are transposed through
// LDS (the solver's row store is dead by now) and written as full wave-wide
// coalesced stores.  Other layouts (explicit strides) are written directly.
constexpr bool OBS_STAGE_FITS = !TEAMED && LDS_WORDS >= LANES*(NOBS > 0 ? NOBS : 1);

DEV void store_outputs(Env& E, const DmcArgs& a, int e, bool accumulate,
                       real* lds_base) {
  const long long n = a.nenv;
  if (TEAMED && TASK == TASK_NONE && NSENSOR == 0 && NTOUCH == 0) {
    // no task: the observation is the (shared) state, every lane copies its share
    tsync();
    for (int k = tlane(); k < NQ; k += TEAM)
      a.obs[(long long)k*a.obs_sk + (long long)e*a.obs_se] = E.qpos[k];
    for (int k = tlane(); k < NV; k += TEAM)
      a.obs[(long long)(NQ + k)*a.obs_sk + (long long)e*a.obs_se] = E.qvel[k];
    if (tlane() == 0) {
      a.reward[e] = 0;
      if (a.xpos) for (int i = 0; i < NBODY*3; i++) a.xpos[i*n + e] = E.xpos[i];
      if (a.xmat) for (int i = 0; i < NBODY*9; i++) a.xmat[i*n + e] = E.xmat[i];
      a.stats[e] = E.ncon; a.stats[n + e] = E.nefc + E.nmerged; a.stats[2*n + e] = E.iters;
    }
    return;
  }
  real obs[NOBS > 0 ? NOBS : 1];
  const real rew = task_outputs(E, a, obs);
  if (TEAMED && tlane() != 0) return;
  if (OBS_STAGE_FITS && a.obs_sk == 1 && a.obs_se == NOBS) {
    const int lane = threadIdx.x;
    DMC_UNROLL
    for (int k = 0; k < NOBS; k++) lds_base[k*LANES + lane] = obs[k];
    __syncthreads();
    const long long base = (long long)blockIdx.x*blockDim.x;
    const long long left = n - base;
    const int nvalid = left < (long long)blockDim.x ? (int)left : (int)blockDim.x;
    real* out = a.obs + base*NOBS;
    // lanes 0..nvalid-1 are exactly the active ones of a partial last block
    for (int w = lane; w < nvalid*NOBS; w += nvalid)
      out[w] = lds_base[(w % NOBS)*LANES + w/NOBS];
  } else {
    DMC_UNROLL
    for (int k = 0; k < NOBS; k++)
      a.obs[(long long)k*a.obs_sk + (long long)e*a.obs_se] = obs[k];
  }
  a.reward[e] = rew;
  if (accumulate) a.episode_return[e] += rew;
  DMC_UNROLL
  for (int s = 0; s < NSENSOR; s++) {
    const int adr = sensor_adr[s], o = sensor_objid[s];
    if (sensor_type[s] == 35)
      DMC_UNROLL
      for (int k = 0; k < 3; k++) a.sensordata[(adr + k)*n + e] = E.subtree_linvel[3*o + k];
    else if (sensor_type[s] == 34)
      DMC_UNROLL
      for (int k = 0; k < 3; k++) a.sensordata[(adr + k)*n + e] = E.subtree_com[3*o + k];
    else if (sensor_type[s] == 8)
      a.sensordata[adr*n + e] = E.qpos[jnt_qposadr[o]];
    else if (sensor_type[s] == 9)
      a.sensordata[adr*n + e] = E.qvel[jnt_dofadr[o]];
  }
  DMC_UNROLL
  for (int t = 0; t < NTOUCH; t++) a.sensordata[touch_adr[t]*n + e] = E.touch[t];
  if (a.xpos) {
    DMC_UNROLL
    for (int i = 0; i < NBODY*3; i++) a.xpos[i*n + e] = E.xpos[i];
  }
  if (a.xmat) {
    DMC_UNROLL
    for (int i = 0; i < NBODY*9; i++) a.xmat[i*n + e] = E.xmat[i];
  }
  // nefc as mj_makeConstraint counts it: a merged edge pair is two rows there
  a.stats[e] = E.ncon; a.stats[n + e] = E.nefc + E.nmerged; a.stats[2*n + e] = E.iters;
}

// nsub x Physics.step, then observation + reward of the new state.
// flags bit0: ctrl given (else reuse ctrl_store); bit1: skip outputs (settle)
#ifndef DMC_WAVES_PER_EU
#define DMC_WAVES_PER_EU 1
#endif
extern "C" __global__ void __launch_bounds__(LANES, DMC_WAVES_PER_EU)
dmc_step(DmcArgs a) {
  const int e = (int)((blockIdx.x*blockDim.x + threadIdx.x)/TEAM);
  if (e >= a.nenv) return;                 // (team mode: a team leaves together)
  Env E;
  real time;
  const long long n = a.nenv;
  __shared__ real lds_rows[TEAMED ? TEAM_LDS_WORDS*(LANES/TEAM) : LDS_WORDS];
  const Work W = TEAMED
      ? Work{lds_rows + (threadIdx.x/TEAM)*TEAM_LDS_WORDS, a.ws + (long long)e*WS_WORDS, 1}
      : Work{lds_rows + threadIdx.x, a.ws + e, n};
  team_bind(E, W);
  load_env(E, a, e, time);
  if (TEAMED) {
    // (a lane applies the actuators i = lane mod TEAM: it fetches those controls)
    const int tl = tlane();
    bool bc = false;
    for (int i = tl; i < NU; i += TEAM) {
      E.ctrl[i] = (a.flags & 1) ? a.ctrl[i*a.ctrl_sk + (long long)e*a.ctrl_se] : a.ctrl_store[i*n + e];
      bc |= bad(E.ctrl[i]);
    }
    if ((a.flags & 1) && tany(bc)) {
      E.warn |= WARN_BADCTRL;
      for (int i = tl; i < NU; i += TEAM) E.ctrl[i] = 0;
    }
    if (a.flags & 1)
      for (int i = tl; i < NU; i += TEAM) a.ctrl_store[i*n + e] = E.ctrl[i];
  } else if (a.flags & 1) {
    bool bc = false;
    DMC_UNROLL
    for (int i = 0; i < NU; i++) {
      E.ctrl[i] = a.ctrl[i*a.ctrl_sk + (long long)e*a.ctrl_se];
      bc |= bad(E.ctrl[i]);
    }
    if (bc) {   // mj_fwdActuation's ctrl check: warn and zero the controls
      E.warn |= WARN_BADCTRL;
      DMC_UNROLL
      for (int i = 0; i < NU; i++) E.ctrl[i] = 0;
    }
    if (!TEAMED || tlane() == 0)
    DMC_UNROLL
    for (int i = 0; i < NU; i++) a.ctrl_store[i*n + e] = E.ctrl[i];
  } else {
    DMC_UNROLL
    for (int i = 0; i < NU; i++) E.ctrl[i] = a.ctrl_store[i*n + e];
  }
  const real tol = R(tolerance_opt > DMC_TOL_FLOOR ? tolerance_opt : DMC_TOL_FLOOR);
#ifdef DMC_STEP_PROFILE
  const long long tk_ = wall_clock64();
#endif
  for (int s = 0; s < a.nsub; s++)
    physics_step(E, W, time, tol, s == 0 && (a.flags & DMC_FLAG_STALE_FIRST));
#ifdef DMC_STEP_PROFILE
  E.prof[7] = (real)(wall_clock64() - tk_);      // all substeps
#endif
  if (a.qacc && (!TEAMED || tlane() == 0)) {
    DMC_UNROLL
    for (int i = 0; i < NV; i++) a.qacc[i*n + e] = E.qacc[i];
  }
  if (!(a.flags & 2)) {
#ifndef DMC_ABLATE_OBS
    // (a model without a task: the observation is qpos and qvel; the frames are
    // only recomputed if someone reads them)
    if (TASK == TASK_NONE && NSENSOR == 0 && NTOUCH == 0 && !a.xpos && !a.xmat) check_state(E, time);
    else observe_stage(E, time);
#endif
    store_outputs(E, a, e, true, lds_rows);
  }
#ifdef DMC_SOLVER_PROFILE
  __syncthreads();
  for (int k = 0; k < 8 && k < NOBS; k++) a.obs[(long long)e*a.obs_se + k] = E.prof[k];
#endif
#ifdef DMC_STEP_PROFILE
  if (a.xpos && tlane() == 0) for (int k = 0; k < 8; k++) a.xpos[(long long)k*n + e] = E.prof[k];
#endif
  store_env(E, a, e, time);
}

// observation / reward / sensors of the current state (reset, after_reset)
extern "C" __global__ void __launch_bounds__(LANES, DMC_WAVES_PER_EU)
dmc_observe(DmcArgs a) {
  const int e = (int)((blockIdx.x*blockDim.x + threadIdx.x)/TEAM);
  if (e >= a.nenv) return;                 // (team mode: a team leaves together)
  Env E;
  real time;
  const long long n = a.nenv;
  __shared__ real lds_rows[TEAMED ? TEAM_LDS_WORDS*(LANES/TEAM) : LDS_WORDS];
  const Work W = TEAMED
      ? Work{lds_rows + (threadIdx.x/TEAM)*TEAM_LDS_WORDS, a.ws + (long long)e*WS_WORDS, 1}
      : Work{lds_rows + threadIdx.x, a.ws + e, n};
  team_bind(E, W);
  load_env(E, a, e, time);
  DMC_UNROLL
  for (int i = 0; i < NU; i++) E.ctrl[i] = a.ctrl_store[i*n + e];
  if (NTOUCH > 0) {
    // acceleration-stage sensors need the constraint forces: the reference's
    // after_reset runs mj_forward with actuation disabled (engine.py:283-295);
    // a bad state is reset first (mj_checkPos), as mj_forward would see it
    const real tol = R(tolerance_opt > DMC_TOL_FLOOR ? tolerance_opt : DMC_TOL_FLOOR);
    check_state(E, time);
    forward(E, W, false, tol);
  }
  const int ncon_forward = E.ncon, nefc_forward = E.nefc;
  observe_stage(E, time);
  if (NTOUCH > 0) {
    E.ncon = ncon_forward; E.nefc = nefc_forward;
  } else if (a.flags & 4) {   // count contacts (humanoid reset rejection test)
    E.ncon = 0; E.nefc = 0; E.nmerged = 0;
    if (NPAIR > 0) detect_contacts(E, W);
  }
  store_outputs(E, a, e, false, lds_rows);
  store_env(E, a, e, time);
}

#endif  // !DMC_COOP_BUILD

// stateless counter-based generator for on-device episode initialisation
DEV uint32_t mix32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (uint32_t)(x >> 16);
}
struct Rng {
  uint64_t key; uint32_t ctr;
  __device__ real uniform() {   // [0, 1)
    return (real)((mix32(key + 0x9e3779b97f4a7c15ULL*(++ctr)) >> 8)*(1.0/16777216.0));
  }
  __device__ real normal() {
    real u1 = uniform(), u2 = uniform();
    if (u1 < R(1e-7)) u1 = R(1e-7);
    return sqrt(-2*log(u1))*cos(R(6.283185307179586)*u2);
  }
};

// task.initialize_episode on device (distributional equivalent of
// cartpole.py:177-195, cheetah.py:63-70, randomizers.py:35-86; the host RNG
// stream of the reference cannot be reproduced in a batched kernel).
// flags bit3: only redraw envs whose stats[ncon] > 0 (humanoid rejection loop)
extern "C" __global__ void __launch_bounds__(64)
dmc_init_episode(DmcArgs a) {
  const int e = blockIdx.x*blockDim.x + threadIdx.x;   // one env per lane in every build
  if (e >= a.nenv) return;
  const long long n = a.nenv;
  if ((a.flags & 8) && a.stats[sidx(0, e, n, 3)] == 0) return;
  Rng rng = {a.seed*0x2545F4914F6CDD1DULL + (uint64_t)e, 0};
  real qpos[NQ > 0 ? NQ : 1], qvel[NVX];
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) qpos[i] = R(qpos0[i]);
  DMC_UNROLL
  for (int i = 0; i < NV; i++) qvel[i] = 0;
  if (a.flags & DMC_FLAG_RESET_ONLY) {
    // mj_resetData only
  } else if (TASK == TASK_CARTPOLE) {
    if (a.task_param_i & 2) {   // swing_up
      qpos[0] = R(0.01)*rng.normal();
      qpos[1] = R(3.141592653589793) + R(0.01)*rng.normal();
      DMC_UNROLL
      for (int i = 2; i < NQ; i++) qpos[i] = R(0.1)*rng.normal();
    } else {
      qpos[0] = R(-0.1) + R(0.2)*rng.uniform();
      DMC_UNROLL
      for (int i = 1; i < NQ; i++) qpos[i] = R(-0.034) + R(0.068)*rng.uniform();
    }
    DMC_UNROLL
    for (int i = 0; i < NV; i++) qvel[i] = R(0.01)*rng.normal();
  } else if (TASK == TASK_CHEETAH || TASK == TASK_HUMANOID ||
             TASK == TASK_WALKER || TASK == TASK_PENDULUM || TASK == TASK_ACROBOT ||
             TASK == TASK_HOPPER || TASK == TASK_REACHER || TASK == TASK_POINTMASS) {
    DMC_UNROLL
    for (int j = 0; j < NJNT; j++) {
      const int qa = jnt_qposadr[j];
      if (jnt_limited[j] && (jnt_type[j] == JNT_HINGE || jnt_type[j] == JNT_SLIDE)) {
        const real lo = R(jnt_range[2*j]), hi = R(jnt_range[2*j + 1]);
        qpos[qa] = lo + (hi - lo)*rng.uniform();
      } else if (TASK != TASK_CHEETAH && !jnt_limited[j]) {
        if (jnt_type[j] == JNT_HINGE) {
          qpos[qa] = R(-3.141592653589793) + R(6.283185307179586)*rng.uniform();
        } else if (jnt_type[j] == JNT_FREE) {
          real q[4];
          DMC_UNROLL
          for (int k = 0; k < 4; k++) q[k] = rng.uniform();
          normalize4(q);
          DMC_UNROLL
          for (int k = 0; k < 4; k++) qpos[qa + 3 + k] = q[k];
        }
      }
    }
  }
  if (a.flags & DMC_FLAG_TASKDATA_DEFAULT) {   // batch creation: the compiled model's values
    DMC_UNROLL
    for (int i = 0; i < NTASKDATA; i++)
      a.taskdata[sidx(i, e, n, NTDX)] = R(task_data_default[i]);
  }
  if (TASK == TASK_POINTMASS && !(a.flags & DMC_FLAG_RESET_ONLY)) {
    // point_mass.py:103-113: each control drives a random direction in the
    // plane ("hard"), or the model's own axes ("easy")
    real d1[2] = {R(task_data_default[0]), R(task_data_default[1])};
    real d2[2] = {R(task_data_default[2]), R(task_data_default[3])};
    if (a.task_param_i & 1) {
      real nrm;
      d1[0] = rng.normal(); d1[1] = rng.normal();
      nrm = sqrt(d1[0]*d1[0] + d1[1]*d1[1]); d1[0] /= nrm; d1[1] /= nrm;
      for (int tries = 0; tries < 64; tries++) {
        d2[0] = rng.normal(); d2[1] = rng.normal();
        nrm = sqrt(d2[0]*d2[0] + d2[1]*d2[1]); d2[0] /= nrm; d2[1] /= nrm;
        if (!(fabs(d1[0]*d2[0] + d1[1]*d2[1]) > R(0.9))) break;
      }
    }
    a.taskdata[sidx(0, e, n, NTDX)] = d1[0]; a.taskdata[sidx(1, e, n, NTDX)] = d1[1];
    a.taskdata[sidx(2, e, n, NTDX)] = d2[0]; a.taskdata[sidx(3, e, n, NTDX)] = d2[1];
  }
  if (TASK == TASK_REACHER && !(a.flags & DMC_FLAG_RESET_ONLY)) {
    // reacher.py:100-104: target on a ring around the shoulder
    const real angle = R(6.283185307179586)*rng.uniform();
    const real radius = R(0.05) + R(0.15)*rng.uniform();
    a.taskdata[sidx(0, e, n, NTDX)] = radius*sin(angle);
    a.taskdata[sidx(1, e, n, NTDX)] = radius*cos(angle);
  }
  DMC_UNROLL
  for (int i = 0; i < NQ; i++) a.qpos[sidx(i, e, n, NQX)] = qpos[i];
  DMC_UNROLL
  for (int i = 0; i < NV; i++) { a.qvel[sidx(i, e, n, NVX)] = qvel[i]; a.warm[sidx(i, e, n, NVX)] = 0; }
  DMC_UNROLL
  for (int i = 0; i < NU; i++) a.ctrl_store[sidx(i, e, n, NUX)] = 0;
  a.time[e] = 0;
  a.episode_return[e] = 0;
}

#ifndef DMC_COOP_BUILD
// self-description read by dmc_api.cpp through hipModuleGetGlobal
extern "C" __device__ const int dmc_info[20] = {
    1 /*abi*/, (int)sizeof(real), NQ, NV, NU, NBODY, NOBS, NSENSORDATA,
    (WS_WORDS > 0 ? WS_WORDS : 1) /*workspace reals per env*/, TASK, NCON_MAX, NEFC_MAX,
    INTEGRATOR, NPAIR, LANES/TEAM /*envs per workgroup of dmc_step/dmc_observe (= its
                                threads unless a team of lanes shares an env);
                                the workspace is sized for the batch rounded up to this*/,
    DMC_ENV_MAJOR /*0: state fields are [k][env]*/, NTASKDATA,
    LANES /*threads per workgroup*/, 0, 0};
#endif
