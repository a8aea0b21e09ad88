// dmc_coop.hip -- the physics step with SEVERAL LANES PER ENVIRONMENT.
//
// csrc/dmc_kernels.hip gives every environment one lane and keeps its whole
// working set in that lane's registers.  That is the right shape for the small
// planar models (nv <= ~10), and the wrong one for the 27-dof humanoid: 378
// mass-matrix entries + 378 Hessian entries + the tree state do not fit one
// lane's register file, and a 1024-env shard (BASELINE configs[3]) is only 16
// waves on a 1024-SIMD chip.  Here a GROUP of G lanes (default G = 64: one env
// per wavefront; 32 = two envs per wavefront is used for mid-size walker /
// hopper batches) advances one environment together:
//
//   * the env's working set lives in LDS (one region per env, ~35 KB fp32 for
//     the humanoid incl. all 149 constraint rows -- no HBM workspace at all);
//   * tree passes run one lane per body, level by level; dof-indexed work
//     (mass-matrix rows, Jacobian columns, Hessian rows) one lane per dof;
//     row-indexed work (reference acceleration, line search sums) one lane per
//     constraint row; the static geom-pair list is strided over the lanes and
//     compacted with an ordered prefix sum, so contacts and rows come out in
//     exactly the order of the one-lane kernel and the CPU oracle;
//   * Cholesky factor / solves of the nv x nv matrices are cooperative: lane
//     i holds row i in registers, pivot rows travel by v_readlane, finished
//     4-column panels through LDS (see `rows_chol`);
//   * the 2-D state fields of these code objects are env-major in HBM
//     ([env][k], `sidx`), so a group's loads and stores are unit-stride.
//
// Lanes of a group communicate through LDS between phases (`gsync`) and
// through DPP / readlane reductions; groups never interact, so the envs of a
// wave may take different numbers of Newton iterations.
//
// Same entry points, argument block and outputs as dmc_kernels.hip; the host
// only needs `dmc_info` (envs and threads per workgroup, state layout) to size
// the grid and to present the fields.
// Reference path replaced: see dmc_kernels.hip (Physics.step, mj_step*, task
// observation/reward of dm_control.suite).

#define DMC_COOP_BUILD 1
#include "dmc_kernels.hip"

#ifndef DMC_GROUP
#define DMC_GROUP 64
#endif
constexpr int G = DMC_GROUP;            // lanes per environment
constexpr int EPB = 64/G;               // environments per 64-lane workgroup
static_assert(G == 8 || G == 16 || G == 32 || G == 64, "group must divide a wave");
constexpr int RNV = (NVX + G - 1)/G;    // lane rounds needed to cover the dofs

constexpr int odd_(int x) { return x | 1; }
constexpr int NJX = NJNT > 0 ? NJNT : 1;
constexpr int NGX = NGEOM > 0 ? NGEOM : 1;
constexpr int NOBSX = NOBS > 0 ? NOBS : 1;
constexpr int NSDX = NSENSORDATA > 0 ? NSENSORDATA : 1;
constexpr int NVP = odd_(NVX);          // row stride of M and H (odd: lane = row is conflict-free)
// constraint row record: J(nv), D, aref, Jaref, Jv, force
enum { CR_D = NV, CR_AREF = NV + 1, CR_JAR = NV + 2, CR_JV = NV + 3, CR_F = NV + 4,
       // the reference acceleration is dead once the warm start has been rated:
       // its word then carries the row's pending Hessian change (+1 enters the
       // active set, -1 leaves it, 0 stays)
       CR_FLIP = CR_AREF };
constexpr int CRW = odd_(NV + 5);
// contact record: pos(3) normal(3) tangent hint(3) dist pair first-row
enum { CC_DIST = 9, CC_PAIR = 10, CC_ROW = 11 };
constexpr int CCW = 13;
constexpr bool RK4 = INTEGRATOR != 0;
// Two wavefronts per env (G = 64, Euler models): between the position stage and
// the solver, wave 0 computes the mass matrix, the velocity stage, the smooth
// forces and qacc_smooth while wave 1 builds the constraint rows (limits,
// collision detection, contact Jacobians) -- two chains that only share
// read-only inputs.  Everything else is wave 0's; wave 1 waits at the next
// workgroup barrier.  Two barriers per step (`wsync`), see forward().
#ifndef DMC_COOP_DUO
#define DMC_COOP_DUO 0
#endif
constexpr bool DUO = DMC_COOP_DUO != 0 && G == 64 && !RK4;
constexpr int NTHREADS = DUO ? 128 : 64;
constexpr bool coop_damped() {
  bool d = false;
  for (int i = 0; i < NV; i++) d = d || dof_damping[i] > 0;
  return d;
}

// word offsets of the per-env LDS region
namespace off {
constexpr int QPOS = 0;
constexpr int QVEL = QPOS + NQX;
constexpr int CTRL = QVEL + NVX;
constexpr int WARM = CTRL + NUX;
constexpr int XPOS = WARM + NVX;
constexpr int XQUAT = XPOS + NBODY*3;
constexpr int XMAT = XQUAT + NBODY*4;
constexpr int XIPOS = XMAT + NBODY*9;
constexpr int SUBCOM = XIPOS + NBODY*3;
constexpr int CDOF = SUBCOM + NBODY*3;
constexpr int CVEL = CDOF + NVX*6;
constexpr int MM = CVEL + NBODY*6;
// tree temporaries that are dead once qfrc_smooth exists; the triangular
// solves reuse their space as the transposition scratch HH
constexpr int TMP0 = ((MM + NVX*NVP + 3)/4)*4;   // 16-byte aligned: wide panel loads
constexpr int XIMAT = TMP0;
constexpr int XANCHOR = XIMAT + NBODY*9;
constexpr int XAXIS = XANCHOR + NJX*3;
constexpr int CINERT = XAXIS + NJX*3;
constexpr int CRB = CINERT + NBODY*10;
constexpr int CDOFDOT = CRB + NBODY*10;
constexpr int CACC = CDOFDOT + NVX*6;
constexpr int CFRC = CACC + NBODY*6;
constexpr int TMP1 = CFRC + NBODY*6;
constexpr int HH = TMP0;
#ifndef DMC_COOP_CB
#define DMC_COOP_CB 4
#endif
constexpr int HH_WORDS = NVX*NVP > 2*NVX*DMC_COOP_CB ? NVX*NVP : 2*NVX*DMC_COOP_CB;   // solve scratch / factor panels
constexpr int TMPEND = TMP1 - TMP0 > HH_WORDS ? TMP1 : TMP0 + HH_WORDS;
constexpr int FS = TMPEND;              // qfrc_smooth
constexpr int FC = FS + NVX;            // qfrc_constraint
constexpr int QAS = FC + NVX;           // qacc_smooth
constexpr int QACC = QAS + NVX;
constexpr int MA = QACC + NVX;
constexpr int MV = MA + NVX;
constexpr int GRAD = MV + NVX;
constexpr int SEARCH = GRAD + NVX;
constexpr int JQ = MA;                  // joint rotations of the position stage (4 per joint; the solver vectors are dead then)
constexpr int GEOM = SEARCH + NVX;
constexpr int CON = GEOM + NGX*12;
constexpr int ROWS = CON + NCON_MAX*CCW;
constexpr int SLV = ROWS + NEFC_MAX*CRW;
constexpr int TOUCH = SLV + NBODY*3;     // touch sensor readings
constexpr int XCH = TOUCH + (NTOUCH > 0 ? NTOUCH : 1);      // nefc, ncon, warn from the row-building wave
constexpr int TASKD = XCH + 4;          // (+ the velocity-stage flag) per-instance task parameters
constexpr int OBSV = TASKD + NTDX;
constexpr int Q0 = OBSV + NOBSX;        // RK4 stage storage
constexpr int V0 = Q0 + (RK4 ? NQX : 0);
constexpr int FV = V0 + (RK4 ? NVX : 0);
constexpr int FA = FV + (RK4 ? 4*NVX : 0);
constexpr int DV = FA + (RK4 ? 4*NVX : 0);
constexpr int END = DV + (RK4 ? NVX : 0);
}  // namespace off
// region stride: groups that share a 32-lane LDS half start G banks apart
constexpr int ENV_WORDS = ((off::END + 31)/32)*32 + (G % 32);
static_assert(NJX <= NVX, "joint rotations are staged in the solver vectors");
// DUO: wave 1 factorises M + h D while wave 0 runs the solver (M is final
// before the second barrier of forward(); the panel scratch is the geom
// frames, dead once the contacts exist) and solves when the forces are final
constexpr bool EULER_OFFLOAD = DUO && coop_damped() && NGX*12 >= 2*NVX*DMC_COOP_CB;
static_assert((long long)ENV_WORDS*EPB*sizeof(real) <= 150*1024,
              "the env working set does not fit in LDS; use the one-lane kernel");

// ---------------------------------------------------------------------------
// group primitives
// ---------------------------------------------------------------------------
#ifndef DMC_HOST_SHIM
// LDS hand-over between phases.  The workgroup is one wave and the lanes of a
// group run in lock step, so no hardware barrier is needed: what must be
// prevented is the compiler moving LDS accesses across the phase boundary.
DEV void gsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
DEV real gxor(real x, int m) { return __shfl_xor(x, m, G); }
DEV int gxor(int x, int m) { return __shfl_xor(x, m, G); }
DEV int gup(int x, int d) { return __shfl_up(x, d, G); }
DEV int gget(int x, int src) { return __shfl(x, src, G); }
DEV real gget(real x, int src) { return __shfl(x, src, G); }
// value of `x` in lane `src` of the caller's group, `src` uniform over the
// wave: v_readlane per group instead of a trip through the LDS crossbar
DEV int gbcast_bits(int x, int src) {
  if (G == 64) return __builtin_amdgcn_readlane(x, src);
  if (G == 32) {
    const int lo = __builtin_amdgcn_readlane(x, src);
    const int hi = __builtin_amdgcn_readlane(x, src + 32);
    return (threadIdx.x & 32) ? hi : lo;
  }
  return __shfl(x, src, G);
}
DEV int gbcast(int x, int src) { return gbcast_bits(x, src); }
DEV float gbcast(float x, int src) {
  return __int_as_float(gbcast_bits(__float_as_int(x), src));
}
DEV double gbcast(double x, int src) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)gbcast_bits((int)(unsigned)(b & 0xffffffffLL), src);
  const unsigned hi = (unsigned)gbcast_bits((int)(unsigned)((unsigned long long)b >> 32), src);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#endif
// a value the optimiser must treat as unknown at this point
#ifndef DMC_HOST_SHIM
DEV int opaque(int x) { asm volatile("" : "+v"(x)); return x; }
#else
DEV int opaque(int x) { return x; }
#endif
// hand-over between the two wavefronts of an env (a real workgroup barrier)
DEV void wsync() { if (DUO) __syncthreads(); }
// Group reductions.  Butterfly sums: every lane ends with the bitwise-identical
// total (each stage adds the same two partial sums in both partner lanes), so
// control flow that depends on them stays uniform inside the group.  On the
// GPU the stages inside a 16-lane row are DPP moves (quad permutes and the two
// row mirrors pair exactly the lanes an xor butterfly would pair once the
// smaller blocks are uniform); only the row-to-row stage crosses through the
// LDS crossbar.
#ifndef DMC_HOST_SHIM
template <int CTRL>
DEV int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xf, 0xf, true); }
template <int CTRL> DEV int dpp(int x) { return dpp_i<CTRL>(x); }
template <int CTRL> DEV float dpp(float x) { return __int_as_float(dpp_i<CTRL>(__float_as_int(x))); }
template <int CTRL> DEV double dpp(double x) {
  const long long b = __double_as_longlong(x);
  const unsigned lo = (unsigned)dpp_i<CTRL>((int)(unsigned)(b & 0xffffffffLL));
  const unsigned hi = (unsigned)dpp_i<CTRL>((int)(unsigned)((unsigned long long)b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
enum { DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140,
       DPP_SHR1 = 0x111, DPP_SHR2 = 0x112, DPP_SHR4 = 0x114, DPP_SHR8 = 0x118 };
template <class T>
DEV T gsum_t(T x) {
  if (G >= 16) {
    x += dpp<DPP_XOR1>(x);
    x += dpp<DPP_XOR2>(x);
    x += dpp<DPP_HALF_MIRROR>(x);
    x += dpp<DPP_MIRROR>(x);
    if (G == 64) {   // four uniform row totals: add them in a fixed order
      const T r0 = gbcast(x, 0), r1 = gbcast(x, 16), r2 = gbcast(x, 32), r3 = gbcast(x, 48);
      x = (r0 + r1) + (r2 + r3);
    } else {
      for (int m = 16; m < G; m <<= 1) x += gxor(x, m);
    }
  } else {
    for (int m = G/2; m > 0; m >>= 1) x += gxor(x, m);
  }
  return x;
}
// exclusive prefix sum in lane order + group total
DEV int gscan(int x, int lane, int& total) {
  int v = x;
  if (G >= 16) {
    v += dpp<DPP_SHR1>(v);      // bound_ctrl: lanes shifted in from outside the row read 0
    v += dpp<DPP_SHR2>(v);
    v += dpp<DPP_SHR4>(v);
    v += dpp<DPP_SHR8>(v);
    for (int r = 16; r < G; r += 16) {   // carry the running total into the next row
      const int t = gbcast_bits(v, r - 1);
      if (lane >= r && lane < r + 16) v += t;
    }
  } else {
    for (int d = 1; d < G; d <<= 1) {
      const int t = gup(v, d);
      if (lane >= d) v += t;
    }
  }
  total = gbcast_bits(v, G - 1);
  return v - x;
}
#else
template <class T>
DEV T gsum_t(T x) {
  for (int m = G/2; m > 0; m >>= 1) x += gxor(x, m);
  return x;
}
DEV int gscan(int x, int lane, int& total) {
  int v = x;
  for (int d = 1; d < G; d <<= 1) {
    const int t = gup(v, d);
    if (lane >= d) v += t;
  }
  total = gget(v, G - 1);
  return v - x;
}
#endif
// which lanes of the group hold `b` (bit = lane within the group)
#ifndef DMC_HOST_SHIM
DEV unsigned long long gballot(bool b) {
  const unsigned long long w = __ballot(b);
  if (G == 64) return w;
  return (w >> ((threadIdx.x/G)*G)) & ((1ull << G) - 1);
}
DEV int first_bit(unsigned long long m) { return __ffsll((unsigned long long)m) - 1; }
#else
DEV unsigned long long gballot(bool b) {
  unsigned long long x = b ? 1ull << (threadIdx.x % G) : 0ull;
  for (int m = G/2; m > 0; m >>= 1) x |= gxor(x, m);
  return x;
}
DEV int first_bit(unsigned long long m) { return __builtin_ctzll(m); }
#endif
DEV real gsum(real x) { return gsum_t(x); }
DEV int gsum(int x) { return gsum_t(x); }
DEV bool gany(bool b) { return gsum(b ? 1 : 0) != 0; }

// -DDMC_COOP_PROFILE: per-phase time (100 MHz ticks) of every env, summed over
// the launch, written over the first words of the env's observation (experiments only)
#ifdef DMC_COOP_PROFILE
enum { PH_KIN, PH_COM, PH_CRB, PH_FACM, PH_VEL, PH_SMOOTH, PH_LIMIT, PH_DETECT, PH_CROWS,
       PH_FINISH, PH_WARM, PH_HESS, PH_FACH, PH_SOLVE, PH_LS, PH_UPD, PH_EULER, PH_OBS, PH_N };
#define PROF(k) do { const long long t_ = wall_clock64(); tprof[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define PROF(k) do {} while (0)
#endif

// Model tables that the tree / row passes index with per-lane values (body,
// joint and dof ids differ from lane to lane, so these cannot be scalar loads).
// They are staged once per workgroup in LDS -- a dependent chain of three
// global loads per tree level (level_body -> body_jntadr -> jnt_qposadr)
// otherwise costs more than the arithmetic of the level.  Inside `Coop` the
// table names below resolve to these LDS mirrors; integers are stored as reals
// (all are < 2^24).
#define COOP_TABLES(X) \
  X(I, level_adr, NLEVEL + 1) X(I, level_body, NBODY > 1 ? NBODY - 1 : 1) \
  X(I, body_parentid, NBODY) X(I, body_rootid, NBODY) X(I, body_jntnum, NBODY) \
  X(I, body_jntadr, NBODY) X(I, body_subtree_n, NBODY) \
  X(R, body_pos, 3*NBODY) X(R, body_quat, 4*NBODY) X(R, body_ipos, 3*NBODY) \
  X(R, body_iquat, 4*NBODY) X(R, body_mass, NBODY) X(R, body_subtreemass, NBODY) \
  X(R, body_inertia, 3*NBODY) \
  X(I, jnt_type, NJX) X(I, jnt_qposadr, NJX) X(I, jnt_dofadr, NJX) X(I, jnt_bodyid, NJX) \
  X(R, jnt_pos, 3*NJX) X(R, jnt_axis, 3*NJX) X(R, jnt_stiffness, NJX) \
  X(R, jnt_range, 2*NJX) X(R, jnt_margin, NJX) \
  X(R, qpos0, NQX) X(R, qpos_spring, NQX) \
  X(I, dof_bodyid, NVX) X(I, dof_jntid, NVX) X(I, dof_anc_len, NVX) X(I, dof_anc, NVX*MAXCHAIN) \
  X(I, body_lastdof, NBODY) \
  X(R, dof_armature, NVX) X(R, dof_damping, NVX) X(R, dof_invweight0, NVX) \
  X(I, geom_bodyid, NGX) X(R, geom_pos, 3*NGX) X(R, geom_quat, 4*NGX)
namespace tb {
enum : int {
#define X(kind, name, n) name##_off, name##_last = name##_off + (n) - 1,
  COOP_TABLES(X)
#undef X
  END
};
}  // namespace tb
__shared__ real coop_tab[tb::END];
template <int OFF> struct LdsTabI {
  __device__ __forceinline__ int operator[](int i) const { return (int)coop_tab[OFF + i]; }
};
template <int OFF> struct LdsTabR {
  __device__ __forceinline__ real operator[](int i) const { return coop_tab[OFF + i]; }
};
// every lane of the workgroup takes part (call before any lane leaves)
DEV void stage_tables() {
#define X(kind, name, n) \
  for (int k = threadIdx.x; k < (n); k += NTHREADS) coop_tab[tb::name##_off + k] = (real)dmc_model::name[k];
  COOP_TABLES(X)
#undef X
  __syncthreads();
}

struct EnvView {   // the fields task_outputs() reads, as LDS pointers
  const real *qpos, *qvel, *ctrl, *xpos, *xmat, *xipos, *subtree_linvel, *touch,
             *taskdata;
};

// ---------------------------------------------------------------------------
// one environment, advanced by the G lanes of its group
// ---------------------------------------------------------------------------
struct Coop {
  real* S;         // LDS region of this env
  int l;           // lane within the group
  real time;
  unsigned warn;
  int ncon, nefc, iters;
  int epoch;       // forward() calls so far (DUO: value of the velocity-stage flag)
#ifdef DMC_COOP_PROFILE
  long long tprof[PH_N], tlast;
#endif
  // LDS mirrors shadow the namespace-scope tables inside the member functions
#define X(kind, name, n) LdsTab##kind<tb::name##_off> name;
  COOP_TABLES(X)
#undef X

  // one-way signal between the two wavefronts of an env (DUO): an LDS word
  // written with release and polled with acquire semantics
#ifndef DMC_HOST_SHIM
  __device__ __forceinline__ void flag_write(int word, int value) {
    __hip_atomic_store(reinterpret_cast<int*>(S + word), value, __ATOMIC_RELEASE,
                       __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ int flag_read(int word) const {
    return __hip_atomic_load(reinterpret_cast<const int*>(S + word), __ATOMIC_ACQUIRE,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
  }
#else
  __device__ __forceinline__ void flag_write(int word, int value) {
    __atomic_store_n(reinterpret_cast<int*>(S + word), value, __ATOMIC_RELEASE);
  }
  __device__ __forceinline__ int flag_read(int word) const {
    return __atomic_load_n(reinterpret_cast<const int*>(S + word), __ATOMIC_ACQUIRE);
  }
#endif
  __device__ __forceinline__ static void flag_pause() {
#ifndef DMC_HOST_SHIM
    __builtin_amdgcn_s_sleep(2);
#else
    sched_yield();
#endif
  }

  // ---- cooperative dense linear algebra -----------------------------------
  // Lane (l + t*G) keeps row (l + t*G) of a symmetric NV x NV matrix in
  // registers (`Rows`); all indices below are compile-time after unrolling and
  // the pivot row reaches the other lanes through register broadcasts, so a
  // factorisation is ~NV^2/2 broadcast+FMA pairs with no memory latency in the
  // dependence chain.  (A first version kept the matrix in LDS and synchronised
  // per column: 25 us per 27 x 27 factorisation, every inner step paying an LDS
  // round trip.)
  struct Rows { real a[RNV][NVX]; };

  __device__ __forceinline__ void rows_load(Rows& A, int src) const {
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) {
      const int i = l + t*G;
      _Pragma("unroll")
      for (int k = 0; k < NV; k++) A.a[t][k] = i < NV ? S[src + i*NVP + k] : R(0);
    }
  }
  // in place A = L L^T; lane i ends with L[i][0..i-1] and 1/L[i][i] in a[i];
  // entries right of the diagonal are dead.  Right-looking in blocks of CB
  // columns: inside a block the pivot row travels by register broadcast; the
  // finished panel (CB values per row) goes to LDS once and every lane reads
  // row k of it with one wide uniform-address load for the update of its
  // column k -- one LDS round trip per block instead of one broadcast per
  // (column, column) pair.  The panel buffer is the solve scratch at `panel`
  // (two halves used alternately, so one phase boundary per block suffices).
  static constexpr int CB = DMC_COOP_CB;
  __device__ __forceinline__ int rows_chol(Rows& A, int panel) {
    int nbad = 0;
    _Pragma("unroll")
    for (int jb = 0; jb < NV; jb += CB) {
      const int w = NV - jb < CB ? NV - jb : CB;
      real pl[RNV][CB];
      // The diagonal block travels to every lane first (independent register
      // broadcasts) and every lane factorises it for itself, repeating the
      // pivot lanes' arithmetic bit for bit: the columns of the block then
      // need no cross-lane traffic inside their dependent chain.
      real Dg[CB][CB], invd[CB];
      _Pragma("unroll")
      for (int a = 0; a < w; a++) {
        _Pragma("unroll")
        for (int b = 0; b <= a; b++)
          Dg[a][b] = gbcast(A.a[(jb + a)/G][jb + b], (jb + a) % G);
      }
      _Pragma("unroll")
      for (int jj = 0; jj < w; jj++) {
        real d = Dg[jj][jj];
        if (!(d >= DMC_MINVAL)) { d = DMC_MINVAL; nbad++; }
        invd[jj] = rsqrt_(d);
        _Pragma("unroll")
        for (int a = jj + 1; a < w; a++) Dg[a][jj] *= invd[jj];
        _Pragma("unroll")
        for (int a = jj + 1; a < w; a++) {
          _Pragma("unroll")
          for (int b = jj + 1; b <= a; b++) Dg[a][b] -= Dg[a][jj]*Dg[b][jj];
        }
      }
      _Pragma("unroll")
      for (int jj = 0; jj < w; jj++) {
        const int j = jb + jj;
        _Pragma("unroll")
        for (int t = 0; t < RNV; t++) {
          pl[t][jj] = A.a[t][j]*invd[jj];
          A.a[t][j] = (l + t*G == j) ? invd[jj] : pl[t][jj];
        }
        _Pragma("unroll")
        for (int k = jj + 1; k < w; k++) {
          _Pragma("unroll")
          for (int t = 0; t < RNV; t++) A.a[t][jb + k] -= pl[t][jj]*Dg[k][jj];
        }
      }
      if (jb + w < NV) {
        const int buf = panel + ((jb/CB) & 1)*NVX*CB;
        _Pragma("unroll")
        for (int t = 0; t < RNV; t++) {
          const int i = l + t*G;
          if (i < NV) {
            _Pragma("unroll")
            for (int m = 0; m < w; m++) S[buf + i*CB + m] = pl[t][m];
          }
        }
        gsync();
        _Pragma("unroll")
        for (int k = jb + w; k < NV; k++) {
          real pk[CB];
          _Pragma("unroll")
          for (int m = 0; m < w; m++) pk[m] = S[buf + k*CB + m];
          _Pragma("unroll")
          for (int t = 0; t < RNV; t++) {
            _Pragma("unroll")
            for (int m = 0; m < w; m++) A.a[t][k] -= pl[t][m]*pk[m];
          }
        }
      }
    }
    return nbad;
  }
  // x = (L L^T)^-1 b, b/x held one entry per lane row.  The backward sweep
  // needs column i of L in lane i: the factor is transposed through the LDS
  // square at `scratch` (one write + one read per entry, all independent).
  __device__ __forceinline__ void rows_solve(const Rows& A, int scratch, real* b) {
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) {
      const int i = l + t*G;
      if (i < NV) {
        _Pragma("unroll")
        for (int k = 0; k < NV; k++) S[scratch + i*NVP + k] = A.a[t][k];
      }
    }
    _Pragma("unroll")
    for (int k = 0; k < NV; k++) {          // forward: L y = b
      const real yk = gbcast(b[k/G]*A.a[k/G][k], k % G);
      _Pragma("unroll")
      for (int t = 0; t < RNV; t++) {
        const int i = l + t*G;
        b[t] = i == k ? yk : (i > k ? b[t] - A.a[t][k]*yk : b[t]);
      }
    }
    gsync();
    real lt[RNV][NVX];
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) {
      const int i = l + t*G;
      _Pragma("unroll")
      for (int k = 0; k < NV; k++) lt[t][k] = i < NV ? S[scratch + k*NVP + i] : R(0);
    }
    _Pragma("unroll")
    for (int k = NV - 1; k >= 0; k--) {     // backward: L^T x = y
      const real xk = gbcast(b[k/G]*A.a[k/G][k], k % G);
      _Pragma("unroll")
      for (int t = 0; t < RNV; t++) {
        const int i = l + t*G;
        b[t] = i == k ? xk : (i < k ? b[t] - lt[t][k]*xk : b[t]);
      }
    }
    gsync();   // `scratch` may be rewritten by the caller from here on
  }
  // y_i = sum_k M_ik x_k for this lane's rows (M stored full and symmetric)
  __device__ __forceinline__ real mrow_dot(int i, int x) const {
    real s = 0;
    _Pragma("unroll")
    for (int k = 0; k < NV; k++) s += S[off::MM + i*NVP + k]*S[x + k];
    return s;
  }

  // ---- position stage ------------------------------------------------------
  // The tree is walked once per level, and the wave waits for the slowest lane
  // of each level -- so everything that does not depend on the parent's world
  // frame is taken out of the walk and done in parallel first:
  //   (1) lane = joint: the joint's own rotation quaternion (the sin/cos);
  //   (2) lane = body:  the body's pose RELATIVE TO ITS PARENT with its joints
  //       applied in order, and the joints' anchors and axes in that frame;
  //   (3) level walk:   world pose = parent's world pose o relative pose;
  //   (4) lane = joint / body: anchors, axes and inertial frames in the world.
  // mj_kinematics composes the same transforms joint by joint in the world
  // frame; the two orders agree to rounding.
  __device__ __forceinline__ void kinematics() {
    for (int j = l; j < NJNT; j += G) {
      const int jt = jnt_type[j], qa = jnt_qposadr[j];
      real q[4] = {1, 0, 0, 0};
      if (jt == JNT_HINGE) {
        const real jax[3] = {R(jnt_axis[3*j]), R(jnt_axis[3*j + 1]), R(jnt_axis[3*j + 2])};
        axisangle2quat(q, jax, S[off::QPOS + qa] - R(qpos0[qa]));
      } else if (jt == JNT_BALL) {
        for (int k = 0; k < 4; k++) q[k] = S[off::QPOS + qa + k];
        normalize4(q);
      } else if (jt == JNT_SLIDE) {
        q[0] = S[off::QPOS + qa] - R(qpos0[qa]);   // displacement along the axis
      }
      for (int k = 0; k < 4; k++) S[off::JQ + 4*j + k] = q[k];
    }
    gsync();
    for (int i = l; i < NBODY; i += G) {
      real lp[3] = {0, 0, 0}, lq[4] = {1, 0, 0, 0};
      if (i > 0) {
        const int jadr = body_jntadr[i], jnum = body_jntnum[i];
        if (jnum == 1 && jnt_type[jadr < 0 ? 0 : jadr] == JNT_FREE) {
          const int qa = jnt_qposadr[jadr];
          for (int k = 0; k < 3; k++) lp[k] = S[off::QPOS + qa + k];
          for (int k = 0; k < 4; k++) lq[k] = S[off::QPOS + qa + 3 + k];
          normalize4(lq);
          for (int k = 0; k < 3; k++) {
            S[off::XANCHOR + 3*jadr + k] = lp[k];
            S[off::XAXIS + 3*jadr + k] = R(jnt_axis[3*jadr + k]);
          }
        } else {
          for (int k = 0; k < 3; k++) lp[k] = R(body_pos[3*i + k]);
          for (int k = 0; k < 4; k++) lq[k] = R(body_quat[4*i + k]);
          for (int j = 0; j < jnum; j++) {
            const int jid = jadr + j, jt = jnt_type[jid];
            const real jax[3] = {R(jnt_axis[3*jid]), R(jnt_axis[3*jid + 1]), R(jnt_axis[3*jid + 2])};
            const real jp[3] = {R(jnt_pos[3*jid]), R(jnt_pos[3*jid + 1]), R(jnt_pos[3*jid + 2])};
            real anchor[3], axis[3], m[9];
            quat2mat(m, lq);
            mulmatvec3(axis, m, jax);
            mulmatvec3(anchor, m, jp);
            for (int k = 0; k < 3; k++) anchor[k] += lp[k];
            for (int k = 0; k < 3; k++) {
              S[off::XANCHOR + 3*jid + k] = anchor[k];
              S[off::XAXIS + 3*jid + k] = axis[k];
            }
            if (jt == JNT_SLIDE) {
              const real q = S[off::JQ + 4*jid];
              for (int k = 0; k < 3; k++) lp[k] += axis[k]*q;
            } else if (jt == JNT_HINGE || jt == JNT_BALL) {
              real qloc[4], r[4], vec[3];
              for (int k = 0; k < 4; k++) qloc[k] = S[off::JQ + 4*jid + k];
              mulquat(r, lq, qloc);
              for (int k = 0; k < 4; k++) lq[k] = r[k];
              rotvecquat(vec, jp, lq);
              for (int k = 0; k < 3; k++) lp[k] = anchor[k] - vec[k];
            }
          }
        }
      }
      for (int k = 0; k < 3; k++) S[off::XPOS + 3*i + k] = lp[k];
      for (int k = 0; k < 4; k++) S[off::XQUAT + 4*i + k] = lq[k];
      if (i == 0)
        for (int k = 0; k < 9; k++) S[off::XMAT + k] = (k == 0 || k == 4 || k == 8) ? R(1) : R(0);
    }
    gsync();
    for (int lev = 0; lev < NLEVEL; lev++) {
      for (int idx = level_adr[lev] + l; idx < level_adr[lev + 1]; idx += G) {
        const int i = level_body[idx], pid = body_parentid[i];
        real pm[9], pq[4], lp[3], lq[4], v[3], xquat[4], xm[9];
        for (int k = 0; k < 9; k++) pm[k] = S[off::XMAT + 9*pid + k];
        for (int k = 0; k < 4; k++) pq[k] = S[off::XQUAT + 4*pid + k];
        for (int k = 0; k < 3; k++) lp[k] = S[off::XPOS + 3*i + k];
        for (int k = 0; k < 4; k++) lq[k] = S[off::XQUAT + 4*i + k];
        mulmatvec3(v, pm, lp);
        mulquat(xquat, pq, lq);
        normalize4(xquat);
        quat2mat(xm, xquat);
        for (int k = 0; k < 3; k++) S[off::XPOS + 3*i + k] = S[off::XPOS + 3*pid + k] + v[k];
        for (int k = 0; k < 4; k++) S[off::XQUAT + 4*i + k] = xquat[k];
        for (int k = 0; k < 9; k++) S[off::XMAT + 9*i + k] = xm[k];
      }
      gsync();
    }
    for (int j = l; j < NJNT; j += G) {
      const int pid = body_parentid[jnt_bodyid[j]];
      real pm[9], a[3], x[3], va[3], vx[3];
      for (int k = 0; k < 9; k++) pm[k] = S[off::XMAT + 9*pid + k];
      for (int k = 0; k < 3; k++) { a[k] = S[off::XANCHOR + 3*j + k]; x[k] = S[off::XAXIS + 3*j + k]; }
      mulmatvec3(va, pm, a);
      mulmatvec3(vx, pm, x);
      for (int k = 0; k < 3; k++) {
        S[off::XANCHOR + 3*j + k] = S[off::XPOS + 3*pid + k] + va[k];
        S[off::XAXIS + 3*j + k] = vx[k];
      }
    }
    for (int i = l; i < NBODY; i += G) {
      real xm[9], xq[4], v[3], q[4], im[9];
      const real ip[3] = {R(body_ipos[3*i]), R(body_ipos[3*i + 1]), R(body_ipos[3*i + 2])};
      const real iq[4] = {R(body_iquat[4*i]), R(body_iquat[4*i + 1]),
                          R(body_iquat[4*i + 2]), R(body_iquat[4*i + 3])};
      for (int k = 0; k < 9; k++) xm[k] = S[off::XMAT + 9*i + k];
      for (int k = 0; k < 4; k++) xq[k] = S[off::XQUAT + 4*i + k];
      mulmatvec3(v, xm, ip);
      for (int k = 0; k < 3; k++) S[off::XIPOS + 3*i + k] = S[off::XPOS + 3*i + k] + v[k];
      mulquat(q, xq, iq);
      quat2mat(im, q);
      for (int k = 0; k < 9; k++) S[off::XIMAT + 9*i + k] = i == 0 ? ((k == 0 || k == 4 || k == 8) ? R(1) : R(0)) : im[k];
    }
    gsync();
  }

  __device__ __forceinline__ void com_pos() {
    // subtree centre of mass: bodies are depth-first ordered, a subtree is the
    // index range [i, i + body_subtree_n[i]).  lane = (body, axis); the range
    // loop runs its full static length with the terms past the end masked, so
    // all loads are in flight together (see com_vel)
    for (int idx = l; idx < NBODY*3; idx += G) {
      const int i = idx/3, k = idx - 3*i;
      const int n = body_subtree_n[i];
      real acc = 0;
      _Pragma("unroll")
      for (int jj = 0; jj < NBODY; jj++) {
        const bool in = jj < n;
        const int j = in ? i + jj : i;
        const real mass = in ? R(body_mass[j]) : R(0);
        acc += mass*S[off::XIPOS + 3*j + k];
      }
      if (body_subtreemass[i] < 1e-15) acc = S[off::XIPOS + idx];
      else acc *= R(1)/R(body_subtreemass[i]);
      S[off::SUBCOM + idx] = acc;
    }
    gsync();
    for (int i = l; i < NBODY; i += G) {
      real* res = S + off::CINERT + 10*i;
      if (i == 0) { for (int k = 0; k < 10; k++) res[k] = 0; continue; }
      const real* com = S + off::SUBCOM + 3*body_rootid[i];
      const real* mat = S + off::XIMAT + 9*i;
      real dif[3], t[9], m[9];
      const real mass = R(body_mass[i]);
      const real in0 = R(body_inertia[3*i]), in1 = R(body_inertia[3*i + 1]),
                 in2 = R(body_inertia[3*i + 2]);
      for (int k = 0; k < 9; k++) m[k] = mat[k];
      for (int k = 0; k < 3; k++) dif[k] = S[off::XIPOS + 3*i + k] - com[k];
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
          t[3*a + b] = m[3*a]*in0*m[3*b] + m[3*a + 1]*in1*m[3*b + 1] +
                       m[3*a + 2]*in2*m[3*b + 2];
      res[0] = t[0] + mass*(dif[1]*dif[1] + dif[2]*dif[2]);
      res[1] = t[4] + mass*(dif[0]*dif[0] + dif[2]*dif[2]);
      res[2] = t[8] + mass*(dif[0]*dif[0] + dif[1]*dif[1]);
      res[3] = t[1] - mass*dif[0]*dif[1];
      res[4] = t[2] - mass*dif[0]*dif[2];
      res[5] = t[5] - mass*dif[1]*dif[2];
      res[6] = mass*dif[0]; res[7] = mass*dif[1]; res[8] = mass*dif[2];
      res[9] = mass;
    }
    for (int j = l; j < NJNT; j += G) {
      const int b = jnt_bodyid[j], da = jnt_dofadr[j], jt = jnt_type[j];
      const real* com = S + off::SUBCOM + 3*body_rootid[b];
      real offv[3];
      for (int k = 0; k < 3; k++) offv[k] = com[k] - S[off::XANCHOR + 3*j + k];
      real* cd = S + off::CDOF + 6*da;
      if (jt == JNT_FREE || jt == JNT_BALL) {
        if (jt == JNT_FREE) {
          for (int k = 0; k < 18; k++) cd[k] = 0;
          cd[3] = 1; cd[6 + 4] = 1; cd[12 + 5] = 1;
          cd += 18;
        }
        for (int k = 0; k < 3; k++) {
          real ax[3] = {S[off::XMAT + 9*b + k], S[off::XMAT + 9*b + 3 + k],
                        S[off::XMAT + 9*b + 6 + k]};
          real c[3];
          cross3(c, ax, offv);
          for (int d = 0; d < 3; d++) { cd[6*k + d] = ax[d]; cd[6*k + 3 + d] = c[d]; }
        }
      } else if (jt == JNT_SLIDE) {
        for (int k = 0; k < 3; k++) { cd[k] = 0; cd[3 + k] = S[off::XAXIS + 3*j + k]; }
      } else {
        real ax[3], c[3];
        for (int k = 0; k < 3; k++) ax[k] = S[off::XAXIS + 3*j + k];
        cross3(c, ax, offv);
        for (int k = 0; k < 3; k++) { cd[k] = ax[k]; cd[3 + k] = c[k]; }
      }
    }
    gsync();
  }

  // composite inertias and the full symmetric mass matrix
  __device__ __forceinline__ void crb_matrix() {
    // composite inertia = range sum of cinert over the subtree, lane = (body, word)
    for (int idx = l; idx < NBODY*10; idx += G) {
      const int i = idx/10, k = idx - 10*i;
      const int n = i > 0 ? body_subtree_n[i] : 0;
      real acc = 0;
      _Pragma("unroll")
      for (int jj = 0; jj < NBODY - 1; jj++) {
        const bool in = jj < n;
        const real v = S[off::CINERT + 10*(in ? i + jj : i) + k];
        acc += in ? v : R(0);
      }
      S[off::CRB + idx] = acc;
    }
    gsync();
    // lane = dof: M[i][i] and M[i][ancestors of i]; the entries between
    // unrelated dofs are structural zeros, written once per launch (load())
    for (int i = l; i < NV; i += G) {
      real buf[6], cd[6], crb[10];
      for (int k = 0; k < 6; k++) cd[k] = S[off::CDOF + 6*i + k];
      for (int k = 0; k < 10; k++) crb[k] = S[off::CRB + 10*dof_bodyid[i] + k];
      mul_inert_vec(buf, crb, cd);
      const int na = dof_anc_len[i];
      int anc[MAXCHAIN];
      // (the table index is made opaque: otherwise the compiler hoists these
      // reads and the 3*MAXCHAIN addresses derived from them out of the substep
      // loop and keeps them live -- or spilled -- through the whole step)
      const int row = opaque(i*MAXCHAIN);
      _Pragma("unroll")
      for (int a = 0; a < MAXCHAIN; a++) anc[a] = dof_anc[row + a];
      const real diag = dot6(cd, buf) + R(dof_armature[i]);
      _Pragma("unroll")
      for (int a = 0; a < MAXCHAIN; a++) {
        // branch-free: the slots past the end of the chain rewrite the diagonal
        const int j = a < na ? anc[a] : i;
        real cj[6];
        for (int k = 0; k < 6; k++) cj[k] = S[off::CDOF + 6*j + k];
        const real v = a < na ? dot6(cj, buf) : diag;
        S[off::MM + i*NVP + j] = v;
        S[off::MM + j*NVP + i] = v;
      }
      S[off::MM + i*NVP + i] = diag;
    }
    gsync();
    PROF(PH_CRB);
  }

  // ---- velocity stage --------------------------------------------------------
  // Body velocities, cdof_dot and the bias accelerations.  In the frame the
  // spatial vectors use (world axes, origin at the subtree's centre of mass) a
  // body's velocity is the plain sum of cdof*qvel over the dofs between the
  // root and the body, so no level-by-level walk is needed:
  //   lane = dof:  cdof_dot = velocity of everything upstream of the dof's
  //                joint  x  cdof   (mj_comVel: cvel "before the joint");
  //   lane = body: cvel and cacc as sums over the body's dof chain, added in
  //                root-to-leaf order like the reference's walk.
  __device__ __forceinline__ void com_vel() {
    // The chain loops run their full static length with the terms past the end
    // weighted by zero: all index loads, then all data loads, are independent
    // and in flight together instead of one LDS round trip per ancestor.
    for (int d = l; d < NV; d += G) {
      const int j = dof_jntid[d], jt = jnt_type[j], da = jnt_dofadr[j];
      // dofs of the same ball / free-rotation joint do not count as upstream;
      // the translational dofs of a free joint have cdof_dot = 0
      const int first = jt == JNT_BALL ? da : (jt == JNT_FREE ? da + 3 : d);
      const int na = (jt == JNT_FREE && d < da + 3) ? 0 : dof_anc_len[d];
      int anc[MAXCHAIN];
      const int row = opaque(d*MAXCHAIN);
      _Pragma("unroll")
      for (int a = 0; a < MAXCHAIN; a++) anc[a] = dof_anc[row + a];
      real cvel[6] = {0, 0, 0, 0, 0, 0}, cd[6], dd[6];
      _Pragma("unroll")
      for (int a = MAXCHAIN - 1; a >= 0; a--) {
        const int k = anc[a];
        const real qv = (a < na && k < first) ? S[off::QVEL + k] : R(0);
        _Pragma("unroll")
        for (int c = 0; c < 6; c++) cvel[c] += S[off::CDOF + 6*k + c]*qv;
      }
      for (int c = 0; c < 6; c++) cd[c] = S[off::CDOF + 6*d + c];
      cross_motion(dd, cvel, cd);
      for (int c = 0; c < 6; c++) S[off::CDOFDOT + 6*d + c] = dd[c];
    }
    gsync();
    for (int i = l; i < NBODY; i += G) {
      real cvel[6] = {0, 0, 0, 0, 0, 0}, cacc[6] = {0, 0, 0, 0, 0, 0};
      if (!(DISABLEFLAGS & DSBL_GRAVITY))
        for (int k = 0; k < 3; k++) cacc[3 + k] = -R(gravity[k]);
      const int last = body_lastdof[i];
      const int lastx = last < 0 ? 0 : last;
      const int na = last < 0 ? -1 : dof_anc_len[lastx];
      int anc[MAXCHAIN];
      const int row = opaque(lastx*MAXCHAIN);
      _Pragma("unroll")
      for (int a = 0; a < MAXCHAIN; a++) anc[a] = dof_anc[row + a];
      _Pragma("unroll")
      for (int a = MAXCHAIN; a >= 0; a--) {
        const int k = a == 0 ? lastx : anc[a - 1];
        const real qv = a <= na ? S[off::QVEL + k] : R(0);
        _Pragma("unroll")
        for (int c = 0; c < 6; c++) {
          cvel[c] += S[off::CDOF + 6*k + c]*qv;
          cacc[c] += S[off::CDOFDOT + 6*k + c]*qv;
        }
      }
      for (int k = 0; k < 6; k++) { S[off::CVEL + 6*i + k] = cvel[k]; S[off::CACC + 6*i + k] = cacc[k]; }
    }
    gsync();
  }

  // qfrc_smooth = passive - bias + actuator
  __device__ __forceinline__ void smooth_forces(bool actuation) {
    for (int i = l; i < NBODY; i += G) {
      real f[6];
      if (i == 0) { for (int k = 0; k < 6; k++) f[k] = 0; }
      else {
        real ci[10], v[6], acc[6], tmp[6], tmp1[6];
        for (int k = 0; k < 10; k++) ci[k] = S[off::CINERT + 10*i + k];
        for (int k = 0; k < 6; k++) { v[k] = S[off::CVEL + 6*i + k]; acc[k] = S[off::CACC + 6*i + k]; }
        mul_inert_vec(f, ci, acc);
        mul_inert_vec(tmp, ci, v);
        cross_force(tmp1, v, tmp);
        for (int k = 0; k < 6; k++) f[k] += tmp1[k];
      }
      for (int k = 0; k < 6; k++) S[off::CFRC + 6*i + k] = f[k];
    }
    gsync();
    // force on the subtree of every body, lane = (body, component) (range sum
    // as in crb_matrix; the composite inertias are dead, their words take it)
    for (int idx = l; idx < NBODY*6; idx += G) {
      const int i = idx/6, k = idx - 6*i;
      const int n = body_subtree_n[i];
      real acc = 0;
      _Pragma("unroll")
      for (int jj = 0; jj < NBODY; jj++) {
        const bool in = jj < n;
        const real v = S[off::CFRC + 6*(in ? i + jj : i) + k];
        acc += in ? v : R(0);
      }
      S[off::CRB + idx] = acc;
    }
    gsync();
    // bias force of dof i = cdof_i . (force on the subtree of its body)
    for (int i = l; i < NV; i += G) {
      const int b = dof_bodyid[i];
      real f[6], cd[6];
      for (int k = 0; k < 6; k++) f[k] = S[off::CRB + 6*b + k];
      for (int k = 0; k < 6; k++) cd[k] = S[off::CDOF + 6*i + k];
      real fs = -dot6(cd, f);
      if (!(DISABLEFLAGS & DSBL_PASSIVE)) {
        const int j = dof_jntid[i], jt = jnt_type[j];
        if (jnt_stiffness[j] != 0 && (jt == JNT_SLIDE || jt == JNT_HINGE)) {
          const int qa = jnt_qposadr[j];
          fs -= R(jnt_stiffness[j])*(S[off::QPOS + qa] - R(qpos_spring[qa]));
        }
        fs -= R(dof_damping[i])*S[off::QVEL + i];
      }
      if (actuation && !(DISABLEFLAGS & DSBL_ACTUATION)) {
        const EnvView V = {S + off::QPOS, S + off::QVEL, S + off::CTRL, S + off::XPOS,
                           S + off::XMAT, S + off::XIPOS, S + off::SLV, S + off::TOUCH,
                           S + off::TASKD};
        // every lane evaluates every actuator (unrolled: all model tables fold
        // to literals, the controls are uniform LDS reads) and keeps the ones
        // whose transmission reaches its dof
        _Pragma("unroll")
        for (int u = 0; u < NU; u++) {
          real moment = 0;
          bool mine = false;
          _Pragma("unroll")
          for (int k = 0; k < act_wrap_num[u]; k++) {
            const int w = act_wrap_adr[u] + k;
            if (act_wrap_dof[w] == i) { moment += wrap_coef(V, w); mine = true; }
          }
          const real gear = R(actuator_gear[u]);
          real c = S[off::CTRL + u];
          if (actuator_ctrllimited[u] && !(DISABLEFLAGS & DSBL_CLAMPCTRL))
            c = clampr(c, R(actuator_ctrlrange[2*u]), R(actuator_ctrlrange[2*u + 1]));
          real force = R(actuator_gainprm[3*u])*c;
          if (actuator_biastype[u] == 1) {
            real length = 0, velocity = 0;
            _Pragma("unroll")
            for (int k = 0; k < act_wrap_num[u]; k++) {
              const int w = act_wrap_adr[u] + k;
              const real coef = wrap_coef(V, w);
              length += coef*S[off::QPOS + act_wrap_qadr[w]];
              velocity += coef*S[off::QVEL + act_wrap_dof[w]];
            }
            force += R(actuator_biasprm[3*u]) + R(actuator_biasprm[3*u + 1])*gear*length +
                     R(actuator_biasprm[3*u + 2])*gear*velocity;
          }
          if (actuator_forcelimited[u])
            force = clampr(force, R(actuator_forcerange[2*u]), R(actuator_forcerange[2*u + 1]));
          if (mine) fs += gear*moment*force;
        }
      }
      S[off::FS + i] = fs;
    }
    gsync();
    PROF(PH_SMOOTH);
  }
  // M = L L^T, and qacc_smooth = M^-1 qfrc_smooth
  __device__ __forceinline__ void factor_mass(Rows& L) {
    rows_load(L, off::MM);
    if (rows_chol(L, off::HH)) warn |= WARN_INERTIA;
    PROF(PH_FACM);
  }
  __device__ __forceinline__ void smooth_acc(const Rows& L) {
    real b[RNV];
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) b[t] = l + t*G < NV ? S[off::FS + l + t*G] : R(0);
    rows_solve(L, off::HH, b);
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) if (l + t*G < NV) S[off::QAS + l + t*G] = b[t];
    gsync();
    PROF(PH_SOLVE);
  }

  __device__ __forceinline__ void subtree_vel() {
    // momentum of every body about the world origin frame, then range sums
    for (int i = l; i < NBODY; i += G) {
      real dif[3], t[3], w[3];
      const real* com = S + off::SUBCOM + 3*body_rootid[i];
      for (int k = 0; k < 3; k++) { dif[k] = S[off::XIPOS + 3*i + k] - com[k]; w[k] = S[off::CVEL + 6*i + k]; }
      cross3(t, w, dif);
      for (int k = 0; k < 3; k++)
        S[off::CFRC + 3*i + k] = R(body_mass[i])*(S[off::CVEL + 6*i + 3 + k] + t[k]);
    }
    gsync();
    for (int i = l; i < NBODY; i += G) {
      real acc[3] = {0, 0, 0};
      const int n = body_subtree_n[i];
      for (int j = i; j < i + n; j++)
        for (int k = 0; k < 3; k++) acc[k] += S[off::CFRC + 3*j + k];
      const real inv = R(1.0/(body_subtreemass[i] < 1e-15 ? 1e-15 : body_subtreemass[i]));
      for (int k = 0; k < 3; k++) S[off::SLV + 3*i + k] = acc[k]*inv;
    }
    gsync();
  }

  // ---- constraints -----------------------------------------------------------
  // Rows are created with their metadata parked in the solver slots
  // (CR_D <- R, CR_AREF <- K*imp*(pos - margin), CR_JAR <- B); finish_rows()
  // turns that into D and aref once J is complete.
  __device__ __forceinline__ void row_meta(int r, real pm, real K, real B, real imp, real Rrow) {
    real* row = S + off::ROWS + r*CRW;
    row[CR_D] = Rrow; row[CR_AREF] = K*imp*pm; row[CR_JAR] = B;
  }

  __device__ __forceinline__ void limit_rows() {
    if (DISABLEFLAGS & (DSBL_LIMIT | DSBL_CONSTRAINT)) return;
    for (int base = 0; base < NLIMIT; base += G) {
      const int li = base + l;
      int cnt = 0;
      real dist[2] = {0, 0};
      int j = 0, dof = 0;
      real margin = 0;
      if (li < NLIMIT) {
        j = limit_jnt[li];
        dof = jnt_dofadr[j];
        margin = R(jnt_margin[j]);
        const real q = S[off::QPOS + jnt_qposadr[j]];
        dist[0] = q - R(jnt_range[2*j]);
        dist[1] = R(jnt_range[2*j + 1]) - q;
        cnt = (dist[0] < margin ? 1 : 0) + (dist[1] < margin ? 1 : 0);
      }
      int total;
      int r = nefc + gscan(cnt, l, total);
      for (int side = 0; side < 2; side++) {
        if (li >= NLIMIT || !(dist[side] < margin)) continue;
        if (r >= NEFC_MAX) { warn |= WARN_CNSTRFULL; continue; }
        real* row = S + off::ROWS + r*CRW;
        for (int k = 0; k < NV; k++) row[k] = 0;
        row[dof] = side == 0 ? R(1) : R(-1);
        const real pm = dist[side] - margin;
        const real imp = impedance(limit_solimp + 5*li, pm);
        row_meta(r, pm, R(limit_K[li]), R(limit_B[li]), imp,
                 (1 - imp)*R(dof_invweight0[dof])/imp);
        r++;
      }
      nefc += total;
      if (nefc > NEFC_MAX) nefc = NEFC_MAX;
    }
    warn = (unsigned)gor_bits(warn);
  }
  __device__ __forceinline__ unsigned gor_bits(unsigned w) {
    int x = (int)w;
#ifndef DMC_HOST_SHIM
    if (G >= 16) {
      x |= dpp<DPP_XOR1>(x);
      x |= dpp<DPP_XOR2>(x);
      x |= dpp<DPP_HALF_MIRROR>(x);
      x |= dpp<DPP_MIRROR>(x);
      if (G == 64) x = gbcast_bits(x, 0) | gbcast_bits(x, 16) | gbcast_bits(x, 32) | gbcast_bits(x, 48);
      else for (int m = 16; m < G; m <<= 1) x |= gxor(x, m);
      return (unsigned)x;
    }
#endif
    for (int m = G/2; m > 0; m >>= 1) x |= gxor(x, m);
    return (unsigned)x;
  }

  // narrowphase over the static pair list, strided over the lanes; an ordered
  // prefix sum per stride keeps the contact list in pair order
  // returns the number of constraint rows once the contacts' rows are counted
  __device__ __forceinline__ int detect_contacts() {
    if (DISABLEFLAGS & (DSBL_CONTACT | DSBL_CONSTRAINT)) return nefc;
    for (int g = l; g < NGEOM; g += G) {
      const int b = geom_bodyid[g];
      real gp[3] = {R(geom_pos[3*g]), R(geom_pos[3*g + 1]), R(geom_pos[3*g + 2])};
      real gq[4] = {R(geom_quat[4*g]), R(geom_quat[4*g + 1]), R(geom_quat[4*g + 2]),
                    R(geom_quat[4*g + 3])};
      real v[3], q[4], bq[4], bm[9], gm[9];
      for (int k = 0; k < 9; k++) bm[k] = S[off::XMAT + 9*b + k];
      for (int k = 0; k < 4; k++) bq[k] = S[off::XQUAT + 4*b + k];
      mulmatvec3(v, bm, gp);
      for (int k = 0; k < 3; k++) S[off::GEOM + 12*g + k] = S[off::XPOS + 3*b + k] + v[k];
      mulquat(q, bq, gq);
      normalize4(q);
      quat2mat(gm, q);
      for (int k = 0; k < 9; k++) S[off::GEOM + 12*g + 3 + k] = gm[k];
    }
    gsync();
    int nrow_total = nefc;
    for (int base = 0; base < NPAIR; base += G) {
      const int p = base + l;
      RawCon rc[4] = {};
      int mask = 0;
      if (p < NPAIR) mask = collide_pair(ArrPoses{S + off::GEOM}, p, rc);
      const int cnt = __builtin_popcount((unsigned)mask);
      int total;
      int slot = ncon + gscan(cnt, l, total);
      // rows of the accepted contacts of this lane
      int nrows = 0;
      const int per = p < NPAIR ? pair_nrow[p] : 0;
      const real incl = p < NPAIR ? R(pair_includemargin[p]) : R(0);
      {
        int s2 = slot;
        _Pragma("unroll")
        for (int c = 0; c < 4; c++) {
          if (!((mask >> c) & 1)) continue;
          if (s2 < NCON_MAX && rc[c].dist < incl) nrows += per;
          s2++;
        }
      }
      int rtotal;
      int rbase = nrow_total + gscan(nrows, l, rtotal);
      _Pragma("unroll")
      for (int c = 0; c < 4; c++) {
        if (!((mask >> c) & 1)) continue;
        if (slot >= NCON_MAX) { warn |= WARN_CONTACTFULL; slot++; continue; }
        real* rec = S + off::CON + slot*CCW;
        for (int k = 0; k < 3; k++) {
          rec[k] = rc[c].pos[k]; rec[3 + k] = rc[c].frame[k]; rec[6 + k] = rc[c].frame[3 + k];
        }
        rec[CC_DIST] = rc[c].dist;
        rec[CC_PAIR] = (real)p;
        const bool has_rows = rc[c].dist < incl;
        rec[CC_ROW] = has_rows ? (real)rbase : R(-1);
        if (has_rows) rbase += per;
        slot++;
      }
      ncon += total;
      if (ncon > NCON_MAX) ncon = NCON_MAX;
      nrow_total += rtotal;
    }
    warn = gor_bits(warn);
    gsync();
    return nrow_total;
  }

  // Jacobian columns of all contact rows: lane = (contact, dof), so a few
  // simultaneous contacts cost one pass instead of one pass each.  Every lane
  // derives its contact's frame and row parameters itself (the per-pair model
  // tables are indexed per lane); the lane of dof 0 writes the row metadata.
  __device__ __forceinline__ void contact_rows() {
    const int rows_after = detect_contacts();
    PROF(PH_DETECT);
    for (int idx = l; idx < ncon*NV; idx += G) {
      const int c = idx/NV, j = idx - c*NV;
      const real* rec = S + off::CON + c*CCW;
      const int r0 = (int)rec[CC_ROW];
      if (r0 < 0) continue;
      const int p = (int)rec[CC_PAIR];
      const real dist = rec[CC_DIST];
      real pos[3], fin[6], f[9];
      for (int k = 0; k < 3; k++) { pos[k] = rec[k]; fin[k] = rec[3 + k]; fin[3 + k] = rec[6 + k]; }
      make_frame(fin, f);
      const int b1 = pair_b1[p], b2 = pair_b2[p], dim = pair_dim[p];
      const unsigned m1 = body_dofmask[NMASKW*b1 + (j >> 5)] >> (j & 31);
      const unsigned m2 = body_dofmask[NMASKW*b2 + (j >> 5)] >> (j & 31);
      const bool in1 = (m1 & 1u) != 0, in2 = (m2 & 1u) != 0;
      real off1[3], off2[3];
      for (int k = 0; k < 3; k++) {
        off1[k] = pos[k] - S[off::SUBCOM + 3*body_rootid[b1] + k];
        off2[k] = pos[k] - S[off::SUBCOM + 3*body_rootid[b2] + k];
      }
      const int nrow = pair_nrow[p];
      if (r0 + nrow > NEFC_MAX) warn |= WARN_CNSTRFULL;
      if (j == 0) {   // per-row metadata; rows past the capacity are dropped
        const real pm = dist - R(pair_includemargin[p]);
        const real imp = impedance(pair_solimp + 5*p, pm);
        const real K = R(pair_K[p]), B = R(pair_B[p]);
        const real mu0 = R(pair_friction[5*p]);
        real R0 = (1 - imp)*R(pair_diag[6*p + 1])/imp;
        if (R0 < DMC_MINVAL) R0 = DMC_MINVAL;
        const real Rrow = dim == 1 ? (1 - imp)*R(pair_diag[6*p])/imp : 2*mu0*mu0*R0;
        for (int r = r0; r < r0 + nrow && r < NEFC_MAX; r++) row_meta(r, pm, K, B, imp, Rrow);
      }
      real cd[6], jb[3], jt[3];
      for (int k = 0; k < 6; k++) cd[k] = S[off::CDOF + 6*j + k];
      for (int d = 0; d < 3; d++) {
        const real* dir = f + 3*d;
        real w1[3], w2[3];
        cross3(w1, off1, dir);
        cross3(w2, off2, dir);
        const real dl = dot3(dir, cd + 3);
        const real v2 = dl + dot3(w2, cd), v1 = dl + dot3(w1, cd);
        jb[d] = (in2 ? v2 : R(0)) - (in1 ? v1 : R(0));
        const real vr = dot3(dir, cd);
        jt[d] = (in2 ? vr : R(0)) - (in1 ? vr : R(0));
      }
      real* col = S + off::ROWS + j;
      if (dim == 1) {
        if (r0 < NEFC_MAX) col[r0*CRW] = jb[0];
      } else {
        for (int k = 1; k < dim; k++) {
          const real mu = R(pair_friction[5*p + k - 1]);
          const real t = k < 3 ? jb[k] : jt[k - 3];
          const int r = r0 + 2*(k - 1);
          if (r < NEFC_MAX) col[r*CRW] = jb[0] + mu*t;
          if (r + 1 < NEFC_MAX) col[(r + 1)*CRW] = jb[0] - mu*t;
        }
      }
    }
    nefc = rows_after < NEFC_MAX ? rows_after : NEFC_MAX;
    warn = gor_bits(warn);
    gsync();
  }

  // D and aref of every row (lane = row): aref = -B (J qvel) - K imp (pos - margin)
  __device__ __forceinline__ void finish_rows() {
    for (int r = l; r < nefc; r += G) {
      real* row = S + off::ROWS + r*CRW;
      real vel = 0;
      _Pragma("unroll")
      for (int j = 0; j < NV; j++) vel += row[j]*S[off::QVEL + j];
      const real Rr = row[CR_D], kip = row[CR_AREF], B = row[CR_JAR];
      row[CR_AREF] = -B*vel - kip;
      row[CR_D] = R(1)/(Rr < DMC_MINVAL ? DMC_MINVAL : Rr);
    }
    gsync();
  }

  // ---- Newton solver (same algorithm and stopping rules as solve_newton in
  // dmc_kernels.hip; sums over dofs and rows are group reductions) -----------
  static constexpr int ROUNDS = (NEFC_MAX + G - 1)/G;   // lane rounds needed to cover the rows
  struct Ls { real alpha, dcost, d0, d1; };
  __device__ __forceinline__ void ls_eval(Ls& P, real alpha, real q1, real q2) const {
    real dcost = 0, d0 = 0, d1 = 0;
    for (int r = l; r < nefc; r += G) {
      const real* row = S + off::ROWS + r*CRW;
      const real x0 = row[CR_JAR], v = row[CR_JV], D = row[CR_D];
      const real x = x0 + alpha*v;
      const real a = x < 0 ? x : R(0), a0 = x0 < 0 ? x0 : R(0);
      dcost += R(0.5)*D*(a*a - a0*a0);
      if (x < 0) { d0 += D*x*v; d1 += D*v*v; }
    }
    dcost = gsum(dcost) + alpha*alpha*q2 + alpha*q1;
    d0 = gsum(d0) + 2*alpha*q2 + q1;
    d1 = gsum(d1) + 2*q2;
    P.alpha = alpha; P.dcost = dcost; P.d0 = d0;
    P.d1 = d1 > DMC_MINVAL ? d1 : DMC_MINVAL;
  }

  // As in dmc_kernels.hip: the Hessian M + J^T D_active J is kept across the
  // iterations and only the rows that changed sides are added or removed, and
  // (fp32) the pass that computes Jv also evaluates the line-search point
  // alpha = 1, the exact Newton step, so the usual iteration needs no further
  // line-search pass.
  __device__ __forceinline__ void solve_newton(real tol) {
    const real scale = R(1.0/(meaninertia*(NV > 1 ? NV : 1)));
    for (int i = l; i < NV; i += G) S[off::MA + i] = mrow_dot(i, off::QACC);
    gsync();
    real improvement = 0;
    bool converged = false;
    int iter = 0;
    Rows H;
    rows_load(H, off::MM);
    for (;; iter++) {
      // constraint forces of the active rows (lane = row); the first iteration
      // adds every active row to H.  The rows the dof lanes have to visit --
      // active ones for the force, flipped ones also for H -- are collected as
      // bit masks, so that pass touches no inactive row.
      unsigned long long visit[ROUNDS], flipped[ROUNDS];
      _Pragma("unroll")
      for (int t = 0; t < ROUNDS; t++) {
        const int r = l + t*G;
        bool act = false, flp = false;
        if (r < nefc) {
          real* row = S + off::ROWS + r*CRW;
          const real jar = row[CR_JAR];
          act = jar < 0;
          row[CR_F] = act ? -row[CR_D]*jar : R(0);
          if (iter == 0) row[CR_FLIP] = act ? R(1) : R(0);
          flp = iter == 0 ? act : row[CR_FLIP] != 0;
        }
        flipped[t] = gballot(flp);
        visit[t] = gballot(act) | flipped[t];
      }
      gsync();
      // lane = dof: qfrc_constraint, gradient, changes of row i of H
      real gn = 0;
      real grad[RNV];
      {
        real fc[RNV];
        _Pragma("unroll")
        for (int t = 0; t < RNV; t++) fc[t] = 0;
        _Pragma("unroll")
        for (int tr = 0; tr < ROUNDS; tr++) {
          // rows that only contribute their force: four at a time, so that the
          // LDS reads of a group are in flight together
          unsigned long long m = visit[tr] & ~flipped[tr];
          while (m) {
            real f4[4], j4[4][RNV];
            _Pragma("unroll")
            for (int u = 0; u < 4; u++) {
              const bool on = m != 0;
              const int b = on ? first_bit(m) : 0;
              m &= m - (on ? 1ull : 0ull);
              const real* row = S + off::ROWS + (b + tr*G)*CRW;
              f4[u] = on ? row[CR_F] : R(0);
              _Pragma("unroll")
              for (int t = 0; t < RNV; t++) {
                const int i = l + t*G;
                j4[u][t] = i < NV ? row[i] : R(0);
              }
            }
            _Pragma("unroll")
            for (int u = 0; u < 4; u++) {
              _Pragma("unroll")
              for (int t = 0; t < RNV; t++) fc[t] += j4[u][t]*f4[u];
            }
          }
          // rows that enter or leave the active set: force and rank-1 change of H
          m = flipped[tr];
          while (m) {
            const int b = first_bit(m);
            m &= m - 1;
            const real* row = S + off::ROWS + (b + tr*G)*CRW;
            const real f = row[CR_F];
            const real D = row[CR_FLIP]*row[CR_D];
            real jr[NVX];
            _Pragma("unroll")
            for (int k = 0; k < NV; k++) jr[k] = row[k];
            _Pragma("unroll")
            for (int t = 0; t < RNV; t++) {
              const int i = l + t*G;
              const real ji = i < NV ? row[i] : R(0);
              fc[t] += ji*f;
              const real s = D*ji;
              _Pragma("unroll")
              for (int k = 0; k < NV; k++) H.a[t][k] += s*jr[k];
            }
          }
        }
        _Pragma("unroll")
        for (int t = 0; t < RNV; t++) {
          const int i = l + t*G;
          grad[t] = 0;
          if (i < NV) {
            S[off::FC + i] = fc[t];
            grad[t] = S[off::MA + i] - S[off::FS + i] - fc[t];
            gn += grad[t]*grad[t];
          }
        }
      }
      gn = gsum(gn);
      gsync();
      PROF(PH_HESS);
      if (iter > 0 && (converged || scale*improvement < tol || scale*sqrt(gn) < tol)) break;
      if (iter >= ITERATIONS) break;
      Rows F = H;                       // the factorisation works in place
      rows_chol(F, off::HH);
      PROF(PH_FACH);
      _Pragma("unroll")
      for (int t = 0; t < RNV; t++) grad[t] = -grad[t];
      rows_solve(F, off::HH, grad);
      _Pragma("unroll")
      for (int t = 0; t < RNV; t++) if (l + t*G < NV) S[off::SEARCH + l + t*G] = grad[t];
      gsync();
      PROF(PH_SOLVE);
      real sn = 0, q1 = 0, q2 = 0;
      for (int i = l; i < NV; i += G) {
        const real si = S[off::SEARCH + i];
        const real mv = mrow_dot(i, off::SEARCH);
        S[off::MV + i] = mv;
        sn += si*si;
        q1 += si*(S[off::MA + i] - S[off::FS + i]);
        q2 += R(0.5)*si*mv;
      }
      sn = sqrt(gsum(sn)); q1 = gsum(q1); q2 = gsum(q2);
      if (sn < DMC_MINVAL) break;
      const real gtol = tol*R(0.01)*sn/scale;
      // Jv, the line-search derivatives at alpha = 0 and (fp32) the point alpha = 1
      Ls p0, p, best;
      {
        real d0 = 0, d1 = 0, c1 = 0, e0 = 0, e1 = 0;
        for (int r = l; r < nefc; r += G) {
          real* row = S + off::ROWS + r*CRW;
          real sacc = 0;
          _Pragma("unroll")
          for (int j = 0; j < NV; j++) sacc += row[j]*S[off::SEARCH + j];
          row[CR_JV] = sacc;
          const real x0 = row[CR_JAR], Dv = row[CR_D]*sacc;
          if (x0 < 0) { d0 += Dv*x0; d1 += Dv*sacc; }
          if (DMC_F32_RULES) {
            const real x = x0 + sacc;
            const real xa = x < 0 ? x : R(0), xa0 = x0 < 0 ? x0 : R(0);
            c1 += R(0.5)*row[CR_D]*(xa*xa - xa0*xa0);
            if (x < 0) { e0 += Dv*x; e1 += Dv*sacc; }
          }
        }
        d0 = gsum(d0) + q1; d1 = gsum(d1) + 2*q2;
        p0.alpha = 0; p0.dcost = 0; p0.d0 = d0;
        p0.d1 = d1 > DMC_MINVAL ? d1 : DMC_MINVAL;
        if (DMC_F32_RULES) {
          c1 = gsum(c1) + q2 + q1; e0 = gsum(e0) + 2*q2 + q1; e1 = gsum(e1) + 2*q2;
          p.alpha = 1; p.dcost = c1; p.d0 = e0;
          p.d1 = e1 > DMC_MINVAL ? e1 : DMC_MINVAL;
        }
      }
      gsync();
      // exact line search: safeguarded Newton on the directional derivative
      if (!(p0.d0 < 0)) break;
      best = p0;
      real lo = 0, hi = 0, a = DMC_F32_RULES ? R(1) : -p0.d0/p0.d1;
      bool have_hi = false;
      const real dtol = DMC_F32_RULES ? fmax(gtol, R(1e-5)*fabs(p0.d0)) : gtol;
      for (int it = 0; it < DMC_LS_MAXIT; it++) {
        if (!(DMC_F32_RULES && it == 0)) ls_eval(p, a, q1, q2);
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
      PROF(PH_LS);
      if (alpha == 0) break;
      improvement = -best.dcost;
      for (int i = l; i < NV; i += G) {
        S[off::QACC + i] += alpha*S[off::SEARCH + i];
        S[off::MA + i] += alpha*S[off::MV + i];
      }
      bool changed = false;
      for (int r = l; r < nefc; r += G) {
        real* row = S + off::ROWS + r*CRW;
        const real x0 = row[CR_JAR];
        const real x1 = x0 + alpha*row[CR_JV];
        const real flip = R((x1 < 0) - (x0 < 0));
        changed |= flip != 0;
        row[CR_JAR] = x1;
        row[CR_FLIP] = flip;
      }
      changed = gany(changed);
      gsync();
      if (DMC_F32_RULES && !changed && fabs(alpha - 1) < R(1e-3)) converged = true;
      PROF(PH_UPD);
    }
    iters = iter;
  }

  // touch sensors (see touch_sensors in dmc_kernels.hip): lane = contact; the
  // row forces CR_F are those of the final iterate
  __device__ __forceinline__ void touch_sensors() {
    real acc[NTOUCH > 0 ? NTOUCH : 1];
    _Pragma("unroll")
    for (int s = 0; s < NTOUCH; s++) acc[s] = 0;
    const EnvView V = {S + off::QPOS, S + off::QVEL, S + off::CTRL, S + off::XPOS,
                       S + off::XMAT, S + off::XIPOS, S + off::SLV, S + off::TOUCH,
                         S + off::TASKD};
    if (nefc > 0) {
      for (int c = l; c < ncon; c += G) {
        const real* rec = S + off::CON + c*CCW;
        const int r0 = (int)rec[CC_ROW];
        if (r0 < 0) continue;
        const int p = (int)rec[CC_PAIR];
        const int nrow = pair_nrow[p];
        real fn = 0;
        for (int j = 0; j < nrow; j++)
          if (r0 + j < nefc) fn += S[off::ROWS + (r0 + j)*CRW + CR_F];
        if (!(fn > 0)) continue;
        real pos[3], normal[3];
        for (int k = 0; k < 3; k++) { pos[k] = rec[k]; normal[k] = rec[3 + k]; }
        normalize3(normal);
        _Pragma("unroll")
        for (int s = 0; s < NTOUCH; s++)
          acc[s] += fn*touch_hit(V, s, pos, normal, pair_b1[p], pair_b2[p]);
      }
    }
    _Pragma("unroll")
    for (int s = 0; s < NTOUCH; s++) {
      const real total = gsum(acc[s]);
      if (l == 0) S[off::TOUCH + s] = total;
    }
    gsync();
  }

  // limits, collision detection, contact Jacobians, reference accelerations
  __device__ __forceinline__ void constraint_rows() {
    ncon = 0; nefc = 0;
    limit_rows();
    PROF(PH_LIMIT);
    if (NPAIR > 0) contact_rows();
    PROF(PH_CROWS);
    finish_rows();
    PROF(PH_FINISH);
  }
  // wave 1's share of one forward() (DUO): the barriers pair with forward()'s
  __device__ __forceinline__ void forward_rows() {
    l = opaque(l);
    wsync();
    com_vel();        // ends with a phase boundary: the results are in LDS
    epoch++;
    if (l == 0) flag_write(off::XCH + 3, epoch);
    warn = 0;
    constraint_rows();
    if (l == 0) { S[off::XCH] = (real)nefc; S[off::XCH + 1] = (real)ncon; S[off::XCH + 2] = (real)(int)warn; }
    wsync();
  }
  // wave 1's share of one physics_step() (DUO)
  __device__ __forceinline__ void step_rows() {
    forward_rows();
    if (EULER_OFFLOAD) {
      Rows A;
      damped_factor(A, off::GEOM);
      wsync();        // qfrc_smooth + qfrc_constraint are final, wave 0 is done with the scratch
      damped_solve(A);
      wsync();
    }
  }

  // forward dynamics at (qpos, qvel, ctrl): qacc, qfrc_smooth, qfrc_constraint
  __device__ __forceinline__ void forward(bool actuation, real tol) {
    PROF(PH_EULER);
    kinematics();
    PROF(PH_KIN);
    com_pos();
    PROF(PH_COM);
    wsync();          // frames, cdof and subtree coms are final: wave 1 starts on the rows
    crb_matrix();
    Rows L;
    if (!DUO) {
      com_vel();
      PROF(PH_VEL);
      smooth_forces(actuation);
      factor_mass(L);
    } else {
      // wave 1 does the velocity stage first and flags it; by the time M is
      // factorised here the flag is up
      factor_mass(L);
      epoch++;
      while (flag_read(off::XCH + 3) != epoch) flag_pause();
      PROF(PH_VEL);
      smooth_forces(actuation);
    }
    smooth_acc(L);
    iters = 0;
    if (!DUO) {
      constraint_rows();
    } else {
      wsync();        // rows, contacts and their counts are in LDS
      nefc = (int)S[off::XCH]; ncon = (int)S[off::XCH + 1];
      warn |= (unsigned)(int)S[off::XCH + 2];
      PROF(PH_FINISH);
    }
    if (nefc == 0) {
      for (int i = l; i < NV; i += G) { S[off::QACC + i] = S[off::QAS + i]; S[off::FC + i] = 0; }
      gsync();
    } else {
      // warmstart: the better of the previous qacc and the unconstrained one
      const bool try_warm = !(DISABLEFLAGS & DSBL_WARMSTART);
      real cw = 0, cs = 0;
      if (try_warm) {
        for (int i = l; i < NV; i += G)
          cw += R(0.5)*(mrow_dot(i, off::WARM) - S[off::FS + i])*
                (S[off::WARM + i] - S[off::QAS + i]);
      }
      for (int r = l; r < nefc; r += G) {
        real* row = S + off::ROWS + r*CRW;
        real jw = 0, js = 0;
        _Pragma("unroll")
        for (int j = 0; j < NV; j++) {
          const real v = row[j];
          jw += v*S[off::WARM + j]; js += v*S[off::QAS + j];
        }
        const real aref = row[CR_AREF], D = row[CR_D];
        jw -= aref; js -= aref;
        if (jw < 0) cw += R(0.5)*D*jw*jw;
        if (js < 0) cs += R(0.5)*D*js*js;
        row[CR_JAR] = jw; row[CR_JV] = js;
      }
      cw = gsum(cw); cs = gsum(cs);
      const bool use_warm = try_warm && !(cw > cs);
      for (int i = l; i < NV; i += G)
        S[off::QACC + i] = use_warm ? S[off::WARM + i] : S[off::QAS + i];
      if (!use_warm)
        for (int r = l; r < nefc; r += G) {
          real* row = S + off::ROWS + r*CRW;
          row[CR_JAR] = row[CR_JV];
        }
      gsync();
      PROF(PH_WARM);
      solve_newton(tol);
    }
    if (NTOUCH > 0) touch_sensors();
    for (int i = l; i < NV; i += G) S[off::WARM + i] = S[off::QACC + i];
    gsync();
  }

  // qpos += h * qvel on the configuration manifold (lane = joint); qvel at `v`
  __device__ __forceinline__ void integrate_pos(int v, real h) {
    for (int j = l; j < NJNT; j += G) {
      const int qa = jnt_qposadr[j], da = jnt_dofadr[j], jt = jnt_type[j];
      if (jt == JNT_FREE || jt == JNT_BALL) {
        int q = qa, d = da;
        if (jt == JNT_FREE) {
          for (int k = 0; k < 3; k++) S[off::QPOS + qa + k] += h*S[v + da + k];
          q += 3; d += 3;
        }
        real quat[4], w[3];
        for (int k = 0; k < 4; k++) quat[k] = S[off::QPOS + q + k];
        for (int k = 0; k < 3; k++) w[k] = S[v + d + k];
        quat_integrate(quat, w, h);
        for (int k = 0; k < 4; k++) S[off::QPOS + q + k] = quat[k];
      } else {
        S[off::QPOS + qa] += h*S[v + da];
      }
    }
    gsync();
  }

  __device__ __forceinline__ void reset_state() {   // mj_resetData
    for (int i = l; i < NQ; i += G) S[off::QPOS + i] = R(qpos0[i]);
    for (int i = l; i < NV; i += G) { S[off::QVEL + i] = 0; S[off::WARM + i] = 0; }
    for (int i = l; i < NU; i += G) S[off::CTRL + i] = 0;
    time = 0;
    gsync();
  }
  __device__ __forceinline__ bool check_state() {   // mj_checkPos / mj_checkVel
    bool bp = false, bv = false;
    for (int i = l; i < NQ; i += G) bp |= bad(S[off::QPOS + i]);
    for (int i = l; i < NV; i += G) bv |= bad(S[off::QVEL + i]);
    bp = gany(bp); bv = gany(bv);
    if (bp) { warn |= WARN_BADQPOS; reset_state(); }
    else if (bv) { warn |= WARN_BADQVEL; reset_state(); }
    return bp || bv;
  }
  __device__ __forceinline__ bool bad_qacc() {
    bool ba = false;
    for (int i = l; i < NV; i += G) ba |= bad(S[off::QACC + i]);
    return gany(ba);
  }

  // Euler's implicit joint damping: Cholesky factor of M + h D, and
  // GRAD <- (M + h D)^-1 (qfrc_smooth + qfrc_constraint)
  static constexpr bool damped() { return coop_damped(); }
  __device__ __forceinline__ void damped_factor(Rows& A, int panel) {
    const real h = R(timestep);
    rows_load(A, off::MM);
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) {
      const int i = l + t*G;
      _Pragma("unroll")
      for (int k = 0; k < NV; k++)
        if (i == k) A.a[t][k] += h*R(dof_damping[k]);
    }
    rows_chol(A, panel);
  }
  __device__ __forceinline__ void damped_solve(const Rows& A) {
    real rhs[RNV];
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) {
      const int i = l + t*G;
      rhs[t] = i < NV ? S[off::FS + i] + S[off::FC + i] : R(0);
    }
    rows_solve(A, off::HH, rhs);
    _Pragma("unroll")
    for (int t = 0; t < RNV; t++) if (l + t*G < NV) S[off::GRAD + l + t*G] = rhs[t];
    gsync();
  }

  // one `Physics.step()`; `stale`: acceleration from the position/velocity
  // stage of the reset state, applied to the current state (the first of the
  // cheetah's settle steps, see physics_step in dmc_kernels.hip)
  __device__ __forceinline__ void physics_step(real tol, bool stale = false) {
    // nothing derived from the lane index is carried across steps: left alone,
    // the compiler precomputes a few dozen per-lane LDS addresses before the
    // substep loop and then spills them
    l = opaque(l);
    const real h = R(timestep);
    check_state();
    if (!RK4) {
      constexpr int KQ = (NQX + G - 1)/G, KV = (NVX + G - 1)/G;
      real qkeep[KQ], vkeep[KV];
      if (stale) {
        _Pragma("unroll")
        for (int t = 0; t < KQ; t++) {
          const int i = l + t*G;
          if (i < NQ) { qkeep[t] = S[off::QPOS + i]; S[off::QPOS + i] = R(qpos0[i]); }
        }
        _Pragma("unroll")
        for (int t = 0; t < KV; t++) {
          const int i = l + t*G;
          if (i < NV) { vkeep[t] = S[off::QVEL + i]; S[off::QVEL + i] = 0; }
        }
        gsync();
      }
      forward(true, tol);
      if (stale) {
        _Pragma("unroll")
        for (int t = 0; t < KQ; t++) if (l + t*G < NQ) S[off::QPOS + l + t*G] = qkeep[t];
        _Pragma("unroll")
        for (int t = 0; t < KV; t++) if (l + t*G < NV) S[off::QVEL + l + t*G] = vkeep[t];
        gsync();
      }
      if (EULER_OFFLOAD) {   // wave 1 solves (M + h D) a = f with the factor it made meanwhile
        wsync();
        wsync();
      }
      if (bad_qacc()) { warn |= WARN_BADQACC; reset_state(); return; }
      int src = off::QACC;
      if (damped()) {   // implicit in the joint damping: (M + h D) a = f
        if (!EULER_OFFLOAD) {
          Rows A;
          damped_factor(A, off::HH);
          damped_solve(A);
        }
        src = off::GRAD;
      }
      for (int i = l; i < NV; i += G) S[off::QVEL + i] += h*S[src + i];
      gsync();
      integrate_pos(off::QVEL, h);
      time += h;
    } else {
      const real t0 = time;
      for (int i = l; i < NQ; i += G) S[off::Q0 + i] = S[off::QPOS + i];
      for (int i = l; i < NV; i += G) S[off::V0 + i] = S[off::QVEL + i];
      gsync();
      forward(true, tol);
      if (bad_qacc()) { warn |= WARN_BADQACC; reset_state(); return; }
      for (int i = l; i < NV; i += G) {
        S[off::FV + i] = S[off::QVEL + i]; S[off::FA + i] = S[off::QACC + i];
      }
      gsync();
      for (int s = 1; s < 4; s++) {
        const real a = s == 3 ? R(1) : R(0.5);
        for (int i = l; i < NV; i += G) {
          S[off::DV + i] = a*S[off::FV + (s - 1)*NV + i];
          S[off::QVEL + i] = S[off::V0 + i] + h*a*S[off::FA + (s - 1)*NV + i];
        }
        for (int i = l; i < NQ; i += G) S[off::QPOS + i] = S[off::Q0 + i];
        gsync();
        integrate_pos(off::DV, h);
        forward(true, tol);
        for (int i = l; i < NV; i += G) {
          S[off::FV + s*NV + i] = S[off::QVEL + i]; S[off::FA + s*NV + i] = S[off::QACC + i];
        }
        gsync();
      }
      for (int i = l; i < NV; i += G) {
        const real* Fv = S + off::FV + i;
        const real* Fa = S + off::FA + i;
        S[off::DV + i] = (Fv[0] + 2*Fv[NV] + 2*Fv[2*NV] + Fv[3*NV])*R(1.0/6.0);
        const real acc = (Fa[0] + 2*Fa[NV] + 2*Fa[2*NV] + Fa[3*NV])*R(1.0/6.0);
        S[off::QVEL + i] = S[off::V0 + i] + h*acc;
      }
      for (int i = l; i < NQ; i += G) S[off::QPOS + i] = S[off::Q0 + i];
      gsync();
      integrate_pos(off::DV, h);
      time = t0 + h;
    }
  }

  __device__ __forceinline__ void observe_stage() {
    check_state();
    kinematics();
    com_pos();
    com_vel();
    subtree_vel();
  }

  // ---- HBM I/O ---------------------------------------------------------------
  __device__ __forceinline__ void load(const DmcArgs& a, int e) {
    const long long n = a.nenv;
    for (int i = l; i < NQ; i += G) S[off::QPOS + i] = a.qpos[sidx(i, e, n, NQX)];
    for (int i = l; i < NV; i += G) {
      S[off::QVEL + i] = a.qvel[sidx(i, e, n, NVX)]; S[off::WARM + i] = a.warm[sidx(i, e, n, NVX)];
    }
    for (int k = l; k < NV*NVP; k += G) S[off::MM + k] = 0;   // structural zeros stay
    if (l == 0) flag_write(off::XCH + 3, 0);
    time = a.time[e];
    warn = 0; ncon = 0; nefc = 0; iters = 0; epoch = 0;
    if (l < (NTOUCH > 0 ? NTOUCH : 1)) S[off::TOUCH + l] = 0;
    for (int i = l; i < NTASKDATA; i += G) S[off::TASKD + i] = a.taskdata[sidx(i, e, n, NTDX)];
  }
  __device__ __forceinline__ void store(const DmcArgs& a, int e) {
    const long long n = a.nenv;
    for (int i = l; i < NQ; i += G) a.qpos[sidx(i, e, n, NQX)] = S[off::QPOS + i];
    for (int i = l; i < NV; i += G) {
      a.qvel[sidx(i, e, n, NVX)] = S[off::QVEL + i]; a.warm[sidx(i, e, n, NVX)] = S[off::WARM + i];
    }
    if (l == 0) {
      a.time[e] = time;
      if (warn) a.warn[e] |= warn;
    }
  }
  __device__ __forceinline__ void outputs(const DmcArgs& a, int e, bool accumulate) {
    const long long n = a.nenv;
    if (l == 0) {
      const EnvView V = {S + off::QPOS, S + off::QVEL, S + off::CTRL, S + off::XPOS,
                         S + off::XMAT, S + off::XIPOS, S + off::SLV, S + off::TOUCH,
                         S + off::TASKD};
      const real rew = task_outputs(V, a, S + off::OBSV);
      a.reward[e] = rew;
      if (accumulate) a.episode_return[e] += rew;
      a.stats[sidx(0, e, n, 3)] = ncon; a.stats[sidx(1, e, n, 3)] = nefc;
      a.stats[sidx(2, e, n, 3)] = iters;
    }
    gsync();
    for (int k = l; k < NOBS; k += G)
      a.obs[(long long)k*a.obs_sk + (long long)e*a.obs_se] = S[off::OBSV + k];
    for (int s = 0; s < NSENSOR; s++) {
      const int adr = sensor_adr[s], o = sensor_objid[s], ty = sensor_type[s];
      if (ty == 35 || ty == 34) {
        const int src = ty == 35 ? off::SLV : off::SUBCOM;
        if (l < 3) a.sensordata[sidx(adr + l, e, n, NSDX)] = S[src + 3*o + l];
      } else if (l == 0) {
        if (ty == 8) a.sensordata[sidx(adr, e, n, NSDX)] = S[off::QPOS + jnt_qposadr[o]];
        else if (ty == 9) a.sensordata[sidx(adr, e, n, NSDX)] = S[off::QVEL + jnt_dofadr[o]];
      }
    }
    if (l < NTOUCH) a.sensordata[sidx(touch_adr[l], e, n, NSDX)] = S[off::TOUCH + l];
    if (a.xpos) for (int i = l; i < NBODY*3; i += G) a.xpos[sidx(i, e, n, NBODY*3)] = S[off::XPOS + i];
    if (a.xmat) for (int i = l; i < NBODY*9; i += G) a.xmat[sidx(i, e, n, NBODY*9)] = S[off::XMAT + i];
  }
};

__shared__ __attribute__((aligned(16))) real coop_lds[ENV_WORDS*EPB];

// Workgroups are handed to the 8 XCDs round-robin (workgroup b runs on XCD
// b % 8) and every XCD has its own L2.  With env = workgroup, the 100-byte
// state rows and the 4-byte scalars of neighbouring envs -- which share
// cache lines -- would be read and written through eight different L2s, each
// moving the whole line.  This permutation gives every XCD one contiguous range
// of envs instead (a bijection for any number of workgroups).
DEV int xcd_contiguous(int b, int nb) {
  const int q = nb/8, r = nb % 8, x = b % 8;
  return x*q + (x < r ? x : r) + b/8;
}

// nsub x Physics.step, then observation + reward of the new state
// DUO: two waves of every env, four envs per CU by LDS => two waves per SIMD,
// so a wave may use half of the SIMD's 512 registers
#if DMC_COOP_DUO && DMC_GROUP == 64
#define DMC_COOP_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2)))
#else
#define DMC_COOP_OCCUPANCY
#endif
extern "C" __global__ void __launch_bounds__(NTHREADS) DMC_COOP_OCCUPANCY
dmc_step(DmcArgs a) {
#ifdef DMC_COOP_PROFILE
  const long long tstart_ = wall_clock64();
#endif
  stage_tables();
  const int slot = DUO ? 0 : threadIdx.x/G;
  const int e = xcd_contiguous(blockIdx.x, (a.nenv + EPB - 1)/EPB)*EPB + slot;
  if (e >= a.nenv) return;               // whole groups (DUO: both waves) leave together
  Coop C;
  C.S = coop_lds + slot*ENV_WORDS;
  C.l = threadIdx.x % G;
  const int l = C.l;
  real* S = C.S;
  const long long n = a.nenv;
#ifdef DMC_COOP_PROFILE
  for (int k = 0; k < PH_N; k++) C.tprof[k] = 0;
  C.tlast = wall_clock64();
#endif
#if !defined(DMC_HOST_SHIM) && !defined(DMC_COOP_NO_PRIO)
  // the two waves of an env (or of neighbouring envs) share a SIMD: the wave
  // on the critical path issues first, the row builder fills the gaps
  if (DUO) { if (threadIdx.x < 64) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
#endif
  if (DUO && threadIdx.x >= 64) {        // the row-building wave: one forward() per step
    C.ncon = 0; C.nefc = 0; C.warn = 0; C.iters = 0; C.time = 0; C.epoch = 0;
    for (int s = 0; s < a.nsub; s++) C.step_rows();
    return;
  }
  C.load(a, e);
  if (a.flags & 1) {
    bool bc = false;
    for (int i = l; i < NU; i += G) {
      const real c = a.ctrl[i*a.ctrl_sk + (long long)e*a.ctrl_se];
      S[off::CTRL + i] = c;
      bc |= bad(c);
    }
    if (gany(bc)) {   // mj_fwdActuation's ctrl check: warn and zero the controls
      C.warn |= WARN_BADCTRL;
      for (int i = l; i < NU; i += G) S[off::CTRL + i] = 0;
    }
    for (int i = l; i < NU; i += G) a.ctrl_store[sidx(i, e, n, NUX)] = S[off::CTRL + i];
  } else {
    for (int i = l; i < NU; i += G) S[off::CTRL + i] = a.ctrl_store[sidx(i, e, n, NUX)];
  }
  gsync();
  const real tol = R(tolerance_opt > DMC_TOL_FLOOR ? tolerance_opt : DMC_TOL_FLOOR);
  for (int s = 0; s < a.nsub; s++)
    C.physics_step(tol, s == 0 && (a.flags & DMC_FLAG_STALE_FIRST));
#ifdef DMC_COOP_PROFILE
  { const long long t_ = wall_clock64(); C.tprof[PH_EULER] += t_ - C.tlast; C.tlast = t_; }
#endif
  if (a.qacc) for (int i = l; i < NV; i += G) a.qacc[sidx(i, e, n, NVX)] = S[off::QACC + i];
  if (!(a.flags & 2)) {
    C.observe_stage();
    C.outputs(a, e, true);
  }
#ifdef DMC_COOP_PROFILE
  { const long long t_ = wall_clock64(); C.tprof[PH_OBS] += t_ - C.tlast; }
  if (l == 0) for (int k = 0; k < PH_N && k < NOBS; k++) a.obs[(long long)e*a.obs_se + k] = (real)C.tprof[k];
  if (l == 0 && PH_N + 4 <= NOBS) {   // wave placement and life time: start, end (10 ns ticks, 20 bits), HW_ID, XCC_ID
    real* o = a.obs + (long long)e*a.obs_se + PH_N;
    o[0] = (real)(int)(tstart_ & 0xFFFFF); o[1] = (real)(int)(wall_clock64() & 0xFFFFF);
    o[2] = (real)(int)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xFFFF);
    o[3] = (real)(int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF);
  }
#endif
  C.store(a, e);
}

// observation / reward / sensors of the current state (reset, after_reset)
extern "C" __global__ void __launch_bounds__(NTHREADS) DMC_COOP_OCCUPANCY
dmc_observe(DmcArgs a) {
  stage_tables();
  const int slot = DUO ? 0 : threadIdx.x/G;
  const int e = xcd_contiguous(blockIdx.x, (a.nenv + EPB - 1)/EPB)*EPB + slot;
  if (e >= a.nenv) return;
  Coop C;
  C.S = coop_lds + slot*ENV_WORDS;
  C.l = threadIdx.x % G;
  const int l = C.l;
  real* S = C.S;
  const long long n = a.nenv;
#ifdef DMC_COOP_PROFILE
  for (int k = 0; k < PH_N; k++) C.tprof[k] = 0;
  C.tlast = 0;
#endif
  if (DUO && threadIdx.x >= 64) {
    C.ncon = 0; C.nefc = 0; C.warn = 0; C.iters = 0; C.time = 0; C.epoch = 0;
    if (NTOUCH > 0) C.forward_rows();
    return;
  }
  C.load(a, e);
  for (int i = l; i < NU; i += G) S[off::CTRL + i] = a.ctrl_store[sidx(i, e, n, NUX)];
  gsync();
  if (NTOUCH > 0) {
    // acceleration-stage sensors need the constraint forces: the reference's
    // after_reset runs mj_forward with actuation disabled (engine.py:283-295);
    // a bad state is reset first (mj_checkPos), as mj_forward would see it
    const real tol = R(tolerance_opt > DMC_TOL_FLOOR ? tolerance_opt : DMC_TOL_FLOOR);
    C.check_state();
    C.forward(false, tol);
  }
  const int ncon_forward = C.ncon, nefc_forward = C.nefc;
  C.observe_stage();
  if (NTOUCH > 0) {
    C.ncon = ncon_forward; C.nefc = nefc_forward;
  } else if (a.flags & 4) {   // count contacts (humanoid reset rejection test)
    C.ncon = 0; C.nefc = 0;
    if (NPAIR > 0) C.detect_contacts();
  }
  C.outputs(a, e, false);
  C.store(a, e);
}

extern "C" __device__ const int dmc_info[20] = {
    1 /*abi*/, (int)sizeof(real), NQ, NV, NU, NBODY, NOBS, NSENSORDATA,
    1 /*workspace reals per env: none, everything is in LDS*/, TASK, NCON_MAX, NEFC_MAX,
    INTEGRATOR, NPAIR, EPB /*envs per 64-lane workgroup*/,
    DMC_ENV_MAJOR /*state fields are [env][k]*/, NTASKDATA, NTHREADS /*threads per workgroup*/,
    0, 0};
