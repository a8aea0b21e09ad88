// Host shim for csrc/dmc_coop.hip: runs the several-lanes-per-env kernel as
// plain C++ with ONE OS THREAD PER LANE.  A group's phase hand-over (gsync) is
// a pthread barrier and the group shuffles go through an exchange buffer, so
// ThreadSanitizer sees every LDS word that one lane writes and another reads
// without a phase boundary in between, and AddressSanitizer sees every index.
// TEST INFRASTRUCTURE ONLY -- nothing in dm_control_amd/ can reach it.
#pragma once
#include <pthread.h>
#include <sched.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#define DMC_HOST_SHIM 1
#define __device__
#define __global__
#define __forceinline__ inline
#define __noinline__
#define __shared__ static
#define __launch_bounds__(...)
#ifndef DMC_GROUP
#define DMC_GROUP 32
#endif
struct Dim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local Dim3 threadIdx;
static thread_local Dim3 blockIdx;
static Dim3 blockDim{64, 1, 1};
using std::sqrt; using std::fabs; using std::pow; using std::exp; using std::log;
using std::cos; using std::sin; using std::fmax; using std::fmin; using std::log1p;

struct ShimTeam {
  pthread_barrier_t bar;
  alignas(8) unsigned char buf[DMC_GROUP][8];
};
static ShimTeam shim_teams[128/DMC_GROUP];   // 64 lanes, or two wavefronts of one env
static pthread_barrier_t shim_block_barrier;   // all lanes of the workgroup
static inline void __syncthreads() { pthread_barrier_wait(&shim_block_barrier); }
static inline ShimTeam& shim_team() { return shim_teams[threadIdx.x/DMC_GROUP]; }
static inline int shim_lane() { return (int)(threadIdx.x % DMC_GROUP); }
static inline void gsync() { pthread_barrier_wait(&shim_team().bar); }
template <class T>
static inline T shim_xchg(T x, int src) {
  static_assert(sizeof(T) <= 8, "exchange slot");
  ShimTeam& t = shim_team();
  memcpy(t.buf[shim_lane()], &x, sizeof x);
  pthread_barrier_wait(&t.bar);
  T r;
  memcpy(&r, t.buf[src], sizeof r);
  pthread_barrier_wait(&t.bar);
  return r;
}
template <class T> static inline T gxor(T x, int m) { return shim_xchg(x, shim_lane() ^ m); }
template <class T> static inline T gup(T x, int d) {
  const int src = shim_lane() - d;
  return shim_xchg(x, src < 0 ? shim_lane() : src);
}
template <class T> static inline T gget(T x, int src) { return shim_xchg(x, src); }
template <class T> static inline T gbcast(T x, int src) { return shim_xchg(x, src); }
