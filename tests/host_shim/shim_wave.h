// Host shim for the row-parallel solver of csrc/dmc_kernels.hip (ROWPAR): runs
// one workgroup of the one-env-per-lane kernel as plain C++ with ONE OS THREAD
// PER LANE.  The wave hand-over `wsync` is a pthread barrier, ballots and
// shuffles go through an exchange buffer, LDS float atomics take a mutex, so
// ThreadSanitizer sees every LDS word that one lane writes and another reads
// without a hand-over in between, and AddressSanitizer sees every index.
// TEST INFRASTRUCTURE ONLY -- nothing in dm_control_amd/ can reach it.
#pragma once
#include <pthread.h>
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
struct Dim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local Dim3 threadIdx;
static thread_local Dim3 blockIdx;
static Dim3 blockDim{64, 1, 1};
using std::sqrt; using std::fabs; using std::pow; using std::exp; using std::log;
using std::cos; using std::sin; using std::fmax; using std::fmin; using std::log1p;

static pthread_barrier_t shim_bar;          // all 64 lanes
static pthread_mutex_t shim_atomic = PTHREAD_MUTEX_INITIALIZER;
static long long shim_buf[64];
static inline void __syncthreads() { pthread_barrier_wait(&shim_bar); }
static inline void wsync() { pthread_barrier_wait(&shim_bar); }
static inline long long shim_xchg(long long x, int src) {
  shim_buf[threadIdx.x] = x;
  pthread_barrier_wait(&shim_bar);
  const long long r = shim_buf[src];
  pthread_barrier_wait(&shim_bar);
  return r;
}
static inline bool wany(bool p) {
  shim_buf[threadIdx.x] = p;
  pthread_barrier_wait(&shim_bar);
  bool r = false;
  for (int i = 0; i < 64; i++) r |= shim_buf[i] != 0;
  pthread_barrier_wait(&shim_bar);
  return r;
}
static inline int wshfl_up(int v, int d) {
  const int src = (int)threadIdx.x - d;
  return (int)shim_xchg(v, src < 0 ? (int)threadIdx.x : src);
}
static inline int wbcast(int v, int src) { return (int)shim_xchg(v, src); }
template <class T>
static inline void lds_add(T* p, T v) {
  pthread_mutex_lock(&shim_atomic);
  *p += v;
  pthread_mutex_unlock(&shim_atomic);
}
