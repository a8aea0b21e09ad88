// Host shim for the several-wavefronts-per-workgroup build of
// csrc/dmc_kernels.hip (-DDMC_WAVES=N): runs ONE workgroup as plain C++ with
// one OS thread per lane (64 x N threads).  `__syncthreads` is a pthread barrier
// over all of them, the wave-level helpers (wany, wsync) one per wavefront, so
// a barrier that one wavefront skips deadlocks (the test times out),
// ThreadSanitizer sees every LDS word that one lane writes and another reads
// without a barrier in between, and AddressSanitizer sees every index.
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
#ifndef DMC_WAVES
#error "build with -DDMC_WAVES=N"
#endif
struct Dim3 { unsigned x = 0, y = 0, z = 0; };
static thread_local Dim3 threadIdx;
static thread_local Dim3 blockIdx;
static Dim3 blockDim{64*DMC_WAVES, 1, 1};
using std::sqrt; using std::fabs; using std::pow; using std::exp; using std::log;
using std::cos; using std::sin; using std::fmax; using std::fmin; using std::log1p;

static pthread_barrier_t shim_block_bar;            // all lanes of the workgroup
static pthread_barrier_t shim_wave_bar[DMC_WAVES];  // the 64 lanes of one wavefront
static int shim_vote[DMC_WAVES][64];
static inline void __syncthreads() { pthread_barrier_wait(&shim_block_bar); }
static inline void wsync() { pthread_barrier_wait(&shim_wave_bar[threadIdx.x/64]); }
static inline bool wany(bool p) {
  const int w = threadIdx.x/64;
  shim_vote[w][threadIdx.x % 64] = p;
  pthread_barrier_wait(&shim_wave_bar[w]);
  bool r = false;
  for (int i = 0; i < 64; i++) r |= shim_vote[w][i] != 0;
  pthread_barrier_wait(&shim_wave_bar[w]);
  return r;
}
