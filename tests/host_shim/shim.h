// Host shim: lets tests compile csrc/dmc_kernels.hip as plain C++ (one lane,
// one workgroup) under AddressSanitizer/UBSan.  TEST INFRASTRUCTURE ONLY -- it
// exists to run sanitizers over the kernel logic (GPU ASan is unavailable);
// nothing in dm_control_amd/ can reach it.
#pragma once
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
static Dim3 threadIdx, blockIdx;
static Dim3 blockDim{64, 1, 1};
static inline void __syncthreads() {}
using std::sqrt; using std::fabs; using std::pow; using std::exp; using std::log;
using std::cos; using std::sin; using std::fmax; using std::fmin; using std::log1p;
// glibc already declares sincos/sincosf with the signatures the kernel uses
