// LDS float-atomic cost on gfx950, one wave per workgroup (design input for the
// row-parallel Newton solver: per-env reduction of 54 products per constraint
// row through ds_add_f32 / ds_add_f64 with c lanes hitting the same address).
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_bench lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <class T, int NACC>
__global__ void __launch_bounds__(64) bench(long long* out, int c, int reps, T* sink) {
  __shared__ T acc[NACC*64];
  const int lane = threadIdx.x;
  for (int k = 0; k < NACC; k++) acc[k*64 + lane] = 0;
  __syncthreads();
  const int slot = lane/c;              // c consecutive lanes share an address
  T v = (T)(lane + 1)*(T)1e-3;
  long long t0 = wall_clock64();
  long long c0 = clock64();
  for (int r = 0; r < reps; r++) {
#pragma unroll
    for (int k = 0; k < NACC; k++)
      __hip_atomic_fetch_add(&acc[k*64 + slot], v, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  }
  long long c1 = clock64();
  long long t1 = wall_clock64();
  __syncthreads();
  if (lane == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = t1 - t0; }
  sink[blockIdx.x*64 + lane] = acc[lane];
}

template <class T>
void run(const char* name) {
  long long* out; T* sink;
  hipMalloc(&out, 16); hipMalloc(&sink, 256*64*sizeof(T));
  const int reps = 2000;
  for (int c = 1; c <= 64; c *= 2) {
    long long h[2];
    hipLaunchKernelGGL((bench<T, 54>), dim3(1), dim3(64), 0, 0, out, c, reps, sink);
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("%s c=%2d: %.1f shader cycles, %.1f ns per 54-atomic group (%.2f cyc/atomic)\n",
           name, c, (double)h[0]/reps, (double)h[1]*10.0/reps, (double)h[0]/reps/54);
  }
}

int main() {
  run<float>("ds_add_f32");
  run<double>("ds_add_f64");
  return 0;
}
