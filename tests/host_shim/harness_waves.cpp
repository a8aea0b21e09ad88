// Runs dmc_step of csrc/dmc_kernels.hip built with -DDMC_WAVES=N for one
// workgroup (nenv <= 64 envs; lanes beyond nenv are the surplus lanes of a
// partial last workgroup) on the host under sanitizers, one thread per lane of
// every wavefront, and prints qpos/qvel of every env after each step (compared
// with the oracle by tests/test_kernel_sanitizers.py).
//   harness_waves <steps> <nenv> [qpos qvel of env 0] [qpos qvel of env 1] ...
#include "shim_waves.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include DMC_KERNEL_SOURCE

static DmcArgs g_args;
static void* lane_main(void* arg) {
  threadIdx.x = (unsigned)(size_t)arg;
  blockIdx.x = 0;
  dmc_step(g_args);
  return nullptr;
}

int main(int argc, char** argv) {
  const int steps = argc > 1 ? atoi(argv[1]) : 5;
  const int n = argc > 2 ? atoi(argv[2]) : 1;
  const int nthreads = 64*NW;
  const int nq = NQ > 0 ? NQ : 1, nv = NV > 0 ? NV : 1, nu = NU > 0 ? NU : 1;
  std::vector<real> qpos(nq*n), qvel(nv*n, 0), warm(nv*n, 0), tm(n, 0), ctrl(nu*n, 0),
      obs((NOBS > 0 ? NOBS : 1)*n), rew(n), ret(n, 0),
      sens((NSENSORDATA > 0 ? NSENSORDATA : 1)*n), xpos(NBODY*3*n), xmat(NBODY*9*n),
      qacc(nv*n), ws((WS_WORDS > 0 ? WS_WORDS : 1)*64, 0);   // padded to a workgroup
  std::vector<unsigned> warn(n, 0);
  std::vector<int> stats(3*n, 0);
  int at = 3;
  for (int e = 0; e < n; e++) {   // state fields are [k][env]
    for (int i = 0; i < NQ; i++)
      qpos[i*n + e] = at < argc ? (real)atof(argv[at++]) : (real)qpos0[i];
    for (int i = 0; i < NV; i++)
      qvel[i*n + e] = at < argc ? (real)atof(argv[at++]) : (real)0;
  }
  DmcArgs& a = g_args;
  memset(&a, 0, sizeof a);
  a.nenv = n; a.nsub = 1; a.flags = 0;
  a.qpos = qpos.data(); a.qvel = qvel.data(); a.warm = warm.data(); a.time = tm.data();
  a.ctrl_store = ctrl.data(); a.obs = obs.data(); a.obs_sk = 1; a.obs_se = NOBS;
  a.reward = rew.data(); a.episode_return = ret.data(); a.sensordata = sens.data();
  a.xpos = xpos.data(); a.xmat = xmat.data(); a.qacc = qacc.data();
  a.warn = warn.data(); a.stats = stats.data(); a.ws = ws.data();
  pthread_barrier_init(&shim_block_bar, nullptr, nthreads);
  for (int w = 0; w < NW; w++) pthread_barrier_init(&shim_wave_bar[w], nullptr, 64);
  printf("CFG waves=%d lds_rows=%d glb_rows=%d lds_cons=%d\n", NW, LDS_ROWS, GLB_ROWS,
         LDS_CONS);
  for (int t = 0; t < steps; t++) {
    std::vector<pthread_t> th(nthreads);
    for (size_t i = 0; i < (size_t)nthreads; i++)
      pthread_create(&th[i], nullptr, lane_main, (void*)i);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], nullptr);
    for (int e = 0; e < n; e++) {
      printf("STEP %d %d", t, e);
      for (int i = 0; i < NQ; i++) printf(" %.17g", (double)qpos[i*n + e]);
      for (int i = 0; i < NV; i++) printf(" %.17g", (double)qvel[i*n + e]);
      printf(" | %d %d %d %u\n", stats[e], stats[n + e], stats[2*n + e], warn[e]);
    }
  }
  return 0;
}
