// Runs dmc_step of csrc/dmc_coop.hip for one workgroup (64/G envs, one thread
// per lane) on the host under sanitizers and prints qpos/qvel of every env
// after each step (compared with the oracle by tests/test_kernel_sanitizers.py).
#include "shim_coop.h"
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
  const int nsub = argc > 2 ? atoi(argv[2]) : 1;
  const int n = EPB;
  const int nq = NQ > 0 ? NQ : 1, nv = NV > 0 ? NV : 1, nu = NU > 0 ? NU : 1;
  std::vector<real> qpos(nq*n), qvel(nv*n, 0), warm(nv*n, 0), tm(n, 0), ctrl(nu*n, 0),
      obs((NOBS > 0 ? NOBS : 1)*n), rew(n), ret(n, 0),
      sens((NSENSORDATA > 0 ? NSENSORDATA : 1)*n), xpos(NBODY*3*n), xmat(NBODY*9*n),
      qacc(nv*n), ws(n);
  std::vector<unsigned> warn(n, 0);
  std::vector<int> stats(3*n, 0);
  // initial states from argv: per env qpos then qvel
  int at = 3;
  // several-lanes-per-env code objects keep the state env-major: [env][k]
  for (int e = 0; e < n; e++) {
    for (int i = 0; i < NQ; i++)
      qpos[e*nq + i] = at < argc ? (real)atof(argv[at++]) : (real)qpos0[i];
    for (int i = 0; i < NV; i++)
      qvel[e*nv + i] = at < argc ? (real)atof(argv[at++]) : (real)0;
  }
  DmcArgs& a = g_args;
  memset(&a, 0, sizeof a);
  a.nenv = n; a.nsub = nsub; a.flags = 0;
  a.qpos = qpos.data(); a.qvel = qvel.data(); a.warm = warm.data(); a.time = tm.data();
  a.ctrl_store = ctrl.data(); a.obs = obs.data(); a.obs_sk = 1; a.obs_se = NOBS;
  a.reward = rew.data(); a.episode_return = ret.data(); a.sensordata = sens.data();
  a.xpos = xpos.data(); a.xmat = xmat.data(); a.qacc = qacc.data();
  a.warn = warn.data(); a.stats = stats.data(); a.ws = ws.data();
  for (int t = 0; t < NTHREADS/G; t++) pthread_barrier_init(&shim_teams[t].bar, nullptr, G);
  pthread_barrier_init(&shim_block_barrier, nullptr, NTHREADS);
  for (int t = 0; t < steps; t++) {
    pthread_t th[NTHREADS];
    for (size_t i = 0; i < NTHREADS; i++) pthread_create(&th[i], nullptr, lane_main, (void*)i);
    for (int i = 0; i < NTHREADS; i++) pthread_join(th[i], nullptr);
    for (int e = 0; e < n; e++) {
      printf("STEP %d %d", t, e);
      for (int i = 0; i < NQ; i++) printf(" %.17g", (double)qpos[e*nq + i]);
      for (int i = 0; i < NV; i++) printf(" %.17g", (double)qvel[e*nv + i]);
      printf(" | %d %d %d %u\n", stats[3*e], stats[3*e + 1], stats[3*e + 2], warn[e]);
    }
  }
  return 0;
}
