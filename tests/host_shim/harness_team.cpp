// Runs dmc_step of csrc/dmc_kernels.hip in team mode (-DDMC_TEAM=T: T lanes
// advance one env together) for ONE env on the host, one OS thread per lane
// (shim_coop.h: pthread barriers for tsync and the lane exchanges), under
// sanitizers, and prints qpos/qvel after each step (compared with the oracle
// by tests/test_kernel_sanitizers.py).  TEST INFRASTRUCTURE ONLY.
#define DMC_GROUP DMC_TEAM
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
  const int nq = NQ > 0 ? NQ : 1, nv = NV > 0 ? NV : 1, nu = NU > 0 ? NU : 1;
  std::vector<real> qpos(nq), qvel(nv, 0), warm(nv, 0), tm(1, 0), ctrl(nu, 0),
      obs(NOBS > 0 ? NOBS : 1), rew(1), ret(1), sens(NSENSORDATA > 0 ? NSENSORDATA : 1),
      xpos(NBODY*3), xmat(NBODY*9), qacc(nv), ws((WS_WORDS > 0 ? WS_WORDS : 1));
  std::vector<unsigned> warn(1, 0);
  std::vector<int> stats(3, 0);
  for (int i = 0; i < NQ; i++) qpos[i] = (real)qpos0[i];
  for (int i = 0; i < NQ && 2 + i < argc; i++) qpos[i] = (real)atof(argv[2 + i]);
  for (int i = 0; i < NV && 2 + NQ + i < argc; i++) qvel[i] = (real)atof(argv[2 + NQ + i]);
  DmcArgs& a = g_args;
  memset(&a, 0, sizeof a);
  a.nenv = 1; a.nsub = 1; a.flags = 0;
  a.qpos = qpos.data(); a.qvel = qvel.data(); a.warm = warm.data(); a.time = tm.data();
  a.ctrl_store = ctrl.data(); a.obs = obs.data(); a.obs_sk = 1; a.obs_se = NOBS;
  a.reward = rew.data(); a.episode_return = ret.data(); a.sensordata = sens.data();
  a.xpos = xpos.data(); a.xmat = xmat.data(); a.qacc = qacc.data();
  a.warn = warn.data(); a.stats = stats.data(); a.ws = ws.data();
  blockDim.x = TEAM;
  pthread_barrier_init(&shim_teams[0].bar, nullptr, TEAM);
  pthread_barrier_init(&shim_block_barrier, nullptr, TEAM);
  pthread_attr_t attr;
  pthread_attr_init(&attr);
  pthread_attr_setstacksize(&attr, 256u << 20);     // the per-lane arrays of a big scene
  for (int t = 0; t < steps; t++) {
    pthread_t th[TEAM];
    for (size_t i = 0; i < TEAM; i++) pthread_create(&th[i], &attr, lane_main, (void*)i);
    for (int i = 0; i < TEAM; i++) pthread_join(th[i], nullptr);
    printf("STEP %d", t);
    for (int i = 0; i < NQ; i++) printf(" %.17g", (double)qpos[i]);
    for (int i = 0; i < NV; i++) printf(" %.17g", (double)qvel[i]);
    printf(" | %d %d %d %u\n", stats[0], stats[1], stats[2], warn[0]);
  }
  return 0;
}
