// dmc_args.h -- kernel argument block shared by the host runtime
// (dmc_api.cpp) and the device code (dmc_kernels.hip).  Plain pointers and
// sizes only; `real`-typed arrays are void* on the host side because the
// element type (float/double) is a property of the loaded code object.
#pragma once
#ifndef DMC_REALPTR
#define DMC_REALPTR void*
#define DMC_CREALPTR const void*
#endif
struct DmcArgs {
  int nenv, nsub, flags, task_param_i;
  unsigned long long seed;
  DMC_REALPTR qpos;        // [NQ][nenv]
  DMC_REALPTR qvel;        // [NV][nenv]
  DMC_REALPTR warm;        // [NV][nenv] qacc_warmstart
  DMC_REALPTR time;        // [nenv]
  DMC_CREALPTR ctrl;       // element (k, env) at ctrl[k*ctrl_sk + env*ctrl_se]
  long long ctrl_sk, ctrl_se;
  DMC_REALPTR ctrl_store;  // [NU][nenv] last applied control (data.ctrl)
  DMC_REALPTR obs;         // element (k, env) at obs[k*obs_sk + env*obs_se]
  long long obs_sk, obs_se;
  DMC_REALPTR reward;      // [nenv]
  DMC_REALPTR episode_return;  // [nenv] sum of rewards since the last reset
  DMC_REALPTR sensordata;  // [NSENSORDATA][nenv]
  DMC_REALPTR xpos;        // [NBODY*3][nenv] (may be null)
  DMC_REALPTR xmat;        // [NBODY*9][nenv] (may be null)
  DMC_REALPTR qacc;        // [NV][nenv] (may be null)
  unsigned* warn;          // [nenv] sticky mjtWarning bit mask
  int* stats;              // [3][nenv]: ncon, nefc, solver iterations
  DMC_REALPTR ws;          // workspace, ws_per_env reals per env, [idx][nenv]
  DMC_REALPTR taskdata;    // [NTASKDATA][nenv] per-instance task parameters
  double task_param_r[4];
};
// dmc_step flags
#define DMC_FLAG_CTRL 1          // ctrl pointer valid (else reuse ctrl_store)
#define DMC_FLAG_NO_OUTPUT 2     // skip observation/reward (settle steps)
#define DMC_FLAG_COUNT_CONTACTS 4
#define DMC_FLAG_ONLY_COLLIDING 8
#define DMC_FLAG_RESET_ONLY 16     // dmc_init_episode: mj_resetData only
#define DMC_FLAG_TASKDATA_DEFAULT 32  // dmc_init_episode: task data <- model values
#define DMC_FLAG_STALE_FIRST 64    // dmc_step: first substep takes its acceleration from the reset state
