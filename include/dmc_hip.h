/* dmc_hip.h -- C ABI of the MI355X batched physics step (libdmc_hip.so).
 *
 * Drop-in boundary for the path the reference reaches through the ctypes
 * handle `mjlib` (/root/reference/dm_control/mujoco/wrapper/util.py:107-120).
 * Every entry point is extern "C", takes plain pointers and sizes, returns an
 * int status (0 = ok) and leaves a message for dmc_last_error() on failure.
 * Per-environment physics faults never abort: they are reported through the
 * sticky `DMC_FIELD_WARN` bit mask whose bit order is mjtWarning
 * (engine.py:307-330 turns new warnings into PhysicsError).
 *
 * Batched counterparts (B = nenv independent instances, struct-of-arrays
 * [k][env] in device memory unless a stride is given):
 *
 *   reference call (file:line)                          this ABI
 *   --------------------------------------------------  ----------------------
 *   mj_version            wrapper/core.py:65            dmc_version
 *   mj_loadXML            wrapper/core.py:312-328       dmc_model_compile /
 *                                                       dmc_model_load (*)
 *   mj_deleteModel        wrapper/core.py:326           dmc_model_free
 *   mj_makeData           wrapper/core.py:646           dmc_batch_create
 *   mj_deleteData         wrapper/core.py:649           dmc_batch_free
 *   mj_resetData          engine.py:280                 dmc_batch_reset
 *   np.copyto(qpos/qvel)  suite tasks' initialize_episode dmc_batch_set_state,
 *                                                       dmc_batch_init_episode
 *   mj_forward            engine.py:305 (after_reset)   dmc_batch_forward
 *   mj_step2/mj_step +    engine.py:162-166             dmc_batch_step
 *     mj_step1
 *   mjData field views    wrapper/core.py:630-776       dmc_batch_read,
 *                                                       dmc_batch_device_ptr
 *   mj_copyData           engine.py:262                 dmc_batch_copy_state
 *
 * (*) MJCF parsing/compilation is host logic in Python
 *     (dm_control_amd/mjcf/compiler.py); what crosses the ABI is the model as a
 *     table of constants (dm_control_amd/codegen.py) -- either already built
 *     into a gfx950 code object (dmc_model_load) or as text that
 *     dmc_model_compile turns into one in-process, without the hipcc toolchain.
 */
#ifndef DMC_HIP_H_
#define DMC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmc_model dmc_model;
typedef struct dmc_batch dmc_batch;

/* fields addressable through dmc_batch_read / dmc_batch_device_ptr */
enum dmc_field {
  DMC_FIELD_QPOS = 0,       /* real [nq][nenv] */
  DMC_FIELD_QVEL = 1,       /* real [nv][nenv] */
  DMC_FIELD_WARMSTART = 2,  /* real [nv][nenv]  (qacc_warmstart) */
  DMC_FIELD_TIME = 3,       /* real [nenv] */
  DMC_FIELD_CTRL = 4,       /* real [nu][nenv]  (last applied data.ctrl) */
  DMC_FIELD_OBS = 5,        /* real [nenv][nobs] (row-major, agent layout) */
  DMC_FIELD_REWARD = 6,     /* real [nenv] */
  DMC_FIELD_SENSORDATA = 7, /* real [nsensordata][nenv] */
  DMC_FIELD_XPOS = 8,       /* real [nbody*3][nenv] */
  DMC_FIELD_XMAT = 9,       /* real [nbody*9][nenv] */
  DMC_FIELD_QACC = 10,      /* real [nv][nenv] */
  DMC_FIELD_WARN = 11,      /* uint32 [nenv] sticky mjtWarning bit mask */
  DMC_FIELD_STATS = 12,     /* int32 [3][nenv]: ncon, nefc, solver iterations */
  DMC_FIELD_RETURN = 13,    /* real [nenv] sum of rewards since the last reset */
  DMC_FIELD_TASKDATA = 14,  /* real [ntaskdata][nenv] per-instance task parameters (e.g.
                               the reacher's target position, which the reference
                               writes into model.geom_pos per episode) */
  DMC_FIELD_COUNT = 15
};

enum dmc_warn_bit {
  DMC_WARN_INERTIA = 1, DMC_WARN_CONTACTFULL = 2, DMC_WARN_CNSTRFULL = 4,
  DMC_WARN_VGEOMFULL = 8, DMC_WARN_BADQPOS = 16, DMC_WARN_BADQVEL = 32,
  DMC_WARN_BADQACC = 64, DMC_WARN_BADCTRL = 128
};

/* sizes of a loaded model (read back from the code object) */
typedef struct dmc_model_info {
  int abi, real_size, nq, nv, nu, nbody, nobs, nsensordata, ws_per_env, task,
      ncon_max, nefc_max, integrator, npair,
      ntaskdata,     /* per-instance task parameters (DMC_FIELD_TASKDATA rows) */
      envs_per_block, /* envs per workgroup of the step kernel */
      lanes_per_env, /* 1, or the group size of a several-lanes-per-env build
                        (8..64; 128 = one env per workgroup of two wavefronts) */
      env_major;     /* 1: the 2-D state fields are [nenv][k] in HBM (what
                        dmc_batch_device_ptr returns); dmc_batch_read and
                        dmc_batch_set_state present [k][nenv] regardless */
} dmc_model_info;

int dmc_version(void);
const char* dmc_last_error(void);
int dmc_device_count(void);

/* model = gfx950 code object specialised for one compiled MJCF */
int dmc_model_load(const char* code_object_path, int device_id, dmc_model** out);
/* The compile half of mj_loadXML (wrapper/core.py:312-328) at the C boundary,
 * for a model that was not pre-built: `source` (the text of csrc/dmc_kernels.hip
 * or csrc/dmc_coop.hip) is compiled for gfx950 against `nheaders` in-memory
 * headers -- the generated model constants and the sources' own includes, each
 * under the name it is #included by -- with the given compiler options, through
 * the HIP runtime-compilation library of the ROCm runtime (libhiprtc, opened on
 * first use): no hipcc executable, no files.  The code object is returned in a
 * buffer owned by the library (dmc_code_free); load it with dmc_model_load_data
 * and/or store it.  On failure the compiler log is copied into `log` (like the
 * `char error[1000]` of mj_loadXML). */
int dmc_model_compile(const char* source, const char* source_name,
                      const char* const* header_names,
                      const char* const* header_texts, int nheaders,
                      const char* const* options, int noptions,
                      void** code, size_t* code_size, char* log, size_t log_size);
void dmc_code_free(void* code);
int dmc_model_load_data(const void* code, size_t code_size, int device_id,
                        dmc_model** out);
int dmc_model_get_info(const dmc_model* model, dmc_model_info* info);
void dmc_model_free(dmc_model* model);

/* batch of nenv instances resident in the HBM of the model's device */
int dmc_batch_create(const dmc_model* model, int nenv, dmc_batch** out);
void dmc_batch_free(dmc_batch* batch);
int dmc_batch_nenv(const dmc_batch* batch);

/* auxiliary per-step outputs (DMC_FIELD_XPOS, _XMAT, _QACC).  Off by default:
 * they are 105 extra words per env-step for cheetah, three times the
 * algorithmic output; switch them on when the caller reads those fields. */
int dmc_batch_set_aux_outputs(dmc_batch* batch, int enabled);

/* task parameters: integer flags + up to 4 reals (meaning is per task, e.g.
 * cartpole {bit0 sparse, bit1 swing_up}; humanoid r[0] = move_speed) */
int dmc_batch_set_task_params(dmc_batch* batch, int iparam, const double* rparam,
                              int nr);

/* mj_resetData for every env: qpos <- qpos0, qvel/ctrl/warmstart/time <- 0,
 * warning mask cleared */
int dmc_batch_reset(dmc_batch* batch);
/* explicit state upload, host pointers, `real`-typed [k][nenv]; NULL = keep */
int dmc_batch_set_state(dmc_batch* batch, const void* qpos, const void* qvel,
                        const void* warmstart, const void* time);
/* generic host -> device write of a writable field (QPOS, QVEL, WARMSTART,
 * TIME, TASKDATA), `real`-typed [k][nenv]; replaces the reference's in-place
 * writes through numpy views on mjData / mjModel (wrapper/core.py:630-776) */
int dmc_batch_write(dmc_batch* batch, int field, const void* src, size_t bytes);
/* task.initialize_episode on device (counter-based RNG keyed by seed, env).
 * only_colliding != 0 redraws only envs whose last contact count was > 0. */
int dmc_batch_init_episode(dmc_batch* batch, uint64_t seed, int only_colliding);
/* position/velocity stage + observation/reward/sensors of the current state
 * (what the task reads after reset_context; count_contacts != 0 also runs the
 * narrowphase so DMC_FIELD_STATS[0] = ncon) */
int dmc_batch_forward(dmc_batch* batch, int count_contacts);

/* bits of the `want_outputs` argument of dmc_batch_step / dmc_batch_step_n */
#define DMC_STEP_OUTPUTS 1       /* compute observation/reward after the last substep
                                  * (0: settle steps, task.initialize_episode) */
#define DMC_STEP_STALE_FIRST 2   /* the FIRST substep takes its acceleration from the
                                  * position/velocity stage of the reset state (qpos0,
                                  * zero velocity): what mj_step2 sees when a task
                                  * rewrites qpos after reset_context's mj_forward and
                                  * steps without a forward pass
                                  * (suite/cheetah.py:63-77, engine.py:149-166) */

/* nsub x Physics.step, then observation + reward.
 * ctrl: element (k, env) at ctrl[k*stride_k + env*stride_env] (in reals);
 * host pointer when on_device == 0 (copied), device pointer otherwise
 * (zero-copy, e.g. a torch tensor).  ctrl == NULL keeps the previous control.
 * want_outputs: DMC_STEP_* bits (0 skips observation/reward). */
int dmc_batch_step(dmc_batch* batch, const void* ctrl, long long stride_k,
                   long long stride_env, int on_device, int nsub,
                   int want_outputs);
/* `nsteps` control steps back to back on the batch's stream, step t reading
 * its controls at ctrl + t*stride_t reals (device memory): the n_sub_steps loop
 * of control.Environment.step (rl/control.py:101-102) over a pre-computed action
 * sequence (open-loop rollouts, benchmarks) without a host round trip per step;
 * the observation / reward fields hold those of the last step */
int dmc_batch_step_n(dmc_batch* batch, const void* ctrl, long long stride_k,
                     long long stride_env, long long stride_t, int nsteps,
                     int nsub, int want_outputs);

/* device -> host copy of a whole field (synchronises the batch stream) */
int dmc_batch_read(dmc_batch* batch, int field, void* dst, size_t bytes);
size_t dmc_batch_field_bytes(const dmc_batch* batch, int field);
/* borrowed device pointer; lifetime = the batch handle */
void* dmc_batch_device_ptr(dmc_batch* batch, int field);
int dmc_batch_clear_warnings(dmc_batch* batch);
int dmc_batch_copy_state(dmc_batch* dst, const dmc_batch* src);
int dmc_batch_sync(dmc_batch* batch);
/* run every later launch/copy of this batch on a caller-owned hipStream_t (for
 * example torch.cuda.current_stream().cuda_stream, so the step is ordered with
 * the caller's own kernels without extra synchronisation).  external != 0:
 * use `stream` as given (NULL is HIP's legacy default stream, which is what
 * torch's default stream is); external == 0: back to the batch's own stream */
int dmc_batch_set_stream(dmc_batch* batch, void* stream, int external);
/* the batch's hipStream_t, for callers that enqueue their own work behind the
 * step (e.g. torch.cuda.ExternalStream) */
void* dmc_batch_stream(dmc_batch* batch);

/* timing of the step kernel on the batch's own stream (HIP events around the
 * launches issued since dmc_batch_timer_start); returns accumulated device
 * milliseconds and the number of launches */
int dmc_batch_timer_start(dmc_batch* batch);
int dmc_batch_timer_stop(dmc_batch* batch, double* ms, long long* launches);

#ifdef __cplusplus
}
#endif
#endif  /* DMC_HIP_H_ */
