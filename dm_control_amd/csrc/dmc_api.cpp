// dmc_api.cpp -- host runtime behind the C ABI of include/dmc_hip.h.
//
// Owns: the HIP module of a model-specialised code object, the HBM-resident
// struct-of-arrays state of a batch, one HIP stream per batch, and the launch
// of the step / observe / init kernels.  No torch types, no CPU fallback: if
// HIP or the code object is unavailable every call fails with a message.
//
// Reference counterparts: wrapper/core.py (MjModel :444-627, MjData :630-776,
// error convention :85-101,312-328) and engine.py:149-166,268-305.

#include "../../include/dmc_hip.h"

#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "dmc_args.h"

namespace {

thread_local std::string g_error;

int fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_error = buf;
  return -1;
}

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t err_ = (expr);                                                \
    if (err_ != hipSuccess) {                                                \
      (void)hipGetLastError(); /* do not leak sticky state to other users */ \
      return fail("%s failed: %s", #expr, hipGetErrorString(err_));          \
    }                                                                        \
  } while (0)

}  // namespace

struct dmc_model {
  int device = 0;
  hipModule_t module = nullptr;
  hipFunction_t k_step = nullptr, k_observe = nullptr, k_init = nullptr;
  dmc_model_info info{};
};

struct dmc_batch {
  const dmc_model* model = nullptr;
  int nenv = 0;
  hipStream_t stream = nullptr;       // stream in use (own or caller's)
  hipStream_t own_stream = nullptr;   // created with the batch
  void* field[DMC_FIELD_COUNT] = {};
  size_t bytes[DMC_FIELD_COUNT] = {};
  size_t rows[DMC_FIELD_COUNT] = {};   // k extent of a [k][nenv] field (1: per-env scalar)
  size_t elem[DMC_FIELD_COUNT] = {};   // bytes per element
  void* ws = nullptr;
  void* ctrl_staging = nullptr;   // device copy of host-provided controls
  size_t ctrl_staging_bytes = 0;
  int task_param_i = 0;
  bool aux_outputs = false;
  double task_param_r[4] = {0, 0, 0, 0};
  // timing
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timing = false;
  long long launches = 0;
};

namespace {

int launch(dmc_batch* b, hipFunction_t fn, DmcArgs& args, int group = -1) {
  size_t size = sizeof(DmcArgs);
  void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args,
                    HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  // Workgroup = one wavefront.  `group` lanes advance one env together (the
  // code object reports its shape in dmc_info); `group` < 0: the task-setup
  // kernel, always one env per lane of a full wave.
  const dmc_model_info& mi = b->model->info;
  unsigned block = 64, per_block = 64;
  if (group > 1) {   // several lanes per env: 64 threads, or two wavefronts per env
    per_block = (unsigned)mi.envs_per_block;
    block = (unsigned)(mi.lanes_per_env*mi.envs_per_block);
  }
  else if (group == 1) { block = per_block = (unsigned)mi.envs_per_block; }
  const unsigned grid = (unsigned)((b->nenv + per_block - 1)/per_block);
  HIP_TRY(hipModuleLaunchKernel(fn, grid, 1, 1, block, 1, 1, 0, b->stream,
                                nullptr, config));
  return 0;
}

// [rows][n] <-> [n][rows] on the host (elements of `elem` bytes): code objects
// with env-major state keep [nenv][k] in HBM while the ABI presents [k][nenv]
void transpose_host(const char* src, char* dst, size_t src_rows, size_t src_cols,
                    size_t elem) {
  for (size_t r = 0; r < src_rows; r++)
    for (size_t c = 0; c < src_cols; c++)
      memcpy(dst + (c*src_rows + r)*elem, src + (r*src_cols + c)*elem, elem);
}

void fill_args(dmc_batch* b, DmcArgs& a) {
  memset(&a, 0, sizeof a);
  a.nenv = b->nenv;
  a.nsub = 1;
  a.task_param_i = b->task_param_i;
  memcpy(a.task_param_r, b->task_param_r, sizeof a.task_param_r);
  a.qpos = b->field[DMC_FIELD_QPOS];
  a.qvel = b->field[DMC_FIELD_QVEL];
  a.warm = b->field[DMC_FIELD_WARMSTART];
  a.time = b->field[DMC_FIELD_TIME];
  a.ctrl_store = b->field[DMC_FIELD_CTRL];
  a.obs = b->field[DMC_FIELD_OBS];
  a.obs_sk = 1;                       // agent layout [nenv][nobs]
  a.obs_se = b->model->info.nobs;
  a.reward = b->field[DMC_FIELD_REWARD];
  a.episode_return = b->field[DMC_FIELD_RETURN];
  a.taskdata = b->field[DMC_FIELD_TASKDATA];
  a.sensordata = b->field[DMC_FIELD_SENSORDATA];
  a.xpos = b->aux_outputs ? b->field[DMC_FIELD_XPOS] : nullptr;
  a.xmat = b->aux_outputs ? b->field[DMC_FIELD_XMAT] : nullptr;
  a.qacc = b->aux_outputs ? b->field[DMC_FIELD_QACC] : nullptr;
  a.warn = (unsigned*)b->field[DMC_FIELD_WARN];
  a.stats = (int*)b->field[DMC_FIELD_STATS];
  a.ws = b->ws;
}

}  // namespace

extern "C" {

int dmc_version(void) { return 100; }

const char* dmc_last_error(void) { return g_error.c_str(); }

int dmc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

namespace {

// shared by dmc_model_load (file) and dmc_model_load_data (memory image)
int load_model(const char* path, const void* image, int device_id, dmc_model** out);

// libhiprtc is opened on first use, so that the library loads (and every other
// entry point works) on systems without it
struct Hiprtc {
  void* lib = nullptr;
  decltype(&hiprtcCreateProgram) create = nullptr;
  decltype(&hiprtcCompileProgram) compile = nullptr;
  decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
  decltype(&hiprtcGetProgramLog) log = nullptr;
  decltype(&hiprtcGetCodeSize) code_size = nullptr;
  decltype(&hiprtcGetCode) code = nullptr;
  decltype(&hiprtcDestroyProgram) destroy = nullptr;
  decltype(&hiprtcGetErrorString) error_string = nullptr;
};

// Opens libhiprtc once (thread-safe: a function-local static is initialised
// exactly once) and keeps the reason when that fails -- dlerror() is cleared by
// the call that reads it and overwritten by any later dl* call, so it is read
// once, right where the failure happened.
struct HiprtcOnce {
  Hiprtc rtc{};
  std::string why;
  HiprtcOnce() {
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      rtc.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (rtc.lib) break;
      const char* e = dlerror();
      if (e) why = e;
    }
    if (!rtc.lib) {
      if (why.empty()) why = "dlopen failed";
      return;
    }
    why.clear();
#define DMC_SYM(field, symbol)                                              \
    if (rtc.lib) {                                                          \
      rtc.field = (decltype(rtc.field))dlsym(rtc.lib, #symbol);             \
      if (!rtc.field) {                                                     \
        const char* e = dlerror();                                          \
        why = e ? e : "symbol " #symbol " missing";                         \
        dlclose(rtc.lib);                                                   \
        rtc.lib = nullptr;                                                  \
      }                                                                     \
    }
    DMC_SYM(create, hiprtcCreateProgram)
    DMC_SYM(compile, hiprtcCompileProgram)
    DMC_SYM(log_size, hiprtcGetProgramLogSize)
    DMC_SYM(log, hiprtcGetProgramLog)
    DMC_SYM(code_size, hiprtcGetCodeSize)
    DMC_SYM(code, hiprtcGetCode)
    DMC_SYM(destroy, hiprtcDestroyProgram)
    DMC_SYM(error_string, hiprtcGetErrorString)
#undef DMC_SYM
  }
};

const HiprtcOnce* hiprtc_once() {
  static const HiprtcOnce once;
  return &once;
}

const Hiprtc* hiprtc() {
  const HiprtcOnce* o = hiprtc_once();
  return o->rtc.lib ? &o->rtc : nullptr;
}

}  // namespace

int dmc_model_compile(const char* source, const char* source_name,
                      const char* const* header_names,
                      const char* const* header_texts, int nheaders,
                      const char* const* options, int noptions,
                      void** code, size_t* code_size, char* log, size_t log_size) {
  if (log && log_size) log[0] = 0;
  if (!source || !code || !code_size || nheaders < 0 || noptions < 0 ||
      (nheaders && (!header_names || !header_texts)) || (noptions && !options))
    return fail("dmc_model_compile: bad argument");
  *code = nullptr;
  *code_size = 0;
  const Hiprtc* rtc = hiprtc();
  if (!rtc)
    return fail("dmc_model_compile: the HIP runtime-compilation library "
                "(libhiprtc) is not available: %s", hiprtc_once()->why.c_str());
  hiprtcProgram prog = nullptr;
  hiprtcResult rc = rtc->create(&prog, source, source_name ? source_name : "dmc_model.hip",
                                nheaders, const_cast<const char**>(header_texts),
                                const_cast<const char**>(header_names));
  if (rc != HIPRTC_SUCCESS)
    return fail("hiprtcCreateProgram failed: %s", rtc->error_string(rc));
  rc = rtc->compile(prog, noptions, const_cast<const char**>(options));
  size_t n = 0;
  if (log && log_size && rtc->log_size(prog, &n) == HIPRTC_SUCCESS && n > 1) {
    std::vector<char> text(n + 1, 0);
    if (rtc->log(prog, text.data()) == HIPRTC_SUCCESS) {
      // keep the END of a long log: that is where the errors are
      const size_t keep = n < log_size ? n : log_size - 1;
      memcpy(log, text.data() + (n - keep), keep);
      log[keep] = 0;
    }
  }
  if (rc != HIPRTC_SUCCESS) {
    rtc->destroy(&prog);
    return fail("hiprtcCompileProgram failed: %s (see the log)", rtc->error_string(rc));
  }
  size_t size = 0;
  rc = rtc->code_size(prog, &size);
  void* buf = (rc == HIPRTC_SUCCESS && size) ? malloc(size) : nullptr;
  if (buf) rc = rtc->code(prog, (char*)buf);
  rtc->destroy(&prog);
  if (!buf || rc != HIPRTC_SUCCESS) {
    free(buf);
    return fail("hiprtcGetCode failed: %s", rtc->error_string(rc));
  }
  *code = buf;
  *code_size = size;
  return 0;
}

void dmc_code_free(void* code) { free(code); }

int dmc_model_load(const char* path, int device_id, dmc_model** out) {
  if (!path || !out) return fail("dmc_model_load: null argument");
  return load_model(path, nullptr, device_id, out);
}

int dmc_model_load_data(const void* code, size_t code_size, int device_id,
                        dmc_model** out) {
  if (!code || !code_size || !out) return fail("dmc_model_load_data: null argument");
  return load_model("<memory image>", code, device_id, out);
}

}  // extern "C"

namespace {

int load_model(const char* path, const void* image, int device_id, dmc_model** out) {
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail("dmc_model_load: no HIP device available (the physics step "
                "has no CPU fallback)");
  if (device_id < 0 || device_id >= ndev)
    return fail("dmc_model_load: device %d out of range (%d devices)",
                device_id, ndev);
  HIP_TRY(hipSetDevice(device_id));
  dmc_model* m = new (std::nothrow) dmc_model;
  if (!m) return fail("out of host memory");
  m->device = device_id;
  hipError_t err = image ? hipModuleLoadData(&m->module, image)
                         : hipModuleLoad(&m->module, path);
  if (err != hipSuccess) {
    delete m;
    (void)hipGetLastError();   // clear the runtime's sticky last-error
    return fail("hipModuleLoad(%s) failed: %s", path, hipGetErrorString(err));
  }
  struct { const char* name; hipFunction_t* fn; } fns[] = {
      {"dmc_step", &m->k_step}, {"dmc_observe", &m->k_observe},
      {"dmc_init_episode", &m->k_init}};
  for (auto& f : fns) {
    err = hipModuleGetFunction(f.fn, m->module, f.name);
    if (err != hipSuccess) {
      (void)hipModuleUnload(m->module);
      delete m;
      (void)hipGetLastError();
      return fail("code object %s lacks kernel %s", path, f.name);
    }
  }
  hipDeviceptr_t dptr = nullptr;
  size_t bytes = 0;
  err = hipModuleGetGlobal(&dptr, &bytes, m->module, "dmc_info");
  int raw[20] = {0};
  if (err == hipSuccess && bytes >= sizeof raw)
    err = hipMemcpy(raw, dptr, sizeof raw, hipMemcpyDeviceToHost);
  if (err != hipSuccess || raw[0] != 1) {
    (void)hipModuleUnload(m->module);
    delete m;
    return fail("code object %s has no valid dmc_info table", path);
  }
  dmc_model_info& i = m->info;
  i.abi = raw[0]; i.real_size = raw[1]; i.nq = raw[2]; i.nv = raw[3];
  i.nu = raw[4]; i.nbody = raw[5]; i.nobs = raw[6]; i.nsensordata = raw[7];
  i.ws_per_env = raw[8]; i.task = raw[9]; i.ncon_max = raw[10];
  i.nefc_max = raw[11]; i.integrator = raw[12]; i.npair = raw[13];
  // raw[14] envs per workgroup, raw[17] threads per workgroup: one env per lane
  // (64, 32 or 16 envs in a 64-wide wave) or a group of lanes per env
  i.envs_per_block = raw[14] > 0 ? raw[14] : 64;
  i.lanes_per_env = raw[17] > i.envs_per_block ? raw[17]/i.envs_per_block : 1;
  i.env_major = raw[15] != 0;
  i.ntaskdata = raw[16];
  *out = m;
  return 0;
}

}  // namespace

extern "C" {

int dmc_model_get_info(const dmc_model* model, dmc_model_info* info) {
  if (!model || !info) return fail("dmc_model_get_info: null argument");
  *info = model->info;
  return 0;
}

void dmc_model_free(dmc_model* m) {
  if (!m) return;
  if (m->module) (void)hipModuleUnload(m->module);
  delete m;
}

int dmc_batch_create(const dmc_model* model, int nenv, dmc_batch** out) {
  if (!model || !out) return fail("dmc_batch_create: null argument");
  if (nenv <= 0) return fail("dmc_batch_create: nenv must be positive");
  *out = nullptr;
  HIP_TRY(hipSetDevice(model->device));
  dmc_batch* b = new (std::nothrow) dmc_batch;
  if (!b) return fail("out of host memory");
  b->model = model;
  b->nenv = nenv;
  const dmc_model_info& i = model->info;
  const size_t rs = (size_t)i.real_size, n = (size_t)nenv;
  auto atleast1 = [](int v) { return (size_t)(v > 0 ? v : 1); };
  b->bytes[DMC_FIELD_QPOS] = atleast1(i.nq)*n*rs;
  b->bytes[DMC_FIELD_QVEL] = atleast1(i.nv)*n*rs;
  b->bytes[DMC_FIELD_WARMSTART] = atleast1(i.nv)*n*rs;
  b->bytes[DMC_FIELD_TIME] = n*rs;
  b->bytes[DMC_FIELD_CTRL] = atleast1(i.nu)*n*rs;
  b->bytes[DMC_FIELD_OBS] = atleast1(i.nobs)*n*rs;
  b->bytes[DMC_FIELD_REWARD] = n*rs;
  b->bytes[DMC_FIELD_SENSORDATA] = atleast1(i.nsensordata)*n*rs;
  b->bytes[DMC_FIELD_XPOS] = (size_t)i.nbody*3*n*rs;
  b->bytes[DMC_FIELD_XMAT] = (size_t)i.nbody*9*n*rs;
  b->bytes[DMC_FIELD_QACC] = atleast1(i.nv)*n*rs;
  b->bytes[DMC_FIELD_WARN] = n*sizeof(unsigned);
  b->bytes[DMC_FIELD_STATS] = 3*n*sizeof(int);
  b->bytes[DMC_FIELD_RETURN] = n*rs;
  b->bytes[DMC_FIELD_TASKDATA] = atleast1(i.ntaskdata)*n*rs;
  for (int f = 0; f < DMC_FIELD_COUNT; f++) {
    b->elem[f] = (f == DMC_FIELD_WARN || f == DMC_FIELD_STATS) ? sizeof(int) : rs;
    b->rows[f] = b->bytes[f]/(n*b->elem[f]);
  }
  b->rows[DMC_FIELD_OBS] = 1;   // agent layout [nenv][nobs] in every code object
  hipError_t err = hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking);
  b->stream = b->own_stream;
  for (int f = 0; f < DMC_FIELD_COUNT && err == hipSuccess; f++) {
    err = hipMalloc(&b->field[f], b->bytes[f]);
    if (err == hipSuccess) err = hipMemset(b->field[f], 0, b->bytes[f]);
  }
  // workspace: ws_per_env reals per env, laid out for the batch rounded up to
  // whole 64-env workgroups (surplus lanes of the last workgroup own a slot)
  const size_t npad = (n + 63)/64*64;
  if (err == hipSuccess)
    err = hipMalloc(&b->ws, atleast1(i.ws_per_env)*npad*rs);
  if (err == hipSuccess)
    err = hipMemset(b->ws, 0, atleast1(i.ws_per_env)*npad*rs);
  if (err == hipSuccess) err = hipEventCreate(&b->ev0);
  if (err == hipSuccess) err = hipEventCreate(&b->ev1);
  if (err != hipSuccess) {
    dmc_batch_free(b);
    (void)hipGetLastError();
    return fail("dmc_batch_create: %s", hipGetErrorString(err));
  }
  *out = b;
  {   // per-instance task data starts from the compiled model's values
    DmcArgs a;
    fill_args(b, a);
    a.flags = DMC_FLAG_RESET_ONLY | DMC_FLAG_TASKDATA_DEFAULT;
    if (launch(b, model->k_init, a)) return -1;
  }
  return dmc_batch_reset(b);
}

void dmc_batch_free(dmc_batch* b) {
  if (!b) return;
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  for (int f = 0; f < DMC_FIELD_COUNT; f++)
    if (b->field[f]) (void)hipFree(b->field[f]);
  if (b->ws) (void)hipFree(b->ws);
  if (b->ctrl_staging) (void)hipFree(b->ctrl_staging);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
  delete b;
}

int dmc_batch_nenv(const dmc_batch* b) { return b ? b->nenv : 0; }

int dmc_batch_set_aux_outputs(dmc_batch* b, int enabled) {
  if (!b) return fail("null batch");
  b->aux_outputs = enabled != 0;
  return 0;
}

int dmc_batch_set_task_params(dmc_batch* b, int iparam, const double* r, int nr) {
  if (!b) return fail("null batch");
  if (nr < 0 || nr > 4) return fail("at most 4 real task parameters");
  b->task_param_i = iparam;
  for (int k = 0; k < nr; k++) b->task_param_r[k] = r[k];
  return 0;
}

int dmc_batch_reset(dmc_batch* b) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  // mj_resetData: an init_episode launch with a task-less code path would
  // need the model tables; instead run dmc_init_episode with TASK bits
  // masked off via flags=16 (reset only).
  DmcArgs a;
  fill_args(b, a);
  a.flags = DMC_FLAG_RESET_ONLY;
  if (launch(b, b->model->k_init, a)) return -1;
  HIP_TRY(hipMemsetAsync(b->field[DMC_FIELD_WARN], 0, b->bytes[DMC_FIELD_WARN],
                         b->stream));
  HIP_TRY(hipMemsetAsync(b->field[DMC_FIELD_STATS], 0,
                         b->bytes[DMC_FIELD_STATS], b->stream));
  return 0;
}

int dmc_batch_set_state(dmc_batch* b, const void* qpos, const void* qvel,
                        const void* warm, const void* time) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  struct { const void* src; int f; } items[] = {
      {qpos, DMC_FIELD_QPOS}, {qvel, DMC_FIELD_QVEL},
      {warm, DMC_FIELD_WARMSTART}, {time, DMC_FIELD_TIME}};
  std::vector<char> tmp;
  for (auto& it : items) {
    if (!it.src) continue;
    const void* src = it.src;
    if (b->model->info.env_major && b->rows[it.f] > 1) {
      tmp.resize(b->bytes[it.f]);
      transpose_host((const char*)it.src, tmp.data(), b->rows[it.f],
                     (size_t)b->nenv, b->elem[it.f]);
      src = tmp.data();
    }
    HIP_TRY(hipMemcpyAsync(b->field[it.f], src, b->bytes[it.f],
                           hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));   // `tmp` is reused
  }
  return 0;
}

int dmc_batch_write(dmc_batch* b, int field, const void* src, size_t bytes) {
  if (!b || !src) return fail("dmc_batch_write: null argument");
  // the integration state, the per-instance task data and what a checkpoint
  // must restore besides (last applied control, episode return, warning mask)
  if (field != DMC_FIELD_QPOS && field != DMC_FIELD_QVEL &&
      field != DMC_FIELD_WARMSTART && field != DMC_FIELD_TIME &&
      field != DMC_FIELD_TASKDATA && field != DMC_FIELD_CTRL &&
      field != DMC_FIELD_RETURN && field != DMC_FIELD_WARN)
    return fail("dmc_batch_write: field %d is not writable", field);
  if (bytes != b->bytes[field])
    return fail("dmc_batch_write: field %d has %zu bytes, caller passed %zu",
                field, b->bytes[field], bytes);
  HIP_TRY(hipSetDevice(b->model->device));
  std::vector<char> tmp;
  if (b->model->info.env_major && b->rows[field] > 1) {
    tmp.resize(bytes);
    transpose_host((const char*)src, tmp.data(), b->rows[field], (size_t)b->nenv,
                   b->elem[field]);
    src = tmp.data();
  }
  HIP_TRY(hipMemcpyAsync(b->field[field], src, bytes, hipMemcpyHostToDevice,
                         b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

int dmc_batch_init_episode(dmc_batch* b, uint64_t seed, int only_colliding) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  DmcArgs a;
  fill_args(b, a);
  a.seed = seed;
  a.flags = only_colliding ? DMC_FLAG_ONLY_COLLIDING : 0;
  return launch(b, b->model->k_init, a);
}

int dmc_batch_forward(dmc_batch* b, int count_contacts) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  DmcArgs a;
  fill_args(b, a);
  a.flags = count_contacts ? DMC_FLAG_COUNT_CONTACTS : 0;
  return launch(b, b->model->k_observe, a, b->model->info.lanes_per_env);
}

int dmc_batch_step(dmc_batch* b, const void* ctrl, long long stride_k,
                   long long stride_env, int on_device, int nsub,
                   int want_outputs) {
  if (!b) return fail("null batch");
  if (nsub < 0) return fail("nsub must be >= 0");
  HIP_TRY(hipSetDevice(b->model->device));
  const dmc_model_info& i = b->model->info;
  DmcArgs a;
  fill_args(b, a);
  a.nsub = nsub;
  a.flags = (want_outputs & DMC_STEP_OUTPUTS) ? 0 : DMC_FLAG_NO_OUTPUT;
  if (want_outputs & DMC_STEP_STALE_FIRST) a.flags |= DMC_FLAG_STALE_FIRST;
  if (ctrl && i.nu > 0) {
    a.flags |= DMC_FLAG_CTRL;
    if (on_device) {
      a.ctrl = ctrl;
    } else {
      // host controls: extent = 1 + (nu-1)*sk + (nenv-1)*se reals
      const size_t extent = (size_t)(1 + (i.nu - 1)*stride_k +
                                     (long long)(b->nenv - 1)*stride_env);
      const size_t bytes = extent*(size_t)i.real_size;
      if (bytes > b->ctrl_staging_bytes) {
        if (b->ctrl_staging) HIP_TRY(hipFree(b->ctrl_staging));
        b->ctrl_staging = nullptr;
        HIP_TRY(hipMalloc(&b->ctrl_staging, bytes));
        b->ctrl_staging_bytes = bytes;
      }
      HIP_TRY(hipMemcpyAsync(b->ctrl_staging, ctrl, bytes,
                             hipMemcpyHostToDevice, b->stream));
      a.ctrl = b->ctrl_staging;
    }
    a.ctrl_sk = stride_k;
    a.ctrl_se = stride_env;
  }
  if (launch(b, b->model->k_step, a, b->model->info.lanes_per_env)) return -1;
  if (b->timing) b->launches++;
  return 0;
}

int dmc_batch_step_n(dmc_batch* b, const void* ctrl, long long stride_k,
                     long long stride_env, long long stride_t, int nsteps,
                     int nsub, int want_outputs) {
  if (!b) return fail("null batch");
  if (!ctrl) return fail("dmc_batch_step_n: device controls required");
  if (nsteps < 0) return fail("nsteps must be >= 0");
  const size_t rs = (size_t)b->model->info.real_size;
  for (int t = 0; t < nsteps; t++)
    if (dmc_batch_step(b, (const char*)ctrl + (size_t)t*(size_t)stride_t*rs, stride_k,
                       stride_env, 1, nsub, want_outputs))
      return -1;
  return 0;
}

size_t dmc_batch_field_bytes(const dmc_batch* b, int field) {
  if (!b || field < 0 || field >= DMC_FIELD_COUNT) return 0;
  return b->bytes[field];
}

int dmc_batch_read(dmc_batch* b, int field, void* dst, size_t bytes) {
  if (!b || !dst) return fail("dmc_batch_read: null argument");
  if (field < 0 || field >= DMC_FIELD_COUNT) return fail("unknown field %d", field);
  if (bytes != b->bytes[field])
    return fail("dmc_batch_read: field %d has %zu bytes, caller asked for %zu",
                field, b->bytes[field], bytes);
  HIP_TRY(hipSetDevice(b->model->device));
  const bool flip = b->model->info.env_major && b->rows[field] > 1;
  std::vector<char> tmp(flip ? bytes : 0);
  HIP_TRY(hipMemcpyAsync(flip ? (void*)tmp.data() : dst, b->field[field], bytes,
                         hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  if (flip)
    transpose_host(tmp.data(), (char*)dst, (size_t)b->nenv, b->rows[field],
                   b->elem[field]);
  return 0;
}

void* dmc_batch_device_ptr(dmc_batch* b, int field) {
  if (!b || field < 0 || field >= DMC_FIELD_COUNT) return nullptr;
  return b->field[field];
}

int dmc_batch_clear_warnings(dmc_batch* b) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  HIP_TRY(hipMemsetAsync(b->field[DMC_FIELD_WARN], 0, b->bytes[DMC_FIELD_WARN],
                         b->stream));
  return 0;
}

int dmc_batch_copy_state(dmc_batch* dst, const dmc_batch* src) {
  if (!dst || !src) return fail("null batch");
  if (dst->nenv != src->nenv || dst->model->info.nq != src->model->info.nq ||
      dst->model->info.real_size != src->model->info.real_size ||
      dst->model->info.env_major != src->model->info.env_major)
    return fail("dmc_batch_copy_state: incompatible batches");
  HIP_TRY(hipSetDevice(dst->model->device));
  HIP_TRY(hipStreamSynchronize(src->stream));
  for (int f = 0; f < DMC_FIELD_COUNT; f++)
    HIP_TRY(hipMemcpyAsync(dst->field[f], src->field[f], dst->bytes[f],
                           hipMemcpyDeviceToDevice, dst->stream));
  dst->task_param_i = src->task_param_i;
  dst->aux_outputs = src->aux_outputs;
  memcpy(dst->task_param_r, src->task_param_r, sizeof dst->task_param_r);
  HIP_TRY(hipStreamSynchronize(dst->stream));
  return 0;
}

int dmc_batch_sync(dmc_batch* b) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return 0;
}

int dmc_batch_set_stream(dmc_batch* b, void* stream, int external) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  HIP_TRY(hipStreamSynchronize(b->stream));
  b->stream = external ? (hipStream_t)stream : b->own_stream;
  return 0;
}

void* dmc_batch_stream(dmc_batch* b) { return b ? (void*)b->stream : nullptr; }

int dmc_batch_timer_start(dmc_batch* b) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  HIP_TRY(hipEventRecord(b->ev0, b->stream));
  b->timing = true;
  b->launches = 0;
  return 0;
}

int dmc_batch_timer_stop(dmc_batch* b, double* ms, long long* launches) {
  if (!b) return fail("null batch");
  HIP_TRY(hipSetDevice(b->model->device));
  HIP_TRY(hipEventRecord(b->ev1, b->stream));
  HIP_TRY(hipEventSynchronize(b->ev1));
  float t = 0;
  HIP_TRY(hipEventElapsedTime(&t, b->ev0, b->ev1));
  if (ms) *ms = t;
  if (launches) *launches = b->launches;
  b->timing = false;
  return 0;
}

}  // extern "C"
