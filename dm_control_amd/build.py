"""hipcc driver: builds libdmc_hip.so and per-model gfx950 code objects.

Everything is built IN-TREE (csrc/libdmc_hip.so, csrc/_build/<key>.hsaco) so
the artefacts travel to the GPU box with the repository snapshot.  hipcc
cross-compiles for gfx950 without a GPU present.
"""

import contextlib
import hashlib
import os
import shutil
import subprocess

from dm_control_amd import codegen

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
_BUILD = os.path.join(_CSRC, '_build')
LIB_PATH = os.path.join(_CSRC, 'libdmc_hip.so')
ARCH = 'gfx950'


def _hipcc():
  exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  if not os.path.exists(exe):
    raise RuntimeError('hipcc not found; cannot build the HIP extension')
  return exe


def backend():
  """How model code objects are built: "hipcc" (the toolchain's driver, what
  `__graft_entry__.build()` uses ahead of time) or "hiprtc" -- in-process
  through `dmc_model_compile` of the C ABI (HIP runtime compilation: needs the
  ROCm runtime only, no hipcc executable).  `$DMC_BUILD_BACKEND` selects;
  default: hipcc when it exists, else hiprtc."""
  choice = os.environ.get('DMC_BUILD_BACKEND')
  if choice in ('hipcc', 'hiprtc'):
    return choice
  exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
  return 'hipcc' if os.path.exists(exe) else 'hiprtc'


def _newer(target, sources):
  if not os.path.exists(target):
    return False
  t = os.path.getmtime(target)
  return all(os.path.getmtime(s) <= t for s in sources)


def build_library(force=False):
  """Compiles the C-ABI runtime (include/dmc_hip.h) into libdmc_hip.so."""
  srcs = [os.path.join(_CSRC, 'dmc_api.cpp'),
          os.path.join(_CSRC, 'dmc_args.h'),
          os.path.join(_CSRC, '..', '..', 'include', 'dmc_hip.h')]
  if not force and _newer(LIB_PATH, srcs):
    return LIB_PATH
  cmd = [_hipcc(), '-O2', '-fPIC', '-shared', '-std=c++17',
         '--offload-arch=' + ARCH, '-o', LIB_PATH, srcs[0]]
  subprocess.check_call(cmd)
  return LIB_PATH


# Spill budget of the fully unrolled ("static") build.  Kernels far beyond it
# (fp64 builds of 20-dof models: ~1900 VGPR + ~200 SGPR spills, 300 KB of code)
# have produced wrong results on gfx950 with ROCm 7.2 while the same source is
# correct on the host under ASan/UBSan and with gcc/clang -O3 (see DESIGN.md,
# "compiler hazard"), so `mode="auto"` falls back to the rolled build, whose
# per-lane arrays are explicit scratch objects instead of register spills.
MAX_VGPR_SPILLS = 128
MAX_SGPR_SPILLS = 128
# The several-lanes kernel (csrc/dmc_coop.hip) keeps per-env data in LDS and its
# uniform address arithmetic in SGPRs: every suite build spills 136-386 SGPRs
# (to VGPR lanes, no scratch traffic) and at most 7 VGPRs, and all of them are
# parity-tested per step against the oracle.  What the guard must refuse there
# is the same thing as above: builds that spill VGPRs wholesale to scratch
# (the 62-dof soccer walker: 1652 fp32 / 6552 fp64).
COOP_MAX_SGPR_SPILLS = 640
SEMI_ROLLED_MAX_NV = 16     # see build_model: the generic tier with the backend unroller on


# fp32 builds: v_rcp / v_rsq based division and sqrt (<= 2.5 ulp) instead of the
# IEEE sequences, and x / y as x * rcp(y).  Without the second flag LLVM lowers
# every fp32 division to a frexp / rcp / ldexp sequence of 8 instructions -- 9 %
# of the cheetah kernel's code; with it the step takes 5 % (cheetah) to 10 %
# (humanoid) less time.  The fp64 build keeps exact division.
_FP32_FLAGS = ('-fno-hip-fp32-correctly-rounded-divide-sqrt', '-freciprocal-math')


# "rolled" has to mean rolled: the AMDGPU backend raises the unroll thresholds
# of loops that index private arrays (to promote them to registers), which
# turned the generic loops of mid-size models back into straight-line code --
# the rolled fp64 build of the 62-dof walker spilled 14591 VGPRs, took 68 s to
# compile and returned a wrong mass-matrix factor on the GPU (round 3).  With
# the loop unroller off the same build spills nothing and compiles in 5 s.
_ROLLED_FLAGS = ('-fno-unroll-loops',)


def model_key(model, task, precision, ncon_max=None, extra_flags=(),
              unroll=True):
  src = os.path.join(_CSRC, 'dmc_kernels.hip')
  h = hashlib.sha1()
  h.update(model.content_hash().encode())
  h.update(('%d/%s/%r/%r/%s/%d/%s/%s' % (
      task, precision, ncon_max, tuple(extra_flags),
      os.environ.get('DMC_PRAGMA_UNROLL_THRESHOLD', ''), int(unroll is True),
      ' '.join(_FP32_FLAGS),
      '' if unroll is True else ('semi' if unroll == 'semi' else ' '.join(_ROLLED_FLAGS)))).encode())
  for path in (src, os.path.join(_CSRC, 'dmc_coop.hip'),
               os.path.join(_CSRC, 'dmc_args.h'), codegen.__file__):
    with open(path, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()[:20]


def _spills(remarks, kernel='dmc_step'):
  """(vgpr, sgpr) spill counts of `kernel` from -Rpass-analysis remarks, or
  None if the remarks do not hold them (the caller then treats the build as
  over budget: the guard fails closed)."""
  vg = sg = None
  inside = False
  for line in remarks.splitlines():
    if 'Function Name:' in line:
      inside = ('Function Name: %s ' % kernel) in line + ' '
    elif inside and 'VGPRs Spill:' in line:
      vg = int(line.split('VGPRs Spill:')[1].split()[0])
    elif inside and 'SGPRs Spill:' in line:
      sg = int(line.split('SGPRs Spill:')[1].split()[0])
  if vg is None or sg is None:
    return None
  return vg, sg


_OVERBUDGET_MSG = (
    'the unrolled build of this model spills %%s (VGPR, SGPR) registers, beyond '
    'the budget of %d / %d within which this kind of build is trusted '
    '(DESIGN.md 3.4: an over-budget build has produced wrong results on '
    'gfx950); use mode="auto" / "rolled" / "coop", or set '
    '$DMC_ALLOW_OVERBUDGET=1 to build it anyway' % (MAX_VGPR_SPILLS, MAX_SGPR_SPILLS))


_COOP_OVERBUDGET_MSG = (
    'the several-lanes build of this model spills %%s (VGPR, SGPR) registers '
    '(budget %d / %d); set $DMC_ALLOW_OVERBUDGET=1 to build it anyway'
    % (MAX_VGPR_SPILLS, COOP_MAX_SGPR_SPILLS))


def _allow_overbudget():
  """Explicit override for experiments and the canary test."""
  return os.environ.get('DMC_ALLOW_OVERBUDGET') == '1'


@contextlib.contextmanager
def allow_overbudget():
  """`with build.allow_overbudget():` -- builds inside may exceed the spill
  budget (the canary test and tools/spill_hazard/ only; never the product)."""
  prev = os.environ.get('DMC_ALLOW_OVERBUDGET')
  os.environ['DMC_ALLOW_OVERBUDGET'] = '1'
  try:
    yield
  finally:
    if prev is None:
      del os.environ['DMC_ALLOW_OVERBUDGET']
    else:
      os.environ['DMC_ALLOW_OVERBUDGET'] = prev


def _within_spill_budget(spills, max_sgpr=MAX_SGPR_SPILLS):
  return (spills is not None and spills[0] <= MAX_VGPR_SPILLS
          and spills[1] <= max_sgpr)


def _compile(model, task, precision, ncon_max, extra_flags, unroll, out,
             keep_temps, source='dmc_kernels.hip', kernel='dmc_step'):
  key = os.path.basename(out)[4:-6]
  if os.environ.get('DMC_BUILD_LOG'):     # which code objects were not pre-built
    with open(os.environ['DMC_BUILD_LOG'], 'a') as f:
      f.write('%s task=%d %s ncon_max=%r flags=%r unroll=%r nv=%d nbody=%d\n' % (
          source, task, precision, ncon_max, tuple(extra_flags), unroll, model.nv, model.nbody))
  header = os.path.join(_BUILD, 'model_%s.h' % key)
  with open(header, 'w') as f:
    f.write(codegen.generate_header(model, task, ncon_max, unroll=unroll is True))
  # -pragma-unroll-threshold: the per-model straight-line code is far beyond
  #   LLVM's default budget; without it the pair loop stays rolled, per-lane
  #   arrays are indexed dynamically and the whole working set lands in scratch.
  # -fno-slp-vectorize: v_pk_*_f32 is not faster on gfx950 and the packing
  #   moves cost ~25 % extra instructions plus spills.
  # -fno-hip-fp32-correctly-rounded-divide-sqrt (fp32 build only): v_rcp/v_rsq
  #   based division and sqrt (<= 2.5 ulp) instead of the 10-instruction
  #   IEEE sequences; the fp64 build keeps exact division.
  # -ffinite-math-only -fno-signed-zeros: lets LLVM fold 0*x and x+0.  In the
  #   unrolled build the world frame, joint axes and body offsets are
  #   constants, so for the planar suite models (cheetah, walker, hopper,
  #   cart-pole, ...) the y components, two quaternion entries and four matrix
  #   entries of every frame are exact zeros that now disappear at compile time
  #   (cheetah: 25.4 k -> 18.6 k VALU instructions per step).  Values are
  #   unchanged for finite inputs; NaN/inf detection is done on bit patterns
  #   (`bad()` in the kernel source), not with comparisons.
  flags = ['--offload-arch=' + ARCH, '-O3', '-std=c++17',
           '-ffinite-math-only', '-fno-signed-zeros',
           '-Rpass-analysis=kernel-resource-usage',
           '-mllvm', '-pragma-unroll-threshold=%s' % os.environ.get(
               'DMC_PRAGMA_UNROLL_THRESHOLD', '10000000'), '-fno-slp-vectorize',
           '-ffp-contract=off' if precision == 'f64' else '-ffp-contract=fast']
  flags[0:0] = list(extra_flags)
  flags[0:0] = ['-DDMC_REAL_IS_DOUBLE'] if precision == 'f64' else list(_FP32_FLAGS)
  if unroll is False:      # (unroll == 'semi': generic source, the backend may unroll)
    flags += list(_ROLLED_FLAGS)
  if backend() == 'hiprtc':
    with open(header) as f:
      code, log = _compile_in_process(f.read(), source, flags)
    with open(out + '.tmp', 'wb') as f:
      f.write(code)
    if keep_temps:
      print(log)
    return _spills(log, kernel)
  cmd = [_hipcc(), '--genco'] + flags + [
      '-DDMC_MODEL_HEADER="%s"' % header, '-I', _CSRC,
      '-o', out + '.tmp', os.path.join(_CSRC, source)]
  if keep_temps:
    cmd.insert(1, '-save-temps')
  proc = subprocess.run(cmd, cwd=_BUILD, stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, universal_newlines=True)
  if proc.returncode != 0:
    raise RuntimeError('hipcc failed for model kernels:\n%s'
                       % proc.stdout[-4000:])
  if keep_temps:
    print(proc.stdout)
  return _spills(proc.stdout, kernel)


def _compile_in_process(header_text, source, flags):
  """No toolchain driver: the C ABI compiles the same source with the same
  flags in-process (`dmc_model_compile`, HIP runtime compilation); the headers
  travel as text."""
  from dm_control_amd import wrapper
  headers = {'model.h': header_text}
  for name in ('dmc_args.h', 'dmc_kernels.hip'):
    if name != source:
      with open(os.path.join(_CSRC, name)) as f:
        headers[name] = f.read()
  with open(os.path.join(_CSRC, source)) as f:
    text = f.read()
  return wrapper.compile_code_object(
      text, source, headers, list(flags) + ['-DDMC_MODEL_HEADER="model.h"'])


def code_object_bytes(model, task=codegen.TASK_NONE, precision='f32',
                      ncon_max=None, unroll=False, coop_group=None):
  """gfx950 code object of `model` as bytes, built in-process (no hipcc, no
  files): the `mj_loadXML` route for a model that was not pre-built --
  `wrapper.HipModel.from_code(build.code_object_bytes(model))`.  The generic
  (rolled) build by default: it compiles in seconds for any model size."""
  flags = (['-DDMC_REAL_IS_DOUBLE'] if precision == 'f64' else list(_FP32_FLAGS)) + [
           '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-ffinite-math-only',
           '-fno-signed-zeros', '-mllvm', '-pragma-unroll-threshold=10000000',
           '-fno-slp-vectorize',
           '-ffp-contract=off' if precision == 'f64' else '-ffp-contract=fast']
  source = 'dmc_kernels.hip'
  if coop_group:
    flags += ['-DDMC_GROUP=%d' % min(coop_group, 64),
              '-DDMC_COOP_DUO=%d' % (coop_group == 128)]
    source, unroll = 'dmc_coop.hip', True
  if not unroll:
    flags += list(_ROLLED_FLAGS)
  header = codegen.generate_header(model, task, ncon_max, unroll=unroll)
  return _compile_in_process(header, source, flags)[0]


def lds_budget_for(nenv):
  """LDS bytes per workgroup for the constraint-row / contact staging area.

  The step kernel is bound by VALU issue of one wave per SIMD, so how many
  workgroups fit on a CU decides the throughput once the batch exceeds one
  wave per CU.  Measured on cheetah-run (M env-steps/s at 128 / 64 / 36-40 KB):
  B = 8192: 102 / 101 / 85;  16384: 201 / 198 / 169;  32768: 217 / 374 / 319;
  65536: 228 / 404 / 583;  262144: 248 / 456 / 657.
  """
  if nenv is None or nenv <= 16384:
    return 128*1024         # 1 workgroup per CU, most rows in LDS
  if nenv <= 32768:
    return 64*1024          # 2 per CU
  return 36*1024            # 4 per CU: one wave on every SIMD


def build_model(model, task=codegen.TASK_NONE, precision='f32',
                ncon_max=None, force=False, keep_temps=False, extra_flags=None,
                mode='auto', lds_budget=None, group=64):
  """Generates the constants header for `model` and compiles its kernels.

  mode: "unrolled" (static indexing, per-lane state in registers), "rolled"
  (generic loops, per-lane arrays in scratch) or "auto" (unrolled unless its
  register spills exceed MAX_*_SPILLS), "team" (the rolled source with the 64
  lanes of a wavefront sharing ONE env: scenes whose matrices live in the HBM
  workspace, e.g. a soccer pitch), or "coop": `group` lanes advance one
  env together with its working set in LDS (csrc/dmc_coop.hip; the shape for
  nv ~ 20+ models and for small shards; 128 = 64 lanes + a helper wavefront).  Returns the path of the gfx950 code
  object; cached in-tree by content hash.
  """
  if precision not in ('f32', 'f64', 'mixed'):
    raise ValueError('precision must be "f32", "f64" or "mixed"')
  if mode not in ('auto', 'unrolled', 'rolled', 'coop', 'team'):
    raise ValueError('mode must be auto, unrolled, rolled, coop or team')
  if extra_flags is None:
    # experiment hook: extra -D flags for ablation builds (never set in tests)
    extra_flags = tuple(os.environ.get('DMC_EXTRA_FLAGS', '').split())
  if mode == 'team':
    # big scenes: the generic source with one wavefront per env (csrc/dmc_kernels.hip,
    # "team mode"): matrices, rows and contacts in the HBM workspace, a tree's
    # diagonal block at a time in LDS
    if precision == 'mixed':
      raise ValueError('precision "mixed" is built for the one-env-per-lane kernel')
    extra_flags = tuple(extra_flags) + ('-DDMC_TEAM=64',)
    mode = 'rolled'
  if precision == 'mixed':
    # fp32 arithmetic, qpos/qvel carried between steps as fp64 (high, low)
    # pairs (csrc/dmc_kernels.hip, DMC_STATE_COMP); one-env-per-lane kernel only
    if mode == 'coop':
      raise ValueError('precision "mixed" is built for the one-env-per-lane kernel')
    precision = 'f32'
    extra_flags = tuple(extra_flags) + ('-DDMC_STATE_COMP=1',)
  if lds_budget is not None and lds_budget != 128*1024:
    extra_flags = tuple(extra_flags) + ('-DDMC_LDS_BUDGET=%d' % lds_budget,)
  os.makedirs(_BUILD, exist_ok=True)
  if mode == 'coop':
    # several lanes per env (csrc/dmc_coop.hip): working set in LDS, generic
    # loops; `lds_budget` does not apply (no row tiers)
    flags = tuple(f for f in extra_flags if not f.startswith('-DDMC_LDS_BUDGET'))
    if group not in (8, 16, 32, 64, 128):
      raise ValueError('group must be 8, 16, 32, 64 or 128 (two wavefronts) lanes per env')
    # 128: one env per 64 lanes plus a second wavefront that builds the
    # constraint rows and factorises M + h D meanwhile (Euler models; the best
    # shape while the batch fits the chip in one round, 4 envs per CU)
    flags += ('-DDMC_GROUP=%d' % min(group, 64), '-DDMC_COOP_DUO=%d' % (group == 128))
    out = os.path.join(_BUILD, 'dmc_%s.hsaco' % model_key(
        model, task, precision, ncon_max, flags, True))
    if force or not os.path.exists(out):
      spills = _compile(model, task, precision, ncon_max, flags, True, out,
                        keep_temps, source='dmc_coop.hip')
      # the several-lanes kernel keeps its working set in LDS; a build that
      # spills beyond the budget is as untrusted as an over-budget unrolled one
      ok = _within_spill_budget(spills, COOP_MAX_SGPR_SPILLS)
      if not ok and not _allow_overbudget():
        os.remove(out + '.tmp')
        raise RuntimeError(_COOP_OVERBUDGET_MSG % (spills,))
      os.replace(out + '.tmp', out)
      with open(out + ('.ok' if ok else '.overbudget'), 'w') as f:
        f.write('%r' % (spills,))
    elif os.path.exists(out + '.overbudget') and not _allow_overbudget():
      with open(out + '.overbudget') as f:
        raise RuntimeError(_COOP_OVERBUDGET_MSG % f.read().strip())
    return out

  def path(unroll):
    return os.path.join(_BUILD, 'dmc_%s.hsaco' % model_key(
        model, task, precision, ncon_max, extra_flags, unroll))
  marker = path(True) + '.rolled'     # "auto" decided against the unrolled build
  vetted = path(True) + '.ok'         # spill counts recorded and within budget
  if not force:
    if mode != 'rolled' and os.path.exists(path(True)) and (
        os.path.exists(vetted) or (mode == 'unrolled' and _allow_overbudget())):
      return path(True)
    over = path(True) + '.overbudget'   # built once with the override: spills known
    if mode == 'unrolled' and os.path.exists(over) and not _allow_overbudget():
      with open(over) as f:
        raise RuntimeError(_OVERBUDGET_MSG % f.read().strip())
  # `auto` decided against the unrolled build earlier (marker): straight to the
  # generic tiers
  skip_unrolled = mode == 'rolled' or (mode == 'auto' and not force
                                       and os.path.exists(marker))
  if not skip_unrolled:
    out = path(True)
    spills = _compile(model, task, precision, ncon_max, extra_flags, True, out,
                      keep_temps)
    ok = _within_spill_budget(spills)
    if mode == 'unrolled' and not ok and not _allow_overbudget():
      os.remove(out + '.tmp')
      raise RuntimeError(_OVERBUDGET_MSG % (spills,))
    if mode == 'unrolled' or ok:
      os.replace(out + '.tmp', out)
      if not ok:
        with open(out + '.overbudget', 'w') as f:
          f.write('%r' % (spills,))
      if ok:
        with open(vetted, 'w') as f:
          f.write('vgpr spills %d, sgpr spills %d\n' % spills)
      return out
    os.remove(out + '.tmp')
    with open(marker, 'w') as f:
      f.write('spills (vgpr, sgpr): %r\n' % (spills,))
  # Generic ("rolled") source, two tiers.  First with the backend's loop unroller
  # left on: for small models it turns the per-lane loops back into mostly
  # straight-line code within the spill budget (cheetah fp64: 14 VGPR / 300 SGPR
  # spills, 0.33 ms per launch against 1.33 ms strictly rolled).  Beyond the
  # budget -- mid-size models: humanoid fp64 3644, the 62-dof walker 14591 spilled
  # VGPRs -- the strictly rolled form (-fno-unroll-loops: no spills at all).
  # Only small models take this tier: it is the regime every GPU parity test of
  # a suite fp64 build covers, and the one time the unroller was let loose on a
  # big model (the 2v2 pitch) it produced wrong code (csrc/dmc_kernels.hip,
  # DMC_KEEP_ROLLED).
  semi = path('semi')
  if (not force and not os.path.exists(semi + '.strict') and model.nv <= SEMI_ROLLED_MAX_NV
      and os.environ.get('DMC_ROLLED_STRICT') != '1'):   # (experiments: skip this tier)
    if os.path.exists(semi) and os.path.exists(semi + '.ok'):
      return semi
    spills = _compile(model, task, precision, ncon_max, extra_flags, 'semi', semi,
                      keep_temps)
    if _within_spill_budget(spills, COOP_MAX_SGPR_SPILLS):
      os.replace(semi + '.tmp', semi)
      with open(semi + '.ok', 'w') as f:
        f.write('%r' % (spills,))
      return semi
    os.remove(semi + '.tmp')
    with open(semi + '.strict', 'w') as f:
      f.write('spills (vgpr, sgpr) with the loop unroller on: %r\n' % (spills,))
  out = path(False)
  if not force and os.path.exists(out):
    return out
  spills = _compile(model, task, precision, ncon_max, extra_flags, False, out,
                    keep_temps)
  if not _within_spill_budget(spills, COOP_MAX_SGPR_SPILLS) and not _allow_overbudget():
    os.remove(out + '.tmp')
    raise RuntimeError(
        'the rolled build of this model spills %r (VGPR, SGPR) registers (budget '
        '%d / %d); set $DMC_ALLOW_OVERBUDGET=1 to build it anyway'
        % (spills, MAX_VGPR_SPILLS, COOP_MAX_SGPR_SPILLS))
  os.replace(out + '.tmp', out)
  return out
