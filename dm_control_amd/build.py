"""hipcc driver: builds libdmc_hip.so and per-model gfx950 code objects.

Everything is built IN-TREE (csrc/libdmc_hip.so, csrc/_build/<key>.hsaco) so
the artefacts travel to the GPU box with the repository snapshot.  hipcc
cross-compiles for gfx950 without a GPU present.
"""

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


def model_key(model, task, precision, ncon_max=None, extra_flags=()):
  src = os.path.join(_CSRC, 'dmc_kernels.hip')
  h = hashlib.sha1()
  h.update(model.content_hash().encode())
  h.update(('%d/%s/%r/%r' % (task, precision, ncon_max,
                             tuple(extra_flags))).encode())
  for path in (src, os.path.join(_CSRC, 'dmc_args.h'),
               codegen.__file__):
    with open(path, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()[:20]


def code_object_path(model, task=codegen.TASK_NONE, precision='f32',
                     ncon_max=None, extra_flags=()):
  return os.path.join(_BUILD, 'dmc_%s.hsaco' % model_key(
      model, task, precision, ncon_max, extra_flags))


def build_model(model, task=codegen.TASK_NONE, precision='f32',
                ncon_max=None, force=False, keep_temps=False, extra_flags=None):
  """Generates the constants header for `model` and compiles its kernels.

  Returns the path of the gfx950 code object.  Built lazily and cached by
  content hash; `__graft_entry__.build()` pre-builds the suite models.
  """
  if precision not in ('f32', 'f64'):
    raise ValueError('precision must be "f32" or "f64"')
  os.makedirs(_BUILD, exist_ok=True)
  if extra_flags is None:
    # experiment hook: extra -D flags for ablation builds (never set in tests)
    extra_flags = tuple(os.environ.get('DMC_EXTRA_FLAGS', '').split())
  out = code_object_path(model, task, precision, ncon_max, extra_flags)
  if os.path.exists(out) and not force:
    return out
  key = os.path.basename(out)[4:-6]
  header = os.path.join(_BUILD, 'model_%s.h' % key)
  with open(header, 'w') as f:
    f.write(codegen.generate_header(model, task, ncon_max))
  # -pragma-unroll-threshold: the per-model straight-line code is far beyond
  #   LLVM's default budget; without it the pair loop stays rolled, per-lane
  #   arrays are indexed dynamically and the whole working set lands in scratch.
  # -fno-slp-vectorize: v_pk_*_f32 is not faster on gfx950 and the packing
  #   moves cost ~25 % extra instructions plus spills.
  # -fno-hip-fp32-correctly-rounded-divide-sqrt (fp32 build only): v_rcp/v_rsq
  #   based division and sqrt (<= 2.5 ulp) instead of the 10-instruction
  #   IEEE sequences; the fp64 build keeps exact division.
  cmd = [_hipcc(), '--genco', '--offload-arch=' + ARCH, '-O3', '-std=c++17',
         '-mllvm', '-pragma-unroll-threshold=10000000', '-fno-slp-vectorize',
         '-ffp-contract=off' if precision == 'f64' else '-ffp-contract=fast',
         '-DDMC_MODEL_HEADER="%s"' % header, '-I', _CSRC,
         '-o', out + '.tmp', os.path.join(_CSRC, 'dmc_kernels.hip')]
  cmd[1:1] = list(extra_flags)
  if precision == 'f64':
    cmd.insert(1, '-DDMC_REAL_IS_DOUBLE')
  else:
    cmd.insert(1, '-fno-hip-fp32-correctly-rounded-divide-sqrt')
  if keep_temps:
    cmd[1:1] = ['-save-temps', '-Rpass-analysis=kernel-resource-usage']
  try:
    subprocess.check_call(cmd, cwd=_BUILD)
  except subprocess.CalledProcessError as e:
    raise RuntimeError('hipcc failed for model kernels: %s' % e)
  os.replace(out + '.tmp', out)
  return out
