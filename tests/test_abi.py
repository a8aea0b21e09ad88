"""The C-ABI library loads and exports every symbol include/dmc_hip.h declares."""

import ctypes
import os
import re

import pytest

from dm_control_amd import build
from dm_control_amd import wrapper

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
  with open(os.path.join(ROOT, 'include', 'dmc_hip.h')) as f:
    text = f.read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(dmc_[a-z_0-9]+)\s*\(', text)))


def test_header_symbols_are_exported_and_bound():
  build.build_library()
  lib = ctypes.CDLL(wrapper.get_lib_path())
  names = _declared_functions()
  assert len(names) >= 20
  for name in names:
    assert hasattr(lib, name), 'libdmc_hip.so does not export %s' % name
  # the ctypes shim binds exactly the declared surface
  assert sorted(wrapper.SIGNATURES) == names


def test_version_and_error_channel_without_device():
  lib = wrapper.get_lib()
  assert lib.dmc_version() >= 100
  assert lib.dmc_device_count() >= 0
  if lib.dmc_device_count() == 0:
    ptr = ctypes.c_void_p()
    rc = lib.dmc_model_load(b'/nonexistent.hsaco', 0, ctypes.byref(ptr))
    assert rc != 0
    assert b'no HIP device' in lib.dmc_last_error()


def test_in_process_compile_needs_no_toolchain_and_reports_errors():
  """dmc_model_compile (the compile half of mj_loadXML at the C boundary): HIP
  runtime compilation inside libdmc_hip.so -- no hipcc executable, no files, no
  GPU.  A model that was never pre-built becomes a gfx950 code object (an ELF
  image); a broken source fails with the compiler's log, like the
  `char error[1000]` of mj_loadXML (wrapper/core.py:312-328)."""
  import kat_models
  from dm_control_amd import build
  from dm_control_amd.mjcf import compiler
  model = compiler.from_xml_string(kat_models.GPU_MODELS['ball_on_floor'])
  code = build.code_object_bytes(model, 0, 'f64')
  assert code[:4] == b'\x7fELF' and len(code) > 10000
  with pytest.raises(wrapper.Error) as err:
    wrapper.compile_code_object('__global__ void k() { this is not C++; }', 'bad.hip',
                                {}, ['--offload-arch=gfx950'])
  assert 'error' in str(err.value)
