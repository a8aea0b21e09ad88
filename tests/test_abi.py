"""The C-ABI library loads and exports every symbol include/dmc_hip.h declares."""

import ctypes
import os
import re

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
