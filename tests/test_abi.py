"""The C-ABI library loads and exports every symbol include/causalimpact_amd.h declares.
(No compute calls: this runs without a GPU.)"""
import ctypes
import os
import re

import pytest

from causalimpact import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
  hdr = open(os.path.join(ROOT, "include", "causalimpact_amd.h")).read()
  hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
  return sorted(set(re.findall(r"\b(ci_[a-z_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
  assert _declared_symbols() == sorted(_native.exported_symbols())


def test_library_exports_every_declared_symbol():
  if not os.path.exists(_native.LIB_PATH):
    pytest.fail(f"{_native.LIB_PATH} missing: run __graft_entry__.build()")
  lib = ctypes.CDLL(_native.LIB_PATH)
  for name in _declared_symbols():
    assert hasattr(lib, name), name
  assert lib.ci_abi_version() == _native.ABI_VERSION


def test_struct_layouts_match_header_sizes():
  # ci_series_params: 20 doubles + 8 doubles; ci_problem: 22 int32/uint32 fields
  assert ctypes.sizeof(_native.SeriesParams) == 8 * 28
  assert ctypes.sizeof(_native.Problem) == 4 * (5 + 8 + 5 + 2 + 2)
  assert ctypes.sizeof(_native.Outputs) == 8 * 10
