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
  return sorted(set(re.findall(r"\b(ci_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_binding_agree():
  assert _declared_symbols() == sorted(_native.exported_symbols())


def test_library_exports_every_declared_symbol():
  if not os.path.exists(_native.LIB_PATH):
    pytest.fail(f"{_native.LIB_PATH} missing: run __graft_entry__.build()")
  lib = ctypes.CDLL(_native.LIB_PATH)
  for name in _declared_symbols():
    assert hasattr(lib, name), name
  assert lib.ci_abi_version() == _native.ABI_VERSION


def test_flag_constants_match_the_header():
  """Every CI_FLAG_* of the header has a FLAG_* twin of the same value in the binding."""
  hdr = open(os.path.join(ROOT, "include", "causalimpact_amd.h")).read()
  flags = dict(re.findall(r"#define\s+CI_(FLAG_[A-Z_]+)\s+(\d+)", hdr))
  assert len(flags) >= 6
  for name, value in flags.items():
    assert getattr(_native, name) == int(value), name
  values = sorted(int(v) for v in flags.values())
  assert values == [1 << i for i in range(len(values))]          # distinct single bits


def test_struct_layouts_match_header_sizes():
  # ci_series_params: 20 doubles + 8 doubles + 1; ci_problem: 24 int32/uint32 fields
  assert ctypes.sizeof(_native.SeriesParams) == 8 * 29
  assert ctypes.sizeof(_native.Problem) == 4 * (5 + 8 + 5 + 2 + 2 + 2)
  assert ctypes.sizeof(_native.Outputs) == 8 * 10


def test_argument_validation_happens_before_any_device_call():
  """The C-ABI checks its arguments first, so these errors are reachable without a GPU."""
  import numpy as np
  import pytest
  from causalimpact import _native
  T = 20
  y = np.zeros((1, T), np.float32)
  m = np.zeros((1, T), np.uint8)
  spec = dict.fromkeys(_native._PARAM_FIELDS, 1.0)   # pylint: disable=protected-access
  prm = _native.make_params([spec])

  def create(**kw):
    base = dict(T=T, P=0, has_slope=0, num_warmup=1, num_results=2)
    base.update(kw)
    pb = _native.make_problem(**base)
    Tn, Pn = base["T"], base["P"]
    X = np.zeros((1, Tn, Pn), np.float32) if Pn else None
    return _native.Session(pb, np.zeros((1, Tn), np.float32), np.zeros((1, Tn), np.uint8), X,
                           np.zeros((max(1, len(base.get("num_seasons", ()))), Tn), np.uint8)
                           if base.get("num_seasons") else None, prm)

  for kw, msg in [
      (dict(T=2), "T must be >= 3"),
      (dict(P=5000), "P must be in"),
      (dict(num_results=0), "num_results >= 1"),
      (dict(num_chains=0), "num_chains >= 1"),
      (dict(num_seasons=(1,)), r"num_seasons\[0\] must be >= 2"),
      (dict(T=70000), "exceeds the longest supported series"),
  ]:
    with pytest.raises(_native.NativeError, match=msg):
      create(**kw)
  pb = _native.make_problem(T=T, P=0, has_slope=0, num_warmup=1, num_results=2)
  pb.abi_version = 999
  with pytest.raises(_native.NativeError, match="ABI mismatch"):
    _native.Session(pb, y, m, None, None, prm)
  pb = _native.make_problem(T=T, P=0, has_slope=0, num_warmup=1, num_results=2)
  pb.num_blocks = 9
  with pytest.raises(_native.NativeError, match="num_blocks must be in"):
    _native.Session(pb, y, m, None, np.zeros((9, T), np.uint8), prm)
  # straight through ctypes: a design matrix is required when P > 0
  import ctypes as C
  pb = _native.make_problem(T=T, P=3, has_slope=0, num_warmup=1, num_results=2)
  h = C.c_void_p()
  rc = _native.load().ci_session_create(C.byref(pb), y.ctypes.data, m.ctypes.data, None, None, prm,
                                        C.byref(h))
  assert rc != 0 and b"X is NULL" in _native.load().ci_last_error()


def test_round_3_entry_points_validate_before_any_device_call():
  """ci_fit_gibbs_f64, ci_ll_session_create2 and ci_series_params.weights_prior_scale: argument
  errors are reported without touching a GPU."""
  import ctypes as C
  import numpy as np
  L = _native.load()
  T = 30
  y64 = np.zeros((1, T), np.float64)
  y32 = np.zeros((1, T), np.float32)
  m = np.zeros((1, T), np.uint8)
  spec = dict.fromkeys(_native._PARAM_FIELDS, 1.0)   # pylint: disable=protected-access
  prm = _native.make_params([spec])
  assert prm[0].weights_prior_scale == 1.0                       # the reference's prior by default
  out = _native.Outputs()
  pb = _native.make_problem(T=T, P=2, has_slope=0, num_warmup=1, num_results=2)
  assert L.ci_fit_gibbs_f64(C.byref(pb), y64.ctypes.data, m.ctypes.data, None, None, prm, C.byref(out)) != 0
  assert b"X is NULL" in L.ci_last_error()
  pb = _native.make_problem(T=T, P=0, has_slope=0, num_seasons=(7,), num_warmup=1, num_results=2)
  assert L.ci_fit_gibbs_f64(C.byref(pb), y64.ctypes.data, m.ctypes.data, None, None, prm, C.byref(out)) != 0
  assert b"season_change is NULL" in L.ci_last_error()
  pb = _native.make_problem(T=T, P=0, has_slope=0, num_seasons=(40, 40), num_warmup=1, num_results=2)
  sc = np.zeros((2, T), np.uint8)
  assert L.ci_fit_gibbs_f64(C.byref(pb), y64.ctypes.data, m.ctypes.data, None, sc.ctypes.data, prm,
                            C.byref(out)) != 0
  assert b"too wide" in L.ci_last_error()
  # log-likelihood sessions: blocks need create2's season_change; P is capped at 128 there (round 5: was 52)
  h = C.c_void_p()
  pb = _native.make_problem(T=T, P=0, has_slope=0, num_seasons=(7,), num_warmup=0, num_results=1)
  assert L.ci_ll_session_create(C.byref(pb), prm, y32.ctypes.data, m.ctypes.data, None, 4, C.byref(h)) != 0
  assert b"ci_ll_session_create2" in L.ci_last_error()
  assert L.ci_ll_session_create2(C.byref(pb), prm, y32.ctypes.data, m.ctypes.data, None, None, 4,
                                 C.byref(h)) != 0
  assert b"season_change is NULL" in L.ci_last_error()
  pb = _native.make_problem(T=T, P=130, has_slope=0, num_warmup=0, num_results=1)
  X = np.zeros((T, 130), np.float32)
  assert L.ci_ll_session_create2(C.byref(pb), prm, y32.ctypes.data, m.ctypes.data, X.ctypes.data, None,
                                 4, C.byref(h)) != 0
  assert b"P must be <= 128" in L.ci_last_error()
  # ... and at 52 on its seasonal / long-series routes
  pb = _native.make_problem(T=T, P=60, has_slope=0, num_seasons=(7,), num_warmup=0, num_results=1)
  X = np.zeros((T, 60), np.float32)
  sc1 = np.zeros((1, T), np.uint8)
  assert L.ci_ll_session_create2(C.byref(pb), prm, y32.ctypes.data, m.ctypes.data, X.ctypes.data,
                                 sc1.ctypes.data, 4, C.byref(h)) != 0
  assert b"seasonal blocks or T > 4096: P must be <= 52" in L.ci_last_error()
  # a non-positive multiplier of the weights-prior precision is rejected
  bad = _native.make_params([dict(spec, weights_prior_scale=0.0)])
  pb = _native.make_problem(T=T, P=0, has_slope=0, num_warmup=0, num_results=1)
  assert L.ci_ll_session_create2(C.byref(pb), bad, y32.ctypes.data, m.ctypes.data, None, None, 4,
                                 C.byref(h)) != 0
  assert b"weights_prior_scale" in L.ci_last_error()


def test_make_params_fills_every_member_by_name():
  """make_params writes the ci_series_params table as one float64 block: every named member, the
  drift-scale array and the weights-prior multiplier must land where the struct declares them."""
  specs = []
  for b in range(3):
    sp = {f: 100.0 * b + i for i, f in enumerate(_native._PARAM_FIELDS)}   # pylint: disable=protected-access
    sp["drift_scale0"] = (0.5 + b, 0.25)
    if b == 1:
      sp["weights_prior_scale"] = 9.0
    specs.append(sp)
  arr = _native.make_params(specs)
  for b, sp in enumerate(specs):
    for f in _native._PARAM_FIELDS:                                        # pylint: disable=protected-access
      assert getattr(arr[b], f) == sp[f], (b, f)
    assert list(arr[b].drift_scale0)[:3] == [0.5 + b, 0.25, 0.0]
    assert arr[b].weights_prior_scale == (9.0 if b == 1 else 1.0)
