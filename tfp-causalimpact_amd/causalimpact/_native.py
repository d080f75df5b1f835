"""ctypes binding of the C-ABI in include/causalimpact_amd.h.

This is the only place the host package touches native code.  There is no CPU
fallback: if lib/libcausalimpact_amd.so is missing or no MI355X is visible the
calls raise (the reference's hot path, causalimpact_lib.py:345-395, is replaced
by these entry points and by nothing else).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np

ABI_VERSION = 3
MAX_BLOCKS = 8
_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_ROOT, "lib", "libcausalimpact_amd.so")

_PARAM_FIELDS = (
    "level_conc", "level_scale", "level_ub", "slope_conc", "slope_scale", "slope_ub",
    "obs_conc", "obs_scale", "obs_ub", "drift_conc", "drift_scale", "drift_ub",
    "nonzero_prob", "init_level_loc", "init_level_scale", "init_slope_scale",
    "init_seasonal_scale", "obs_scale0", "level_scale0", "slope_scale0")


class SeriesParams(C.Structure):
  _fields_ = ([(f, C.c_double) for f in _PARAM_FIELDS] + [("drift_scale0", C.c_double * MAX_BLOCKS)]
              + [("weights_prior_scale", C.c_double)])


class Problem(C.Structure):
  _fields_ = [
      ("abi_version", C.c_int32), ("T", C.c_int32), ("P", C.c_int32), ("has_slope", C.c_int32),
      ("num_blocks", C.c_int32), ("num_seasons", C.c_int32 * MAX_BLOCKS),
      ("num_warmup", C.c_int32), ("num_results", C.c_int32), ("num_chains", C.c_int32),
      ("chain_offset", C.c_int32), ("num_series", C.c_int32), ("seed", C.c_uint32 * 2),
      ("device", C.c_int32), ("flags", C.c_int32), ("series_offset", C.c_int32),
      ("reserved", C.c_int32)]


_OUT_FIELDS = ("observation_noise_scale", "level_scale", "slope_scale", "seasonal_drift_scales",
               "weights", "level", "slope", "seasonal_levels", "posterior_means",
               "posterior_trajectories")


class Outputs(C.Structure):
  _fields_ = [(f, C.c_void_p) for f in _OUT_FIELDS]


class HmcOptions(C.Structure):
  _fields_ = [
      ("num_chains", C.c_int32), ("chain_offset", C.c_int32), ("num_warmup", C.c_int32),
      ("num_results", C.c_int32), ("num_leapfrog", C.c_int32), ("prior", C.c_int32),
      ("target_accept", C.c_double), ("initial_step_size", C.c_double),
      ("horseshoe_scale", C.c_double), ("seed", C.c_uint32 * 2)]


HMC_PRIORS = {"slab": 0, "horseshoe": 1}   # == CI_HMC_PRIOR_*


class NativeError(RuntimeError):
  pass


_lib = None


def load():
  """Loads the HIP library; raises NativeError when it has not been built."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise NativeError(
        f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(there is no CPU fallback for the Gibbs hot path)")
  L = C.CDLL(LIB_PATH)
  L.ci_last_error.restype = C.c_char_p
  L.ci_abi_version.restype = C.c_int
  L.ci_device_count.argtypes = [C.POINTER(C.c_int)]
  L.ci_series_stream_key.argtypes = [C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_uint32)]
  L.ci_series_stream_key.restype = None
  L.ci_device_synchronize.argtypes = [C.c_int]
  L.ci_fit_gibbs.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.POINTER(SeriesParams), C.POINTER(Outputs)]
  L.ci_fit_gibbs_f64.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(SeriesParams), C.POINTER(Outputs)]
  L.ci_fit_gibbs_f64_kernel_ms.argtypes = [C.POINTER(C.c_float)]
  L.ci_session_create.argtypes = [C.POINTER(Problem), C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.POINTER(SeriesParams), C.POINTER(C.c_void_p)]
  L.ci_session_run.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
  L.ci_session_fetch.argtypes = [C.c_void_p, C.POINTER(Outputs)]
  L.ci_session_algorithmic_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
  L.ci_session_destroy.argtypes = [C.c_void_p]
  L.ci_session_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
  L.ci_session_summarize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
  L.ci_summarize_draws.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double,
                                   C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  L.ci_summarize_draws_f64.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_double,
                                   C.c_double, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  L.ci_kalman_loglik.argtypes = [C.POINTER(Problem), C.POINTER(SeriesParams), C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
  L.ci_ll_session_create.argtypes = [C.POINTER(Problem), C.POINTER(SeriesParams), C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_void_p)]
  L.ci_ll_session_create2.argtypes = [C.POINTER(Problem), C.POINTER(SeriesParams), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                      C.POINTER(C.c_void_p)]
  L.ci_ll_session_eval.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
  L.ci_ll_session_draw_latents.argtypes = [C.c_void_p, C.c_int32, C.c_void_p,
                                           C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  L.ci_ll_session_destroy.argtypes = [C.c_void_p]
  L.ci_ll_session_hmc_run.argtypes = [C.c_void_p, C.POINTER(HmcOptions), C.c_void_p,
                                      C.POINTER(C.c_float)]
  L.ci_ll_session_hmc_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(Outputs)]
  L.ci_ll_session_algorithmic_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
  L.ci_pool_trim.argtypes = []
  L.ci_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
  L.ci_host_free.argtypes = [C.c_void_p]
  L.ci_session_run_streamed.argtypes = [C.c_void_p, C.POINTER(Outputs), C.c_int32,
                                        C.POINTER(C.c_float)]
  L.ci_session_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
  L.ci_ll_session_kernel_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
  L.ci_test_rng.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32, C.c_uint32,
                            C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
  L.ci_test_dk_draw.argtypes = [C.POINTER(Problem), C.POINTER(SeriesParams), C.c_void_p, C.c_void_p,
                                C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p,
                                C.c_uint32, C.c_void_p]
  if L.ci_abi_version() != ABI_VERSION:
    raise NativeError(f"ABI mismatch: library {L.ci_abi_version()}, binding {ABI_VERSION}")
  _lib = L
  return L


def exported_symbols() -> Sequence[str]:
  """Every entry point include/causalimpact_amd.h declares."""
  return ("ci_last_error", "ci_abi_version", "ci_device_count", "ci_series_stream_key", "ci_device_synchronize", "ci_pool_trim", "ci_host_alloc",
          "ci_host_free", "ci_fit_gibbs", "ci_fit_gibbs_f64", "ci_fit_gibbs_f64_kernel_ms",
          "ci_session_create", "ci_session_run", "ci_session_run_streamed", "ci_session_fetch",
          "ci_session_algorithmic_bytes", "ci_session_kernel_name", "ci_session_destroy",
          "ci_session_profile", "ci_ll_session_kernel_name",
          "ci_session_summarize", "ci_summarize_draws", "ci_summarize_draws_f64",
          "ci_kalman_loglik", "ci_ll_session_create", "ci_ll_session_create2", "ci_ll_session_eval",
          "ci_ll_session_draw_latents", "ci_ll_session_hmc_run", "ci_ll_session_hmc_fetch",
          "ci_ll_session_algorithmic_bytes", "ci_ll_session_destroy",
          "ci_comm_unique_id", "ci_comm_create", "ci_comm_info", "ci_comm_set_timeout", "ci_comm_barrier",
          "ci_comm_all_reduce", "ci_comm_all_gather", "ci_comm_session_all_gather",
          "ci_comm_ll_session_all_gather", "ci_comm_destroy", "ci_test_rng",
          "ci_test_dk_draw")


def _check(rc: int):
  if rc != 0:
    raise NativeError(load().ci_last_error().decode("utf-8", "replace"))


def series_stream_key(seed, series_id: int):
  """(k0, k1): the Philox key of series `series_id` of a batch fitted with `seed` -- a
  single-series fit (device or oracle) with seed = this key reproduces that series' draws."""
  s = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in seed_pair(seed)])
  k = (C.c_uint32 * 2)()
  load().ci_series_stream_key(s, int(series_id), k)
  return (int(k[0]), int(k[1]))


def device_count() -> int:
  n = C.c_int(0)
  _check(load().ci_device_count(C.byref(n)))
  return n.value


def device_synchronize(device: int = 0):
  """Waits for all work queued on `device` (ci_device_synchronize)."""
  _check(load().ci_device_synchronize(int(device)))


def pool_trim():
  """Returns the device buffers parked by finished sessions to the driver (ci_pool_trim)."""
  _check(load().ci_pool_trim())


def seed_pair(seed):
  """int s -> (0, s); pairs pass through (causalimpact_lib.py:535-539)."""
  if isinstance(seed, (int, np.integer)):
    return (0, int(seed) & 0xFFFFFFFF)
  a, b = seed
  return (int(a) & 0xFFFFFFFF, int(b) & 0xFFFFFFFF)


def make_params(specs: Sequence[Dict]) -> "C.Array":
  """ci_series_params[len(specs)] -- every member is a double, so the table is filled as one
  [n, 29] float64 block (a batch of 512 series: 15k ctypes attribute writes otherwise)."""
  n, nf = len(specs), len(_PARAM_FIELDS)
  buf = np.zeros((n, nf + MAX_BLOCKS + 1), np.float64)
  for i, sp in enumerate(specs):
    buf[i, :nf] = [sp[f] for f in _PARAM_FIELDS]
    d0 = sp.get("drift_scale0", ())
    if len(d0):
      buf[i, nf:nf + len(d0)] = d0
    buf[i, nf + MAX_BLOCKS] = sp.get("weights_prior_scale", 1.0)
  arr = (SeriesParams * n)()
  assert C.sizeof(arr) == buf.nbytes
  C.memmove(arr, buf.ctypes.data, buf.nbytes)
  return arr


FLAG_SEQUENTIAL_SEASONAL = 1   # == CI_FLAG_SEQUENTIAL_SEASONAL
FLAG_SHARED_SERIES_STREAMS = 2  # == CI_FLAG_SHARED_SERIES_STREAMS
FLAG_FOUR_WAVES = 4             # == CI_FLAG_FOUR_WAVES
FLAG_SEASONAL_WORKSPACE = 8     # == CI_FLAG_SEASONAL_WORKSPACE
FLAG_NO_CLUSTER = 16            # == CI_FLAG_NO_CLUSTER
FLAG_TEST_DROP_HELPER = 32      # == CI_FLAG_TEST_DROP_HELPER
FLAG_CLUSTER_SEASONAL = 64      # == CI_FLAG_CLUSTER_SEASONAL


def make_problem(*, T, P, has_slope, num_seasons=(), num_warmup, num_results, num_chains=1,
                 chain_offset=0, num_series=1, seed=(0, 0), device=0, flags=0,
                 series_offset=0) -> Problem:
  pb = Problem()
  pb.abi_version = ABI_VERSION
  pb.T, pb.P, pb.has_slope = int(T), int(P), int(bool(has_slope))
  pb.num_blocks = len(num_seasons)
  for k, n in enumerate(num_seasons):
    pb.num_seasons[k] = int(n)
  pb.num_warmup, pb.num_results = int(num_warmup), int(num_results)
  pb.num_chains, pb.chain_offset, pb.num_series = int(num_chains), int(chain_offset), int(num_series)
  s = seed_pair(seed)
  pb.seed[0], pb.seed[1] = s
  pb.device = int(device)
  pb.flags = int(flags)
  pb.series_offset = int(series_offset)
  return pb


def _stage_inputs(pb: Problem, y, mask, X, season_change):
  B, T, P, K = pb.num_series, pb.T, pb.P, pb.num_blocks
  mask8 = np.ascontiguousarray(np.asarray(mask, dtype=bool).reshape(B, T).astype(np.uint8))
  y32 = np.asarray(y, dtype=np.float32).reshape(B, T)
  y32 = np.ascontiguousarray(np.where(mask8 != 0, np.float32(0), y32))
  X32 = None
  if P > 0:
    X32 = np.ascontiguousarray(np.asarray(X, dtype=np.float32).reshape(B, T, P))
  sc = None
  if K > 0:
    sc = np.ascontiguousarray(np.asarray(season_change, dtype=np.uint8).reshape(K, T))
  return y32, mask8, X32, sc


def summarize_draws(trajectories, scale, shift, observed, flags, ranks, device=0):
  """ci_summarize_draws / ci_summarize_draws_f64: on-device summary of host-resident [draws, T]
  trajectories -- float64 draws (the float64 kernels) are summarised as float64, anything else as
  float32."""
  f64 = np.asarray(trajectories).dtype == np.float64
  tr = np.ascontiguousarray(trajectories, dtype=np.float64 if f64 else np.float32)
  N, T = tr.shape
  obs = np.ascontiguousarray(observed, dtype=np.float64).reshape(T)
  fl = np.ascontiguousarray(flags, dtype=np.uint8).reshape(T)
  rk = np.ascontiguousarray(ranks, dtype=np.int32)
  vo = np.empty((rk.size, T), np.float64)
  co = np.empty((rk.size, T), np.float64)
  pd_ = np.empty((2, N), np.float64)
  do = np.empty((2, rk.size), np.float64)
  fn = load().ci_summarize_draws_f64 if f64 else load().ci_summarize_draws
  _check(fn(int(device), N, T, tr.ctypes.data, float(scale), float(shift),
            obs.ctypes.data, fl.ctypes.data, int(rk.size), rk.ctypes.data,
            vo.ctypes.data, co.ctypes.data, pd_.ctypes.data, do.ctypes.data))
  return dict(value_order=vo, cum_order=co, per_draw=pd_, per_draw_order=do)


class _PinnedBlock:
  """Owner of one ci_host_alloc buffer: returned to the library's pool when the last numpy view
  of it is collected."""

  def __init__(self, nbytes: int):
    self.ptr = C.c_void_p()
    _check(load().ci_host_alloc(C.byref(self.ptr), max(int(nbytes), 1)))
    self.nbytes = int(nbytes)

  def __del__(self):
    try:
      if self.ptr:
        load().ci_host_free(self.ptr)
        self.ptr = C.c_void_p()
    except Exception:  # pylint: disable=broad-except
      pass


def pinned_empty(shape, dtype=np.float32) -> np.ndarray:
  """An uninitialised numpy array in pinned host memory (ci_host_alloc): device-to-host copies
  into it run at PCIe rate and overlap the fit (`Session.run_streamed`)."""
  dtype = np.dtype(dtype)
  n = int(np.prod(shape, dtype=np.int64))
  block = _PinnedBlock(n * dtype.itemsize)
  buf = (C.c_char * max(n * dtype.itemsize, 1)).from_address(block.ptr.value)
  buf._pinned_owner = block   # keeps the allocation alive as long as any view of `buf`
  return np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)


def output_shapes(pb: Problem) -> Dict[str, tuple]:
  """Shapes of the members of ci_outputs for this problem (chain-major, per device)."""
  B, C_, S, T, P, K = pb.num_series, pb.num_chains, pb.num_results, pb.T, pb.P, pb.num_blocks
  return dict(
      observation_noise_scale=(B, C_, S), level_scale=(B, C_, S), slope_scale=(B, C_, S),
      seasonal_drift_scales=(B, C_, S, K), weights=(B, C_, S, P), level=(B, C_, S, T),
      slope=(B, C_, S, T), seasonal_levels=(B, C_, S, T, K), posterior_means=(B, C_, T),
      posterior_trajectories=(B, C_, S, T))


def _alloc_outputs(pb: Problem, want: Optional[Sequence[str]] = None, pinned: bool = False):
  shapes = output_shapes(pb)
  out, arrs = Outputs(), {}
  for name, shp in shapes.items():
    if want is not None and name not in want:
      continue
    a = pinned_empty(shp) if (pinned and int(np.prod(shp)) > 0) else np.zeros(shp, dtype=np.float32)
    arrs[name] = a
    setattr(out, name, a.ctypes.data if a.size else None)
  return out, arrs


def _ptr(a):
  return None if a is None else a.ctypes.data


def fit_gibbs(pb: Problem, y, mask, X, season_change, params, want=None) -> Dict[str, np.ndarray]:
  """One-shot upload -> W+S Gibbs iterations for B*C chains -> download."""
  L = load()
  y32, mask8, X32, sc = _stage_inputs(pb, y, mask, X, season_change)
  out, arrs = _alloc_outputs(pb, want)
  _check(L.ci_fit_gibbs(C.byref(pb), _ptr(y32), _ptr(mask8), _ptr(X32), _ptr(sc), params,
                        C.byref(out)))
  return arrs


def fit_gibbs_f64(pb: Problem, y, mask, X, season_change, params, want=None) -> Dict[str, np.ndarray]:
  """The float64 fit (ci_fit_gibbs_f64): every input and result array float64, any model."""
  L = load()
  B, T, P, K = pb.num_series, pb.T, pb.P, pb.num_blocks
  mask8 = np.ascontiguousarray(np.asarray(mask, dtype=bool).reshape(B, T).astype(np.uint8))
  y64 = np.ascontiguousarray(np.where(mask8 != 0, 0.0, np.asarray(y, np.float64).reshape(B, T)))
  X64 = np.ascontiguousarray(np.asarray(X, np.float64).reshape(B, T, P)) if P > 0 else None
  sc = (np.ascontiguousarray(np.asarray(season_change, dtype=np.uint8).reshape(K, T))
        if K > 0 else None)
  out, arrs = Outputs(), {}
  for name, shp in output_shapes(pb).items():
    if want is not None and name not in want:
      continue
    a = np.zeros(shp, dtype=np.float64)
    arrs[name] = a
    setattr(out, name, a.ctypes.data if a.size else None)
  _check(L.ci_fit_gibbs_f64(C.byref(pb), _ptr(y64), _ptr(mask8), _ptr(X64), _ptr(sc), params,
                            C.byref(out)))
  return arrs


def fit_gibbs_f64_kernel_ms() -> float:
  """Duration of the sampling kernel of this thread's last fit_gibbs_f64 (HIP events)."""
  ms = C.c_float()
  _check(load().ci_fit_gibbs_f64_kernel_ms(C.byref(ms)))
  return float(ms.value)


class Session:
  """Device-resident fit (inputs and outputs live in HBM between run() calls)."""

  def __init__(self, pb: Problem, y, mask, X, season_change, params):
    self._lib = load()
    self.pb = pb
    y32, mask8, X32, sc = _stage_inputs(pb, y, mask, X, season_change)
    self._h = C.c_void_p()
    _check(self._lib.ci_session_create(C.byref(pb), _ptr(y32), _ptr(mask8), _ptr(X32), _ptr(sc),
                                       params, C.byref(self._h)))

  def run(self) -> float:
    """Runs all W+S iterations; returns the Gibbs kernel's duration in ms (HIP events)."""
    ms = C.c_float(0)
    _check(self._lib.ci_session_run(self._h, C.byref(ms)))
    return float(ms.value)

  def fetch(self, want=None) -> Dict[str, np.ndarray]:
    out, arrs = _alloc_outputs(self.pb, want)
    _check(self._lib.ci_session_fetch(self._h, C.byref(out)))
    return arrs

  def run_streamed(self, want=None, chunk_draws: int = 125, pinned: bool = True, into=None):
    """Runs the fit and copies the results to the host WHILE it runs (ci_session_run_streamed).
    Returns (kernel_ms, arrays); the arrays live in pinned memory unless pinned=False.  `into`
    re-uses the (Outputs, arrays) pair of an earlier call instead of allocating."""
    out, arrs = into if into is not None else _alloc_outputs(self.pb, want, pinned=pinned)
    ms = C.c_float(0)
    _check(self._lib.ci_session_run_streamed(self._h, C.byref(out), int(chunk_draws), C.byref(ms)))
    self._last_streamed = (out, arrs)
    return float(ms.value), arrs

  def algorithmic_bytes(self) -> float:
    b = C.c_double(0)
    _check(self._lib.ci_session_algorithmic_bytes(self._h, C.byref(b)))
    return float(b.value)

  def kernel_name(self) -> str:
    buf = C.create_string_buffer(128)
    _check(self._lib.ci_session_kernel_name(self._h, buf, 128))
    return buf.value.decode()

  def profile(self, enable=True):
    """Enables per-phase cycle counters for the next run(); returns the previous run's."""
    cyc = np.zeros(32, np.int64)
    _check(self._lib.ci_session_profile(self._h, int(enable), cyc.ctypes.data))
    return cyc

  def summarize(self, scale, shift, observed, flags, ranks) -> Dict[str, np.ndarray]:
    """On-device order statistics / running effect sums of the pooled predictive draws of every
    series (ci_session_summarize).  scale, shift: scalars or [B]; observed, flags: [T] or [B,T].
    Returns value_order [B,R,T], cum_order [B,R,T], per_draw [B,2,N], per_draw_order [B,2,R]
    (leading axis dropped when the session holds one series)."""
    B, T, N = self.pb.num_series, self.pb.T, self.pb.num_chains * self.pb.num_results
    sc = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, np.float64), (B,)))
    sh = np.ascontiguousarray(np.broadcast_to(np.asarray(shift, np.float64), (B,)))
    obs = np.ascontiguousarray(np.broadcast_to(np.asarray(observed, np.float64), (B, T)))
    fl = np.ascontiguousarray(np.broadcast_to(np.asarray(flags, np.uint8), (B, T)))
    rk = np.ascontiguousarray(ranks, dtype=np.int32)
    # big batches: the result blocks in pinned host memory from the library's pool (512 series: 24 MB
    # -- pageable destinations cost the copies a staging pass and the arrays their first-touch faults)
    big = B * max(rk.size * T, 2 * N) * 8 > (1 << 20)
    empty = (lambda shp: pinned_empty(shp, np.float64)) if big else (lambda shp: np.empty(shp, np.float64))
    vo = empty((B, rk.size, T))
    co = empty((B, rk.size, T))
    pd_ = empty((B, 2, N))
    do = np.empty((B, 2, rk.size), np.float64)
    _check(self._lib.ci_session_summarize(self._h, sc.ctypes.data, sh.ctypes.data, obs.ctypes.data,
                                          fl.ctypes.data, int(rk.size), rk.ctypes.data,
                                          vo.ctypes.data, co.ctypes.data, pd_.ctypes.data,
                                          do.ctypes.data))
    if B == 1:
      vo, co, pd_, do = vo[0], co[0], pd_[0], do[0]
    return dict(value_order=vo, cum_order=co, per_draw=pd_, per_draw_order=do)

  def close(self):
    if self._h:
      self._lib.ci_session_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


def kalman_loglik(pb: Problem, params, y, mask, X, theta) -> np.ndarray:
  """Log-likelihood l(theta_e) for every row theta_e = (sigma_obs, sigma_level, sigma_slope,
  weights...) of `theta` (row H of SURVEY.md section 8)."""
  L = load()
  T, P = pb.T, pb.P
  m8 = np.ascontiguousarray(np.asarray(mask, bool).astype(np.uint8))
  y32 = np.ascontiguousarray(np.where(m8 != 0, np.float32(0), np.asarray(y, np.float32)))
  X32 = np.ascontiguousarray(np.asarray(X, np.float32).reshape(T, P)) if P > 0 else None
  th = np.ascontiguousarray(np.asarray(theta, np.float64).reshape(-1, 3 + P))
  out = np.zeros(th.shape[0], np.float64)
  _check(L.ci_kalman_loglik(C.byref(pb), params, y32.ctypes.data, m8.ctypes.data, _ptr(X32),
                            th.shape[0], th.ctypes.data, out.ctypes.data))
  return out


class LogLikSession:
  """Device-resident log-likelihood / score evaluator and latent-path drawer for one series."""

  def __init__(self, pb: Problem, params, y, mask, X, max_evals: int, season_change=None):
    """season_change [K, T] uint8 for models with seasonal blocks (pb.num_blocks = K): those, and
    series longer than 4096 steps, run on the sequential route (csrc/ci_score_seq.h); parameter
    rows are then (sigma_obs, sigma_level, sigma_slope, sigma_drift[K], weights[P])."""
    self._lib = load()
    self.T, self.P, self.D, self.K = pb.T, pb.P, 2 if pb.has_slope else 1, pb.num_blocks
    self.num_seasons = [int(pb.num_seasons[k]) for k in range(pb.num_blocks)]
    self.max_evals = int(max_evals)
    m8 = np.ascontiguousarray(np.asarray(mask, bool).astype(np.uint8))
    y32 = np.ascontiguousarray(np.where(m8 != 0, np.float32(0), np.asarray(y, np.float32)))
    X32 = (np.ascontiguousarray(np.asarray(X, np.float32).reshape(self.T, self.P))
           if self.P > 0 else None)
    sc = None
    if self.K > 0:
      sc = np.ascontiguousarray(np.asarray(season_change, dtype=np.uint8).reshape(self.K, self.T))
    self._h = C.c_void_p()
    _check(self._lib.ci_ll_session_create2(C.byref(pb), params, y32.ctypes.data, m8.ctypes.data,
                                           _ptr(X32), _ptr(sc), self.max_evals, C.byref(self._h)))

  def evaluate(self, theta, want_grad=True):
    th = np.ascontiguousarray(np.asarray(theta, np.float64).reshape(-1, 3 + self.K + self.P))
    ll = np.zeros(th.shape[0], np.float64)
    grad = np.zeros_like(th) if want_grad else None
    _check(self._lib.ci_ll_session_eval(self._h, th.shape[0], th.ctypes.data, ll.ctypes.data,
                                        _ptr(grad)))
    return ll, grad

  def draw_latents(self, theta, seed, rng_chain=0, iter0=0):
    th = np.ascontiguousarray(np.asarray(theta, np.float64).reshape(-1, 3 + self.P))
    E = th.shape[0]
    out = {k: np.zeros((E, self.T), np.float32) for k in ("level", "slope", "loc", "traj")}
    s = (C.c_uint32 * 2)(*seed_pair(seed))
    _check(self._lib.ci_ll_session_draw_latents(
        self._h, E, th.ctypes.data, s, int(rng_chain), int(iter0), out["level"].ctypes.data,
        out["slope"].ctypes.data, out["loc"].ctypes.data, out["traj"].ctypes.data))
    return out

  def hmc_run(self, *, num_chains, num_warmup, num_results, num_leapfrog=15, target_accept=0.75,
              initial_step_size=0.05, seed=(0, 0), chain_offset=0, init_theta=None, prior="slab",
              horseshoe_scale=0.1):
    """The whole HMC fit on the device (ci_ll_session_hmc_run): chain, latent paths and
    predictive trajectories stay in HBM.  Returns (hmc_kernel_ms, latents_ms)."""
    o = HmcOptions()
    o.num_chains, o.chain_offset = int(num_chains), int(chain_offset)
    o.num_warmup, o.num_results, o.num_leapfrog = int(num_warmup), int(num_results), int(num_leapfrog)
    if prior not in HMC_PRIORS:
      raise ValueError(f"prior must be one of {sorted(HMC_PRIORS)}, got {prior!r}")
    o.prior = HMC_PRIORS[prior]
    o.target_accept, o.initial_step_size = float(target_accept), float(initial_step_size)
    o.horseshoe_scale = float(horseshoe_scale)
    o.seed[0], o.seed[1] = seed_pair(seed)
    init = None if init_theta is None else np.ascontiguousarray(init_theta, dtype=np.float64)
    if init is not None:
      dim = (3 * self.P + 2 if prior == "horseshoe" else self.P) + self.D + 1 + self.K
      if init.shape != (int(num_chains), dim):
        raise ValueError(f"init_theta must be [{int(num_chains)}, {dim}], got {init.shape}")
    ms = (C.c_float * 2)()
    _check(self._lib.ci_ll_session_hmc_run(self._h, C.byref(o), _ptr(init), ms))
    self._hmc_shape = (int(num_chains), int(num_results))
    return float(ms[0]), float(ms[1])

  def hmc_fetch(self, want=None):
    """Host copies of the finished fit: draws [C, S, 3+K+P] float64 (sigma_obs, sigma_level,
    sigma_slope, sigma_drift[K], weights), accept_rate, step_size [C] and the float32 sample
    container of `fit_gibbs` (leading series axis of 1)."""
    Cn, S = self._hmc_shape
    pb = make_problem(T=self.T, P=self.P, has_slope=self.D == 2, num_seasons=self.num_seasons,
                      num_warmup=0, num_results=S, num_chains=Cn)
    if want is None:
      want = [f for f in _OUT_FIELDS
              if self.K > 0 or f not in ("seasonal_drift_scales", "seasonal_levels")]
    out, arrs = _alloc_outputs(pb, want)
    draws = np.zeros((Cn, S, 3 + self.K + self.P), np.float64)
    acc = np.zeros(Cn, np.float64)
    eps = np.zeros(Cn, np.float64)
    _check(self._lib.ci_ll_session_hmc_fetch(self._h, draws.ctypes.data, acc.ctypes.data,
                                             eps.ctypes.data, C.byref(out)))
    return draws, acc, eps, arrs

  def hmc(self, **kw):
    """hmc_run + the parameter draws only: draws [C, S, 3+P], accept_rate, step_size [C]."""
    self.hmc_run(**kw)
    Cn, S = self._hmc_shape
    draws = np.zeros((Cn, S, 3 + self.K + self.P), np.float64)
    acc = np.zeros(Cn, np.float64)
    eps = np.zeros(Cn, np.float64)
    _check(self._lib.ci_ll_session_hmc_fetch(self._h, draws.ctypes.data, acc.ctypes.data,
                                             eps.ctypes.data, None))
    return draws, acc, eps

  def algorithmic_bytes(self) -> float:
    b = C.c_double(0)
    _check(self._lib.ci_ll_session_algorithmic_bytes(self._h, C.byref(b)))
    return float(b.value)

  def kernel_name(self) -> str:
    buf = C.create_string_buffer(128)
    _check(self._lib.ci_ll_session_kernel_name(self._h, buf, 128))
    return buf.value.decode()

  def close(self):
    if self._h:
      self._lib.ci_ll_session_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


def test_rng(seed, chain, it, site, sub, n, alpha, device=0):
  L = load()
  s = (C.c_uint32 * 2)(*seed_pair(seed))
  u = np.zeros(n, np.float32)
  z = np.zeros(2 * n, np.float32)
  g = np.zeros(1, np.float64)
  _check(L.ci_test_rng(device, s, chain, it, site, sub, n, u.ctypes.data, z.ctypes.data,
                       float(alpha), g.ctypes.data))
  return u, z[:n], z[n:], float(g[0])


def test_dk_draw(pb: Problem, params, resid, mask, obs_scale, level_scale, slope_scale=0.0, it=0):
  L = load()
  T, D = pb.T, 2 if pb.has_slope else 1
  r32 = np.ascontiguousarray(np.asarray(resid, np.float32))
  m8 = np.ascontiguousarray(np.asarray(mask, bool).astype(np.uint8))
  out = np.zeros((T, D), np.float32)
  _check(L.ci_test_dk_draw(C.byref(pb), params, r32.ctypes.data, m8.ctypes.data, None,
                           float(obs_scale), float(level_scale), float(slope_scale), None, int(it),
                           out.ctypes.data))
  return out
