"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It restates, independently of the product package, how the
reference turns data into the sampler's inputs:

  * priors            /root/reference/causalimpact/causalimpact_lib.py:424-489
  * initial state     /root/reference/causalimpact/causalimpact_lib.py:563-581
  * masked extension  /root/reference/causalimpact/causalimpact_lib.py:548-562
  * season calendar   tfp.sts.Seasonal `is_last_day_of_season` (un-vendored TFP;
                      call site causalimpact_lib.py:477-482)

and drives oracle/ci_oracle.c (float64, single thread).  Parity status: see
ci_oracle.h ("per-draw parity with TFP unpinned").
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libci_oracle.so")
_MAX_BLOCKS = 8

SITES = dict(PERM=1, FLIP=2, OBSVAR=3, WEIGHTS=4, PRIOR_INIT=5, PRIOR_LEVEL=6,
             PRIOR_SLOPE=7, PRIOR_OBS=8, PRIOR_SEAS=9, LEVEL_SCALE=10,
             SLOPE_SCALE=11, DRIFT_SCALE=12, OBS_SCALE=13, PRED=14)


def build(force: bool = False) -> str:
  """Compiles oracle/ci_oracle.c with gcc (a few hundred ms)."""
  src = os.path.join(_HERE, "ci_oracle.c")
  hdr = os.path.join(_HERE, "ci_oracle.h")
  stale = (not os.path.exists(_LIB_PATH) or
           os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
  if force or stale:
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
  return _LIB_PATH


class _Problem(C.Structure):
  _fields_ = [
      ("T", C.c_int32), ("P", C.c_int32), ("has_slope", C.c_int32),
      ("num_blocks", C.c_int32), ("num_seasons", C.c_int32 * _MAX_BLOCKS),
      ("num_warmup", C.c_int32), ("num_results", C.c_int32),
      ("seed", C.c_uint32 * 2), ("chain", C.c_int32), ("flags", C.c_int32),
      ("y", C.c_void_p), ("mask", C.c_void_p), ("X", C.c_void_p),
      ("season_change", C.c_void_p),
      ("level_conc", C.c_double), ("level_scale", C.c_double), ("level_ub", C.c_double),
      ("slope_conc", C.c_double), ("slope_scale", C.c_double), ("slope_ub", C.c_double),
      ("obs_conc", C.c_double), ("obs_scale", C.c_double), ("obs_ub", C.c_double),
      ("drift_conc", C.c_double), ("drift_scale", C.c_double), ("drift_ub", C.c_double),
      ("nonzero_prob", C.c_double),
      ("init_level_loc", C.c_double), ("init_level_scale", C.c_double),
      ("init_slope_scale", C.c_double), ("init_seasonal_scale", C.c_double),
      ("obs_scale0", C.c_double), ("level_scale0", C.c_double), ("slope_scale0", C.c_double),
      ("drift_scale0", C.c_double * _MAX_BLOCKS),
      ("weights0", C.c_void_p), ("latents0", C.c_void_p),
      ("weights_prior_scale", C.c_double),
  ]


FLAG_NO_WEIGHT_ADJUSTMENT = 1   # == CI_ORACLE_FLAG_NO_WEIGHT_ADJUSTMENT (test-only)
FLAG_EXACT_MARGINAL = 2         # == CI_ORACLE_FLAG_EXACT_MARGINAL (test-only)


class _Outputs(C.Structure):
  _fields_ = [(n, C.c_void_p) for n in (
      "obs_scale", "level_scale", "slope_scale", "drift_scales", "weights", "level",
      "slope", "seasonal", "pred_mean", "trajectories", "nonzeros")]


class _SSM(C.Structure):
  _fields_ = [
      ("T", C.c_int32), ("d", C.c_int32), ("has_slope", C.c_int32), ("num_blocks", C.c_int32),
      ("num_seasons", C.c_int32 * _MAX_BLOCKS),
      ("mask", C.c_void_p), ("season_change", C.c_void_p),
      ("obs_scale", C.c_double), ("level_scale", C.c_double), ("slope_scale", C.c_double),
      ("drift_scale", C.c_double * _MAX_BLOCKS),
      ("init_level_loc", C.c_double), ("init_level_scale", C.c_double),
      ("init_slope_scale", C.c_double), ("init_seasonal_scale", C.c_double),
  ]


_lib = None


def lib():
  global _lib
  if _lib is None:
    build()
    L = C.CDLL(_LIB_PATH)
    L.ci_oracle_fit_gibbs.restype = C.c_int
    L.ci_oracle_fit_gibbs.argtypes = [C.POINTER(_Problem), C.POINTER(_Outputs)]
    L.ci_oracle_philox.restype = None
    L.ci_oracle_uniform.restype = C.c_double
    L.ci_oracle_uniform.argtypes = [C.POINTER(C.c_uint32)] + [C.c_uint32] * 5
    L.ci_oracle_normal.restype = C.c_double
    L.ci_oracle_normal.argtypes = [C.POINTER(C.c_uint32)] + [C.c_uint32] * 5
    L.ci_oracle_gamma.restype = C.c_double
    L.ci_oracle_gamma.argtypes = [C.c_double, C.POINTER(C.c_uint32)] + [C.c_uint32] * 4
    L.ci_oracle_state_dim.restype = C.c_int
    L.ci_oracle_kalman_loglik.restype = C.c_double
    L.ci_oracle_kalman_loglik.argtypes = [C.POINTER(_SSM), C.c_void_p]
    L.ci_oracle_smoothed_mean.restype = None
    L.ci_oracle_smoothed_mean.argtypes = [C.POINTER(_SSM), C.c_void_p, C.c_void_p]
    L.ci_oracle_dk_draw.restype = None
    L.ci_oracle_dk_draw.argtypes = [C.POINTER(_SSM), C.c_void_p, C.POINTER(C.c_uint32),
                                    C.c_uint32, C.c_uint32, C.c_void_p]
    L.ci_oracle_spike_slab_logp.restype = C.c_double
    L.ci_oracle_spike_slab_logp.argtypes = [C.c_int] + [C.c_void_p] * 3 + [C.c_double, C.c_void_p,
                                                                          C.c_double, C.c_double,
                                                                          C.c_double]
    _lib = L
  return _lib


# ----------------------------------------------------------------------------
# calendar / model specification (independent restatement of the reference)
# ----------------------------------------------------------------------------
def season_change_flags(T: int, num_seasons: int, num_steps_per_season) -> np.ndarray:
  """flags[t] = 1 iff step t is the last step of a season (t -> t+1 rotates).

  tfp.sts.Seasonal: changepoints = cumsum(ravel(num_steps_per_season)) - 1 within
  a cycle of sum(num_steps_per_season) steps; an int is tiled over the seasons
  (reference options: causalimpact_lib.py:162-180).
  """
  steps = np.asarray(num_steps_per_season, dtype=np.int64)
  if steps.ndim == 0:
    steps = np.full([num_seasons], int(steps), dtype=np.int64)
  if steps.shape[-1] != num_seasons:
    raise ValueError("num_steps_per_season must have num_seasons entries per cycle")
  flat = steps.ravel()
  cycle = int(flat.sum())
  change = np.zeros(cycle, dtype=np.uint8)
  change[np.cumsum(flat) - 1] = 1
  t = np.arange(T)
  return change[t % cycle].astype(np.uint8)


def default_spec(y, mask, X, *, prior_level_sd=0.01, seasons=(), has_slope=False,
                 outcome_sd: Optional[float] = None, prior_slope_sd: Optional[float] = None):
  """Priors + initial state exactly as the reference fixes them.

  y: [T] outcome (standardised), NaN/ignored where mask; mask: [T] bool;
  X: [T,P] or None; seasons: sequence of (num_seasons, num_steps_per_season).
  """
  y = np.asarray(y, dtype=np.float64)
  mask = np.asarray(mask, dtype=bool)
  T = y.shape[0]
  P = 0 if X is None else int(np.asarray(X).shape[1])
  if outcome_sd is None:
    # causalimpact_lib.py:563-564 -- nanstd of the pre-period (= observed) outcome.
    outcome_sd = float(np.nanstd(np.where(mask, np.nan, y), ddof=1))
  sd = float(outcome_sd)
  sigma0 = prior_level_sd * sd                       # :572
  spec = dict(
      T=T, P=P, has_slope=int(has_slope),
      num_seasons=[int(s[0]) for s in seasons],
      season_change=[season_change_flags(T, int(s[0]), s[1]) for s in seasons],
      outcome_sd=sd,
      level_conc=16.0, level_scale=16.0 * sigma0 * sigma0, level_ub=sd,     # :424-432
      obs_conc=25.0 if P > 0 else 0.005,                                    # :434-441
      obs_scale=(5.0 if P > 0 else 0.005) * sd * sd,
      obs_ub=1.2 * sd,                                                      # :442-443
      drift_conc=0.005, drift_scale=5e-7 * sd * sd, drift_ub=sd,           # :472-474
      nonzero_prob=min(1.0, 3.0 / P) if P > 0 else 1.0,                      # :449-450
      init_level_loc=float(y[~mask][0]) if mask[0] else float(y[0]),        # :467-469
      init_level_scale=sd, init_seasonal_scale=sd,                          # :469, :489
      obs_scale0=(math.sqrt(1.0 - 0.8) * sd) if P > 0 else sd,              # :566-571
      level_scale0=sigma0,                                                  # :572
      drift_scale0=[0.01 * sd] * len(seasons),                              # :573-574
  )
  # LocalLinearTrend is not built by the reference's default model (":496
  # slope_variance_prior=None"); BASELINE cfg2 asks for it, so the slope block
  # mirrors the level block's prior family (documented extension).
  s0 = (prior_slope_sd if prior_slope_sd is not None else prior_level_sd) * sd
  spec.update(slope_conc=16.0, slope_scale=16.0 * s0 * s0, slope_ub=sd,
              init_slope_scale=sd, slope_scale0=s0 if has_slope else 0.0)
  return spec


def _u32pair(seed) -> "C.Array":
  s = (C.c_uint32 * 2)()
  if isinstance(seed, (int, np.integer)):
    seed = (0, int(seed))   # causalimpact_lib.py:535-539
  s[0], s[1] = int(seed[0]) & 0xFFFFFFFF, int(seed[1]) & 0xFFFFFFFF
  return s


def fit_gibbs(y, mask, X, spec, *, num_results, num_warmup, seed, chain=0,
              want=("obs_scale", "level_scale", "slope_scale", "drift_scales", "weights", "level",
                    "slope", "seasonal", "pred_mean", "trajectories", "nonzeros"),
              weights0=None, latents0=None, flags=0):
  """Runs the float64 oracle for one chain; returns a dict of numpy arrays.  weights0 [P] /
  latents0 [T, d] start the chain somewhere else than the reference's zeros (tests only)."""
  L = lib()
  pb, keep = _make_problem(y, mask, X, spec, num_results=num_results, num_warmup=num_warmup,
                           seed=seed, chain=chain, weights0=weights0, latents0=latents0, flags=flags)
  T, P, K = spec["T"], spec["P"], len(spec["num_seasons"])
  S = int(num_results)
  shapes = dict(obs_scale=(S,), level_scale=(S,), slope_scale=(S,), drift_scales=(S, K),
                weights=(S, P), level=(S, T), slope=(S, T), seasonal=(S, T, K),
                pred_mean=(T,), trajectories=(S, T), nonzeros=(S, P))
  res, out = {}, _Outputs()
  for name, shp in shapes.items():
    if name in want:
      arr = np.zeros(shp, dtype=np.int32 if name == "nonzeros" else np.float64)
      res[name] = arr
      setattr(out, name, arr.ctypes.data if arr.size else None)
  rc = L.ci_oracle_fit_gibbs(C.byref(pb), C.byref(out))
  del keep
  if rc != 0:
    raise RuntimeError(f"ci_oracle_fit_gibbs failed rc={rc}")
  return res


def _make_problem(y, mask, X, spec, *, num_results, num_warmup, seed, chain=0, weights0=None,
                  latents0=None, flags=0):
  """(_Problem, the arrays it points into)."""
  T, P = spec["T"], spec["P"]
  K = len(spec["num_seasons"])
  y64 = np.ascontiguousarray(np.where(np.asarray(mask, bool), 0.0, np.asarray(y, np.float64)))
  m8 = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8))
  X64 = np.ascontiguousarray(np.asarray(X, np.float64)) if P > 0 else np.zeros((T, 0))
  sc = (np.ascontiguousarray(np.stack(spec["season_change"]).astype(np.uint8))
        if K > 0 else np.zeros((0, T), np.uint8))
  pb = _Problem()
  pb.T, pb.P, pb.has_slope, pb.num_blocks = T, P, spec["has_slope"], K
  for k in range(K):
    pb.num_seasons[k] = spec["num_seasons"][k]
    pb.drift_scale0[k] = spec["drift_scale0"][k]
  pb.num_warmup, pb.num_results = int(num_warmup), int(num_results)
  s = _u32pair(seed)
  pb.seed[0], pb.seed[1] = s[0], s[1]
  pb.chain = int(chain)
  pb.flags = int(flags)
  w0 = None if weights0 is None else np.ascontiguousarray(weights0, dtype=np.float64)
  l0 = None if latents0 is None else np.ascontiguousarray(latents0, dtype=np.float64)
  pb.weights0 = None if w0 is None else w0.ctypes.data
  pb.latents0 = None if l0 is None else l0.ctypes.data
  pb.y = y64.ctypes.data
  pb.mask = m8.ctypes.data
  pb.X = X64.ctypes.data if P > 0 else None
  pb.season_change = sc.ctypes.data if K > 0 else None
  for f in ("level_conc", "level_scale", "level_ub", "slope_conc", "slope_scale", "slope_ub",
            "obs_conc", "obs_scale", "obs_ub", "drift_conc", "drift_scale", "drift_ub",
            "nonzero_prob", "init_level_loc", "init_level_scale", "init_slope_scale",
            "init_seasonal_scale", "obs_scale0", "level_scale0", "slope_scale0"):
    setattr(pb, f, float(spec[f]))
  pb.weights_prior_scale = float(spec.get("weights_prior_scale", 1.0))
  return pb, (y64, m8, X64, sc, w0, l0)


# ----------------------------------------------------------------------------
# cpu_baseline leg of bench.py: the build of BASELINE.md section 2 (-O3 -march=native, OpenMP over
# chains), compiled on the host that times it
# ----------------------------------------------------------------------------
_native = None


def native_lib():
  """(CDLL, compiler + flags) of `make native`, built into a fresh temporary directory ON THIS
  HOST (a -march=native binary must not travel between machines)."""
  global _native
  if _native is None:
    import tempfile  # pylint: disable=import-outside-toplevel
    out = tempfile.mkdtemp(prefix="ci_oracle_native_")
    subprocess.check_call(["make", "-s", "-C", _HERE, "native", f"OUT={out}"])
    flags = subprocess.check_output(["make", "-s", "-C", _HERE, "print-native-flags"], text=True).strip()
    L = C.CDLL(os.path.join(out, "libci_oracle_native.so"))
    L.ci_oracle_fit_gibbs_chains.restype = C.c_int
    L.ci_oracle_fit_gibbs_chains.argtypes = [C.POINTER(_Problem), C.c_int, C.c_int]
    L.ci_oracle_max_threads.restype = C.c_int
    _native = (L, flags)
  return _native


def fit_gibbs_chains_native(y, mask, X, spec, *, num_results, num_warmup, seed, first_chain,
                            n_chains, threads=None) -> int:
  """n_chains whole Gibbs fits (every output array produced and dropped) on the native build,
  OpenMP-parallel over chains on `threads` cores (default: all).  Returns the threads used."""
  L, _ = native_lib()
  if threads:
    L.ci_oracle_set_threads(int(threads))
  pb, keep = _make_problem(y, mask, X, spec, num_results=num_results, num_warmup=num_warmup, seed=seed)
  failed = L.ci_oracle_fit_gibbs_chains(C.byref(pb), int(first_chain), int(n_chains))
  del keep
  if failed:
    raise RuntimeError(f"{failed} oracle chains failed")
  return int(L.ci_oracle_max_threads())


def make_ssm(spec, mask, *, obs_scale, level_scale, slope_scale=0.0, drift_scale=()):
  K = len(spec["num_seasons"])
  T = spec["T"]
  m = _SSM()
  m.T, m.has_slope, m.num_blocks = T, spec["has_slope"], K
  for k in range(K):
    m.num_seasons[k] = spec["num_seasons"][k]
    m.drift_scale[k] = float(drift_scale[k])
  m.d = lib().ci_oracle_state_dim(m.has_slope, K, m.num_seasons)
  keep = dict(mask=np.ascontiguousarray(np.asarray(mask, np.uint8)),
              sc=(np.ascontiguousarray(np.stack(spec["season_change"]).astype(np.uint8))
                  if K > 0 else np.zeros((0, T), np.uint8)))
  m.mask = keep["mask"].ctypes.data
  m.season_change = keep["sc"].ctypes.data if K > 0 else None
  m.obs_scale, m.level_scale, m.slope_scale = float(obs_scale), float(level_scale), float(slope_scale)
  for f in ("init_level_loc", "init_level_scale", "init_slope_scale", "init_seasonal_scale"):
    setattr(m, f, float(spec[f]))
  m._keep = keep  # keep the buffers alive
  return m


def kalman_loglik(ssm, data) -> float:
  d64 = np.ascontiguousarray(np.asarray(data, np.float64))
  return float(lib().ci_oracle_kalman_loglik(C.byref(ssm), d64.ctypes.data))


def smoothed_mean(ssm, data) -> np.ndarray:
  d64 = np.ascontiguousarray(np.asarray(data, np.float64))
  out = np.zeros((ssm.T, ssm.d))
  lib().ci_oracle_smoothed_mean(C.byref(ssm), d64.ctypes.data, out.ctypes.data)
  return out


def dk_draw(ssm, data, seed, chain=0, it=0) -> np.ndarray:
  d64 = np.ascontiguousarray(np.asarray(data, np.float64))
  out = np.zeros((ssm.T, ssm.d))
  lib().ci_oracle_dk_draw(C.byref(ssm), d64.ctypes.data, _u32pair(seed), int(chain), int(it),
                          out.ctypes.data)
  return out


def philox(ctr: Sequence[int], key: Sequence[int]) -> np.ndarray:
  c = (C.c_uint32 * 4)(*[int(v) & 0xFFFFFFFF for v in ctr])
  k = (C.c_uint32 * 2)(*[int(v) & 0xFFFFFFFF for v in key])
  o = (C.c_uint32 * 4)()
  lib().ci_oracle_philox(c, k, o)
  return np.array(list(o), dtype=np.uint32)


def uniform(seed, chain, it, site, sub, idx) -> float:
  return float(lib().ci_oracle_uniform(_u32pair(seed), chain, it, site, sub, idx))


def normal(seed, chain, it, site, sub, idx) -> float:
  return float(lib().ci_oracle_normal(_u32pair(seed), chain, it, site, sub, idx))


def gamma(alpha, seed, chain, it, site, sub=0) -> float:
  return float(lib().ci_oracle_gamma(float(alpha), _u32pair(seed), chain, it, site, sub))


def spike_slab_logp(xtx, prior_prec, xty, yty, nonzeros, nonzero_prob, post_conc, prior_scale):
  P = len(xty)
  a = np.ascontiguousarray(np.asarray(xtx, np.float64))
  b = np.ascontiguousarray(np.asarray(prior_prec, np.float64))
  c = np.ascontiguousarray(np.asarray(xty, np.float64))
  nz = np.ascontiguousarray(np.asarray(nonzeros, np.uint8))
  return float(lib().ci_oracle_spike_slab_logp(P, a.ctypes.data, b.ctypes.data, c.ctypes.data,
                                               float(yty), nz.ctypes.data, float(nonzero_prob),
                                               float(post_conc), float(prior_scale)))


# ----------------------------------------------------------------------------
# row H (extension): score of the log-likelihood, HMC sampler of csrc/ci_hmc.h
# ----------------------------------------------------------------------------
SITES.update(HMC_MOMENTUM=15, HMC_ACCEPT=16, HMC_INIT=17)


class _HmcProblem(C.Structure):
  _fields_ = [
      ("T", C.c_int32), ("P", C.c_int32), ("has_slope", C.c_int32), ("num_blocks", C.c_int32),
      ("num_seasons", C.c_int32 * _MAX_BLOCKS),
      ("num_warmup", C.c_int32), ("num_results", C.c_int32), ("num_leapfrog", C.c_int32),
      ("prior_mode", C.c_int32), ("seed", C.c_uint32 * 2), ("chain", C.c_int32),
      ("reserved", C.c_int32),
      ("y", C.c_void_p), ("mask", C.c_void_p), ("X", C.c_void_p), ("season_change", C.c_void_p),
      ("omega", C.c_void_p), ("init", C.c_void_p),
      ("ig_a", C.c_double * (3 + _MAX_BLOCKS)), ("ig_b", C.c_double * (3 + _MAX_BLOCKS)),
      ("init_log", C.c_double * (3 + _MAX_BLOCKS)),
      ("hs_scale0", C.c_double), ("target_accept", C.c_double), ("eps0", C.c_double),
      ("init_level_loc", C.c_double), ("init_level_scale", C.c_double),
      ("init_slope_scale", C.c_double), ("init_seasonal_scale", C.c_double),
  ]


def loglik_score(ssm, data):
  """(loglik, e[T] = -dl/d data, dl/d(sigma_obs, sigma_level, sigma_slope, sigma_drift[K]))."""
  L = lib()
  L.ci_oracle_loglik_score.restype = C.c_double
  L.ci_oracle_loglik_score.argtypes = [C.POINTER(_SSM), C.c_void_p, C.c_void_p, C.c_void_p]
  d64 = np.ascontiguousarray(np.asarray(data, np.float64))
  e = np.zeros(ssm.T)
  gs = np.zeros(3 + _MAX_BLOCKS)
  ll = float(L.ci_oracle_loglik_score(C.byref(ssm), d64.ctypes.data, e.ctypes.data, gs.ctypes.data))
  return ll, e, gs[:3 + ssm.num_blocks]


def slab_precision(X) -> np.ndarray:
  """Omega = 0.01 (X'X/2 + diag(X'X)/2) / T over all rows (causalimpact_lib.py:451-453)."""
  X64 = np.asarray(X, np.float64)
  xtx = X64.T @ X64
  return 0.01 * (0.5 * xtx + 0.5 * np.diag(np.diag(xtx))) / X64.shape[0]


def hmc_windows(W: int):
  L = lib()
  v = [C.c_int(0) for _ in range(4)]
  L.ci_oracle_hmc_windows.restype = None
  L.ci_oracle_hmc_windows(C.c_int(int(W)), *[C.byref(x) for x in v])
  return tuple(x.value for x in v)       # slow_begin, slow_end, first_end, base


def _hmc_problem(y, mask, X, spec, *, num_results, num_warmup, num_leapfrog, prior, seed, chain,
                 target_accept, initial_step_size, horseshoe_scale, init):
  T, P = spec["T"], spec["P"]
  K = len(spec["num_seasons"])
  keep = dict(
      y=np.ascontiguousarray(np.where(np.asarray(mask, bool), 0.0, np.asarray(y, np.float64))),
      m=np.ascontiguousarray(np.asarray(mask, dtype=np.uint8)),
      X=np.ascontiguousarray(np.asarray(X, np.float64)) if P > 0 else np.zeros((T, 0)),
      sc=(np.ascontiguousarray(np.stack(spec["season_change"]).astype(np.uint8))
          if K > 0 else np.zeros((0, T), np.uint8)))
  keep["omega"] = (np.ascontiguousarray(slab_precision(keep["X"]) * float(spec.get("weights_prior_scale", 1.0)))
                   if P > 0 else np.zeros((0, 0)))
  pb = _HmcProblem()
  pb.T, pb.P, pb.has_slope, pb.num_blocks = T, P, int(spec["has_slope"]), K
  for k in range(K):
    pb.num_seasons[k] = spec["num_seasons"][k]
  pb.num_warmup, pb.num_results, pb.num_leapfrog = int(num_warmup), int(num_results), int(num_leapfrog)
  pb.prior_mode = {"slab": 0, "horseshoe": 1}[prior]
  s = _u32pair(seed)
  pb.seed[0], pb.seed[1] = s[0], s[1]
  pb.chain = int(chain)
  pb.y, pb.mask = keep["y"].ctypes.data, keep["m"].ctypes.data
  pb.X = keep["X"].ctypes.data if P > 0 else None
  pb.season_change = keep["sc"].ctypes.data if K > 0 else None
  pb.omega = keep["omega"].ctypes.data if P > 0 else None
  if init is not None:
    keep["init"] = np.ascontiguousarray(init, dtype=np.float64)
    pb.init = keep["init"].ctypes.data
  igs = [(spec["obs_conc"], spec["obs_scale"], spec["obs_scale0"]),
         (spec["level_conc"], spec["level_scale"], max(spec["level_scale0"], 1e-4))]
  if spec["has_slope"]:
    igs.append((spec["slope_conc"], spec["slope_scale"], max(spec["slope_scale0"], 1e-4)))
  for k in range(K):
    igs.append((spec["drift_conc"], spec["drift_scale"], max(spec["drift_scale0"][k], 1e-4)))
  for i, (a, b, s0) in enumerate(igs):
    pb.ig_a[i], pb.ig_b[i], pb.init_log[i] = float(a), float(b), math.log(s0)
  pb.hs_scale0 = float(horseshoe_scale)
  pb.target_accept, pb.eps0 = float(target_accept), float(initial_step_size)
  for f in ("init_level_loc", "init_level_scale", "init_slope_scale", "init_seasonal_scale"):
    setattr(pb, f, float(spec[f]))
  pb._keep = keep
  return pb


def hmc_logp(y, mask, X, spec, theta, *, prior="slab", horseshoe_scale=0.1):
  """log posterior (up to a constant) and gradient at the unconstrained point theta."""
  L = lib()
  L.ci_oracle_hmc_logp.restype = C.c_double
  L.ci_oracle_hmc_logp.argtypes = [C.POINTER(_HmcProblem), C.c_void_p, C.c_void_p]
  L.ci_oracle_hmc_dim.argtypes = [C.POINTER(_HmcProblem)]
  pb = _hmc_problem(y, mask, X, spec, num_results=1, num_warmup=0, num_leapfrog=1, prior=prior,
                    seed=0, chain=0, target_accept=0.75, initial_step_size=0.05,
                    horseshoe_scale=horseshoe_scale, init=None)
  dim = L.ci_oracle_hmc_dim(C.byref(pb))
  th = np.ascontiguousarray(theta, dtype=np.float64)
  assert th.shape == (dim,), (th.shape, dim)
  g = np.zeros(dim)
  lp = float(L.ci_oracle_hmc_logp(C.byref(pb), th.ctypes.data, g.ctypes.data))
  return lp, g


def fit_hmc(y, mask, X, spec, *, num_results, num_warmup, seed, chain=0, num_leapfrog=15,
            prior="slab", target_accept=0.75, initial_step_size=0.05, horseshoe_scale=0.1,
            init=None, latents=True):
  """The HMC sampler of csrc/ci_hmc.h for one chain, float64.  Returns draws [S, 3 + K + P]
  (sigma_obs, sigma_level, sigma_slope, drift[K], beta[P]), accept_rate, step_size and -- with
  latents=True, trend + regression models -- level / slope / loc / trajectories [S, T] drawn as
  the device does after the chain (Durbin-Koopman draw with iteration = draw index)."""
  L = lib()
  L.ci_oracle_fit_hmc.restype = C.c_int
  L.ci_oracle_fit_hmc.argtypes = [C.POINTER(_HmcProblem), C.c_void_p, C.c_void_p, C.c_void_p]
  pb = _hmc_problem(y, mask, X, spec, num_results=num_results, num_warmup=num_warmup,
                    num_leapfrog=num_leapfrog, prior=prior, seed=seed, chain=chain,
                    target_accept=target_accept, initial_step_size=initial_step_size,
                    horseshoe_scale=horseshoe_scale, init=init)
  S, T, P, K = int(num_results), spec["T"], spec["P"], len(spec["num_seasons"])
  draws = np.zeros((S, 3 + K + P))
  acc, eps = C.c_double(0), C.c_double(0)
  rc = L.ci_oracle_fit_hmc(C.byref(pb), draws.ctypes.data, C.byref(acc), C.byref(eps))
  if rc != 0:
    raise RuntimeError(f"ci_oracle_fit_hmc failed rc={rc}")
  out = dict(draws=draws, accept_rate=acc.value, step_size=eps.value)
  if latents:
    L.ci_oracle_hmc_latents.restype = C.c_int
    L.ci_oracle_hmc_latents.argtypes = [C.POINTER(_HmcProblem), C.c_void_p, C.c_int] + [C.c_void_p] * 4
    arrs = {k: np.zeros((S, T)) for k in ("level", "slope", "loc", "trajectories")}
    rc = L.ci_oracle_hmc_latents(C.byref(pb), draws.ctypes.data, S, arrs["level"].ctypes.data,
                                 arrs["slope"].ctypes.data, arrs["loc"].ctypes.data,
                                 arrs["trajectories"].ctypes.data)
    if rc != 0:
      raise RuntimeError(f"ci_oracle_hmc_latents failed rc={rc}")
    out.update(arrs)
  return out
