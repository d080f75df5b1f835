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
      ("seed", C.c_uint32 * 2), ("chain", C.c_int32), ("reserved", C.c_int32),
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
  ]


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
                    "slope", "seasonal", "pred_mean", "trajectories", "nonzeros")):
  """Runs the float64 oracle for one chain; returns a dict of numpy arrays."""
  L = lib()
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
  pb.y = y64.ctypes.data
  pb.mask = m8.ctypes.data
  pb.X = X64.ctypes.data if P > 0 else None
  pb.season_change = sc.ctypes.data if K > 0 else None
  for f in ("level_conc", "level_scale", "level_ub", "slope_conc", "slope_scale", "slope_ub",
            "obs_conc", "obs_scale", "obs_ub", "drift_conc", "drift_scale", "drift_ub",
            "nonzero_prob", "init_level_loc", "init_level_scale", "init_slope_scale",
            "init_seasonal_scale", "obs_scale0", "level_scale0", "slope_scale0"):
    setattr(pb, f, float(spec[f]))
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
  if rc != 0:
    raise RuntimeError(f"ci_oracle_fit_gibbs failed rc={rc}")
  return res


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
