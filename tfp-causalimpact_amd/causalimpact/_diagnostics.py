"""Convergence diagnostics for multi-chain fits (SURVEY.md section 8(f) N3 -- absent from the
reference, which runs one chain; required by `north_star` "chain gather/diagnostics").

Split-R-hat, bulk ESS and tail ESS (Gelman et al. 2013 ch. 11; Vehtari, Gelman, Simpson,
Carpenter & Buerkner 2021).  Everything is written as PARTIAL SUMS over (split) chains that add
across ranks, so that one small all-reduce combines the chains of all GPUs
(`_distributed.fit_sharded`):

  per split chain m of length n:   mean_m,  acov_m[k] (biased autocovariance, k = 0..n-1)
  partial sums over local chains:  sum_m acov_m[k]  [n],  sum_m mean_m,  sum_m mean_m^2,  M

  W       = n/(n-1) * mean_m acov_m[0]
  var+    = mean_m acov_m[0] + var_m(mean_m)
  rho_k   = 1 - (W - n/(n-1) mean_m acov_m[k]) / var+
  ESS     = M n / tau,  tau = -1 + 2 sum of Geyer's initial monotone positive pair sums

Bulk ESS applies this to the rank-normalised draws, tail ESS is the smaller of the ESS of the
indicators I(x <= q05) and I(x <= q95); both need pooled order statistics of a SCALAR parameter,
for which the [chains, draws] scalars themselves are all-gathered first (a few hundred KB).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
from scipy import special


def split_chains(draws: np.ndarray) -> np.ndarray:
  """[C, S] -> [2C, S // 2]: first and second halves as separate chains."""
  x = np.asarray(draws, np.float64)
  half = x.shape[1] // 2
  return np.concatenate([x[:, :half], x[:, half:2 * half]], axis=0)


def partial_sums(z: np.ndarray) -> Dict[str, np.ndarray]:
  """Additive statistics of the (already split) chains z [M, n]."""
  z = np.asarray(z, np.float64)
  m, n = z.shape
  mean = z.mean(axis=1) if n else np.zeros(m)
  if m == 0 or n == 0:
    return dict(acov=np.zeros(n), s1=np.zeros(()), s2=np.zeros(()), m=np.zeros(()))
  zc = z - mean[:, None]
  size = 1 << int(np.ceil(np.log2(max(2 * n, 2))))
  f = np.fft.rfft(zc, size, axis=1)
  acov = np.fft.irfft(f * np.conj(f), size, axis=1)[:, :n] / n           # biased, per chain
  return dict(acov=acov.sum(axis=0), s1=np.asarray(mean.sum()), s2=np.asarray((mean * mean).sum()),
              m=np.asarray(float(m)))


def pack(ps: Dict[str, np.ndarray]) -> np.ndarray:
  """One flat float64 vector [n + 3] for an all-reduce(sum)."""
  return np.concatenate([np.asarray(ps["acov"], np.float64).ravel(),
                         [float(ps["s1"]), float(ps["s2"]), float(ps["m"])]])


def unpack(v: np.ndarray) -> Dict[str, np.ndarray]:
  v = np.asarray(v, np.float64)
  return dict(acov=v[:-3], s1=v[-3], s2=v[-2], m=v[-1])


def ess_from_sums(ps: Dict[str, np.ndarray]) -> float:
  acov_sum, m = np.asarray(ps["acov"], np.float64), float(ps["m"])
  n = acov_sum.shape[0]
  if m < 1 or n < 4:
    return float("nan")
  acov = acov_sum / m
  mean_var = (float(ps["s2"]) - float(ps["s1"]) ** 2 / m) / (m - 1.0) if m > 1 else 0.0
  within = acov[0] * n / (n - 1.0)
  var_plus = acov[0] + mean_var
  if not np.isfinite(var_plus) or var_plus <= 0:
    return float("nan")
  rho = 1.0 - (within - acov * n / (n - 1.0)) / var_plus
  rho[0] = 1.0
  tau, prev = -1.0, np.inf
  for k in range(0, n - 1, 2):       # pair sums must be positive and non-increasing
    pair = rho[k] + rho[k + 1]
    if pair < 0:
      break
    pair = min(pair, prev)
    tau += 2.0 * pair
    prev = pair
  return float(m * n / max(tau, 1.0 / np.log10(max(m * n, 10))))


def rhat_from_sums(ps: Dict[str, np.ndarray]) -> float:
  acov_sum, m = np.asarray(ps["acov"], np.float64), float(ps["m"])
  n = acov_sum.shape[0]
  if m < 2 or n < 2:
    return float("nan")
  within = acov_sum[0] / m * n / (n - 1.0)
  between = n * (float(ps["s2"]) - float(ps["s1"]) ** 2 / m) / (m - 1.0)
  if within <= 0:
    return float("nan")
  return float(np.sqrt(((n - 1.0) / n * within + between / n) / within))


def rank_normalize(local: np.ndarray, pooled: np.ndarray) -> np.ndarray:
  """z-scores of the average ranks of `local` among `pooled` (Vehtari et al. 2021 eq. 14)."""
  pooled = np.sort(np.asarray(pooled, np.float64).ravel())
  x = np.asarray(local, np.float64)
  lo = np.searchsorted(pooled, x, side="left")
  hi = np.searchsorted(pooled, x, side="right")
  rank = 0.5 * (lo + hi + 1)                              # average rank, 1-based
  return special.ndtri((rank - 0.375) / (pooled.size + 0.25))


def bulk_partial(local: np.ndarray, pooled: np.ndarray) -> Dict[str, np.ndarray]:
  return partial_sums(split_chains(rank_normalize(local, pooled)))


def tail_partials(local: np.ndarray, pooled: np.ndarray):
  q05, q95 = np.quantile(np.asarray(pooled, np.float64), [0.05, 0.95])
  x = np.asarray(local, np.float64)
  return (partial_sums(split_chains((x <= q05).astype(np.float64))),
          partial_sums(split_chains((x <= q95).astype(np.float64))))


def split_rhat(draws: np.ndarray) -> float:
  """Split-R-hat of [chains, draws]; NaN for degenerate input."""
  x = np.asarray(draws, np.float64)
  if x.shape[1] // 2 < 2:
    return float("nan")
  return rhat_from_sums(partial_sums(split_chains(x)))


def ess_plain(draws: np.ndarray) -> float:
  """ESS of the draws as they are (no rank normalisation)."""
  x = np.asarray(draws, np.float64)
  if x.shape[1] // 2 < 4:
    return float("nan")
  return ess_from_sums(partial_sums(split_chains(x)))


def ess_bulk(draws: np.ndarray) -> float:
  x = np.asarray(draws, np.float64)
  if x.shape[1] // 2 < 4 or not np.isfinite(x).all() or np.ptp(x) == 0:
    return float("nan")
  return ess_from_sums(bulk_partial(x, x))


def ess_tail(draws: np.ndarray) -> float:
  x = np.asarray(draws, np.float64)
  if x.shape[1] // 2 < 4 or not np.isfinite(x).all() or np.ptp(x) == 0:
    return float("nan")
  lo, hi = tail_partials(x, x)
  both = np.array([ess_from_sums(lo), ess_from_sums(hi)], np.float64)
  ok = np.isfinite(both)          # (nanmin of two NaNs would warn "All-NaN axis")
  return float(both[ok].min()) if ok.any() else float("nan")


def summarize(draws_by_key: Dict[str, np.ndarray]) -> Dict[str, Dict[str, float]]:
  """{"split_rhat", "ess_bulk", "ess_tail"} -> {key: value} for [chains, draws] scalars."""
  return {"split_rhat": {k: split_rhat(v) for k, v in draws_by_key.items()},
          "ess_bulk": {k: ess_bulk(v) for k, v in draws_by_key.items()},
          "ess_tail": {k: ess_tail(v) for k, v in draws_by_key.items()}}
