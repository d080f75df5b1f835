"""Seeded synthetic workloads (SURVEY.md section 8(d) recipe).

Mirrors the reference's own generators -- `_create_test_data`
(causalimpact_lib_test.py:35-45) and the quickstart recipe
(docs/quickstart.ipynb:279-298): AR(1) covariates around 100, a linear outcome,
a local-level random walk, unit observation noise and a step effect after 70 %
of the series.  Used by bench.py and the tests; nothing here touches the GPU.
"""
from __future__ import annotations

import numpy as np


def make_raw_series(T: int, num_covariates: int, seed: int, *, phi: float = 0.999,
                    effect: float = 10.0, level_sd: float = 0.1):
  """Returns (y[T], X[T,p]) on the original scale, effect added after 0.7 T."""
  rng = np.random.default_rng(seed)
  p = int(num_covariates)
  X = np.empty((T, p))
  for j in range(p):
    e = rng.normal(size=T)
    x = np.empty(T)
    x[0] = e[0]
    for t in range(1, T):
      x[t] = phi * x[t - 1] + e[t]
    X[:, j] = 100.0 + x
  beta = rng.uniform(0.2, 1.2, size=p) * (rng.random(p) < 0.5) if p else np.zeros(0)
  if p:
    beta[0] = 1.2
  y = X @ beta + np.cumsum(rng.normal(scale=level_sd, size=T)) + rng.normal(size=T)
  y[int(0.7 * T):] += effect
  return y, X


def standardize_for_sampler(y, X, pre_len: int):
  """What CausalImpactData hands to the sampler (data.py:105-137, standardize.py:42-55):
  pre-period z-scores (ddof=1) applied to all rows, outcome masked after the pre-period,
  intercept column appended last."""
  T = y.shape[0]
  mu_y, sd_y = y[:pre_len].mean(), y[:pre_len].std(ddof=1)
  ys = (y - mu_y) / sd_y
  mask = np.zeros(T, bool)
  mask[pre_len:] = True
  if X.shape[1] > 0:
    mu, sd = X[:pre_len].mean(0), X[:pre_len].std(0, ddof=1)
    Xs = np.where(sd > 0, (X - mu) / np.where(sd > 0, sd, 1.0), X)
    Xs = np.concatenate([Xs, np.ones((T, 1))], axis=1)
  else:
    Xs = None
  return ys, mask, Xs, (mu_y, sd_y)


def make_sampler_inputs(T: int, num_covariates: int, seed: int, **kw):
  y, X = make_raw_series(T, num_covariates, seed, **kw)
  return standardize_for_sampler(y, X, int(0.7 * T))
