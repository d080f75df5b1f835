"""Mean-field Gaussian surrogate posterior of the model parameters (variational inference).

EXTENSION (BASELINE.json's north_star lists
`tfp.sts.build_factored_surrogate_posterior` among the subsystems to replace -- upstream it is what
`tfp.sts.fit_with_hmc` initialises its chains from; the reference itself never calls it).
Parity with TFP: unpinned; validated against this build's HMC / Gibbs posteriors
(tests/test_gpu_hmc.py).

The surrogate is q(theta) = prod_i N(theta_i; m_i, s_i^2) over the same unconstrained
parameters as the HMC target (`_hmc._Target`: weights, log scales).  The ELBO is maximised with
reparameterisation gradients: theta_k = m + s * eps_k, and all `num_mc` Monte-Carlo points of a
step are ONE device call (`ci_ll_session_eval`: log-likelihood and score of every theta_k by the
time-parallel Kalman scans).  Adam on the host; nothing else is host arithmetic.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from causalimpact import _hmc
from causalimpact import _native


def fit_surrogate_posterior(y, mask, X, spec: Dict, *, has_slope: bool, num_steps: int = 300,
                            num_mc: int = 32, learning_rate: float = 0.05, seed=0, device: int = 0,
                            sess: Optional[_native.LogLikSession] = None, num_blocks: int = 0,
                            num_seasons=(), season_change=None) -> Dict[str, np.ndarray]:
  """Returns {"mean": [dim], "log_sd": [dim], "elbo": [num_steps]} in the unconstrained
  parameterisation theta = (weights[P], log sigma_obs, log sigma_level[, log sigma_slope],
  log sigma_drift[K]) -- K = `num_blocks` seasonal blocks of the session `sess` (round 5), or of
  the session built here from `num_seasons` / `season_change` (as in `_native.fit_gibbs`)."""
  y = np.asarray(y, np.float64)
  mask = np.asarray(mask, bool)
  T = y.shape[0]
  P = 0 if X is None else int(np.asarray(X).shape[1])
  own = sess is None
  K = int(num_blocks)
  if own:
    K = len(num_seasons)
    pb = _native.make_problem(T=T, P=P, has_slope=has_slope, num_seasons=num_seasons, num_warmup=0,
                              num_results=1, seed=seed, device=device)
    sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=num_mc,
                                 season_change=season_change)
  try:
    omega = None
    if P:
      X64 = np.asarray(X, np.float64)
      xtx = X64.T @ X64
      omega = 0.01 * (0.5 * xtx + 0.5 * np.diag(np.diag(xtx))) / T          # :451-453
    target = _hmc._Target(sess, spec, omega, P, has_slope, K)   # pylint: disable=protected-access
    dim = target.dim
    s0, s1 = _native.seed_pair(seed)
    rng = np.random.Generator(np.random.Philox(key=[(s0 << 32) | s1, 0x5649]))
    mean = np.zeros(dim)
    mean[P] = np.log(spec["obs_scale0"])
    mean[P + 1] = np.log(max(spec["level_scale0"], 1e-4))
    ntr = 3 if has_slope else 2
    if has_slope:
      mean[P + 2] = np.log(max(spec["slope_scale0"], 1e-4))
    for j in range(K):                   # the Gibbs sampler's initial drift scales (:573-574)
      mean[P + ntr + j] = np.log(max(float(np.atleast_1d(spec["drift_scale0"])[j]), 1e-4))
    log_sd = np.full(dim, np.log(0.05))
    m1 = np.zeros(2 * dim)
    m2 = np.zeros(2 * dim)
    elbo = np.zeros(num_steps)
    b1, b2, tiny = 0.9, 0.999, 1e-8
    for step in range(num_steps):
      eps = rng.normal(size=(num_mc, dim))
      sd = np.exp(log_sd)
      lp, g = target(mean + sd * eps)                     # one launch for all num_mc points
      ok = np.isfinite(lp)
      if not ok.any():
        log_sd -= 0.5                                      # every point diverged: shrink
        continue
      w = ok / ok.sum()
      grad_mean = (w[:, None] * g).sum(axis=0)
      grad_log_sd = (w[:, None] * g * eps * sd).sum(axis=0) + 1.0          # + entropy gradient
      elbo[step] = float((w * np.where(ok, lp, 0.0)).sum() + log_sd.sum())
      grad = np.concatenate([grad_mean, grad_log_sd])
      m1 = b1 * m1 + (1 - b1) * grad
      m2 = b2 * m2 + (1 - b2) * grad * grad
      upd = learning_rate * (m1 / (1 - b1 ** (step + 1))) / (np.sqrt(m2 / (1 - b2 ** (step + 1))) + tiny)
      mean += upd[:dim]
      log_sd = np.clip(log_sd + upd[dim:], -12.0, 3.0)
    return {"mean": mean, "log_sd": log_sd, "elbo": elbo}
  finally:
    if own:
      sess.close()


def sample_surrogate(vi: Dict[str, np.ndarray], num_draws: int, seed=0) -> np.ndarray:
  """[num_draws, dim] draws of the fitted surrogate (unconstrained parameterisation)."""
  s0, s1 = _native.seed_pair(seed)
  rng = np.random.Generator(np.random.Philox(key=[(s0 << 32) | s1, 0x5650]))
  return vi["mean"] + np.exp(vi["log_sd"]) * rng.normal(size=(num_draws, vi["mean"].shape[0]))
