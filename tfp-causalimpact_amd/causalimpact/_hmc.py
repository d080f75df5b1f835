"""Hamiltonian Monte Carlo over the model's parameters, likelihood and score on the GPU.

EXTENSION: the reference release is Gibbs-only -- there is no
`_run_hmc` in /root/reference -- but BASELINE.json's configs 1 and 3 ask for an HMC fit, so
this mirrors what `tfp.sts.fit_with_hmc` does upstream, with the pieces re-designed for the
device: the target's expensive part, the Kalman-filter log-likelihood l(theta) and its score,
are ONE kernel launch per leapfrog step for ALL chains (`ci_ll_session_eval`: associative
filter scan + two suffix scans); latent paths and posterior-predictive trajectories are drawn
afterwards for every retained theta (`ci_ll_session_draw_latents`).  Parity with TFP: unpinned
(no reference call site, no reference test); it is validated against this build's Gibbs
posterior on the same data (tests/test_gpu_hmc.py).

Model / target.  theta = (beta, log sigma_obs, log sigma_level[, log sigma_slope]);
  log p(theta | y) = l(sigma, beta) + sum_k log IG(sigma_k^2; a_k, b_k) + log|d sigma^2/d log sigma|
                     - 1/2 beta' Omega beta,
with the reference's inverse-gamma variance priors (causalimpact_lib.py:424-443) and, because
the spike-and-slab prior has no density, the Gaussian slab of that prior alone:
Omega = 0.01 (X'X/2 + diag(X'X)/2) / T (causalimpact_lib.py:451-453).  The reference's hard
upper bounds on the scales (:432, :442) are not part of this target.

Sampler.  Fixed-length leapfrog trajectories, per-chain step size by dual averaging
(Nesterov 2009 / Hoffman & Gelman 2014, target acceptance 0.75) during warm-up, diagonal mass
matrix estimated once from the middle of warm-up (pooled over chains).  Momentum and
accept/reject randomness is host-side numpy (Philox bit generator keyed by the seed pair).
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from causalimpact import _native


def _unpack(theta_u: np.ndarray, P: int, has_slope: bool, K: int = 0):
  """unconstrained [C, dim] -> device layout [C, 3 + K + P] = (s_obs, s_level, s_slope, drift[K], beta);
  theta_u = (beta[P], log s_obs, log s_level[, log s_slope], log s_drift[K]) -- the order of
  csrc/ci_hmc.h / ci_score_seq.h::hmc_drive."""
  C = theta_u.shape[0]
  out = np.zeros((C, 3 + K + P))
  lam = np.clip(theta_u[:, P:], -30.0, 30.0)   # diverging trajectories are rejected anyway
  out[:, 0] = np.exp(lam[:, 0])
  out[:, 1] = np.exp(lam[:, 1])
  ntr = 3 if has_slope else 2
  if has_slope:
    out[:, 2] = np.exp(lam[:, 2])
  for j in range(K):
    out[:, 3 + j] = np.exp(lam[:, ntr + j])
  out[:, 3 + K:] = theta_u[:, :P]
  return out


class _Target:
  """log posterior and gradient for all chains at once (one device call)."""

  def __init__(self, sess, spec, omega, P, has_slope, K=0):
    self.sess, self.P, self.has_slope, self.omega, self.K = sess, P, has_slope, omega, K
    self.ig = [(spec["obs_conc"], spec["obs_scale"]), (spec["level_conc"], spec["level_scale"])]
    if has_slope:
      self.ig.append((spec["slope_conc"], spec["slope_scale"]))
    # device slot of each of theta's scales: trend scales in front, one drift scale per block
    self.slot = list(range(len(self.ig))) + [3 + j for j in range(K)]
    self.ig += [(spec["drift_conc"], spec["drift_scale"])] * K          # causalimpact_lib.py:472-474
    self.dim = P + len(self.ig)
    self.calls = 0

  def __call__(self, theta_u):
    P, K = self.P, self.K
    dev = _unpack(theta_u, P, self.has_slope, K)
    ll, g = self.sess.evaluate(dev, want_grad=True)
    self.calls += 1
    lp = ll.copy()
    grad = np.zeros_like(theta_u)
    beta = theta_u[:, :P]
    if P:
      ob = beta @ self.omega
      lp -= 0.5 * np.sum(beta * ob, axis=1)
      grad[:, :P] = g[:, 3 + K:] - ob
    for k, (a, b) in enumerate(self.ig):
      lam = theta_u[:, P + k]
      lam = np.clip(lam, -30.0, 30.0)
      lp += -2.0 * a * lam - b * np.exp(-2.0 * lam)
      grad[:, P + k] = dev[:, self.slot[k]] * g[:, self.slot[k]] - 2.0 * a + 2.0 * b * np.exp(-2.0 * lam)
    bad = ~np.isfinite(lp)
    lp[bad] = -np.inf
    grad[bad] = 0.0
    return lp, grad


def _package(sess, dev, seed, chain_offset, C, S, T, P):
  """Latent paths + predictive trajectories for every retained draw; the arrays of
  `_native.fit_gibbs` (leading series axis of 1)."""
  level = np.zeros((C * S, T), np.float32)
  slope = np.zeros((C * S, T), np.float32)
  loc = np.zeros((C * S, T), np.float32)
  traj = np.zeros((C * S, T), np.float32)
  step = sess.max_evals
  for c in range(C):
    for lo in range(0, S, step):
      hi = min(S, lo + step)
      rows = slice(c * S + lo, c * S + hi)
      out = sess.draw_latents(dev[rows], seed, rng_chain=chain_offset + c, iter0=lo)
      level[rows], slope[rows], loc[rows], traj[rows] = (out["level"], out["slope"], out["loc"],
                                                          out["traj"])
  dev3 = dev.reshape(C, S, 3 + P)
  return dict(
      observation_noise_scale=dev3[None, :, :, 0].astype(np.float32),
      level_scale=dev3[None, :, :, 1].astype(np.float32),
      slope_scale=dev3[None, :, :, 2].astype(np.float32),
      seasonal_drift_scales=np.zeros((1, C, S, 0), np.float32),
      weights=dev3[None, :, :, 3:].astype(np.float32),
      level=level.reshape(1, C, S, T), slope=slope.reshape(1, C, S, T),
      seasonal_levels=np.zeros((1, C, S, T, 0), np.float32),
      posterior_means=loc.reshape(C, S, T).mean(axis=1)[None].astype(np.float32),
      posterior_trajectories=traj.reshape(1, C, S, T))


def fit_hmc(y, mask, X, spec: Dict, *, has_slope: bool, num_results: int, num_warmup: int,
            num_chains: int, seed, device: int = 0, chain_offset: int = 0, num_leapfrog: int = 15,
            target_accept: float = 0.75, initial_step_size: float = 0.05,
            init: str = "gibbs", prior: str = "slab",
            horseshoe_scale: float = 0.1, num_seasons=(), season_change=None) -> Dict[str, np.ndarray]:
  """HMC with the whole fit on the device (csrc/ci_hmc.h: one workgroup per chain runs the
  windowed warm-up and all sampling iterations; one more launch draws the latent path and the
  predictive trajectory of every retained draw).  Returns the arrays of `_native.fit_gibbs`
  (leading series axis of 1) plus `hmc_accept_rate`, `hmc_step_size` [C], `hmc_target_calls`
  and `hmc_kernel_ms`.  Models with seasonal blocks (`num_seasons`, `season_change` [K, T] as in
  `_native.fit_gibbs`) and series longer than 4096 steps run on the sequential route
  (csrc/ci_score_seq.h: same target, same adaptation, one wavefront evaluates the score).

  prior: "slab" (the Gaussian slab of the reference's spike-and-slab prior) or "horseshoe" (the
  prior of `tfp.sts.SparseLinearRegression`, weights_prior_scale = horseshoe_scale).
  init: "gibbs" starts every chain at the Gibbs sampler's initial state (jittered); "vi" first
  fits the mean-field surrogate posterior (`_vi.fit_surrogate_posterior`, what
  `tfp.sts.fit_with_hmc` does upstream) and starts chain c at its c-th draw (slab prior only;
  since round 5 for models with seasonal blocks too: the drift scales are coordinates of the
  surrogate like the other scales)."""
  y = np.asarray(y, np.float64)
  mask = np.asarray(mask, bool)
  T = y.shape[0]
  P = 0 if X is None else int(np.asarray(X).shape[1])
  C, S, W = int(num_chains), int(num_results), int(num_warmup)
  K = len(num_seasons)
  pb = _native.make_problem(T=T, P=P, has_slope=has_slope, num_seasons=num_seasons, num_warmup=0,
                            num_results=1, seed=seed, device=device)
  sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=64,
                               season_change=season_change)
  try:
    init_theta = None
    if init == "vi":
      if prior != "slab":
        raise NotImplementedError("the surrogate posterior is built for the slab prior only")
      from causalimpact import _vi  # pylint: disable=import-outside-toplevel
      vi = _vi.fit_surrogate_posterior(y, mask, X, spec, has_slope=has_slope, seed=seed,
                                       device=device, sess=sess,
                                       num_mc=min(32, sess.max_evals), num_blocks=K)
      # chain c starts at the (chain_offset + c)-th draw, whatever the split over devices
      init_theta = _vi.sample_surrogate(vi, chain_offset + C, seed=seed)[chain_offset:]
    elif init != "gibbs":
      raise ValueError(f"init must be 'gibbs' or 'vi', got {init!r}")
    ms = sess.hmc_run(num_chains=C, num_warmup=W, num_results=S, num_leapfrog=num_leapfrog,
                      target_accept=target_accept, initial_step_size=initial_step_size, seed=seed,
                      chain_offset=chain_offset, init_theta=init_theta, prior=prior,
                      horseshoe_scale=horseshoe_scale)
    _, acc, eps, out = sess.hmc_fetch()
  finally:
    sess.close()
  if K == 0:
    out["seasonal_drift_scales"] = np.zeros((1, C, S, 0), np.float32)
    out["seasonal_levels"] = np.zeros((1, C, S, T, 0), np.float32)
  out.update(hmc_accept_rate=acc, hmc_step_size=eps, hmc_kernel_ms=np.asarray(ms),
             hmc_target_calls=np.int64((W + S) * num_leapfrog + 1))
  return out


def fit_hmc_host(y, mask, X, spec: Dict, *, has_slope: bool, num_results: int, num_warmup: int,
                 num_chains: int, seed, device: int = 0, chain_offset: int = 0,
                 num_leapfrog: int = 15, target_accept: float = 0.75,
                 initial_step_size: float = 0.05) -> Dict[str, np.ndarray]:
  """The same sampler with momentum / accept logic in numpy and one device call per leapfrog
  step (`ci_ll_session_eval`): the statistical reference of the device kernel
  (tests/test_gpu_hmc.py), ~14x slower.  Mass matrix pooled over chains here, per chain on the
  device; different random streams -- the two agree in distribution, not per draw."""
  y = np.asarray(y, np.float64)
  mask = np.asarray(mask, bool)
  T = y.shape[0]
  P = 0 if X is None else int(np.asarray(X).shape[1])
  C, S, W = int(num_chains), int(num_results), int(num_warmup)
  pb = _native.make_problem(T=T, P=P, has_slope=has_slope, num_warmup=0, num_results=1,
                            seed=seed, device=device)
  sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X,
                               max_evals=max(C, min(256, C * S)))
  omega = None
  if P:
    X64 = np.asarray(X, np.float64)
    xtx = X64.T @ X64
    omega = 0.01 * (0.5 * xtx + 0.5 * np.diag(np.diag(xtx))) / T        # :451-453
  target = _Target(sess, spec, omega, P, has_slope)
  dim = target.dim
  s0, s1 = _native.seed_pair(seed)
  rng = np.random.Generator(np.random.Philox(key=[(s0 << 32) | s1, 0x484D43 + chain_offset]))

  # start from the initial Gibbs state (causalimpact_lib.py:566-581) with a little jitter
  theta = np.zeros((C, dim))
  theta[:, P] = np.log(spec["obs_scale0"])
  theta[:, P + 1] = np.log(max(spec["level_scale0"], 1e-4))
  if has_slope:
    theta[:, P + 2] = np.log(max(spec["slope_scale0"], 1e-4))
  theta += 0.01 * rng.normal(size=theta.shape)
  lp, grad = target(theta)

  inv_mass = np.ones(dim)
  # dual averaging state per chain
  eps = np.full(C, initial_step_size)
  mu = np.log(10.0 * eps)
  hbar = np.zeros(C)
  log_eps_bar = np.zeros(C)
  t_da = np.zeros(C)
  gamma_da, t0_da, kappa_da = 0.05, 10.0, 0.75
  window = (int(0.25 * W), int(0.75 * W))
  collected = []
  draws = np.zeros((C, S, dim))
  accepted = np.zeros(C)

  def restart_dual_averaging():
    nonlocal mu, hbar, log_eps_bar, t_da
    mu = np.log(10.0 * eps)
    hbar = np.zeros(C)
    log_eps_bar = np.zeros(C)
    t_da = np.zeros(C)

  for it in range(W + S):
    mom = rng.normal(size=(C, dim)) / np.sqrt(inv_mass)
    h0 = -lp + 0.5 * np.sum(mom * mom * inv_mass, axis=1)
    th, g, p = theta.copy(), grad.copy(), mom.copy()
    e = eps[:, None]
    lp_new = lp
    for _ in range(num_leapfrog):
      p = p + 0.5 * e * g
      th = th + e * inv_mass * p
      lp_new, g = target(th)
      p = p + 0.5 * e * g
    h1 = -lp_new + 0.5 * np.sum(p * p * inv_mass, axis=1)
    log_acc = np.where(np.isfinite(h1), h0 - h1, -np.inf)
    acc_prob = np.exp(np.minimum(0.0, log_acc))
    take = np.log(rng.random(C)) < log_acc
    theta[take], lp[take], grad[take] = th[take], lp_new[take], g[take]
    if it < W:
      # dual averaging of log step size
      t_da += 1.0
      hbar = (1 - 1 / (t_da + t0_da)) * hbar + (target_accept - acc_prob) / (t_da + t0_da)
      log_eps = mu - np.sqrt(t_da) / gamma_da * hbar
      eta = t_da ** (-kappa_da)
      log_eps_bar = eta * log_eps + (1 - eta) * log_eps_bar
      eps = np.exp(log_eps)
      if window[0] <= it < window[1]:
        collected.append(theta.copy())
      if it == window[1] - 1 and len(collected) >= 10:
        pooled = np.concatenate(collected, axis=0)
        var = pooled.var(axis=0) + 1e-8
        inv_mass = var / var.mean() if np.all(np.isfinite(var)) else inv_mass
        eps = np.exp(log_eps_bar)
        restart_dual_averaging()
      if it == W - 1:
        eps = np.exp(log_eps_bar)
    else:
      draws[:, it - W] = theta
      accepted += take

  dev = _unpack(draws.reshape(C * S, dim), P, has_slope)
  out = _package(sess, dev, seed, chain_offset, C, S, T, P)
  calls = target.calls
  sess.close()
  out.update(hmc_accept_rate=accepted / max(S, 1), hmc_step_size=eps.copy(),
             hmc_target_calls=np.array(calls))
  return out
