"""The reference-faithful regression step of the oracle -- weight adjustment ON
(causalimpact_lib.py:388), exponent (a_post - 1), variance CLIPPED (not truncated) at
upper_bound -- against an independent numpy transcription of the same transition.

tests/test_geweke.py pins the oracle to the MODEL but has to switch the weight adjustment off (the
adjusted chain has no fixed-prior invariant law).  What the adjusted transition actually is -- one
step of a Markov chain on (gamma, sigma^2, beta) whose slab precision is sigma^2_prev * Omega --
is written out here with dense linear algebra (slogdet / solve / inv instead of the oracle's
Cholesky route), fed the SAME uniforms / normals / gamma variates through the oracle's stream
primitives, and must reproduce ONE oracle transition from arbitrary starting states: visiting
order, every accept / reject, sigma^2 (including starts where the clip binds) and the weights."""
import numpy as np
import pytest

from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc

SITE_PERM, SITE_FLIP, SITE_OBSVAR, SITE_WEIGHTS = 1, 2, 3, 4      # oracle/ci_oracle.h


def _transition(y, mask, X, spec, level, w_prev, sigma_prev, seed, chain, it=0):
  obs = ~mask
  P = X.shape[1]
  Xo = X[obs]
  target = (y - level)[obs]
  xtx = Xo.T @ Xo
  xty = Xo.T @ target
  yty = float(target @ target)
  n = int(obs.sum())
  omega = orc.slab_precision(X) * sigma_prev ** 2          # the adjustment: x previous sigma^2
  a_post = spec["obs_conc"] + 0.5 * n
  pi = spec["nonzero_prob"]

  def logp(g):
    idx = np.flatnonzero(g)
    quad, ld_prior, ld_post = 0.0, 0.0, 0.0
    if idx.size:
      Og = omega[np.ix_(idx, idx)]
      Ag = Og + xtx[np.ix_(idx, idx)]
      quad = float(xty[idx] @ np.linalg.solve(Ag, xty[idx]))
      ld_prior = 0.5 * np.linalg.slogdet(Og)[1]
      ld_post = 0.5 * np.linalg.slogdet(Ag)[1]
    beta = spec["obs_scale"] + 0.5 * (yty - quad)
    prior = 0.0 if pi >= 1.0 else float(np.where(g, np.log(pi), np.log1p(-pi)).sum())
    return ld_prior - ld_post + prior - (a_post - 1.0) * np.log(2.0 * beta), beta

  if pi >= 1.0:
    g = np.ones(P, bool)
  else:
    g = w_prev != 0
    cur = logp(g)[0]
    u = np.array([orc.uniform(seed, chain, it, SITE_PERM, 0, j) for j in range(P)])
    for s, j in enumerate(np.argsort(u, kind="stable")):
      g[j] = ~g[j]
      prop = logp(g)[0]
      if orc.uniform(seed, chain, it, SITE_FLIP, 0, s) < 1.0 / (1.0 + np.exp(-(prop - cur))):
        cur = prop
      else:
        g[j] = ~g[j]
  beta = logp(g)[1]
  var = beta / orc.gamma(a_post, seed, chain, it, SITE_OBSVAR)
  clipped = var > spec["obs_ub"]
  var = min(var, spec["obs_ub"])                  # the VARIANCE against upper_bound, by clipping
  idx = np.flatnonzero(g)
  w = np.zeros(P)
  if idx.size:
    Ag = omega[np.ix_(idx, idx)] + xtx[np.ix_(idx, idx)]
    mean = np.linalg.solve(Ag, xty[idx])
    L = np.linalg.cholesky(Ag)
    z = np.array([orc.normal(seed, chain, it, SITE_WEIGHTS, 0, int(j)) for j in idx])
    w[idx] = mean + np.sqrt(var) * np.linalg.solve(L.T, z)      # cov = var * Ag^-1
  return g, np.sqrt(var), w, clipped


@pytest.mark.parametrize("p,case", [(7, 0), (7, 1), (7, 2), (12, 3), (2, 4)])
def test_one_adjusted_transition_equals_the_numpy_transcription(p, case):
  T = 90
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 100 + case)
  spec = orc.default_spec(y, mask, X)
  P = p + 1
  rng = np.random.default_rng(case)
  level = 0.3 * rng.normal(size=T).cumsum() / np.sqrt(T)
  w0 = np.where(rng.random(P) < 0.5, rng.normal(size=P), 0.0)
  seed, chain = (9, case), 3 + case
  clips = 0
  # the chain's sigma_prev is its initial obs_scale0: vary it through the spec, and shrink the
  # upper bound in one case so that the clip binds
  for sigma_prev, ub in ((spec["obs_scale0"], spec["obs_ub"]), (1.7, spec["obs_ub"]), (0.05, 0.004)):
    sp = dict(spec, obs_scale0=sigma_prev, obs_ub=ub)
    lat0 = level[:, None].copy()
    got = orc.fit_gibbs(y, mask, X, sp, num_results=1, num_warmup=0, seed=seed, chain=chain,
                        weights0=w0, latents0=lat0)
    g, sig, w, clipped = _transition(np.where(mask, 0.0, y), mask, X, sp, level, w0, sigma_prev,
                                     seed, chain)
    clips += int(clipped)
    np.testing.assert_array_equal(got["nonzeros"][0].astype(bool), g)
    np.testing.assert_allclose(got["obs_scale"][0], sig, rtol=1e-10)
    np.testing.assert_allclose(got["weights"][0], w, rtol=1e-8, atol=1e-10)
  assert clips >= 1                # the clipped branch was exercised
