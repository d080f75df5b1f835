"""Geweke (2004) "Getting it right" joint-distribution test of the ORACLE's Gibbs transition
(oracle/ci_oracle.c::ci_oracle_fit_gibbs), so that the sampler is pinned to the MODEL the
reference specifies (/root/reference/causalimpact/causalimpact_lib.py:398-500: inverse-gamma scale
priors, N(loc, sd) initial level, constrained seasonal block, Bernoulli(pi) x Gaussian-slab
regression prior with precision 0.01 (X'X/2 + diag(X'X)/2) / T), not merely to itself.

Two simulators of the joint p(theta, y):
  * marginal-conditional: theta ~ prior (written out below in numpy, independently of the
    oracle), y ~ p(y | theta);
  * successive-conditional: theta' ~ K(theta | y) -- ONE oracle Gibbs iteration started from the
    current state -- then y' ~ p(y | theta'), repeated.
If K leaves p(theta | y) invariant, both produce the same joint; every test function g(theta, y)
must then have equal means (z-scores with batch-means standard errors for the chain).

Settings that make the test exact rather than approximate: the upper bounds on the scales are
far away (the reference clips, which has no generative model), and the regression case runs with
CI_ORACLE_FLAG_NO_WEIGHT_ADJUSTMENT (the reference's experimental_use_weight_adjustment rescales
the slab by the PREVIOUS sigma^2, which has no fixed invariant law; the flag removes exactly that
one multiplication).  The design has small covariate values at the observed steps and large ones
at the masked steps, so the 1 %-information slab prior is not swamped by the likelihood and
the chain mixes over the prior in a few iterations.

Finding recorded by this test: with the reference's exponent (a_post - 1) in the collapsed
marginal (Scott & Varian 2013 eq. 8, as TFP and bsts write it) the transition UNDER-includes
features at n = 4 observations (z = -5.7 on the inclusion frequency); with the exact a_post it
passes.  The two differ by a factor SS_g'/SS_g = 1 + O(1/n) on the inclusion odds, so the
reference-faithful formula stays the default and the test runs CI_ORACLE_FLAG_EXACT_MARGINAL;
`test_reference_exponent_is_a_one_over_n_perturbation` bounds the difference at realistic n.
"""
import numpy as np
import pytest

from oracle import ci_oracle as orc


def _spec(T, P, has_slope, seasons, X, mask):
  y0 = np.zeros(T)
  spec = orc.default_spec(y0, mask, X if P else None, has_slope=has_slope, seasons=seasons,
                          outcome_sd=1.0)
  # proper, moderately informative priors; bounds out of reach
  spec.update(level_conc=6.0, level_scale=6.0 * 0.3 ** 2, level_ub=1e9,
              slope_conc=6.0, slope_scale=6.0 * 0.1 ** 2, slope_ub=1e9,
              obs_conc=6.0, obs_scale=6.0 * 0.5 ** 2, obs_ub=1e18,
              drift_conc=6.0, drift_scale=6.0 * 0.4 ** 2, drift_ub=1e9,
              init_level_loc=0.3, init_level_scale=0.7, init_slope_scale=0.2,
              init_seasonal_scale=0.6, nonzero_prob=0.5 if P else 1.0)
  return spec


def _inv_gamma(rng, conc, scale):
  return scale / rng.gamma(conc)


def _prior_draw(rng, spec, X, omega):
  """theta ~ prior.  Returns dict(obs, level, slope, drift[K], w[P], lat[T, d])."""
  T, P, K = spec["T"], spec["P"], len(spec["num_seasons"])
  th = dict(obs=np.sqrt(_inv_gamma(rng, spec["obs_conc"], spec["obs_scale"])),
            level=np.sqrt(_inv_gamma(rng, spec["level_conc"], spec["level_scale"])),
            slope=(np.sqrt(_inv_gamma(rng, spec["slope_conc"], spec["slope_scale"]))
                   if spec["has_slope"] else 0.0),
            drift=[np.sqrt(_inv_gamma(rng, spec["drift_conc"], spec["drift_scale"]))
                   for _ in range(K)])
  w = np.zeros(P)
  if P:
    gam = rng.random(P) < spec["nonzero_prob"]
    idx = np.flatnonzero(gam)
    if idx.size:
      prec = omega[np.ix_(idx, idx)] / th["obs"] ** 2      # beta_g | sigma^2 ~ N(0, sigma^2 Omega_g^-1)
      w[idx] = np.linalg.cholesky(np.linalg.inv(prec)) @ rng.normal(size=idx.size)
  th["w"] = w
  d = 1 + int(spec["has_slope"]) + sum(n - 1 for n in spec["num_seasons"])
  lat = np.zeros((T, d))
  x = np.zeros(d)
  x[0] = spec["init_level_loc"] + spec["init_level_scale"] * rng.normal()
  o = 1
  if spec["has_slope"]:
    x[1] = spec["init_slope_scale"] * rng.normal()
    o = 2
  for n in spec["num_seasons"]:
    # n effects iid N(0, s^2) centred to sum zero; the latent keeps the first n-1 (Appendix F)
    e = spec["init_seasonal_scale"] * rng.normal(size=n)
    e -= e.mean()
    x[o:o + n - 1] = e[:n - 1]
    o += n - 1
  for t in range(T):
    lat[t] = x
    if t + 1 == T:
      break
    nx = x.copy()
    nx[0] = x[0] + (x[1] if spec["has_slope"] else 0.0) + th["level"] * rng.normal()
    o = 1
    if spec["has_slope"]:
      nx[1] = x[1] + th["slope"] * rng.normal()
      o = 2
    for k, n in enumerate(spec["num_seasons"]):
      if spec["season_change"][k][t]:
        r = x[o:o + n - 1]
        wd = th["drift"][k] * rng.normal()
        rot = np.concatenate([r[1:], [-r.sum()]])
        nx[o:o + n - 1] = rot - wd / n
      o += n - 1
    x = nx
  th["lat"] = lat
  return th


def _observe(rng, spec, th, X, mask):
  loc = th["lat"][:, 0].copy()
  o = 1 + int(spec["has_slope"])
  for n in spec["num_seasons"]:
    loc += th["lat"][:, o]
    o += n - 1
  if spec["P"]:
    loc += X @ th["w"]
  y = loc + th["obs"] * rng.normal(size=spec["T"])
  return np.where(mask, 0.0, y)


def _transition(spec, th, y, X, mask, seed, flags):
  sp = dict(spec)
  sp.update(obs_scale0=th["obs"], level_scale0=th["level"], slope_scale0=th["slope"],
            drift_scale0=list(th["drift"]))
  K = len(spec["num_seasons"])
  r = orc.fit_gibbs(y, mask, X if spec["P"] else None, sp, num_results=1, num_warmup=0, seed=seed,
                    weights0=th["w"] if spec["P"] else None, latents0=th["lat"], flags=flags,
                    want=("obs_scale", "level_scale", "slope_scale", "drift_scales", "weights",
                          "level", "slope", "seasonal"))
  lat = np.zeros_like(th["lat"])
  lat[:, 0] = r["level"][0]
  o = 1
  if spec["has_slope"]:
    lat[:, 1] = r["slope"][0]
    o = 2
  for k, n in enumerate(spec["num_seasons"]):
    # only the block's first latent (its contribution to y) is returned; with P = 0 nothing in
    # the next transition reads the others (the latent draw does not depend on the old path)
    lat[:, o] = r["seasonal"][0][:, k]
    o += n - 1
  return dict(obs=float(r["obs_scale"][0]), level=float(r["level_scale"][0]),
              slope=float(r["slope_scale"][0]), drift=[float(v) for v in r["drift_scales"][0]],
              w=r["weights"][0].copy() if spec["P"] else np.zeros(0), lat=lat), o


def _stats(spec, th, y, mask):
  g = [th["obs"], th["obs"] ** 2, th["level"], th["level"] ** 2, th["lat"][0, 0], th["lat"][-1, 0],
       th["lat"][-1, 0] ** 2, y[~mask].mean(), (y[~mask] ** 2).mean()]
  if spec["has_slope"]:
    g += [th["slope"], th["slope"] ** 2, th["lat"][-1, 1], th["lat"][-1, 1] ** 2]
  o = 1 + int(spec["has_slope"])
  for k, n in enumerate(spec["num_seasons"]):
    g += [th["drift"][k], th["drift"][k] ** 2, th["lat"][-1, o], th["lat"][-1, o] ** 2,
          th["lat"][0, o] * th["lat"][4, o]]
    o += n - 1
  if spec["P"]:
    w = th["w"]
    g += list(w) + list(w ** 2) + list((w != 0).astype(float)) + [w[0] * w[1], float((w != 0).all())]
  return np.array(g, float)


CASES = {
    "local_level": dict(T=10, P=0, has_slope=False, seasons=()),
    "local_linear_trend": dict(T=10, P=0, has_slope=True, seasons=()),
    "regression_spike_slab": dict(T=8, P=2, has_slope=False, seasons=()),
    "seasonal_3_seasons": dict(T=11, P=0, has_slope=False, seasons=((3, 1),)),
    "seasonal_2x2_and_4": dict(T=12, P=0, has_slope=True, seasons=((2, 2), (4, 1))),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_gibbs_transition_preserves_the_joint_distribution(name):
  c = CASES[name]
  T, P = c["T"], c["P"]
  rng = np.random.default_rng(sum(map(ord, name)))
  mask = np.zeros(T, bool)
  mask[T - 3:] = True                       # the "post-period": masked, propagated by the prior
  mask[2] = True                            # and one missing pre-period step
  X = None
  omega = None
  if P:
    X = rng.normal(size=(T, P))
    X[~mask] *= 0.3
    X[mask] *= 3.0
    omega = orc.slab_precision(X)
  spec = _spec(T, P, c["has_slope"], c["seasons"], X, mask)
  flags = (orc.FLAG_NO_WEIGHT_ADJUSTMENT | orc.FLAG_EXACT_MARGINAL) if P else 0
  N = 24000
  # marginal-conditional
  mc = np.array([_stats(spec, th, _observe(rng, spec, th, X, mask), mask)
                 for th in (_prior_draw(rng, spec, X, omega) for _ in range(N))])
  # successive-conditional, started from an exact draw of the joint
  th = _prior_draw(rng, spec, X, omega)
  y = _observe(rng, spec, th, X, mask)
  sc = np.zeros_like(mc)
  for i in range(N):
    th, o = _transition(spec, th, y, X, mask, seed=(i + 1, 77), flags=flags)
    y = _observe(rng, spec, th, X, mask)
    sc[i] = _stats(spec, th, y, mask)
  nb = 60
  bm = sc[: (N // nb) * nb].reshape(nb, -1, sc.shape[1]).mean(axis=1)
  se = np.sqrt(bm.var(axis=0, ddof=1) / nb + mc.var(axis=0, ddof=1) / N)
  z = (sc.mean(axis=0) - mc.mean(axis=0)) / se
  assert np.abs(z).max() < 4.5, (name, np.round(z, 2))


def test_reference_exponent_is_a_one_over_n_perturbation():
  """Inclusion log-odds under the reference's (a_post - 1) exponent vs the exact a_post differ by
  log(SS_without / SS_with): for a pre-period of realistic length and an irrelevant covariate
  that is ~1/n, far below the Monte-Carlo error of any posterior summary."""
  rng = np.random.default_rng(0)
  n, P = 700, 4
  X = rng.normal(size=(n, P))
  y = X[:, 0] * 0.8 + rng.normal(size=n)
  xtx, xty, yty = X.T @ X, X.T @ y, float(y @ y)
  omega = orc.slab_precision(X)
  post_conc = 25.0 + 0.5 * n
  lp = {}
  for nz in ((1, 0, 0, 0), (1, 1, 0, 0)):
    lp[nz] = orc.spike_slab_logp(xtx, omega, xty, yty, nz, 0.75, post_conc, 5.0)
    # the exact exponent: post_conc + 1 in the reference's formula
    lp[nz + ("exact",)] = orc.spike_slab_logp(xtx, omega, xty, yty, nz, 0.75, post_conc + 1.0, 5.0)
  odds_ref = lp[(1, 1, 0, 0)] - lp[(1, 0, 0, 0)]
  odds_exact = lp[(1, 1, 0, 0, "exact")] - lp[(1, 0, 0, 0, "exact")]
  assert abs(odds_ref - odds_exact) < 5.0 / n
