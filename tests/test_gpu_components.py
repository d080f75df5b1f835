"""GPU parity of the kernel building blocks against the float64 CPU oracle.

Everything goes through the C-ABI (include/causalimpact_amd.h).  Tolerances are
float32-vs-float64 tolerances on identical specified random numbers, stated per test.
"""
import numpy as np
import pytest

from causalimpact import _native
from oracle import ci_oracle as orc

pytestmark = pytest.mark.gpu


def test_library_loaded_and_device_present():
  assert _native.device_count() >= 1


@pytest.mark.parametrize("site,sub", [(2, 0), (6, 0), (9, 3), (14, 0)])
def test_rng_stream_matches_oracle(site, sub):
  seed, chain, it, n = (123, 456), 5, 17, 256
  u, z1, z4, g = _native.test_rng(seed, chain, it, site, sub, n, alpha=40.5)
  uo = np.array([orc.uniform(seed, chain, it, site, sub, i) for i in range(n)])
  zo = np.array([orc.normal(seed, chain, it, site, sub, i) for i in range(n)])
  np.testing.assert_allclose(u, uo, atol=1e-7)           # float32 rounding of a float64 uniform
  np.testing.assert_allclose(z1, zo, atol=3e-5)          # f32 Box-Muller vs f64 Box-Muller
  np.testing.assert_array_equal(z1, z4)                  # 1-wide and 4-wide paths are one stream
  np.testing.assert_allclose(g, orc.gamma(40.5, seed, chain, it, site, sub), rtol=2e-6)   # f32 proposal normal on device


@pytest.mark.parametrize("alpha", [0.3, 1.0, 16.0, 366.0, 5012.5])
def test_gamma_draws_match_oracle(alpha):
  seed = (9, 8)
  for it in range(6):
    _, _, _, g = _native.test_rng(seed, 2, it, 3, 0, 4, alpha=alpha)
    np.testing.assert_allclose(g, orc.gamma(alpha, seed, 2, it, 3, 0), rtol=2e-6)


def _dk_case(T, has_slope, seed=(7, 11)):
  rng = np.random.default_rng(T + has_slope)
  resid = (rng.normal(size=T).cumsum() * 0.05 + rng.normal(size=T) * 0.4).astype(np.float32)
  mask = np.zeros(T, bool)
  mask[[1, 3, 7]] = True
  mask[int(0.7 * T):] = True
  spec = orc.default_spec(resid.astype(np.float64), mask, None, has_slope=bool(has_slope),
                          outcome_sd=1.0)
  return resid, mask, spec, seed


@pytest.mark.parametrize("has_slope", [0, 1])
@pytest.mark.parametrize("T", [37, 100, 300, 1000, 2000, 4000])
def test_dk_draw_matches_oracle(T, has_slope):
  resid, mask, spec, seed = _dk_case(T, has_slope)
  obs, lvl, slp = 0.45, 0.03, 0.004 if has_slope else 0.0
  pb = _native.make_problem(T=T, P=0, has_slope=has_slope, num_warmup=0, num_results=1,
                            seed=seed, chain_offset=3)
  params = _native.make_params([spec])
  got = _native.test_dk_draw(pb, params, resid, mask, obs, lvl, slp, it=5)
  ssm = orc.make_ssm(spec, mask, obs_scale=obs, level_scale=lvl, slope_scale=slp)
  want = orc.dk_draw(ssm, np.where(mask, 0.0, resid.astype(np.float64)), seed, chain=3, it=5)
  # float32 scans over <= 4096 steps vs float64 sequential recursions: 2e-3 absolute on the
  # O(1) fitted part; in the masked forecast tail of a local linear trend the draw is the
  # difference of O(50) components, so there the error is bounded relative to the posterior
  # spread of the draw itself (estimated from 12 oracle draws).
  spread = np.std([orc.dk_draw(ssm, np.where(mask, 0.0, resid.astype(np.float64)), seed,
                               chain=3, it=100 + i) for i in range(12)], axis=0)
  err = np.abs(got - want)
  assert np.isfinite(got).all()
  tol = 2e-3 + 2e-3 * spread
  worst = np.unravel_index(np.argmax(err / tol), err.shape)
  assert (err <= tol).all(), (worst, err[worst], want[worst], got[worst], spread[worst])


def test_dk_draw_observed_everywhere_and_strong_signal():
  # no missing data, large signal-to-noise: exercises the update branch at every step
  T = 513
  rng = np.random.default_rng(5)
  resid = rng.normal(size=T).cumsum().astype(np.float32)
  mask = np.zeros(T, bool)
  spec = orc.default_spec(resid.astype(np.float64), mask, None, has_slope=True, outcome_sd=1.0)
  pb = _native.make_problem(T=T, P=0, has_slope=1, num_warmup=0, num_results=1, seed=(1, 2))
  got = _native.test_dk_draw(pb, _native.make_params([spec]), resid, mask, 0.1, 1.0, 0.2, it=0)
  ssm = orc.make_ssm(spec, mask, obs_scale=0.1, level_scale=1.0, slope_scale=0.2)
  want = orc.dk_draw(ssm, resid.astype(np.float64), (1, 2), chain=0, it=0)
  assert np.abs(got - want).max() < 5e-3 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("T,p,has_slope", [(100, 0, 0), (100, 1, 1), (1000, 10, 1), (3000, 3, 0)])
def test_kalman_loglik_matches_oracle(T, p, has_slope):
  """device log-likelihood vs the oracle's (which is pinned to the
  dense multivariate-normal log-density in tests/test_oracle_math.py).  float32 filter and a
  float32 sum of <= 0.7 T terms: relative tolerance 2e-5, absolute 2e-3."""
  from causalimpact import _synthetic as syn
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 11)
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope))
  P = spec["P"]
  rng = np.random.default_rng(0)
  E = 7
  theta = np.zeros((E, 3 + P))
  theta[:, 0] = rng.uniform(0.2, 1.0, E)
  theta[:, 1] = rng.uniform(0.005, 0.2, E)
  theta[:, 2] = rng.uniform(0.001, 0.02, E) if has_slope else 0.0
  theta[:, 3:] = 0.3 * rng.normal(size=(E, P))
  pb = _native.make_problem(T=T, P=P, has_slope=has_slope, num_warmup=0, num_results=1)
  got = _native.kalman_loglik(pb, _native.make_params([spec]), y, mask, X, theta)
  for e in range(E):
    ssm = orc.make_ssm(spec, mask, obs_scale=theta[e, 0], level_scale=theta[e, 1],
                       slope_scale=theta[e, 2])
    resid = np.where(mask, 0.0, y) - (X @ theta[e, 3:] if P else 0.0)
    want = orc.kalman_loglik(ssm, resid)
    np.testing.assert_allclose(got[e], want, rtol=2e-5, atol=2e-3)


def _ll_setup(T, p, has_slope, E=5, seed=0):
  from causalimpact import _synthetic as syn
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 11)
  mask[[3, 9]] = True
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope))
  P = spec["P"]
  rng = np.random.default_rng(seed)
  theta = np.zeros((E, 3 + P))
  theta[:, 0] = rng.uniform(0.3, 0.9, E)
  theta[:, 1] = rng.uniform(0.02, 0.2, E)
  theta[:, 2] = rng.uniform(0.005, 0.03, E) if has_slope else 0.0
  theta[:, 3:] = 0.3 * rng.normal(size=(E, P))
  pb = _native.make_problem(T=T, P=P, has_slope=has_slope, num_warmup=0, num_results=1, seed=(4, 2))
  sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=64)
  return y, mask, X, spec, theta, sess


def _oracle_ll(spec, y, mask, X, th):
  ssm = orc.make_ssm(spec, mask, obs_scale=th[0], level_scale=th[1], slope_scale=th[2])
  resid = np.where(mask, 0.0, y) - (X @ th[3:] if spec["P"] else 0.0)
  return orc.kalman_loglik(ssm, resid)


@pytest.mark.parametrize("T,p,has_slope", [
    (80, 2, 1), (300, 0, 0), (1000, 10, 1),
    (1000, 40, 1),     # more than 12 covariates: further rounds of 16 sums, float4 rows of the design
    (998, 37, 0),      # T % 4 != 0: the design read row-scalar, ragged last chunk
    (600, 80, 1),      # round 5: more than 52 design columns on the log-likelihood path (limit 128)
])
def test_loglik_score_matches_finite_differences_of_oracle(T, p, has_slope):
  """Row H: device score (two suffix scans) vs central differences of the float64 oracle
  log-likelihood.  float32 recursions: 2 % relative on each component's scale."""
  y, mask, X, spec, theta, sess = _ll_setup(T, p, has_slope)
  ll, grad = sess.evaluate(theta)
  for e in range(theta.shape[0]):
    np.testing.assert_allclose(ll[e], _oracle_ll(spec, y, mask, X, theta[e]), rtol=2e-5, atol=2e-3)
    fd = np.zeros(theta.shape[1])
    for i in range(theta.shape[1]):
      if i == 2 and not has_slope:
        continue
      h = 1e-6 * max(1.0, abs(theta[e, i]))
      a, b = theta[e].copy(), theta[e].copy()
      a[i] += h
      b[i] -= h
      fd[i] = (_oracle_ll(spec, y, mask, X, a) - _oracle_ll(spec, y, mask, X, b)) / (2 * h)
    scale = np.maximum(np.abs(fd), 1e-2 * np.abs(fd).max())
    assert (np.abs(grad[e] - fd) <= 2e-2 * scale + 1e-2).all(), (grad[e], fd)
  sess.close()


def test_latent_draws_for_given_parameters_match_oracle():
  T, p, has_slope = 200, 3, 1
  y, mask, X, spec, theta, sess = _ll_setup(T, p, has_slope, E=3)
  out = sess.draw_latents(theta, seed=(4, 2), rng_chain=7, iter0=10)
  for e in range(3):
    th = theta[e]
    ssm = orc.make_ssm(spec, mask, obs_scale=th[0], level_scale=th[1], slope_scale=th[2])
    resid = np.where(mask, 0.0, y - X @ th[3:])
    want = orc.dk_draw(ssm, resid, (4, 2), chain=7, it=10 + e)
    np.testing.assert_allclose(out["level"][e], want[:, 0], atol=3e-3)
    np.testing.assert_allclose(out["slope"][e], want[:, 1], atol=3e-3)
    loc = want[:, 0] + X @ th[3:]
    np.testing.assert_allclose(out["loc"][e], loc, atol=3e-3)
    zp = np.array([orc.normal((4, 2), 7, 10 + e, orc.SITES["PRED"], 0, t) for t in range(T)])
    np.testing.assert_allclose(out["traj"][e], loc + th[0] * zp, atol=5e-3)
  sess.close()


@pytest.mark.parametrize("T,p,has_slope,seasons", [
    (120, 2, 0, ((7, 1),)),                      # weekly block
    (200, 3, 1, ((4, 3), (7, 1))),               # trend + two blocks
    (90, 0, 0, ((2, 5),)),                       # n = 2, no regression
    (300, 1, 0, ((6, (1, 1, 2, 1, 1, 3)),)),     # per-season lengths
    (5000, 2, 1, ()),                            # no block, T > 4096: the sequential route too
    (6000, 1, 0, ((7, 1),)),
    (37, 1, 1, ((3, 1),)),                       # shorter than the workgroup: most chunks empty
    (30000, 1, 1, ((7, 2),)),                    # 118 steps per thread, trend + block (d = 8)
])
@pytest.mark.parametrize("route", ["auto", "sequential"])
def test_sequential_loglik_and_score_match_the_oracle(T, p, has_slope, seasons, route):
  """Row H for seasonal models and long series against the oracle's ANALYTIC log-likelihood and
  score (ci_oracle_loglik_score, itself pinned to central differences in
  tests/test_oracle_hmc.py): l, dl/d(sigma_obs, sigma_level, sigma_slope, sigma_drift[K]) and
  dl/d beta = X'e.  route "auto": trend + one block of 2-7 seasons and long trend-only series take
  the TIME-PARALLEL scans (csrc/ci_wide_score.h: filter scan + one merged (r, N) suffix scan), the
  rest the sequential one-wavefront route (csrc/ci_score_seq.h); "sequential" forces the latter."""
  from causalimpact import _model
  from causalimpact import _synthetic as syn
  y, mask, X, _ = syn.make_sampler_inputs(T, max(p, 1), 21)
  X = X if p > 0 else None
  y = y + 0.6 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  mask = mask.copy()
  mask[[3, 9, T // 3]] = True
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  P, K = spec["P"], len(seasons)
  rng = np.random.default_rng(1)
  E = 4
  theta = np.zeros((E, 3 + K + P))
  theta[:, 0] = rng.uniform(0.3, 0.9, E)
  theta[:, 1] = rng.uniform(0.02, 0.2, E)
  theta[:, 2] = rng.uniform(0.005, 0.03, E) if has_slope else 0.0
  theta[:, 3:3 + K] = rng.uniform(0.01, 0.1, (E, K))
  theta[:, 3 + K:] = 0.3 * rng.normal(size=(E, P))
  pb = _native.make_problem(T=T, P=P, has_slope=has_slope, num_seasons=counts, num_warmup=0,
                            num_results=1,
                            flags=_native.FLAG_SEQUENTIAL_SEASONAL if route == "sequential" else 0)
  sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=8,
                               season_change=flg)
  wide = route == "auto" and K <= 1
  assert ("hmc_wide_kernel" in sess.kernel_name()) == wide, sess.kernel_name()
  ll, grad = sess.evaluate(theta)
  ll_only, _ = sess.evaluate(theta, want_grad=False)
  np.testing.assert_array_equal(ll, ll_only)
  sess.close()
  for e in range(E):
    th = theta[e]
    ssm = orc.make_ssm(spec, mask, obs_scale=th[0], level_scale=th[1], slope_scale=th[2],
                       drift_scale=th[3:3 + K])
    resid = np.where(mask, 0.0, y) - (X @ th[3 + K:] if P else 0.0)
    want_ll, ee, gs = orc.loglik_score(ssm, resid)
    np.testing.assert_allclose(ll[e], want_ll, rtol=3e-5, atol=3e-3)
    want = np.concatenate([gs[:3], gs[3:3 + K], (X.T @ ee) if P else np.zeros(0)])
    if not has_slope:
      want[2] = 0.0
    scale = np.maximum(np.abs(want), 1e-2 * np.abs(want).max())
    assert (np.abs(grad[e] - want) <= 2e-2 * scale + 1e-2).all(), (grad[e], want)
