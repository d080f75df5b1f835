"""Row H oracle (oracle/ci_oracle.c: ci_oracle_loglik_score, ci_oracle_hmc_logp,
ci_oracle_fit_hmc), pinned without a GPU: the score against central differences of
ci_oracle_kalman_loglik (itself pinned to the dense MVN log-density in test_oracle_math.py), the
log-posterior gradient of both regression priors against central differences, the warm-up
schedule, and the sampler against the exact Gibbs posterior of a model both target."""
import numpy as np
import pytest

from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc


def _data(T=120, p=3, seed=5):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, seed)
  mask = mask.copy()
  mask[[3, 17]] = True
  return y, mask, X


@pytest.mark.parametrize("has_slope,seasons", [(False, ()), (True, ()), (False, ((4, 3),)),
                                               (True, ((7, 1), (3, 2)))])
def test_score_matches_central_differences_of_the_loglik(has_slope, seasons):
  y, mask, X = _data()
  spec = orc.default_spec(y, mask, X, has_slope=has_slope, seasons=seasons)
  K = len(seasons)
  data = np.where(mask, 0, y - X @ np.array([0.1, -0.2, 0.05, 0.3]))
  so, sl, ss, dr = 0.6, 0.08, 0.03, [0.05, 0.02][:K]

  def ll(so, sl, ss, dr, data=data):
    return orc.kalman_loglik(orc.make_ssm(spec, mask, obs_scale=so, level_scale=sl, slope_scale=ss,
                                          drift_scale=dr), data)

  l0, e, gs = orc.loglik_score(orc.make_ssm(spec, mask, obs_scale=so, level_scale=sl,
                                            slope_scale=ss, drift_scale=dr), data)
  assert abs(l0 - ll(so, sl, ss, dr)) < 1e-9
  h = 1e-5
  fd = [(ll(so + h, sl, ss, dr) - ll(so - h, sl, ss, dr)) / (2 * h),
        (ll(so, sl + h, ss, dr) - ll(so, sl - h, ss, dr)) / (2 * h),
        (ll(so, sl, ss + h, dr) - ll(so, sl, ss - h, dr)) / (2 * h) if has_slope else 0.0]
  for k in range(K):
    d1, d2 = list(dr), list(dr)
    d1[k] += h
    d2[k] -= h
    fd.append((ll(so, sl, ss, d1) - ll(so, sl, ss, d2)) / (2 * h))
  np.testing.assert_allclose(gs, fd, rtol=1e-6, atol=1e-6)
  for t in (0, 5, 50, 70):            # e_t = -dl/d data_t
    d1, d2 = data.copy(), data.copy()
    d1[t] += h
    d2[t] -= h
    np.testing.assert_allclose(-e[t], (ll(so, sl, ss, dr, d1) - ll(so, sl, ss, dr, d2)) / (2 * h),
                               rtol=1e-6, atol=1e-7)
  assert (e[mask] == 0).all()


@pytest.mark.parametrize("prior", ["slab", "horseshoe"])
@pytest.mark.parametrize("has_slope,seasons", [(True, ()), (False, ((4, 3),))])
def test_log_posterior_gradient_matches_central_differences(prior, has_slope, seasons):
  y, mask, X = _data()
  spec = orc.default_spec(y, mask, X, has_slope=has_slope, seasons=seasons)
  P, K = spec["P"], len(seasons)
  dim = (3 * P + 2 if prior == "horseshoe" else P) + 2 + int(has_slope) + K
  rng = np.random.default_rng(1)
  th = 0.3 * rng.normal(size=dim)
  th[-(2 + int(has_slope) + K):] -= 1.5
  lp, g = orc.hmc_logp(y, mask, X, spec, th, prior=prior)
  fd = np.zeros(dim)
  for i in range(dim):
    a, b = th.copy(), th.copy()
    a[i] += 1e-5
    b[i] -= 1e-5
    fd[i] = (orc.hmc_logp(y, mask, X, spec, a, prior=prior)[0] -
             orc.hmc_logp(y, mask, X, spec, b, prior=prior)[0]) / 2e-5
  np.testing.assert_allclose(g, fd, rtol=1e-6, atol=1e-6)


def test_warmup_schedule():
  # (slow_begin, slow_end, first_end, base): Stan-style 75 / 50 / 25 when they fit, else 15 % /
  # 10 % buffers around one window; no mass adaptation for very short warm-ups
  assert orc.hmc_windows(1000) == (75, 950, 100, 25)
  assert orc.hmc_windows(112) == (16, 101, 101, 85)
  assert orc.hmc_windows(160) == (75, 110, 110, 25)
  assert orc.hmc_windows(10) == (10, 10, 10, 0)


def test_hmc_recovers_the_gibbs_posterior_when_both_target_the_same_model():
  """No covariates: Gibbs and HMC then differ only in the hard upper bounds on the scales
  (inactive here).  sigma_obs / sigma_level posterior means agree within Monte-Carlo error."""
  T = 150
  rng = np.random.default_rng(3)
  level = np.cumsum(0.15 * rng.normal(size=T))
  y = level + 0.5 * rng.normal(size=T)
  mask = np.zeros(T, bool)
  mask[110:] = True
  spec = orc.default_spec(y, mask, None, prior_level_sd=0.2)
  hm = [orc.fit_hmc(y, mask, None, spec, num_results=600, num_warmup=300, seed=(7, 1), chain=c)
        for c in range(4)]
  gb = [orc.fit_gibbs(y, mask, None, spec, num_results=3000, num_warmup=300, seed=(7, 1), chain=c,
                      want=("obs_scale", "level_scale", "level")) for c in range(4)]
  assert all(0.5 < h["accept_rate"] < 0.99 for h in hm)
  for col, key in ((0, "obs_scale"), (1, "level_scale")):
    h = np.array([x["draws"][:, col].mean() for x in hm])
    g = np.array([x[key].mean() for x in gb])
    se = np.hypot(h.std(ddof=1) / 2, g.std(ddof=1) / 2)
    assert abs(h.mean() - g.mean()) < 4 * se + 0.01 * g.mean(), (key, h, g)
  lh = np.mean([x["level"][:, 100] for x in hm])
  lg = np.mean([x["level"][:, 100] for x in gb])
  assert abs(lh - lg) < 0.06


def test_host_target_layout_for_seasonal_blocks_without_a_gpu():
  """`_hmc._Target` (the log posterior the mean-field surrogate of `_vi.py` is fitted to) with
  seasonal blocks, round 5: theta_u = (beta[P], log s_obs, log s_level[, log s_slope], log s_drift[K])
  is unpacked to the device layout (s_obs, s_level, s_slope, drift[K], beta) and the gradient comes
  back through the same map.  A fake session returns an analytic l(theta) = -1/2 |theta_dev - c|^2:
  the target's value and gradient must equal the closed form, drift priors (causalimpact_lib.py:
  472-474) included."""
  from causalimpact import _hmc      # (conftest.py puts the package on the path; no library call below)
  P, K = 3, 2
  rng = np.random.default_rng(0)
  c = rng.normal(size=3 + K + P)

  class FakeSession:
    def evaluate(self, dev, want_grad=True):
      d = dev - c
      return -0.5 * np.sum(d * d, axis=1), -d

  spec = dict(obs_conc=0.01, obs_scale=0.02, level_conc=16.0, level_scale=0.03, slope_conc=16.0,
              slope_scale=0.04, drift_conc=0.01, drift_scale=0.05)
  omega = np.diag([0.5, 0.25, 0.125])
  tgt = _hmc._Target(FakeSession(), spec, omega, P, True, K)
  assert tgt.dim == P + 3 + K
  th = rng.normal(size=(4, tgt.dim)) * 0.3
  lp, g = tgt(th)
  dev = _hmc._unpack(th, P, True, K)
  np.testing.assert_allclose(dev[:, 3 + K:], th[:, :P])                      # beta last
  np.testing.assert_allclose(dev[:, 3:3 + K], np.exp(th[:, P + 3:]))         # drift scales after the trend's
  ig = [(0.01, 0.02), (16.0, 0.03), (16.0, 0.04)] + [(0.01, 0.05)] * K

  def closed_form(t):
    d = _hmc._unpack(t[None], P, True, K)[0]
    v = -0.5 * np.sum((d - c) ** 2) - 0.5 * t[:P] @ omega @ t[:P]
    for k, (a, b) in enumerate(ig):
      v += -2.0 * a * t[P + k] - b * np.exp(-2.0 * t[P + k])
    return v

  for e in range(th.shape[0]):
    np.testing.assert_allclose(lp[e], closed_form(th[e]), rtol=1e-12)
    fd = np.zeros(tgt.dim)
    for i in range(tgt.dim):
      h = 1e-6
      a_, b_ = th[e].copy(), th[e].copy()
      a_[i] += h
      b_[i] -= h
      fd[i] = (closed_form(a_) - closed_form(b_)) / (2 * h)
    np.testing.assert_allclose(g[e], fd, rtol=1e-6, atol=1e-7)
