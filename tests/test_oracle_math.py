"""Pins the CPU oracle against TFP-independent exact mathematics.

* Philox4x32-10 known-answer vectors (Random123 kat_vectors).
* Kalman log-likelihood == dense multivariate-normal log-density.
* Smoothed means == dense Gaussian conditional means.
* Durbin-Koopman draws: empirical mean/cov == dense posterior mean/cov.
* Spike-and-slab collapsed log posterior == direct evaluation of
  Scott & Varian (2013) eq. 8 as TFP states it.
* Gamma / normal / uniform streams pass KS tests.
"""
import numpy as np
import pytest
import scipy.linalg
import scipy.stats

from oracle import ci_oracle as orc


def _dense_model(spec, mask, obs_scale, level_scale, slope_scale=0.0, drift_scale=()):
  """Joint Gaussian of (x_0..x_{T-1}) built by explicit matrix recursion."""
  T, K = spec["T"], len(spec["num_seasons"])
  has_slope = spec["has_slope"]
  d = 1 + has_slope + sum(n - 1 for n in spec["num_seasons"])
  offs, o = [], 1 + has_slope
  for n in spec["num_seasons"]:
    offs.append(o)
    o += n - 1
  Z = np.zeros(d)
  Z[0] = 1.0
  for o in offs:
    Z[o] = 1.0
  a1 = np.zeros(d)
  a1[0] = spec["init_level_loc"]
  P1 = np.zeros((d, d))
  P1[0, 0] = spec["init_level_scale"] ** 2
  if has_slope:
    P1[1, 1] = spec["init_slope_scale"] ** 2
  for k, n in enumerate(spec["num_seasons"]):
    o = offs[k]
    P1[o:o + n - 1, o:o + n - 1] = spec["init_seasonal_scale"] ** 2 * (np.eye(n - 1) - 1.0 / n)
  means = [a1]
  # cov[(s,t)] blocks via recursion: x_{t+1} = F_t x_t + w_t
  Fs, Qs = [], []
  for t in range(T - 1):
    F = np.eye(d)
    Q = np.zeros((d, d))
    if has_slope:
      F[0, 1] = 1.0
      Q[1, 1] = slope_scale ** 2
    Q[0, 0] = level_scale ** 2
    for k, n in enumerate(spec["num_seasons"]):
      if spec["season_change"][k][t]:
        o = offs[k]
        B = np.zeros((n - 1, n - 1))
        B[:-1, 1:] = np.eye(n - 2)
        B[-1, :] = -1.0
        F[o:o + n - 1, o:o + n - 1] = B
        Q[o:o + n - 1, o:o + n - 1] = (drift_scale[k] / n) ** 2
    Fs.append(F)
    Qs.append(Q)
  mean = np.zeros((T, d))
  cov = np.zeros((T, d, T, d))
  mean[0] = a1
  cov[0, :, 0, :] = P1
  for t in range(T - 1):
    mean[t + 1] = Fs[t] @ mean[t]
    cov[t + 1, :, t + 1, :] = Fs[t] @ cov[t, :, t, :] @ Fs[t].T + Qs[t]
    for s in range(t + 1):
      cov[t + 1, :, s, :] = Fs[t] @ cov[t, :, s, :]
      cov[s, :, t + 1, :] = cov[t + 1, :, s, :].T
  mean = mean.reshape(T * d)
  cov = cov.reshape(T * d, T * d)
  H = np.zeros((T, T * d))
  for t in range(T):
    H[t, t * d:(t + 1) * d] = Z
  obs = ~np.asarray(mask, bool)
  Ho = H[obs]
  Sy = Ho @ cov @ Ho.T + obs_scale ** 2 * np.eye(obs.sum())
  return mean, cov, Ho, Sy, d


def _posterior(mean, cov, Ho, Sy, yobs):
  G = cov @ Ho.T @ np.linalg.inv(Sy)
  pm = mean + G @ (yobs - Ho @ mean)
  pc = cov - G @ Ho @ cov
  return pm, pc


CASES = {
    "local_level": dict(has_slope=False, seasons=()),
    "linear_trend": dict(has_slope=True, seasons=()),
    "seasonal": dict(has_slope=False, seasons=((4, (2, 1, 1, 1)), (3, 1))),
    "trend_seasonal2": dict(has_slope=True, seasons=((2, 3),)),
}


def _setup(case, T=23, seed=0):
  rng = np.random.default_rng(seed)
  y = rng.normal(size=T).cumsum() * 0.3 + rng.normal(size=T)
  mask = np.zeros(T, bool)
  mask[[2, 5]] = True
  mask[T - 6:] = True
  spec = orc.default_spec(y, mask, None, outcome_sd=1.3, **CASES[case])
  K = len(spec["num_seasons"])
  scales = dict(obs_scale=0.7, level_scale=0.21, slope_scale=0.05 if spec["has_slope"] else 0.0,
                drift_scale=[0.3, 0.17][:K])
  return y, mask, spec, scales


def test_philox_known_answers():
  # Random123 kat_vectors: philox4x32 10 rounds
  np.testing.assert_array_equal(
      orc.philox([0, 0, 0, 0], [0, 0]),
      np.array([0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8], np.uint32))
  np.testing.assert_array_equal(
      orc.philox([0xffffffff] * 4, [0xffffffff] * 2),
      np.array([0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd], np.uint32))
  np.testing.assert_array_equal(
      orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]),
      np.array([0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1], np.uint32))


@pytest.mark.parametrize("case", list(CASES))
def test_kalman_loglik_matches_dense(case):
  y, mask, spec, sc = _setup(case)
  mean, cov, Ho, Sy, _ = _dense_model(spec, mask, **sc)
  yobs = y[~mask]
  dense = scipy.stats.multivariate_normal(Ho @ mean, Sy).logpdf(yobs)
  ssm = orc.make_ssm(spec, mask, **sc)
  np.testing.assert_allclose(orc.kalman_loglik(ssm, y), dense, rtol=1e-10)


@pytest.mark.parametrize("case", list(CASES))
def test_smoothed_mean_matches_dense(case):
  y, mask, spec, sc = _setup(case)
  mean, cov, Ho, Sy, d = _dense_model(spec, mask, **sc)
  pm, _ = _posterior(mean, cov, Ho, Sy, y[~mask])
  ssm = orc.make_ssm(spec, mask, **sc)
  np.testing.assert_allclose(orc.smoothed_mean(ssm, y), pm.reshape(-1, d), rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("case", list(CASES))
def test_dk_draws_match_dense_posterior(case):
  y, mask, spec, sc = _setup(case, T=14)
  mean, cov, Ho, Sy, d = _dense_model(spec, mask, **sc)
  pm, pc = _posterior(mean, cov, Ho, Sy, y[~mask])
  ssm = orc.make_ssm(spec, mask, **sc)
  n = 6000
  draws = np.stack([orc.dk_draw(ssm, y, seed=(3, 4), chain=1, it=i).ravel() for i in range(n)])
  sd = np.sqrt(np.maximum(np.diag(pc), 1e-300))
  ok = np.diag(pc) > 1e-12
  zmean = (draws.mean(0) - pm)[ok] / (sd[ok] / np.sqrt(n))
  assert np.abs(zmean).max() < 4.5
  emp = np.cov(draws.T)
  scale = np.sqrt(np.outer(np.diag(pc), np.diag(pc))) + 1e-12
  assert np.abs((emp - pc) / scale).max() < 0.08
  # degenerate (zero-variance) directions must be reproduced exactly
  if (~ok).any():
    np.testing.assert_allclose(draws[:, ~ok], np.broadcast_to(pm[~ok], draws[:, ~ok].shape),
                               atol=1e-8)


def test_spike_slab_logp_matches_direct_formula():
  rng = np.random.default_rng(1)
  T, P = 40, 5
  X = rng.normal(size=(T, P))
  y = X[:, 0] * 1.5 + rng.normal(size=T)
  xtx = X.T @ X
  omega = 0.01 * (0.5 * xtx + 0.5 * np.diag(np.diag(xtx))) / T
  xty, yty = X.T @ y, y @ y
  a0, b0, pi = 25.0, 5.0, 0.6
  a_post = a0 + T / 2
  for bits in range(2 ** P):
    nz = np.array([(bits >> j) & 1 for j in range(P)], bool)
    M = (omega + xtx)[np.ix_(nz, nz)]
    Om = omega[np.ix_(nz, nz)]
    b = xty[nz]
    quad = b @ np.linalg.solve(M, b) if nz.any() else 0.0
    expect = (0.5 * np.linalg.slogdet(Om)[1] - 0.5 * np.linalg.slogdet(M)[1]
              + nz.sum() * np.log(pi) + (~nz).sum() * np.log1p(-pi)
              - (a_post - 1.0) * np.log(2.0 * (b0 + 0.5 * (yty - quad))))
    got = orc.spike_slab_logp(xtx, omega, xty, yty, nz, pi, a_post, b0)
    np.testing.assert_allclose(got, expect, rtol=1e-10)


def test_rng_streams_are_distributed_correctly():
  seed = (11, 22)
  u = np.array([orc.uniform(seed, 0, 7, orc.SITES["FLIP"], 0, i) for i in range(4000)])
  z = np.array([orc.normal(seed, 0, 7, orc.SITES["PRED"], 0, i) for i in range(4000)])
  assert scipy.stats.kstest(u, "uniform").pvalue > 1e-3
  assert scipy.stats.kstest(z, "norm").pvalue > 1e-3
  assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 0.06
  for alpha in (0.3, 1.0, 16.5, 375.0):
    g = np.array([orc.gamma(alpha, seed, 0, i, orc.SITES["OBSVAR"]) for i in range(3000)])
    assert scipy.stats.kstest(g, "gamma", args=(alpha,)).pvalue > 1e-3
