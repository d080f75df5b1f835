"""The oracle's constrained-seasonal block against an independent transcription of TFP's
*published construction* of `tfp.sts.Seasonal(constrain_mean_effect_to_zero=True)`
(reference call site: causalimpact_lib.py:477-489).

oracle/ci_oracle.c carries a seasonal block in closed form ((n-1) free effects: "shift up, last =
minus the sum", drift covariance (sigma/n)^2 11', drift residual n (r_t[1] - r_{t+1}[0]);
ci_oracle.c: apply_transition / propagate_cov / initial_moments and the scale step of
ci_oracle_fit_gibbs).  tests/test_oracle_math.py checks the oracle's recursions against a dense
Gaussian built from the SAME closed forms, so a misreading of TFP shared by both would pass.  Here
the block is written out the way TFP documents it, in the UNREDUCED n-effect space:

  * effects e in R^n, the current season's effect first; observation row (1, 0, ..., 0);
  * at the end of a season the effects are cycled (new[i] = old[i+1], new[n-1] = old[0]) and the
    effect that just ended -- now last -- receives the drift shock sigma * eta
    (SeasonalStateSpaceModel: transition noise scale diag (0, ..., 0, drift_scale));
  * the constrained model keeps the effects on the zero-mean subspace: with
    E = (I - 11'/n)[:-1, :]  ("effects_to_residuals") and R = pinv(E) ("residuals_to_effects") the
    state is r = E e, its transition E Perm R, its noise factor E L, its initial covariance
    v E E' (ConstrainedSeasonalStateSpaceModel).

Part 1 derives the reduced transition, noise covariance, initial covariance and observation row
from those matrices and compares them with the closed forms.  Part 2 builds the dense joint
Gaussian of a whole series IN EFFECT SPACE (n dims per block, singular covariance, no reduction
anywhere) and requires the oracle's Kalman log-likelihood, smoothed observed effects and smoothed
level to agree with it.  Part 3 recovers every drift shock of an oracle Durbin-Koopman draw by a
least-squares projection in effect space (no closed form) and requires the oracle's drift-scale
draw of that Gibbs iteration -- fed the same gamma variate -- to be the conjugate update on those
shocks.
"""
import numpy as np
import pytest

from oracle import ci_oracle as orc

SITE_DRIFT_SCALE, SITE_LEVEL_SCALE = 12, 10     # oracle/ci_oracle.h


def _tfp_block(n):
  """(E, R, Perm, C) of one block as TFP constructs them."""
  C = np.eye(n) - 1.0 / n
  E = C[:-1, :]
  R = np.linalg.pinv(E)
  Perm = np.zeros((n, n))
  for i in range(n - 1):
    Perm[i, i + 1] = 1.0
  Perm[n - 1, 0] = 1.0
  return E, R, Perm, C


@pytest.mark.parametrize("n", [2, 3, 4, 7, 12])
def test_reduced_forms_follow_from_the_published_construction(n):
  E, R, Perm, C = _tfp_block(n)
  # residuals -> effects appends minus the sum (the minimum-norm solution has zero mean)
  r = np.random.default_rng(n).normal(size=n - 1)
  np.testing.assert_allclose(R @ r, np.concatenate([r, [-r.sum()]]), atol=1e-12)
  # transition: shift up, last = -sum  (oracle: apply_transition)
  Tc = E @ Perm @ R
  B = np.zeros((n - 1, n - 1))
  if n > 2:
    B[:-1, 1:] = np.eye(n - 2)
  B[-1, :] = -1.0
  np.testing.assert_allclose(Tc, B, atol=1e-12)
  # noise: factor E L with L = diag(0, ..., 0, sigma)  ->  covariance (sigma/n)^2 11'  (propagate_cov)
  sigma = 0.37
  L = np.zeros((n, n))
  L[-1, -1] = sigma
  M = E @ L
  np.testing.assert_allclose(M @ M.T, (sigma / n) ** 2 * np.ones((n - 1, n - 1)), atol=1e-14)
  # initial covariance v E E' = v (I - 11'/n) on the free effects  (initial_moments)
  np.testing.assert_allclose(E @ E.T, np.eye(n - 1) - 1.0 / n, atol=1e-14)
  # observation row (1, 0, ..., 0) R = first free effect
  z = np.zeros(n)
  z[0] = 1.0
  want = np.zeros(n - 1)
  want[0] = 1.0
  np.testing.assert_allclose(z @ R, want, atol=1e-12)


def _effect_space_model(spec, mask, obs_scale, level_scale, slope_scale, drift_scale):
  """Dense joint Gaussian of the whole path with every block in its n-effect form: returns the
  prior mean / covariance of the stacked states, the observation matrix of the observed steps and
  index helpers.  Nothing here uses a reduced coordinate."""
  T, has_slope = spec["T"], spec["has_slope"]
  ns = spec["num_seasons"]
  tr = 1 + has_slope
  D = tr + sum(ns)
  offs, o = [], tr
  for n in ns:
    offs.append(o)
    o += n
  z = np.zeros(D)
  z[0] = 1.0
  for o in offs:
    z[o] = 1.0
  a1 = np.zeros(D)
  a1[0] = spec["init_level_loc"]
  P1 = np.zeros((D, D))
  P1[0, 0] = spec["init_level_scale"] ** 2
  if has_slope:
    P1[1, 1] = spec["init_slope_scale"] ** 2
  for k, n in enumerate(ns):
    _, _, _, C = _tfp_block(n)
    o = offs[k]
    # N(0, v I) effects projected onto the zero-mean subspace: v C C' = v C
    P1[o:o + n, o:o + n] = spec["init_seasonal_scale"] ** 2 * C
  Fs, Qs = [], []
  for t in range(T - 1):
    F = np.eye(D)
    Lq = np.zeros((D, D))                 # noise factor
    Lq[0, 0] = level_scale
    if has_slope:
      F[0, 1] = 1.0
      Lq[1, 1] = slope_scale
    for k, n in enumerate(ns):
      if spec["season_change"][k][t]:
        _, _, Perm, C = _tfp_block(n)
        o = offs[k]
        F[o:o + n, o:o + n] = C @ Perm     # (C Perm = Perm on the zero-mean subspace)
        Lb = np.zeros((n, n))
        Lb[-1, -1] = drift_scale[k]
        Lq[o:o + n, o:o + n] = C @ Lb
    Fs.append(F)
    Qs.append(Lq @ Lq.T)
  mean = np.zeros((T, D))
  cov = np.zeros((T, D, T, D))
  mean[0] = a1
  cov[0, :, 0, :] = P1
  for t in range(T - 1):
    mean[t + 1] = Fs[t] @ mean[t]
    cov[t + 1, :, t + 1, :] = Fs[t] @ cov[t, :, t, :] @ Fs[t].T + Qs[t]
    for s in range(t + 1):
      cov[t + 1, :, s, :] = Fs[t] @ cov[t, :, s, :]
      cov[s, :, t + 1, :] = cov[t + 1, :, s, :].T
  H = np.zeros((T, T * D))
  for t in range(T):
    H[t, t * D:(t + 1) * D] = z
  obs = ~np.asarray(mask, bool)
  return mean.reshape(T * D), cov.reshape(T * D, T * D), H[obs], D, offs


CASES = {
    "weekly": dict(has_slope=False, seasons=((7, 1),)),
    "two_blocks_ragged": dict(has_slope=False, seasons=((4, (2, 1, 1, 1)), (3, 1))),
    "trend_and_n2": dict(has_slope=True, seasons=((2, 3),)),
    "reference_test_model": dict(has_slope=False, seasons=((4, 1), (7, 2), (6, 1))),   # causalimpact_lib_test.py:738-752 shape
}


def _setup(case, T=31, seed=0):
  rng = np.random.default_rng(seed)
  y = rng.normal(size=T).cumsum() * 0.3 + rng.normal(size=T) + np.sin(np.arange(T))
  mask = np.zeros(T, bool)
  mask[[2, 5, 11]] = True
  mask[T - 7:] = True
  spec = orc.default_spec(y, mask, None, outcome_sd=1.3, **CASES[case])
  K = len(spec["num_seasons"])
  scales = dict(obs_scale=0.7, level_scale=0.21, slope_scale=0.05 if spec["has_slope"] else 0.0,
                drift_scale=[0.3, 0.17, 0.23][:K])
  return y, mask, spec, scales


@pytest.mark.parametrize("case", sorted(CASES))
def test_kalman_filter_and_smoother_match_the_effect_space_gaussian(case):
  y, mask, spec, sc = _setup(case)
  mean, cov, Ho, D, offs = _effect_space_model(spec, mask, **sc)
  obs = ~mask
  Sy = Ho @ cov @ Ho.T + sc["obs_scale"] ** 2 * np.eye(int(obs.sum()))
  resid = y[obs] - Ho @ mean
  _, logdet = np.linalg.slogdet(Sy)
  want_ll = -0.5 * (resid @ np.linalg.solve(Sy, resid) + logdet + obs.sum() * np.log(2 * np.pi))
  ssm = orc.make_ssm(spec, mask, **sc)
  np.testing.assert_allclose(orc.kalman_loglik(ssm, y), want_ll, rtol=1e-9, atol=1e-9)
  # smoothed level and OBSERVED effect of every block (what the fit emits), every time step
  pm = (mean + cov @ Ho.T @ np.linalg.solve(Sy, resid)).reshape(spec["T"], D)
  got = orc.smoothed_mean(ssm, y)                       # reduced coordinates
  tr = 1 + spec["has_slope"]
  np.testing.assert_allclose(got[:, 0], pm[:, 0], atol=1e-8)
  ro = tr
  for k, n in enumerate(spec["num_seasons"]):
    np.testing.assert_allclose(got[:, ro], pm[:, offs[k]], atol=1e-8, err_msg=f"block {k}")
    # and the other free effects are the next seasons' effects, in order
    np.testing.assert_allclose(got[:, ro:ro + n - 1], pm[:, offs[k]:offs[k] + n - 1], atol=1e-8)
    ro += n - 1


@pytest.mark.parametrize("case", ["weekly", "two_blocks_ragged", "trend_and_n2"])
def test_drift_scale_update_is_the_conjugate_draw_on_the_effect_space_shocks(case):
  """One oracle Gibbs iteration without covariates: the latent path is the Durbin-Koopman draw at
  the initial scales; the drift shocks of that path are recovered in effect space by projecting
  e_{t+1} - Perm e_t on the constrained shock direction C e_{n-1} (the definition, no closed
  form); the oracle's new drift scale must be min(sqrt((b0 + ss/2) / Gamma(a0 + m/2)), ub) with
  its own gamma variate.  Pins ci_oracle.c's residual (n (r_t[1] - r_{t+1}[0]); the n = 2 form)
  and the count of changes."""
  y, mask, spec, _ = _setup(case, T=64, seed=3)
  seed = (11, 5)
  res = orc.fit_gibbs(y, mask, None, spec, num_results=1, num_warmup=0, seed=seed)
  ssm = orc.make_ssm(spec, mask, obs_scale=spec["obs_scale0"], level_scale=spec["level_scale0"],
                     slope_scale=spec["slope_scale0"], drift_scale=spec["drift_scale0"])
  path = orc.dk_draw(ssm, y, seed, chain=0, it=0)        # [T, d] reduced coordinates
  np.testing.assert_allclose(res["level"][0], path[:, 0], atol=1e-12)
  T = spec["T"]
  ro = 1 + spec["has_slope"]
  for k, n in enumerate(spec["num_seasons"]):
    E, R, Perm, C = _tfp_block(n)
    dirn = C[:, n - 1]                                   # constrained image of a unit shock
    ss, m = 0.0, 0
    for t in range(T - 1):
      if not spec["season_change"][k][t]:
        continue
      e0, e1 = R @ path[t, ro:ro + n - 1], R @ path[t + 1, ro:ro + n - 1]
      delta = e1 - Perm @ e0
      eta = float(delta @ dirn) / float(dirn @ dirn)
      np.testing.assert_allclose(delta, eta * dirn, atol=1e-10)    # nothing but the shock moved
      ss += eta * eta
      m += 1
    g = orc.gamma(spec["drift_conc"] + 0.5 * m, seed, 0, 0, SITE_DRIFT_SCALE, k)
    want = min(np.sqrt((spec["drift_scale"] + 0.5 * ss) / g), spec["drift_ub"])
    np.testing.assert_allclose(res["drift_scales"][0, k], want, rtol=1e-10, err_msg=f"block {k}")
    ro += n - 1
  # the level scale of the same iteration, for completeness (increments of the trend)
  dl = np.diff(path[:, 0]) - (path[:-1, 1] if spec["has_slope"] else 0.0)
  g = orc.gamma(spec["level_conc"] + 0.5 * (T - 1), seed, 0, 0, SITE_LEVEL_SCALE, 0)
  want = min(np.sqrt((spec["level_scale"] + 0.5 * float(dl @ dl)) / g), spec["level_ub"])
  np.testing.assert_allclose(res["level_scale"][0], want, rtol=1e-10)
