"""`standardize_data=False`: the outcome is conditioned INTERNALLY (y -> (y - mu) / s) before the
float32 kernels see it and the draws are mapped back in float64
(causalimpact_lib._internal_conditioning).  That is only legitimate if the model is exactly
equivariant under the map; this file proves it on the float64 oracle: the chain on the raw series
and the chain on the conditioned series, driven by the same random numbers, are the same chain up
to the affine map -- inclusion patterns identical, every continuous output equal to ~1e-9 -- when
the conditioned chain is started at the image of the raw chain's start (level = 0  <->
level' = -mu / s).  The product starts the conditioned chain at level' = 0, i.e. at the pre-period
mean instead of at 0: another starting point of the SAME Markov chain (warm-up discards it)."""
import numpy as np
import pandas as pd
import pytest

from causalimpact import causalimpact_lib as lib
from causalimpact import data as cid
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc


@pytest.mark.parametrize("has_slope,seasons,p", [(False, (), 4), (True, (), 3), (False, ((7, 1),), 2),
                                                 (False, (), 0)])
def test_default_model_is_equivariant_under_affine_maps_of_the_outcome(has_slope, seasons, p):
  T, W, S = 140, 6, 25
  y0, mask, X, _ = syn.make_sampler_inputs(T, max(p, 1), 9)
  X = X if p > 0 else None
  rng = np.random.default_rng(3)
  raw = 100.0 + 3.7 * np.where(mask, np.nan, y0 + 0.05 * rng.normal(size=T))
  obs = raw[~mask]
  mu, s = float(obs.mean()), float(obs.std(ddof=1))
  cond = (raw - mu) / s
  kw = dict(has_slope=has_slope, seasons=list(seasons))
  a = orc.fit_gibbs(raw, mask, X, orc.default_spec(raw, mask, X, **kw), num_results=S, num_warmup=W,
                    seed=(2, 5))
  # the raw chain starts at level = 0 (:580-581), whose image is level' = -mu / s
  spec_c = orc.default_spec(cond, mask, X, **kw)
  spec_c["weights_prior_scale"] = s * s      # Omega (built from X alone) in the conditioned units
  if p > 0:
    # the reference clips the VARIANCE of the spike-and-slab branch at upper_bound = 1.2 sd (a
    # scale, :442-443): var <= 1.2 sd  <=>  var' <= 1.2 sd / s^2
    spec_c["obs_ub"] = spec_c["obs_ub"] / s
  d = 1 + int(has_slope) + sum(n - 1 for n, _ in seasons)
  start = np.zeros((T, d))
  start[:, 0] = -mu / s
  b = orc.fit_gibbs(cond, mask, X, spec_c, num_results=S, num_warmup=W, seed=(2, 5), latents0=start)
  if p > 0:
    np.testing.assert_array_equal(a["weights"] != 0, b["weights"] != 0)
    np.testing.assert_allclose(a["weights"], b["weights"] * s, rtol=1e-8, atol=1e-9)
  for k in ("obs_scale", "level_scale", "slope_scale", "drift_scales", "slope", "seasonal"):
    if k in a and np.size(a[k]):
      np.testing.assert_allclose(a[k], np.asarray(b[k]) * s, rtol=1e-8, atol=1e-9, err_msg=k)
  for k in ("level", "pred_mean", "trajectories"):
    np.testing.assert_allclose(a[k], b[k] * s + mu, rtol=1e-9, atol=1e-8, err_msg=k)


def test_conditioning_is_the_identity_for_standardised_data_and_affine_otherwise():
  n = 60
  y = 250.0 + 0.01 * np.arange(n) + 0.002 * np.random.default_rng(0).normal(size=n)
  df = pd.DataFrame({"y": y}, index=pd.date_range("2020-01-01", periods=n, freq="D"))
  std = cid.CausalImpactData(df, (df.index[0], df.index[39]), (df.index[40], df.index[-1]),
                             standardize_data=True)
  assert lib._internal_conditioning(std) == (0.0, 1.0)          # pylint: disable=protected-access
  raw = cid.CausalImpactData(df, (df.index[0], df.index[39]), (df.index[40], df.index[-1]),
                             standardize_data=False)
  mu, s = lib._internal_conditioning(raw)                        # pylint: disable=protected-access
  np.testing.assert_allclose(mu, y[:40].mean(), rtol=1e-14)
  np.testing.assert_allclose(s, y[:40].std(ddof=1), rtol=1e-14)
