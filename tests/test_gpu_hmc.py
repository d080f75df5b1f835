"""HMC extension (no reference call site => parity unpinned).
Validated against this build's own Gibbs posterior on the same data: for P <= 3 the
spike-and-slab prior includes every feature, so both samplers target (nearly) the same
posterior -- the Gibbs one adds the hard upper bounds and the sigma^2-coupled slab."""
import numpy as np
import pandas as pd
import pytest

import ref_pins_common as rp
from causalimpact import causalimpact_lib as lib

pytestmark = pytest.mark.gpu


def test_hmc_matches_gibbs_on_quickstart_shape():
  # BASELINE cfg1 shape: T=100, 1 covariate, LocalLevel + regression, 100 HMC draws (x 8 chains)
  df = rp.create_test_data(10.0, 70, num_timesteps=100, seed=6)
  pre, post = (df.index[0], df.index[70]), (df.index[71], df.index[-1])
  gibbs = lib.fit_causalimpact(df, pre, post, seed=(1, 2),
                               inference_options=lib.InferenceOptions(num_results=500, num_chains=4))
  hmc = lib.fit_causalimpact(
      df, pre, post, seed=(1, 2),
      inference_options=lib.InferenceOptions(num_results=100, num_warmup_steps=150, num_chains=8,
                                             sampler="hmc"))
  g, h = gibbs.summary, hmc.summary
  np.testing.assert_allclose(h.loc["average", "abs_effect"], g.loc["average", "abs_effect"],
                             rtol=0.03)
  np.testing.assert_allclose(h.loc["average", "predicted"], g.loc["average", "predicted"],
                             rtol=0.01)
  # interval widths agree within Monte-Carlo error of 800 vs 2000 draws
  wg = g.loc["average", "abs_effect_upper"] - g.loc["average", "abs_effect_lower"]
  wh = h.loc["average", "abs_effect_upper"] - h.loc["average", "abs_effect_lower"]
  assert 0.75 < wh / wg < 1.3, (wh, wg)
  pg, ph = gibbs.posterior_samples, hmc.posterior_samples
  np.testing.assert_allclose(np.mean(ph.observation_noise_scale), np.mean(pg.observation_noise_scale),
                             rtol=0.08)
  # the covariate's weight is identified; intercept and level trade off along a ridge (both
  # samplers mix slowly there), so compare their identified sum instead
  np.testing.assert_allclose(np.mean(ph.weights[:, 0]), np.mean(pg.weights[:, 0]), atol=0.03)
  np.testing.assert_allclose(np.mean(ph.weights[:, 1]) + np.mean(ph.level[:, :70]),
                             np.mean(pg.weights[:, 1]) + np.mean(pg.level[:, :70]), atol=0.05)
  assert ph.level.shape == (800, 100) and ph.weights.shape == (800, 2)
  assert hmc.diagnostics["split_rhat"]["observation_noise_scale"] < 1.1


def test_hmc_rejects_unsupported_options():
  df = rp.create_test_data(5.0, 50, seed=1)
  with pytest.raises(NotImplementedError):
    lib.fit_causalimpact(df, (df.index[0], df.index[49]), (df.index[50], df.index[-1]),
                         model_options=lib.ModelOptions(seasons=[lib.Seasons(7)]),
                         inference_options=lib.InferenceOptions(num_results=10, sampler="hmc"))
  from causalimpact import _native
  big = rp.create_test_data(5.0, 4000, num_timesteps=5000, seed=1)
  with pytest.raises(_native.NativeError, match="exceeds the register-resident scans"):
    lib.fit_causalimpact(big, (big.index[0], big.index[3999]), (big.index[4000], big.index[-1]),
                         inference_options=lib.InferenceOptions(num_results=10, sampler="hmc"))
  with pytest.raises(ValueError, match="sampler must be"):
    lib.fit_causalimpact(df, (df.index[0], df.index[49]), (df.index[50], df.index[-1]),
                         inference_options=lib.InferenceOptions(num_results=10, sampler="nuts"))


def test_device_hmc_agrees_with_host_driven_hmc_and_splits_like_gibbs():
  """The on-device chain (csrc/ci_hmc.h) against the numpy-driven sampler it replaced (same
  target, one device call per leapfrog step), and the multi-GPU rule: chain c's draws do not
  depend on which launch ran it."""
  from causalimpact import _hmc, _model
  from causalimpact import _synthetic as syn
  T, p = 300, 3
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 12)
  spec = _model.series_params(y, mask, X, has_slope=True)
  kw = dict(has_slope=True, num_results=300, num_warmup=300, seed=(3, 4))
  dev = _hmc.fit_hmc(y, mask, X, spec, num_chains=6, **kw)
  host = _hmc.fit_hmc_host(y, mask, X, spec, num_chains=6, **kw)
  assert (dev["hmc_accept_rate"] > 0.5).all() and (dev["hmc_accept_rate"] < 0.99).all()
  for key, tol in (("observation_noise_scale", 0.05), ("level_scale", 0.25)):
    np.testing.assert_allclose(dev[key].mean(), host[key].mean(), rtol=tol, err_msg=key)
  np.testing.assert_allclose(dev["weights"].mean(axis=(0, 1, 2))[:p],
                             host["weights"].mean(axis=(0, 1, 2))[:p], atol=0.03)
  sd_d, sd_h = dev["weights"][0, :, :, 0].std(), host["weights"][0, :, :, 0].std()
  assert 0.7 < sd_d / sd_h < 1.4
  np.testing.assert_allclose(dev["posterior_means"].mean(axis=1), host["posterior_means"].mean(axis=1),
                             atol=0.1)
  # chains {0..5} in one launch == chains {0,1,2} and {3,4,5} in two launches
  a = _hmc.fit_hmc(y, mask, X, spec, num_chains=3, chain_offset=0, **kw)
  b = _hmc.fit_hmc(y, mask, X, spec, num_chains=3, chain_offset=3, **kw)
  for key in ("observation_noise_scale", "weights", "level", "posterior_trajectories"):
    np.testing.assert_array_equal(np.concatenate([a[key], b[key]], axis=1), dev[key], err_msg=key)


def test_surrogate_posterior_tracks_the_hmc_posterior_and_can_initialise_it():
  """`build_factored_surrogate_posterior` analogue (mean-field VI on the device log-likelihood /
  score): its mean sits inside the HMC posterior, its spread is not larger (mean-field
  under-dispersion is expected), and chains started from its draws sample the same posterior."""
  from causalimpact import _hmc, _model, _vi
  from causalimpact import _synthetic as syn
  T, p = 300, 3
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 12)
  spec = _model.series_params(y, mask, X, has_slope=False)
  vi = _vi.fit_surrogate_posterior(y, mask, X, spec, has_slope=False, seed=(3, 4))
  assert vi["elbo"][-50:].mean() > vi["elbo"][:20].mean()
  kw = dict(has_slope=False, num_results=300, num_warmup=300, num_chains=6, seed=(3, 4))
  hmc = _hmc.fit_hmc(y, mask, X, spec, **kw)
  hmc_vi = _hmc.fit_hmc(y, mask, X, spec, init="vi", **kw)
  w_mean = hmc["weights"].mean(axis=(0, 1, 2))
  w_sd = hmc["weights"].std(axis=(0, 1, 2))
  np.testing.assert_allclose(vi["mean"][:p], w_mean[:p], atol=3 * w_sd[:p].max() + 0.02)
  np.testing.assert_allclose(np.exp(vi["mean"][p + 1]), hmc["observation_noise_scale"].mean(), rtol=0.1)
  assert (np.exp(vi["log_sd"][:p]) < 2.0 * w_sd[:p] + 0.01).all()
  np.testing.assert_allclose(hmc_vi["weights"].mean(axis=(0, 1, 2))[:p], w_mean[:p], atol=0.03)
  np.testing.assert_allclose(hmc_vi["observation_noise_scale"].mean(),
                             hmc["observation_noise_scale"].mean(), rtol=0.05)
  assert (hmc_vi["hmc_accept_rate"] > 0.5).all()
