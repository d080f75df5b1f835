"""The float64 Gibbs kernel (csrc/ci_gibbs64.h, `ci_fit_gibbs_f64`, DataOptions.dtype=float64)
against the float64 oracle: same algorithm, same random numbers, float64 on both sides, so the two
agree DRAW FOR DRAW over whole fits up to summation order -- 1e-8, not the 5e-3 of the float32
kernels.  This is the tightest pin of the device algorithm in the suite: inclusion patterns,
every scale, every weight, every latent path and predictive trajectory of every iteration."""
import numpy as np
import pandas as pd
import pytest

from causalimpact import _model, _native
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,p,has_slope,seasons,W,S", [
    (200, 4, 1, (), 10, 60),                    # trend + slope + spike-and-slab
    (150, 2, 0, ((7, 1),), 5, 40),              # weekly block
    (120, 0, 0, (), 5, 40),                     # no covariates: conjugate sigma_obs
    (180, 3, 1, ((4, 3), (7, 1)), 0, 25),       # two blocks + slope
    (300, 70, 0, (), 0, 12),                    # P = 71
    (5000, 2, 0, (), 0, 6),                     # long series
    # the time-parallel trend path (lane j owns an odd number of consecutive steps): fewer steps
    # than lanes, one step per lane, lengths that leave the last lanes empty or half full
    (13, 0, 0, (), 3, 20),
    (64, 1, 1, (), 3, 20),
    (65, 2, 1, (), 3, 20),
    (191, 0, 1, (), 3, 20),
    (1000, 3, 1, (), 0, 8),
    # the eight-wave trend kernel's regression routes: register-resident block with float64
    # transcendentals (P <= 16, here 11 and exactly 16 columns), dense sweeps beyond
    (400, 10, 1, (), 10, 40),
    (300, 15, 0, (), 5, 30),
    (250, 20, 1, (), 5, 20),
])
def test_float64_kernel_equals_the_oracle_draw_for_draw(T, p, has_slope, seasons, W, S):
  y, mask, X, _ = syn.make_sampler_inputs(T, max(p, 1), 31)
  X = X if p > 0 else None
  y = y + 0.5 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  mask = mask.copy()
  mask[[2, 11]] = True
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  K = len(seasons)
  C = 2
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_seasons=counts, num_warmup=W,
                            num_results=S, num_chains=C, chain_offset=3, seed=(7, 1))
  got = _native.fit_gibbs_f64(pb, y[None], mask[None], None if X is None else X[None], flg,
                              _native.make_params([spec]))
  assert all(v.dtype == np.float64 for v in got.values())
  for c in range(C):
    w = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=(7, 1), chain=3 + c)
    tol = dict(rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(got["observation_noise_scale"][0, c], w["obs_scale"], **tol)
    np.testing.assert_allclose(got["level_scale"][0, c], w["level_scale"], **tol)
    if has_slope:
      np.testing.assert_allclose(got["slope_scale"][0, c], w["slope_scale"], **tol)
      np.testing.assert_allclose(got["slope"][0, c], w["slope"], **tol)
    if spec["P"]:
      np.testing.assert_array_equal(got["weights"][0, c] != 0, w["weights"] != 0)
      np.testing.assert_allclose(got["weights"][0, c], w["weights"], **tol)
    if K:
      np.testing.assert_allclose(got["seasonal_drift_scales"][0, c], w["drift_scales"], **tol)
      np.testing.assert_allclose(got["seasonal_levels"][0, c], w["seasonal"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(got["level"][0, c], w["level"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(got["posterior_trajectories"][0, c], w["trajectories"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(got["posterior_means"][0, c], w["pred_mean"], rtol=1e-8, atol=1e-8)


def test_float32_and_float64_kernels_follow_the_same_chain():
  """Same random stream: the float32 latency kernel and the float64 kernel are one chain at two
  precisions (first iterations within float32 round-off, inclusion patterns equal)."""
  T, p = 400, 6
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 5)
  spec = _model.series_params(y, mask, X, has_slope=True)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=0, num_results=6, seed=(2, 2))
  a = _native.fit_gibbs(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  b = _native.fit_gibbs_f64(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  np.testing.assert_array_equal(a["weights"] != 0, b["weights"] != 0)
  np.testing.assert_allclose(a["weights"], b["weights"], atol=5e-3)
  np.testing.assert_allclose(a["level"], b["level"], atol=5e-3)
  np.testing.assert_allclose(a["observation_noise_scale"], b["observation_noise_scale"], rtol=5e-3)


def test_fit_causalimpact_float64_computes_in_float64():
  import causalimpact as ci
  rng = np.random.default_rng(11)
  n, start, effect = 100, 50, 5.0
  y = rng.normal(size=n, scale=0.0001)
  y[start:] += effect
  df = pd.DataFrame({"y": y}, index=pd.date_range("2018-01-01", periods=n, freq="D"))
  res = ci.fit_causalimpact(df, (df.index[0], df.index[start - 1]), (df.index[start], df.index[-1]),
                            seed=5, inference_options=ci.InferenceOptions(num_results=300),
                            data_options=ci.DataOptions(dtype=np.float64))
  assert res.posterior_samples.level.dtype == np.float64
  np.testing.assert_allclose(res.summary["abs_effect"], (effect, effect * (n - start)), rtol=1e-3, atol=1e-3)
