"""Size-independent properties of the whole pipeline at BASELINE cfg2's size (T=1000, 10
covariates, local linear trend, 8 chains): things that must hold whatever the sampler draws.
With standardize_data=True an affine change of units of the outcome, or of any covariate, leaves
the standardised model inputs unchanged (up to float64 round-off before the float32 cast), so the
Gibbs draws are the same and every output transforms exactly like the data."""
import numpy as np
import pandas as pd
import pytest

import causalimpact as ci
from causalimpact import _synthetic as syn

pytestmark = pytest.mark.gpu

T, P_COV = 1000, 10
OPTS = dict(num_results=300, num_chains=8)


def _frame(scale_y=1.0, shift_y=0.0, scale_x=None):
  y, X = syn.make_raw_series(T, P_COV, 5)
  if scale_x is not None:
    X = X * scale_x[None, :] + 3.0
  return pd.DataFrame(np.column_stack([scale_y * y + shift_y, X]),
                      columns=["y"] + [f"x{j}" for j in range(P_COV)])


def _fit(df):
  return ci.fit_causalimpact(df, (0, 699), (700, 999), seed=17,
                             inference_options=ci.InferenceOptions(**OPTS),
                             model_options=ci.ModelOptions(local_linear_trend=True))


def test_outcome_units_do_not_matter():
  base = _fit(_frame())
  a, b = 3.5, -120.0
  moved = _fit(_frame(scale_y=a, shift_y=b))
  for col in ("posterior_mean", "posterior_lower", "posterior_upper"):
    np.testing.assert_allclose(moved.series[col], a * base.series[col] + b, rtol=2e-4, atol=1e-3)
  for col in ("point_effects_mean", "point_effects_lower", "cumulative_effects_upper"):
    np.testing.assert_allclose(moved.series[col].to_numpy()[700:], a * base.series[col].to_numpy()[700:],
                               rtol=2e-3, atol=2e-2)
  # relative effects and the tail probability carry no units
  np.testing.assert_allclose(moved.summary["p_value"], base.summary["p_value"], atol=2e-3)
  np.testing.assert_allclose(moved.summary["abs_effect"], a * base.summary["abs_effect"], rtol=2e-3)
  np.testing.assert_allclose(moved.posterior_samples.weights, base.posterior_samples.weights, atol=2e-4)


def test_covariate_units_do_not_matter():
  base = _fit(_frame())
  rng = np.random.default_rng(0)
  moved = _fit(_frame(scale_x=rng.uniform(0.1, 50.0, P_COV)))
  np.testing.assert_allclose(moved.series["posterior_mean"], base.series["posterior_mean"], rtol=2e-4,
                             atol=1e-3)
  np.testing.assert_allclose(moved.summary.to_numpy(float), base.summary.to_numpy(float), rtol=5e-3,
                             atol=2e-3)
  np.testing.assert_array_equal(moved.posterior_samples.weights != 0, base.posterior_samples.weights != 0)


def test_draws_do_not_depend_on_what_follows_the_post_period():
  """Rows after the post-period are forecast steps like any other (data.py:107-112): dropping
  them cannot change the pre- and post-period results by more than Monte-Carlo error (the
  regression block sees the prior precision of ALL design rows, :458-459, so the draws are not
  identical)."""
  df = _frame()
  full = ci.fit_causalimpact(df, (0, 699), (700, 899), seed=4, inference_options=ci.InferenceOptions(
      num_results=100, num_chains=2))
  cut = ci.fit_causalimpact(df.iloc[:900], (0, 699), (700, 899), seed=4,
                            inference_options=ci.InferenceOptions(num_results=100, num_chains=2))
  np.testing.assert_allclose(full.summary["abs_effect"], cut.summary["abs_effect"], rtol=0.05)
  assert full.series.shape[0] == 1000 and cut.series.shape[0] == 900
  assert full.series["point_effects_mean"].iloc[900:].isna().all()
