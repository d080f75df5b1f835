"""`standardize_data=False` with a LARGE offset (1e7 + N(0, 1)): every route that pools the draws on
the host -- DataOptions.dtype=float64, sampler="hmc", two device shares -- must keep the precision
the internal conditioning was added for.  The on-device summary reads float32 trajectories; it gets
them in the sampler's INTERNAL units with the map to the caller's scale folded into (scale, shift)
in float64 (`causalimpact_lib._run_sampler`).  Summarising the rescaled draws would quantise the
bands to the float32 spacing at 1e7 (= 1.0) -- as large as the noise itself.  Checked against the
host post-processing of the same draws in float64 (`summarize_on_device=False`; reference
:635-1093)."""
import numpy as np
import pandas as pd
import pytest

import causalimpact as ci

pytestmark = pytest.mark.gpu


def _frame(n=120, seed=4):
  rng = np.random.default_rng(seed)
  x = 100.0 + np.cumsum(rng.normal(size=n)) * 0.3
  y = 1e7 + 0.8 * (x - 100.0) + rng.normal(size=n)
  y[90:] += 3.0
  return pd.DataFrame({"y": y, "x": x}, index=pd.date_range("2021-01-01", periods=n, freq="D"))


@pytest.mark.parametrize("route", ["float64", "hmc", "two_shares"])
def test_pooled_summaries_keep_float64_precision_at_a_large_offset(route):
  df = _frame()
  pre, post = (df.index[0], df.index[89]), (df.index[90], df.index[-1])
  data_opts = dict(standardize_data=False)
  inf = dict(num_results=300, num_warmup_steps=60)
  if route == "float64":
    data_opts["dtype"] = np.float64
  elif route == "hmc":
    inf.update(sampler="hmc", num_chains=2, num_results=100, num_warmup_steps=100)
  else:
    inf.update(num_chains=2, devices=[0, 0])
  out = {}
  for on_dev in (True, False):
    res = ci.fit_causalimpact(df, pre, post, seed=(1, 2), data_options=ci.DataOptions(**data_opts),
                              inference_options=ci.InferenceOptions(summarize_on_device=on_dev, **inf))
    out[on_dev] = res
  a, b = out[True].series, out[False].series
  # the bands are order statistics of the same draws: equal up to the float32 rounding of O(1)
  # internal values times the conditioning scale (~1e-6), not up to the float32 spacing at 1e7 (1.0)
  for col in ("posterior_lower", "posterior_upper", "point_effects_lower", "point_effects_upper"):
    np.testing.assert_allclose(a[col].to_numpy(), b[col].to_numpy(), rtol=0, atol=2e-4, err_msg=col)
  for col in ("cumulative_effects_lower", "cumulative_effects_upper"):
    np.testing.assert_allclose(a[col].to_numpy(), b[col].to_numpy(), rtol=0, atol=5e-3, err_msg=col)
  sa, sb = out[True].summary, out[False].summary
  np.testing.assert_allclose(sa["abs_effect"].to_numpy(), sb["abs_effect"].to_numpy(), atol=2e-4)
  assert 1.5 < float(sa.loc["average", "abs_effect"]) < 4.5
