"""Shared harness: run the reference's statistical tolerance tests through either backend.

backend="oracle": float64 CPU oracle behind the SAME host wrapper (data prep + impact
post-processing) -- this is how the oracle itself is pinned by the reference's tests.
backend="gpu": the product path (C-ABI -> HIP kernel).
"""
import numpy as np
import pandas as pd
import scipy.signal

from causalimpact import _model
from causalimpact import causalimpact_lib as lib
from causalimpact import data as cid
from oracle import ci_oracle as orc


def fit(backend, data, pre, post, *, seed, num_results, num_warmup=None, prior_level_sd=0.01,
        seasons=(), alpha=0.05, standardize=True, dtype=np.float32):
  inf = lib.InferenceOptions(num_results=num_results, num_warmup_steps=num_warmup)
  if backend == "gpu":
    return lib.fit_causalimpact(
        data, pre, post, alpha=alpha, seed=seed, inference_options=inf,
        data_options=lib.DataOptions(standardize_data=standardize, dtype=dtype),
        model_options=lib.ModelOptions(prior_level_sd=prior_level_sd, seasons=list(seasons)))
  ci = cid.CausalImpactData(data, pre, post, standardize_data=standardize, dtype=np.float64)
  design = None if ci.feature_ts is None else ci.feature_ts.values.astype(np.float64)
  n_after = ci.model_after_pre_data.shape[0]
  y = np.concatenate([ci.outcome_ts.time_series, np.full(n_after, np.nan)])
  mask = np.concatenate([ci.outcome_ts.is_missing, np.ones(n_after, bool)])
  spec = orc.default_spec(y, mask, design, prior_level_sd=prior_level_sd,
                          seasons=[(s.num_seasons, s.num_steps_per_season) for s in seasons])
  res = orc.fit_gibbs(y, mask, design, spec, num_results=inf.num_results,
                      num_warmup=inf.num_warmup_steps, seed=lib._sanitize_seed(seed))
  series, summary = lib._compute_impact(res["pred_mean"], res["trajectories"], ci, alpha)
  post_s = lib.CausalImpactPosteriorSamples(
      observation_noise_scale=res["obs_scale"], level_scale=res["level_scale"],
      level=res["level"], weights=res["weights"] if spec["P"] else None,
      seasonal_drift_scales=res["drift_scales"] if len(seasons) else None,
      seasonal_levels=res["seasonal"])
  return lib.CausalImpactAnalysis(series, summary, post_s)


def create_test_data(treat_amt, treat_index, num_timesteps=100, seed=0):
  """causalimpact_lib_test.py:35-45: ArmaProcess(ar=[1, .9]) == lfilter([1], [1, .9], e)."""
  rng = np.random.default_rng(seed)
  x = 100 + scipy.signal.lfilter([1.0], [1.0, 0.9], rng.normal(size=num_timesteps))
  y = 1.2 * x + rng.normal(size=num_timesteps)
  df = pd.DataFrame({"y": y, "x": x}, index=pd.date_range("2018-01-01", periods=num_timesteps,
                                                           freq="D"))
  df.loc[df.index > df.index[treat_index], "y"] += treat_amt
  return df


def load_datacsv(golden_dir):
  """causalimpact_lib_test.py:204-220."""
  import os
  df = pd.read_csv(os.path.join(golden_dir, "ref_testdata", "data.csv"))
  df = df.set_index(pd.to_datetime(df["t"])).drop(columns=["t"])
  df.loc[df.index[[1, 3, 7]], "y"] = np.nan
  return df
