"""Generates the golden fixtures in this directory from the REAL reference.

Run in the build container only (needs /root/reference; the GPU box never has it):
    python tests/golden/make_golden.py
TensorFlow / TFP / altair are absent here, so the reference's pandas code is imported
under import-only stubs.  Only data is written: inputs and the
reference's outputs for data preparation (CausalImpactData), impact post-processing
(_compute_impact) and summary() text.  No reference source is copied.
"""
import collections
import json
import os
import sys
import types
import zlib

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _install_stubs():
  tf = types.ModuleType("tensorflow")
  tf.float32, tf.float64 = np.float32, np.float64
  tf.dtypes = types.SimpleNamespace(DType=type)
  tf.Tensor = np.ndarray
  tf.types = types.SimpleNamespace(experimental=types.SimpleNamespace(TensorLike=object))
  tf.function = lambda **kw: (lambda f: f)
  tf.convert_to_tensor = lambda v, dtype=None: np.asarray(v, dtype=dtype)
  tf.math = types.SimpleNamespace(is_nan=np.isnan)
  tfp = types.ModuleType("tensorflow_probability")
  tfp.sts = types.SimpleNamespace(
      MaskedTimeSeries=collections.namedtuple("MaskedTimeSeries", ["time_series", "is_missing"]),
      StructuralTimeSeries=object)
  tfp.bijectors = types.SimpleNamespace()
  tfp.distributions = types.SimpleNamespace()
  mods = {"tensorflow": tf, "tensorflow_probability": tfp}
  for name in ("tensorflow_probability.python", "tensorflow_probability.python.experimental",
               "tensorflow_probability.python.experimental.distributions",
               "tensorflow_probability.python.experimental.sts_gibbs",
               "tensorflow_probability.python.experimental.sts_gibbs.gibbs_sampler",
               "tensorflow_probability.python.internal",
               "tensorflow_probability.python.internal.prefer_static", "altair"):
    mods[name] = types.ModuleType(name)
  mods["altair"].Chart = object
  mods["tensorflow_probability.python.experimental.distributions"].MultivariateNormalPrecisionFactorLinearOperator = object
  gs = mods["tensorflow_probability.python.experimental.sts_gibbs.gibbs_sampler"]
  gs.GibbsSamplerState = collections.namedtuple("GibbsSamplerState", ["x"])
  mods["tensorflow_probability.python.experimental.sts_gibbs"].gibbs_sampler = gs
  mods["tensorflow_probability.python.internal"].prefer_static = mods[
      "tensorflow_probability.python.internal.prefer_static"]
  sys.modules.update(mods)
  sys.path.insert(0, REF)


def _frame_to_json(df: pd.DataFrame):
  out = {"columns": [str(c) for c in df.columns], "index": [str(i) for i in df.index], "data": {}}
  for c in df.columns:
    col = df[c]
    if np.issubdtype(col.dtype, np.datetime64):
      out["data"][str(c)] = [str(v) for v in col]
    else:
      out["data"][str(c)] = [None if (isinstance(v, float) and np.isnan(v)) else float(v) for v in col]
  return out


def _cases():
  """(name, dataframe, pre_period, post_period, standardize, alpha)."""
  df = pd.read_csv(os.path.join(HERE, "ref_testdata", "data.csv"))
  df = df.set_index(pd.to_datetime(df["t"])).drop(columns=["t"])
  df.loc[df.index[[1, 3, 7]], "y"] = np.nan
  yield ("datacsv_nan", df, (df.index[0], df.index[59]), (df.index[60], df.index[-1]), True, 0.05)
  yield ("datacsv_gap_tail_nostd", df, (df.index[2], df.index[49]), (df.index[55], df.index[80]),
         False, 0.1)
  rng = np.random.default_rng(0)
  n = 120
  x = 100 + rng.normal(size=n).cumsum()
  d2 = pd.DataFrame({"y": 1.2 * x + rng.normal(size=n), "x": x, "const": 3.0})
  d2.loc[80:, "y"] += 5
  yield ("int_index_constcol", d2, (0, 79), (80, 119), True, 0.05)
  d3 = pd.DataFrame({"y": rng.normal(size=60) + 10.0},
                    index=pd.date_range("2018-01-01", periods=60, freq="D"))
  yield ("no_covariates_str_periods", d3, ("2018-01-01", "2018-02-09"),
         ("2018-02-10", "2018-03-01"), True, 0.2)


def main():
  _install_stubs()
  import causalimpact  # the reference, under stubs
  from causalimpact import causalimpact_lib as ref_lib
  from causalimpact import data as ref_data
  import importlib
  ref_summary = importlib.import_module('causalimpact.summary')
  assert causalimpact.__version__ == "0.2.0"
  index = []
  for name, df, pre, post, std, alpha in _cases():
    ci_data = ref_data.CausalImpactData(df, pre, post, standardize_data=std, dtype=np.float64)
    T = ci_data.model_pre_data.shape[0] + ci_data.model_after_pre_data.shape[0]
    S = 25
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    # fixed fake sampler outputs (standardised scale): deterministic, seeded
    base = np.linspace(-0.5, 0.7, T)
    means = base + 0.05 * np.sin(np.arange(T))
    traj = base[None, :] + 0.3 * rng.normal(size=(S, T))
    series, summ = ref_lib._compute_impact(means, traj, ci_data, alpha)
    prep = dict(
        y=np.asarray(ci_data.outcome_ts.time_series, dtype=np.float64),
        is_missing=np.asarray(ci_data.outcome_ts.is_missing, dtype=bool),
        feature_values=(np.zeros((T, 0)) if ci_data.feature_ts is None
                        else ci_data.feature_ts.values.astype(np.float64)),
        feature_columns=np.array([] if ci_data.feature_ts is None
                                 else [str(c) for c in ci_data.feature_ts.columns]),
        outcome_mean=np.float64(np.nan if ci_data.outcome_scaler is None
                                else ci_data.outcome_scaler.mean_),
        outcome_std=np.float64(np.nan if ci_data.outcome_scaler is None
                               else ci_data.outcome_scaler.stddev_),
        num_steps_forecast=np.int64(ci_data.num_steps_forecast),
        fake_means=means, fake_trajectories=traj)
    np.savez(os.path.join(HERE, f"{name}_prep.npz"), **prep)
    df.to_csv(os.path.join(HERE, f"{name}_input.csv"), index=True,
              date_format="%Y-%m-%d %H:%M:%S")
    analysis = ref_lib.CausalImpactAnalysis(series, summ, None)
    meta = dict(
        name=name, standardize=std, alpha=alpha,
        index_kind="datetime" if isinstance(df.index, pd.DatetimeIndex) else "int",
        pre_period_in=[str(p) for p in pre], post_period_in=[str(p) for p in post],
        pre_period=[str(p) for p in ci_data.pre_period],
        post_period=[str(p) for p in ci_data.post_period],
        outcome_column=str(ci_data.outcome_column),
        series=_frame_to_json(series), summary=_frame_to_json(summ),
        summary_text=ref_summary.summary(analysis, output_format="summary"),
        report_text=ref_summary.summary(analysis, output_format="report"))
    with open(os.path.join(HERE, f"{name}.json"), "w") as f:
      json.dump(meta, f, indent=1)
    index.append(name)
  with open(os.path.join(HERE, "index.json"), "w") as f:
    json.dump(index, f)
  print("wrote", index)


if __name__ == "__main__":
  main()
