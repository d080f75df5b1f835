# Attribution: this module restates the interface, validation rules and error messages of
# google/tfp-causalimpact causalimpact/data.py (Copyright 2019-2023 The TFP CausalImpact Authors,
# Copyright 2014 Google Inc.; Apache License 2.0, http://www.apache.org/licenses/LICENSE-2.0) so that
# this package is a drop-in for it; the implementation is TensorFlow-free and written for this build.
"""Input validation, pre/after-pre split and standardisation (host side).

Mirror of /root/reference/causalimpact/data.py:77-190 without TensorFlow: the masked
outcome series is a (time_series, is_missing) pair of numpy arrays
(tfp.sts.MaskedTimeSeries at data.py:125-128).
"""
import collections
from typing import Optional, Text, Tuple, Union

import numpy as np
import pandas as pd

from causalimpact import indices
from causalimpact import standardize

MaskedTimeSeries = collections.namedtuple("MaskedTimeSeries", ["time_series", "is_missing"])


def _as_numpy_dtype(dtype):
  """float32 / float64 given as numpy dtype, python type or a TF-like object with a name."""
  if dtype is None:
    return np.dtype(np.float32)
  name = getattr(dtype, "name", None)
  if isinstance(name, str) and name in ("float32", "float64"):
    return np.dtype(name)
  return np.dtype(dtype)


def _validate_data_and_columns(data: pd.DataFrame, outcome_column: Optional[str]):
  """Outcome defaults to the first column; every other column is a covariate."""
  if outcome_column is None:
    outcome_column = data.columns[0]
  if outcome_column not in data.columns:
    raise KeyError(f"Specified `outcome_column` ({outcome_column}) not found in data")
  if standardize._all_float64(data) and data.columns.is_unique:
    # the same checks, in the same order, on the values (float64 frames: every ordinary call)
    y = data[outcome_column].to_numpy()
    seen = y[~np.isnan(y)]
    if seen.size > 0 and seen.std() == 0:
      raise ValueError("Input response cannot be constant.")
    feature_columns = [c for c in data.columns if c != outcome_column] if data.shape[1] > 1 else None
    data = data[[outcome_column] + (feature_columns or [])]
    if seen.size < 3:
      raise ValueError("Input data must have at least 3 observations.")
    if feature_columns and np.isnan(data.to_numpy()[:, 1:]).any():
      raise ValueError("Input data cannot have any missing values.")
    return data, outcome_column, feature_columns
  if data[outcome_column].std(skipna=True, ddof=0) == 0:
    raise ValueError("Input response cannot be constant.")
  feature_columns = [c for c in data.columns if c != outcome_column] if data.shape[1] > 1 else None
  data = data[[outcome_column] + (feature_columns or [])]
  if data[outcome_column].count() < 3:
    raise ValueError("Input data must have at least 3 observations.")
  if data[feature_columns or []].isna().values.any():
    raise ValueError("Input data cannot have any missing values.")
  if not data.dtypes.map(pd.api.types.is_numeric_dtype).all():
    raise ValueError("Input data must contain only numeric values.")
  return data, outcome_column, feature_columns


class CausalImpactData:
  """Holds the validated data and what the sampler consumes.

  Attributes (same names as the reference class): data, pre_period, post_period,
  outcome_column, feature_columns, standardize_data, pre_data, after_pre_data,
  num_steps_forecast, model_pre_data, model_after_pre_data, outcome_scaler, feature_ts,
  outcome_ts.
  """

  def __init__(self,
               data: Union[pd.DataFrame, pd.Series],
               pre_period: Tuple[indices.InputDateType, indices.InputDateType],
               post_period: Tuple[indices.InputDateType, indices.InputDateType],
               outcome_column: Optional[Text] = None,
               standardize_data=True,
               dtype=np.float32):
    data = pd.DataFrame(data)
    self.pre_period, self.post_period = indices.parse_and_validate_date_data(
        data=data, pre_period=pre_period, post_period=post_period)
    self.data, self.outcome_column, self.feature_columns = _validate_data_and_columns(
        data, outcome_column)
    self.standardize_data = standardize_data
    idx = self.data.index
    self.pre_data = self.data.loc[(idx >= self.pre_period[0]) & (idx <= self.pre_period[1])]
    # everything after the pre-period -- gap, post-period and tail -- is forecast (data.py:107-112)
    self.after_pre_data = self.data.loc[idx > self.pre_period[1]]
    self.num_steps_forecast = len(self.after_pre_data.index)
    if standardize_data:
      scaler = standardize.Scaler().fit(self.pre_data)
      self.outcome_scaler = standardize.Scaler().fit(self.pre_data[self.outcome_column])
      self.model_pre_data = scaler.transform(self.pre_data)
      self.model_after_pre_data = scaler.transform(self.after_pre_data)
    else:
      self.outcome_scaler = None
      self.model_pre_data = self.pre_data
      self.model_after_pre_data = self.after_pre_data
    series = np.asarray(self.model_pre_data[self.outcome_column], dtype=_as_numpy_dtype(dtype))
    self.outcome_ts = MaskedTimeSeries(time_series=series, is_missing=np.isnan(series))
    if self.feature_columns is not None:
      # the covariates over pre-period + forecast steps, then the intercept as the LAST column
      # (data.py:130-135) -- one frame from the values instead of concat + setitem
      pre_f = self.model_pre_data[self.feature_columns]
      aft_f = self.model_after_pre_data[self.feature_columns]
      if standardize._all_float64(pre_f) and standardize._all_float64(aft_f) \
          and "intercept_" not in self.feature_columns:
        vals = np.concatenate([pre_f.to_numpy(), aft_f.to_numpy()], axis=0)
        vals = np.concatenate([vals, np.ones((vals.shape[0], 1), dtype=vals.dtype)], axis=1)
        self.feature_ts = pd.DataFrame(vals, index=pre_f.index.append(aft_f.index),
                                       columns=list(self.feature_columns) + ["intercept_"])
      else:
        self.feature_ts = pd.concat([pre_f, aft_f], axis=0)
        self.feature_ts["intercept_"] = 1.0
    else:
      self.feature_ts = None
