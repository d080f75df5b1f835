"""Back-transform and per-timestep quantiles of posterior draws (host side).

Mirror of /root/reference/causalimpact/posterior_processing.py:25-98.
"""
from typing import List, Text, Tuple

import numpy as np
import pandas as pd


def model_index(ci_data) -> pd.Index:
  """Index of the T steps handed to the sampler: pre-period then everything after it."""
  return ci_data.model_pre_data.index.union(ci_data.model_after_pre_data.index).sort_values()


def calculate_trajectory_quantiles(trajectories: pd.DataFrame, column_prefix: Text = "predicted",
                                   quantiles: Tuple[float, float] = (0.025, 0.975)) -> pd.DataFrame:
  """Linear-interpolated quantiles across the sample columns of a T x S frame."""
  values = trajectories.to_numpy(dtype=np.float64)
  with np.errstate(invalid="ignore"):
    q = np.quantile(values, quantiles, axis=1)   # NaN rows stay NaN, like DataFrame.quantile
  return pd.DataFrame({column_prefix + "_lower": q[0], column_prefix + "_upper": q[1]},
                      index=trajectories.index)


def process_posterior_quantities(ci_data, vals_to_process: np.ndarray,
                                 col_names: List[Text]) -> pd.DataFrame:
  """[samples, T] (or [T]) on the model scale -> T x samples frame on the data scale."""
  vals = np.asarray(vals_to_process)
  if ci_data.standardize_data:
    vals = ci_data.outcome_scaler.inverse_transform(vals)
  return pd.DataFrame(np.transpose(vals), columns=col_names, index=model_index(ci_data))
