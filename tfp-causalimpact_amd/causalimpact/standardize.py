"""NaN-aware z-scoring of DataFrames (mirror of /root/reference/causalimpact/standardize.py:26-64).

ddof=1 by default; columns with zero spread pass through unscaled; NaNs are ignored when
fitting and preserved when transforming.
"""
import numpy as np
import pandas as pd


class NotFittedError(ValueError, AttributeError):
  """Scaler used before fit()."""


class Scaler:
  """fit / transform / inverse_transform with pandas in, pandas out."""

  def __init__(self, ddof=1):
    self.ddof = ddof
    self._is_fit = False

  def fit(self, df) -> "Scaler":
    self.mean_ = np.nanmean(df, axis=0)
    self.stddev_ = np.nanstd(df, axis=0, ddof=self.ddof)
    self._is_fit = True
    return self

  def _require_fit(self):
    if not self._is_fit:
      raise NotFittedError("Must call `.fit(df)` before using Scaler to transform!")

  def transform(self, df: pd.DataFrame) -> pd.DataFrame:
    self._require_fit()
    scaled = np.where(self.stddev_ > 0, (df - self.mean_) / self.stddev_, df)
    return pd.DataFrame(scaled, index=df.index, columns=df.columns)

  def fit_transform(self, df: pd.DataFrame) -> pd.DataFrame:
    return self.fit(df).transform(df)

  def inverse_transform(self, values):
    self._require_fit()
    return values * self.stddev_ + self.mean_
