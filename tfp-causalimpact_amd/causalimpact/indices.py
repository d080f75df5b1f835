# Attribution: this module restates the interface, validation rules and error messages of
# google/tfp-causalimpact causalimpact/indices.py (Copyright 2019-2023 The TFP CausalImpact Authors,
# Copyright 2014 Google Inc.; Apache License 2.0, http://www.apache.org/licenses/LICENSE-2.0) so that
# this package is a drop-in for it; the implementation is TensorFlow-free and written for this build.
"""Period parsing/alignment (host side, pandas only).

Behavioural mirror of /root/reference/causalimpact/indices.py:30-149: same accepted
inputs (str / int position / datetime), same alignment rule (a period that does not
start/end on an index value is shrunk: start rounds forward, end rounds back), same
ValueError texts (asserted by the reference's indices_test.py:49-93).
"""
import datetime
from typing import Tuple, Union

import numpy as np
import pandas as pd

InputDateType = Union[str, int, datetime.datetime]
OutputDateType = Union[int, datetime.datetime]
InputPeriodType = Tuple[InputDateType, InputDateType]
OutputPeriodType = Tuple[OutputDateType, OutputDateType]


def _to_index_value(value: InputDateType, index: pd.Index) -> OutputDateType:
  if isinstance(value, str):
    return pd.to_datetime(value)
  if isinstance(value, (int, np.integer)):
    return index[value]           # integers are POSITIONS into the index
  if isinstance(value, datetime.datetime):
    return value
  raise ValueError(f"Expected argument to be str, int, or datetime. Got {type(value)}")


def _align(period: OutputPeriodType, index: pd.Index) -> OutputPeriodType:
  start, end = period
  if start > end:
    raise ValueError(f"Period end must be after period start. Got {period}")
  first = index.get_indexer([start], method="bfill")[0]
  if first < 0:
    raise ValueError("Aligned period start not found in the index.")
  last = index.get_indexer([end], method="ffill")[0]
  if last < 0:
    raise ValueError("Aligned period end not found in the index.")
  return index[first], index[last]


def parse_and_validate_date_data(
    data: pd.DataFrame, pre_period: InputPeriodType,
    post_period: InputPeriodType) -> Tuple[OutputPeriodType, OutputPeriodType]:
  """Returns (pre_period, post_period) as values of `data.index`."""
  index = data.index
  pre = _align(tuple(_to_index_value(v, index) for v in pre_period), index)
  post = _align(tuple(_to_index_value(v, index) for v in post_period), index)
  if pre[1] >= post[0]:
    raise ValueError("pre_period and post_period cannot overlap.")
  n_pre = int(((index >= pre[0]) & (index <= pre[1])).sum())
  if n_pre < 3:
    raise ValueError("pre_period must span at least 3 time points. Got %s" % n_pre)
  if pre[1] < pre[0]:
    raise ValueError("pre_period last number must be bigger than its first.")
  if post[1] < post[0]:
    raise ValueError("post_period last number must be bigger than its first.")
  return pre, post
