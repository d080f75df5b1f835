"""Generates tests/golden/plot/plot_golden.json from the REAL reference (build container only; needs
/root/reference).  TensorFlow / TFP / altair / absl are absent, so the reference's modules are
imported under import-only stubs (as make_golden.py does).  Only DATA is written:

  * the fake impact `series` frames of the reference's plot test (plot_test.py:687-762: one /
    two / four vertical rules, integer index) and, for each, the long-form frame its
    `_create_plot_df` (plot.py:245-322) produces -- computed here by RUNNING the reference;
  * the expected chart-dict fragments its tests assert (`expected_*` literals of
    plot_test.py:27-685: facet / spec / resolve of the classic chart, top / bottom / legend of
    the interactive one) -- the reference's own golden values, read from the imported module.
No reference source text is copied.
"""
import json
import os
import sys
import types
import unittest

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402  pylint: disable=wrong-import-position


def _frame(df):
  out = {"columns": [str(c) for c in df.columns], "index": [str(i) for i in df.index],
         "index_kind": "datetime" if isinstance(df.index, pd.DatetimeIndex) else "int", "data": {}}
  for c in df.columns:
    col = df[c]
    if str(col.dtype) == "category":
      col = col.astype(object)
    if col.dtype != object and np.issubdtype(col.dtype, np.datetime64):
      out["data"][str(c)] = [str(v) for v in col]
    elif col.dtype == object or str(col.dtype).startswith("str"):
      out["data"][str(c)] = [None if (isinstance(v, float) and np.isnan(v)) else str(v) for v in col]
    else:
      out["data"][str(c)] = [None if (isinstance(v, float) and np.isnan(v)) else float(v) for v in col]
  return out


def main():
  make_golden._install_stubs()   # pylint: disable=protected-access
  absl = types.ModuleType("absl")
  testing = types.ModuleType("absl.testing")
  absltest = types.ModuleType("absl.testing.absltest")
  absltest.TestCase = unittest.TestCase
  absltest.main = unittest.main
  parameterized = types.ModuleType("absl.testing.parameterized")
  parameterized.TestCase = unittest.TestCase
  parameterized.named_parameters = lambda *a, **k: (lambda f: f)
  testing.absltest, testing.parameterized = absltest, parameterized
  absl.testing = testing
  sys.modules.update({"absl": absl, "absl.testing": testing, "absl.testing.absltest": absltest,
                      "absl.testing.parameterized": parameterized})
  import importlib
  ref_plot = importlib.import_module("causalimpact.plot")
  ref_test = importlib.import_module("causalimpact.plot_test")
  ref_test.PlotTest.setUpClass()
  cases = {"one_vline": ref_test.PlotTest.ci_data_1, "two_vlines": ref_test.PlotTest.ci_data_2,
           "four_vlines": ref_test.PlotTest.ci_data_4,
           "one_vline_integer_index": ref_test.PlotTest.ci_data_integer_index}
  out = {"cases": {}, "interactive": {
      "case": "two_vlines", "top": ref_test.expected_top_dict, "bottom": ref_test.expected_bot_dict,
      "legend": ref_test.expected_legend_dict}}
  for name, an in cases.items():
    series = an.series.copy()
    # the reference's frames carry *_std / *_median columns; the std component needs
    # tfp.distributions.Normal, absent here -- fixtures cover the quantile bands (this build's
    # `series` has no std columns either)
    series = series[[c for c in series.columns if "std" not in c]]
    plot_df = ref_plot._create_plot_df(series.copy(), 0.05)   # pylint: disable=protected-access
    out["cases"][name] = {
        "series": _frame(series), "plot_df": _frame(plot_df.reset_index(drop=True)),
        "classic": getattr(ref_test, f"expected_classic_dict_{name}")}
  with open(os.path.join(HERE, "plot", "plot_golden.json"), "w") as f:
    json.dump(out, f)
  print("wrote plot/plot_golden.json:", list(out["cases"]))


if __name__ == "__main__":
  main()
