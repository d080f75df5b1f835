"""plot() against fixtures produced by running the reference under stubs
(tests/golden/make_plot_golden.py): the long-form plotting frame row for row
(reference plot.py:245-426), and the chart-dict fragments the reference's own tests assert
(plot_test.py:27-685, :799-884) for the static and the interactive chart."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import importlib

import causalimpact as ci

plot_lib = importlib.import_module("causalimpact.plot")   # `ci.plot` is the function, as upstream

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "plot", "plot_golden.json")))


def _frame(fx):
  data = {}
  for c in fx["columns"]:
    v = fx["data"][c]
    if c in ("time", "pre_period_start", "pre_period_end", "post_period_start", "post_period_end"):
      data[c] = pd.to_datetime(v) if fx["index_kind"] == "datetime" else [int(x) for x in v]
    elif v and isinstance(next((x for x in v if x is not None), 0.0), str):
      data[c] = v
    else:
      data[c] = [np.nan if x is None else x for x in v]
  df = pd.DataFrame(data, columns=fx["columns"])
  if "time" not in fx["columns"]:
    df.index = pd.to_datetime(fx["index"]) if fx["index_kind"] == "datetime" else [int(i) for i in fx["index"]]
  return df


def _analysis(name):
  series = _frame(GOLD["cases"][name]["series"])
  return ci.CausalImpactAnalysis(series=series, summary=pd.DataFrame(), posterior_samples=None)


@pytest.mark.parametrize("name", sorted(GOLD["cases"]))
def test_plot_frame_equals_the_reference_row_for_row(name):
  fx = GOLD["cases"][name]
  want = _frame({**fx["plot_df"], "index_kind": fx["series"]["index_kind"]})
  got = plot_lib._create_plot_df(_analysis(name).series, 0.05).reset_index(drop=True)
  assert list(got.columns) == list(want.columns)
  for c in want.columns:
    if c in ("scale", "stat", "band_method", "scale_pretty", "stat_pretty"):
      assert [None if (isinstance(v, float) and np.isnan(v)) else str(v) for v in got[c]] == \
             [None if (isinstance(v, float) and np.isnan(v)) else str(v) for v in want[c]], c
    elif c in ("value", "lower", "upper", "zero"):
      np.testing.assert_allclose(got[c].astype(float), want[c].astype(float), rtol=1e-12, err_msg=c)
    else:
      assert list(got[c]) == list(want[c]), c
  assert list(got["scale_pretty"].cat.categories) == ["Original", "Pointwise", "Cumulative"]
  assert list(got["stat_pretty"].cat.categories) == ["Observed", "Mean", "Median"]


@pytest.mark.parametrize("name", sorted(GOLD["cases"]))
def test_classic_chart_matches_the_fragments_the_reference_asserts(name):
  d = ci.plot(_analysis(name)).to_dict()
  assert {k: d[k] for k in ("facet", "spec", "resolve")} == GOLD["cases"][name]["classic"]
  assert d["config"]["axis"] == {"titleFontSize": 18, "labelFontSize": 16}
  assert d["config"]["header"] == {"labelFontSize": 20} and d["config"]["background"] == "white"
  stats = {r["stat"] for r in d["data"]["values"]}
  assert stats == {"observed", "mean"}                       # the median line is not drawn
  assert len(d["data"]["values"]) == 40
  json.dumps(d)                                              # a valid JSON document


def test_interactive_chart_matches_the_fragments_the_reference_asserts():
  d = ci.plot(_analysis(GOLD["interactive"]["case"]), static_plot=False).to_dict()
  top, bot, legend = d["hconcat"][0]["vconcat"][0], d["hconcat"][0]["vconcat"][1], d["hconcat"][1]
  for part in (top, bot, legend):
    del part["data"]
  assert top == GOLD["interactive"]["top"]
  assert bot == GOLD["interactive"]["bottom"]
  assert legend == GOLD["interactive"]["legend"]
  assert [p["name"] for p in d["params"]] == ["param_1", "param_2"]


def test_options_and_errors():
  an = _analysis("two_vlines")
  d = ci.plot(an, chart_width=300, chart_height=100, axis_label_font_size=10).to_dict()
  assert d["spec"]["width"] == 300 and d["spec"]["height"] == 100
  assert d["spec"]["layer"][0]["encoding"]["color"]["legend"]["symbolSize"] == 100
  with pytest.raises(ValueError, match="backend must be one of"):
    ci.plot(an, backend="bokeh")
  with pytest.raises(ValueError, match="`component` must be one of"):
    plot_lib._create_plot_component_df(an.series, "ribbons")


def test_matplotlib_backend_draws_three_panels(tmp_path):
  import matplotlib
  matplotlib.use("Agg")
  an = _analysis("four_vlines")
  fig = ci.plot(an, backend="matplotlib")
  assert fig is not None and len(fig.axes) == 3
  assert [ax.get_ylabel() for ax in fig.axes] == ["Original", "Pointwise", "Cumulative"]
  assert fig.axes[2].get_xlabel() == "Time"
  obs = [l for l in fig.axes[0].get_lines() if l.get_label() == "Observed"][0]
  np.testing.assert_allclose(obs.get_ydata(), an.series["observed"].to_numpy())
  # four period rules per panel (+ the zero line on the effect panels)
  dashed = [l for l in fig.axes[1].get_lines() if l.get_linestyle() == "--"]
  assert len(dashed) == 4
  fig.savefig(tmp_path / "impact.png")
  assert (tmp_path / "impact.png").stat().st_size > 1000


def test_plot_of_a_real_series_frame_from_the_wrapper():
  """The wrapper's `series` (golden post-processing fixture: gap and tail around the periods)."""
  meta = json.load(open(os.path.join(HERE, "golden", "datacsv_gap_tail_nostd.json")))
  fx = meta["series"]
  data = {c: [np.nan if v is None else v for v in fx["data"][c]] for c in fx["columns"]}
  for c in ("pre_period_start", "pre_period_end", "post_period_start", "post_period_end"):
    data[c] = pd.to_datetime(data[c])
  series = pd.DataFrame(data, index=pd.to_datetime(fx["index"]))
  d = ci.plot(ci.CausalImpactAnalysis(series, pd.DataFrame(), None)).to_dict()
  marks = [l["encoding"]["x"]["field"] for l in d["spec"]["layer"][3:]]
  assert marks == ["pre_period_start", "pre_period_end", "post_period_start", "post_period_end"]
