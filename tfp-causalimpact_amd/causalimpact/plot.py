"""Plotting causalimpact results (reference `causalimpact/plot.py`).

`plot(ci_model, **kwargs)` keeps the reference's signature and options (`plot.py:164-242`):
`backend` "altair" (default) or "matplotlib", `static_plot`, `alpha`, `show_median`,
`use_std_intervals`, `chart_width`, `chart_height`, `axis_title_font_size`,
`axis_label_font_size`, `strip_title_font_size`.

Own implementation, not a copy: the long-form plotting frame is assembled column by column
(`_create_plot_df`; the reference melts / pivots, `plot.py:245-426`) and pinned row for row by
fixtures produced by RUNNING the reference (`tests/golden/make_plot_golden.py`); the Altair
backend does not need the `altair` package -- it writes the Vega-Lite specification Altair would
emit directly (`VegaLiteChart`, with `to_dict()` / `to_json()` / `save()` / a Jupyter mime
bundle), pinned by the chart-dict fragments the reference's own tests assert
(`plot_test.py:27-685`, `:799-884`).  When `altair` is importable `VegaLiteChart.to_altair()`
hands back a genuine `alt.Chart`-family object.
"""
from __future__ import annotations

import json
from typing import Any, Dict, List

import numpy as np
import pandas as pd
from scipy import special

_PERIOD_COLS = ["pre_period_start", "pre_period_end", "post_period_start", "post_period_end"]
_SCALES = (("posterior", "original"), ("point_effects", "point_effects"),
           ("cumulative_effects", "cumulative_effects"))
_PRETTY = {"original": "Original", "point_effects": "Pointwise", "cumulative_effects": "Cumulative"}
_DEFAULTS = {
    "static_plot": True, "backend": "altair", "alpha": 0.05, "show_median": False,
    "use_std_intervals": False, "chart_width": 600, "chart_height": 200,
    "axis_title_font_size": 18, "axis_label_font_size": 16, "strip_title_font_size": 20}


def _scale_and_stat(column: str):
  """("original" | "point_effects" | "cumulative_effects", stat) of a `series` column."""
  if column == "observed":
    return "original", "observed"
  for prefix, scale in _SCALES:
    if column.startswith(prefix + "_"):
      return scale, column[len(prefix) + 1:]
  return None, None


def _create_plot_component_df(series: pd.DataFrame, component: str,
                              alpha: float = 0.05) -> pd.DataFrame:
  """Long-form frame of one plot component: "lines" (observed / mean / median values),
  "bands" (quantile lower / upper) or "std" (mean -/+ z std).  Reference `plot.py:325-426`."""
  if component not in ("lines", "bands", "std"):
    raise ValueError("`component` must be one of 'lines', 'bands', or 'std'."
                     "Got %s." % component)
  if "time" not in series.columns:
    series = series.assign(time=series.index)
  periods = {c: series[c].to_numpy() for c in _PERIOD_COLS}
  time = series["time"].to_numpy()
  if component == "lines":
    parts = []
    for col in series.columns:
      scale, stat = _scale_and_stat(str(col))
      if stat not in ("mean", "median", "observed"):
        continue
      parts.append(pd.DataFrame({"time": time, **periods, "value": series[col].to_numpy(),
                                 "scale": scale, "stat": stat}))
    return pd.concat(parts, axis=0, ignore_index=True)
  # bands / std: one row per (time, scale) with both bounds present, ordered by time then scale
  wanted = ("lower", "upper") if component == "bands" else ("mean", "std")
  parts = []
  for _, scale in sorted(_SCALES, key=lambda ps: ps[1]):
    cols = {}
    for col in series.columns:
      sc, stat = _scale_and_stat(str(col))
      if sc == scale and stat in wanted:
        cols[stat] = series[col].to_numpy(dtype=float)
    if len(cols) != 2:
      continue
    a, b = cols[wanted[0]], cols[wanted[1]]
    if component == "std":
      z = float(special.ndtri(1.0 - alpha / 2.0))
      a, b = a - z * b, a + z * b
    keep = ~(np.isnan(a) & np.isnan(b))          # a pivot drops rows with no value at all
    parts.append(pd.DataFrame({"time": time[keep], "scale": scale,
                               **{c: v[keep] for c, v in periods.items()},
                               "lower": a[keep], "upper": b[keep]}))
  out = pd.concat(parts, axis=0, ignore_index=True)
  out = out.sort_values(["time", "scale"], kind="stable").reset_index(drop=True)
  out["band_method"] = "quantile" if component == "bands" else "std"
  return out


def _create_plot_df(series: pd.DataFrame, alpha: float = 0.05) -> pd.DataFrame:
  """The frame every backend draws from: columns time, the four period marks, value, scale,
  stat, lower, upper, band_method, zero, scale_pretty, stat_pretty (reference `plot.py:245-322`)."""
  series = series.copy()
  series["time"] = series.index
  lines = _create_plot_component_df(series, "lines")
  bands = _create_plot_component_df(series, "bands")
  if any("std" in str(c) for c in series.columns):
    bands = pd.concat([bands, _create_plot_component_df(series, "std", alpha)], axis=0, sort=True)
  plot_df = lines.merge(bands, on=["time", "scale"] + _PERIOD_COLS, how="left")
  plot_df["zero"] = np.where(plot_df["scale"] == "original", np.nan, 0.0)
  plot_df["scale_pretty"] = pd.Categorical([_PRETTY[s] for s in plot_df["scale"]],
                                           categories=["Original", "Pointwise", "Cumulative"],
                                           ordered=True)
  plot_df["stat_pretty"] = pd.Categorical(plot_df["stat"].str.capitalize(),
                                          categories=["Observed", "Mean", "Median"], ordered=True)
  return plot_df


def _rules_to_draw(plot_df: pd.DataFrame) -> List[str]:
  """Which period marks get a vertical rule: the pre-period start only if points precede it, its
  end only if points lie strictly between it and the post-period start, the post-period start
  always, its end only if points follow (reference `plot.py:470-499`)."""
  t = plot_df["time"]
  first = plot_df.iloc[0]
  out = []
  if (t < first["pre_period_start"]).any():
    out.append("pre_period_start")
  if ((t > first["pre_period_end"]) & (t < first["post_period_start"])).any():
    out.append("pre_period_end")
  out.append("post_period_start")
  if (t > first["post_period_end"]).any():
    out.append("post_period_end")
  return out


# ---------------------------------------------------------------------------------------------
# Vega-Lite backend
# ---------------------------------------------------------------------------------------------
class VegaLiteChart:
  """A Vega-Lite v5 specification with the small part of `alt.Chart`'s surface notebooks use."""
  SCHEMA = "https://vega.github.io/schema/vega-lite/v5.json"

  def __init__(self, spec: Dict[str, Any]):
    self._spec = spec

  def to_dict(self) -> Dict[str, Any]:
    return json.loads(json.dumps(self._spec))

  def to_json(self, indent: int = 2) -> str:
    return json.dumps(self._spec, indent=indent)

  def save(self, path: str):
    """.json writes the specification; .html a page that renders it with vega-embed."""
    if str(path).endswith(".json"):
      with open(path, "w") as f:
        f.write(self.to_json())
      return
    if not str(path).endswith(".html"):
      raise ValueError("save() writes .json or .html")
    page = ("<!DOCTYPE html><html><head>"
            '<script src="https://cdn.jsdelivr.net/npm/vega@5"></script>'
            '<script src="https://cdn.jsdelivr.net/npm/vega-lite@5"></script>'
            '<script src="https://cdn.jsdelivr.net/npm/vega-embed@6"></script>'
            '</head><body><div id="vis"></div><script>vegaEmbed("#vis", %s);</script>'
            "</body></html>") % self.to_json(indent=0)
    with open(path, "w") as f:
      f.write(page)

  def _repr_mimebundle_(self, include=None, exclude=None):   # pylint: disable=unused-argument
    return {"application/vnd.vegalite.v5+json": self.to_dict(), "text/plain": "<VegaLiteChart>"}

  def to_altair(self):
    import altair as alt   # pylint: disable=import-outside-toplevel
    return alt.Chart.from_dict(self.to_dict(), validate=False)


def _records(plot_df: pd.DataFrame) -> List[Dict[str, Any]]:
  out = []
  cols = list(plot_df.columns)
  for row in plot_df.itertuples(index=False):
    rec = {}
    for c, v in zip(cols, row):
      if isinstance(v, (pd.Timestamp, np.datetime64)):
        rec[c] = pd.Timestamp(v).isoformat()
      elif isinstance(v, (float, np.floating)):
        rec[c] = None if np.isnan(v) else float(v)
      elif isinstance(v, (int, np.integer)):
        rec[c] = int(v)
      else:
        rec[c] = None if v is None else str(v)
    out.append(rec)
  return out


def _time_type(plot_df: pd.DataFrame) -> str:
  return "temporal" if np.issubdtype(plot_df["time"].dtype, np.datetime64) else "quantitative"


def _base_layers(plot_df: pd.DataFrame, color: Dict[str, Any], x_scale=None, **kw):
  """lines, band, zero rule and the period rules of one panel stack."""
  ttype = _time_type(plot_df)

  def x(field, title=None):
    enc = {"type": ttype, "field": field}
    if x_scale is not None:
      enc["scale"] = x_scale
    if title is not None:
      enc["title"] = title
    return enc

  layers = [
      {"mark": {"type": "line"},
       "encoding": {"color": color, "x": x("time", "Time"),
                    "y": {"type": "quantitative", "field": "value", "scale": {"zero": False},
                          "title": ""}}},
      {"mark": {"type": "area", "opacity": 0.3},
       "encoding": {"x": x("time", "Time"), "y": {"type": "quantitative", "field": "upper"},
                    "y2": {"field": "lower"}}},
      {"mark": {"type": "rule"}, "encoding": {"y": {"type": "quantitative", "field": "zero"}}},
  ]
  for mark in _rules_to_draw(plot_df):
    layers.append({"mark": {"type": "rule", "strokeDash": [5, 5]},
                   "encoding": {"color": {"value": "grey"}, "x": x(mark)}})
  return layers


def _facet(layers, **kw):
  return {
      "facet": {"row": {"type": "nominal", "field": "scale_pretty",
                        "sort": ["Original", "Pointwise", "Cumulative"], "title": ""}},
      "spec": {"height": kw["chart_height"], "width": kw["chart_width"], "layer": layers},
      "resolve": {"scale": {"y": "independent"}}}


def _config(**kw):
  return {"background": "white",
          "axis": {"titleFontSize": kw["axis_title_font_size"],
                   "labelFontSize": kw["axis_label_font_size"]},
          "header": {"labelFontSize": kw["strip_title_font_size"]}}


def _legend_color(**kw):
  return {"type": "nominal", "field": "stat_pretty",
          "legend": {"labelFontSize": kw["axis_label_font_size"],
                     "symbolSize": 10 * kw["axis_label_font_size"], "title": ""}}


def _draw_classic_plot(plot_df: pd.DataFrame, **kw) -> VegaLiteChart:
  """The static three-panel chart of the R package (reference `plot.py:508-552`)."""
  spec = {"$schema": VegaLiteChart.SCHEMA, "config": _config(**kw),
          "data": {"values": _records(plot_df)}}
  spec.update(_facet(_base_layers(plot_df, _legend_color(**kw), **kw), **kw))
  return VegaLiteChart(spec)


def _draw_interactive_plot(plot_df: pd.DataFrame, **kw) -> VegaLiteChart:
  """Static top panel with an x-brush that zooms the three panels below, and a point legend
  that picks the statistic shown (reference `plot.py:555-665`)."""
  brush, pick = "param_1", "param_2"
  select_color = {"condition": {"type": "nominal", "field": "stat_pretty", "legend": None,
                                "param": pick}, "value": "lightgray"}
  static_df = plot_df.loc[plot_df["scale"] == "original"].reset_index(drop=True)
  top = _facet(_base_layers(static_df, _legend_color(**kw), **kw), **kw)
  top["data"] = {"values": _records(static_df)}
  bottom = _facet(_base_layers(plot_df, select_color, x_scale={"domain": {"param": brush}}, **kw),
                  **kw)
  bottom["data"] = {"values": _records(plot_df)}
  legend = {"mark": {"type": "point"},
            "encoding": {"color": select_color,
                         "y": {"type": "nominal", "axis": {"orient": "right"},
                               "field": "stat_pretty", "title": ""}},
            "name": "view_2", "data": {"values": _records(plot_df)}}
  spec = {"$schema": VegaLiteChart.SCHEMA, "config": _config(**kw),
          "hconcat": [{"vconcat": [top, bottom]}, legend],
          "params": [
              {"name": brush, "select": {"type": "interval", "encodings": ["x"]},
               "views": ["view_1"]},
              {"name": pick, "select": {"type": "point", "fields": ["stat_pretty"]},
               "views": ["view_2"]}]}
  spec["hconcat"][0]["vconcat"][0]["spec"]["layer"][1]["name"] = "view_1"
  return VegaLiteChart(spec)


# ---------------------------------------------------------------------------------------------
# matplotlib backend
# ---------------------------------------------------------------------------------------------
def _draw_matplotlib_plot(plot_df: pd.DataFrame, **kw):
  """Three stacked axes sharing the time axis: observed + prediction, pointwise effect,
  cumulative effect, each with its uncertainty band and the period rules."""
  try:
    import matplotlib   # pylint: disable=import-outside-toplevel
    import matplotlib.pyplot as mplt   # pylint: disable=import-outside-toplevel
  except ImportError as e:
    raise ImportError("matplotlib is required for using it as plotting backend. Please"
                      " install it first.") from e
  del matplotlib
  fig, axes = mplt.subplots(3, 1, sharex=True, figsize=(kw["chart_width"] / 100,
                                                       3 * kw["chart_height"] / 100))
  fig.tight_layout(pad=3.0)
  rules = _rules_to_draw(plot_df)
  first = plot_df.iloc[0]
  panels = (("original", "Original"), ("point_effects", "Pointwise"),
            ("cumulative_effects", "Cumulative"))
  for ax, (scale, label) in zip(axes, panels):
    ax.grid()
    for mark in rules:
      ax.axvline(first[mark], color="grey", linestyle="--")
    ax.set_ylabel(label, rotation=90, fontsize=kw["axis_title_font_size"], fontweight="bold")
    part = plot_df.loc[plot_df["scale"] == scale]
    mean = part.loc[part["stat"] == "mean"]
    ax.plot(mean["time"], mean["value"], label="Mean" if scale == "original" else label)
    if scale == "original":
      obs = part.loc[part["stat"] == "observed"]
      ax.plot(obs["time"], obs["value"], label="Observed")
    else:
      ax.axhline(0, color="grey", linestyle="-")
    ax.fill_between(mean["time"], mean["lower"].astype(float), mean["upper"].astype(float),
                    alpha=0.2)
    ax.legend(loc="upper left")
  axes[2].set_xlabel("Time", fontsize=kw["axis_title_font_size"], fontweight="bold")
  return fig


def plot(ci_model, **kwargs):
  """Draws the impact analysis: observed vs counterfactual, pointwise and cumulative effects.

  ci_model: the `CausalImpactAnalysis` returned by `fit_causalimpact`.  Keyword options as in the
  reference (`plot.py:164-242`); returns a `VegaLiteChart` (backend "altair") or a matplotlib
  `Figure` (backend "matplotlib")."""
  params = dict(_DEFAULTS)
  for k in params:
    if k in kwargs:
      params[k] = kwargs[k]
  plot_df = _create_plot_df(ci_model.series, params["alpha"])
  drop = "quantile" if params["use_std_intervals"] else "std"
  plot_df = plot_df.loc[plot_df["band_method"] != drop]
  if params["show_median"]:
    plot_df = plot_df.loc[plot_df["stat"] != "median"].copy()
    plot_df["stat_pretty"] = pd.Categorical(plot_df["stat_pretty"].astype(str),
                                            categories=["Observed", "Mean"], ordered=True)
  plot_df = plot_df.reset_index(drop=True)
  if params["backend"] == "altair":
    if params["static_plot"]:
      return _draw_classic_plot(plot_df.loc[plot_df["stat"] != "median"].reset_index(drop=True),
                                **params)
    return _draw_interactive_plot(plot_df, **params)
  if params["backend"] == "matplotlib":
    return _draw_matplotlib_plot(plot_df, **params)
  raise ValueError("backend must be one of 'altair' or 'matplotlib'. Got"
                   f" {params['backend']}.")
