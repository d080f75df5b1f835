"""Text summaries of a fitted analysis -- mirror of the reference's `causalimpact.summary`
(summary.py:133-178).

The reference renders two jinja2 templates; here the same text is assembled with plain string
formatting (no template engine on the path).  The wording and number formatting are the
reference's *output format*: they are pinned byte for byte by the reference's own golden texts
(`testdata/test_summary_output.txt`, `test_report_text_[1-4].txt`, kept under
tests/golden/ref_testdata) and by the texts rendered from the reference on this build's golden
cases (tests/golden/*.json).

Formatting rules reproduced (jinja2 semantics): `x | round(n)` is Python's round() followed by
str(); a `[a, b] | sort` pair prints as a Python list; percentages use `'{0:.1%}'`; the
cumulative relative-effect s.d. is rounded to 2 decimals before being formatted; the
confidence label drops trailing zeros of (1 - alpha) * 100; the second column starts 19
characters after the first value's start.
"""
from __future__ import annotations

from typing import Optional

_COL = 19


def _r(x, n) -> str:
  return str(round(float(x), n))


def _pct(x) -> str:
  return "{0:.1%}".format(float(x))


def _ci_label(alpha: float) -> str:
  return str((1 - alpha) * 100).rstrip("0").rstrip(".") + "%"


def _pad(width_used: int) -> str:
  return " " * (_COL - width_used)


def _sorted_pair(a, b) -> str:
  return str(sorted([round(float(a), 1), round(float(b), 1)]))


def _summary_table(s, alpha: float, p_value: float) -> str:
  avg, cum = s["average"], s["cumulative"]
  ci = _ci_label(alpha) + " CI"
  gap = " " * 20
  out = ["", "Posterior Inference {CausalImpact}",
         "                          Average            Cumulative"]
  a = _r(avg["actual"], 1)
  out.append("Actual                    " + a + _pad(len(a)) + _r(cum["actual"], 1))
  p, psd = _r(avg["predicted"], 1), _r(avg["predicted_sd"], 2)
  out.append("Prediction (s.d.)         " + f"{p} ({psd})" + _pad(len(p) + 3 + len(psd)) +
             f"{_r(cum['predicted'], 1)} ({_r(cum['predicted_sd'], 2)})")
  lo, hi = _r(avg["predicted_lower"], 1), _r(avg["predicted_upper"], 1)
  out.append(ci + gap + f"[{lo}, {hi}]" + _pad(4 + len(lo) + len(hi)) +
             f"[{_r(cum['predicted_lower'], 1)}, {_r(cum['predicted_upper'], 1)}]")
  out.append("")
  e, esd = _r(avg["abs_effect"], 1), _r(avg["abs_effect_sd"], 2)
  out.append("Absolute effect (s.d.)    " + f"{e} ({esd})" + _pad(3 + len(e) + len(esd)) +
             f"{_r(cum['abs_effect'], 1)} ({_r(cum['abs_effect_sd'], 2)})")
  lo, hi = _r(avg["abs_effect_lower"], 1), _r(avg["abs_effect_upper"], 1)
  out.append(ci + gap + _sorted_pair(avg["abs_effect_lower"], avg["abs_effect_upper"]) +
             _pad(4 + len(lo) + len(hi)) +
             _sorted_pair(cum["abs_effect_lower"], cum["abs_effect_upper"]))
  out.append("")
  r, rsd = _pct(avg["rel_effect"]), _pct(avg["rel_effect_sd"])
  out.append("Relative effect (s.d.)    " + f"{r} ({rsd})" + _pad(3 + len(r) + len(rsd)) +
             f"{_pct(cum['rel_effect'])} ({_pct(round(float(cum['rel_effect_sd']), 2))})")
  lo = _pct(min(avg["rel_effect_lower"], avg["rel_effect_upper"]))
  hi = _pct(max(avg["rel_effect_lower"], avg["rel_effect_upper"]))
  clo = _pct(min(cum["rel_effect_lower"], cum["rel_effect_upper"]))
  chi = _pct(max(cum["rel_effect_lower"], cum["rel_effect_upper"]))
  out.append(ci + gap + f"[{lo}, {hi}]" + _pad(4 + len(lo) + len(hi)) + f"[{clo}, {chi}]")
  out.append("")
  out.append("Posterior tail-area probability p: " + _r(p_value, 3))
  out.append("Posterior probability of an effect: " + "{0:.2%}".format(1 - p_value))
  out.append("")
  out.append('For more details run the command: summary(impact, output_format="report")')
  return "\n".join(out)


def _report(s, alpha: float, p_value: float) -> str:
  avg, cum = s["average"], s["cumulative"]
  sig = not (avg["rel_effect_lower"] < 0 and avg["rel_effect_upper"] > 0)
  pos = avg["rel_effect"] > 0
  ci = _ci_label(alpha)
  rel_lo = _pct(min(avg["rel_effect_lower"], avg["rel_effect_upper"]))
  rel_hi = _pct(max(avg["rel_effect_lower"], avg["rel_effect_upper"]))
  t = ["", "Analysis report {CausalImpact}", "", "",
       "During the post-intervention period, the response variable had",
       f"an average value of approx. {_r(avg['actual'], 1)}. " +
       ("By contrast, in" if sig else "In") + " the absence of an",
       f"intervention, we would have expected an average response of {_r(avg['predicted'], 1)}.",
       f"The {ci} interval of this counterfactual prediction is "
       f"[{_r(avg['predicted_lower'], 1)}, {_r(avg['predicted_upper'], 1)}].",
       "Subtracting this prediction from the observed response yields",
       "an estimate of the causal effect the intervention had on the",
       f"response variable. This effect is {_r(avg['abs_effect'], 1)} with a {ci} interval of",
       _sorted_pair(avg["abs_effect_lower"], avg["abs_effect_upper"]) +
       ". For a discussion of the significance of this effect,",
       "see below.", "", "",
       "Summing up the individual data points during the post-intervention",
       "period (which can only sometimes be meaningfully interpreted), the",
       f"response variable had an overall value of {_r(cum['actual'], 1)}.",
       ("By contrast, had" if sig else "Had") +
       " the intervention not taken place, we would have expected",
       f"a sum of {_r(cum['predicted'], 1)}. The {ci} interval of this prediction is " +
       _sorted_pair(cum["predicted_lower"], cum["predicted_upper"]) + ".", "", "",
       "The above results are given in terms of absolute numbers. In relative",
       "terms, the response variable showed " +
       ("an increase of +" if pos else "a decrease of ") + _pct(avg["rel_effect"]) + f". The {ci}",
       f"interval of this percentage is [{rel_lo}, {rel_hi}]."]
  if sig and pos:
    t += ["", "",
          "This means that the positive effect observed during the intervention",
          "period is statistically significant and unlikely to be due to random",
          "fluctuations. It should be noted, however, that the question of whether",
          "this increase also bears substantive significance can only be answered",
          f"by comparing the absolute effect ({_r(avg['abs_effect'], 1)}) to the original goal",
          "of the underlying intervention.", ""]
  elif sig and not pos:
    t += ["", "",
          "This means that the negative effect observed during the intervention",
          "period is statistically significant.",
          "If the experimenter had expected a positive effect, it is recommended",
          "to double-check whether anomalies in the control variables may have",
          "caused an overly optimistic expectation of what should have happened",
          "in the response variable in the absence of the intervention.", ""]
  elif pos:
    t += ["", "",
          "This means that, although the intervention appears to have caused a",
          "positive effect, this effect is not statistically significant when",
          "considering the entire post-intervention period as a whole. Individual",
          "days or shorter stretches within the intervention period may of course",
          "still have had a significant effect, as indicated whenever the lower",
          "limit of the impact time series (lower plot) was above zero.", ""]
  else:
    t += ["This means that, although it may look as though the intervention has",
          "exerted a negative effect on the response variable when considering",
          "the intervention period as a whole, this effect is not statistically",
          "significant and so cannot be meaningfully interpreted.", ""]
  body = "\n".join(t)
  if not sig:
    body += "\n".join(["", "",
                       "The apparent effect could be the result of random fluctuations that",
                       "are unrelated to the intervention. This is often the case when the",
                       "intervention period is very long and includes much of the time when",
                       "the effect has already worn off. It can also be the case when the",
                       "intervention period is too short to distinguish the signal from the",
                       "noise. Finally, failing to find a significant effect can happen when",
                       "there are not enough control variables or when these variables do not",
                       "correlate well with the response variable during the learning period.",
                       ""])
  if p_value < alpha:
    body += "\n".join(["", "",
                       "The probability of obtaining this effect by chance is very small",
                       f"(Bayesian one-sided tail-area probability p = {_r(p_value, 3)}).",
                       "This means the effect is statistically significant. It can be",
                       "considered causal if the model assumptions are satisfied."])
  else:
    body += "\n".join(["", "",
                       "The probability of obtaining this effect by chance is p = " +
                       "{0:.0%}".format(p_value) + ".",
                       "This means the effect may be spurious and would generally not be",
                       "considered statistically significant."])
  body += "\n".join(["", "", "",
                     "For more details, including the model assumptions behind the method, see",
                     "https://google.github.io/CausalImpact/."])
  return body


def summary(ci_model, output_format: str = "summary", alpha: Optional[float] = None) -> str:
  """Text summary ('summary') or long-form description ('report') of a CausalImpactAnalysis.

  Same arguments and errors as the reference (summary.py:133-178): `alpha` is inferred from the
  fitted analysis; passing a different one raises DeprecationWarning."""
  inferred_alpha = ci_model.summary.alpha.mean()
  if alpha is not None and alpha != inferred_alpha:
    raise DeprecationWarning("Supplying an argument to `alpha` is deprecated, "
                             "since it is inferred from `ci_model`. Set "
                             f"`alpha=None` to use alpha={inferred_alpha:.2f}, "
                             f"or retrain the model with alpha={alpha}.")
  alpha = inferred_alpha
  if output_format not in ["summary", "report"]:
    raise ValueError("`format` must be either 'summary' or 'report'. "
                     "Got %s" % output_format)
  if alpha <= 0. or alpha >= 1.:
    raise ValueError("`alpha` must be in (0, 1). Got %s" % alpha)
  p_value = float(ci_model.summary["p_value"].iloc[0])
  s = ci_model.summary.transpose().to_dict()
  if output_format == "summary":
    return _summary_table(s, float(alpha), p_value)
  return _report(s, float(alpha), p_value)


def summary_numbers(ci_model) -> dict:
  """The figures of the text summary as a nested dict of floats, rounded the way the text shows
  them -- for callers that want the table without parsing it:

      {"alpha": 0.05, "p_value": 0.038, "probability_of_effect": 0.9615,
       "average":    {"actual": 155.2, "predicted": 124.5, "predicted_sd": 0.55,
                      "predicted_interval": (123.5, 125.5), "abs_effect": 30.6, ...},
       "cumulative": {...}}

  Intervals are (lower, upper) with lower <= upper even when the stored columns are reversed
  (effects of negative sign), as in the text; relative quantities are fractions, not percent.
  """
  s = ci_model.summary.transpose().to_dict()
  p_value = float(ci_model.summary["p_value"].iloc[0])
  out = {"alpha": float(ci_model.summary.alpha.mean()), "p_value": round(p_value, 3),
         "probability_of_effect": round(1.0 - p_value, 4)}
  for row in ("average", "cumulative"):
    r = s[row]

    def pair(lo, hi, nd):
      a, b = round(float(r[lo]), nd), round(float(r[hi]), nd)
      return (min(a, b), max(a, b))

    out[row] = {
        "actual": round(float(r["actual"]), 1),
        "predicted": round(float(r["predicted"]), 1),
        "predicted_sd": round(float(r["predicted_sd"]), 2),
        "predicted_interval": pair("predicted_lower", "predicted_upper", 1),
        "abs_effect": round(float(r["abs_effect"]), 1),
        "abs_effect_sd": round(float(r["abs_effect_sd"]), 2),
        "abs_effect_interval": pair("abs_effect_lower", "abs_effect_upper", 1),
        "rel_effect": round(float(r["rel_effect"]), 3),
        "rel_effect_sd": round(float(r["rel_effect_sd"]), 3),
        "rel_effect_interval": pair("rel_effect_lower", "rel_effect_upper", 3),
    }
  return out
