"""summary() text against (a) the reference's own golden texts (its summary_test.py inputs,
restated here; texts under tests/golden/ref_testdata) and (b) the texts the reference rendered on
this build's golden cases (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pandas as pd
import pytest

import causalimpact as ci

HERE = os.path.dirname(__file__)
REF = os.path.join(HERE, "golden", "ref_testdata")


def _fake_analysis(p_value, rel=None):
  # summary_test.py:31-62 (values are the fixture the golden texts were rendered from)
  cols = ["actual", "predicted", "predicted_lower", "predicted_upper", "predicted_sd",
          "abs_effect", "abs_effect_lower", "abs_effect_upper", "abs_effect_sd", "rel_effect",
          "rel_effect_lower", "rel_effect_upper", "rel_effect_sd", "p_value", "alpha"]
  avg = [5.343, 4.343, 3.343, 6.343, 0.001, 3.343, 2.343, 6.343, 0.001, 0.123, 0.143, 0.343,
         0.001, 0.001, 0.100]
  cum = [10.343, 9.343, 8.343, 9.343, 0.100, 10.343, 4.343, 9.343, 0.100, 0.233, 0.133, 0.333,
         0.100, 0.001, 0.100]
  summary = pd.DataFrame([avg, cum], columns=cols, index=["average", "cumulative"])
  summary["p_value"] = p_value
  if rel is not None:
    summary.loc["average", ["rel_effect", "rel_effect_lower", "rel_effect_upper"]] = rel
  return ci.CausalImpactAnalysis(series=pd.DataFrame(), summary=summary, posterior_samples=[])


@pytest.mark.parametrize("p_value,rel,num", [
    (0.5, [0.41, -0.30, 0.30], 1),
    (0.05, [0.41, 0.434, 0.234], 2),
    (0.5, [-0.343, -0.434, 0.234], 3),
    (0.05, [-0.343, -0.434, -0.234], 4),
])
def test_report_matches_reference_golden_text(p_value, rel, num):
  out = ci.summary(_fake_analysis(p_value, rel), output_format="report", alpha=0.1).strip()
  with open(os.path.join(REF, f"test_report_text_{num}.txt")) as f:
    assert out == f.read().strip()


def test_summary_matches_reference_golden_text():
  out = ci.summary(_fake_analysis(0.459329), output_format="summary", alpha=0.1).strip()
  with open(os.path.join(REF, "test_summary_output.txt")) as f:
    assert out == f.read().strip()


@pytest.mark.parametrize("fmt", ["summary", "report"])
def test_texts_rendered_by_the_reference_on_golden_cases(fmt):
  names = [n[:-5] for n in sorted(os.listdir(os.path.join(HERE, "golden")))
           if n.endswith(".json") and n != "index.json"]
  assert names
  for name in names:
    with open(os.path.join(HERE, "golden", name + ".json")) as f:
      g = json.load(f)
    sm = g["summary"]
    summary = pd.DataFrame({c: np.array(sm["data"][c], dtype=np.float64) for c in sm["columns"]},
                           index=sm["index"])
    an = ci.CausalImpactAnalysis(series=pd.DataFrame(), summary=summary, posterior_samples=[])
    want = g["summary_text"] if fmt == "summary" else g["report_text"]
    assert ci.summary(an, output_format=fmt) == want, name


def test_argument_errors_match_reference():
  an = _fake_analysis(0.5)
  with pytest.raises(DeprecationWarning):
    ci.summary(an, alpha=0.3)
  with pytest.raises(ValueError, match="must be either 'summary' or 'report'"):
    ci.summary(an, output_format="nope")


def test_summary_numbers_agree_with_the_text():
  an = _fake_analysis(0.459329)
  num = ci.summary_numbers(an)
  text = ci.summary(an, alpha=0.1)
  assert str(num["average"]["actual"]) in text and str(num["cumulative"]["predicted"]) in text
  lo, hi = num["average"]["abs_effect_interval"]
  assert lo <= hi and f"[{lo}, {hi}]" in text
  assert num["p_value"] == 0.459 and abs(num["alpha"] - 0.1) < 1e-12
