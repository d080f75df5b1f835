"""Distributional parity with TFP-generated fixtures (tests/golden/tfp/*.json, written by
tests/golden/make_tfp_fixtures.py from the reference's own `_train_causalimpact_sts`).  No fixture
exists until that script has been run where TFP imports: the tests then SKIP and say so -- parity
of the Gibbs body with TFP is unpinned (DESIGN.md "Oracle")."""
import glob
import json
import os

import numpy as np
import pandas as pd
import pytest

import ref_pins_common as rp

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "tfp", "*.json")))
BACKENDS = ["oracle", pytest.param("gpu", marks=pytest.mark.gpu)]


def test_generator_is_guarded_and_writes_nothing_without_tfp(tmp_path):
  import subprocess
  import sys
  r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_tfp_fixtures.py"), "--out",
                      str(tmp_path)], capture_output=True, text=True, check=False)
  try:
    import tensorflow_probability  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
    have_tfp = True
  except ImportError:
    have_tfp = False
  if not have_tfp:
    assert r.returncode == 2 and "unpinned" in r.stdout
    assert not os.listdir(tmp_path)


@pytest.mark.skipif(not FIXTURES, reason="no TFP-generated fixture committed: parity with TFP unpinned")
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("path", FIXTURES or ["-"])
def test_posterior_summaries_match_tfp_within_its_own_seed_spread(backend, path):
  fx = json.load(open(path))
  df = pd.DataFrame(fx["data"], columns=fx["columns"], dtype=float)
  try:
    df.index = pd.to_datetime(fx["index"])
    conv = pd.Timestamp
  except (ValueError, TypeError):
    df.index = [int(i) for i in fx["index"]]
    conv = int
  pre, post = tuple(conv(v) for v in fx["pre"]), tuple(conv(v) for v in fx["post"])
  ours = []
  for s in range(len(fx["seeds"])):
    an = rp.fit(backend, df, pre, post, seed=(1, s), num_results=fx["num_results"],
                num_warmup=fx["num_warmup"], prior_level_sd=fx["prior_level_sd"])
    ps = an.posterior_samples
    ours.append(dict(sigma_obs=float(np.mean(ps.observation_noise_scale)),
                     sigma_level=float(np.mean(ps.level_scale)),
                     w=None if ps.weights is None else np.asarray(ps.weights).mean(axis=0),
                     incl=None if ps.weights is None else (np.asarray(ps.weights) != 0).mean(axis=0)))
  ref = fx["per_seed"]

  def check(name, a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    se = np.sqrt(a.var(axis=0, ddof=1) / len(a) + b.var(axis=0, ddof=1) / len(b))
    diff = np.abs(a.mean(axis=0) - b.mean(axis=0))
    assert (diff <= np.maximum(4 * se, 0.01 * np.abs(b.mean(axis=0)) + 1e-3)).all(), (name, diff, se)

  check("sigma_obs", [o["sigma_obs"] for o in ours], [r["sigma_obs"]["mean"] for r in ref])
  check("sigma_level", [o["sigma_level"] for o in ours], [r["sigma_level"]["mean"] for r in ref])
  if ours[0]["w"] is not None:
    check("weights", [o["w"] for o in ours], [r["weights_mean"] for r in ref])
    check("inclusion", [o["incl"] for o in ours], [r["weights_inclusion"] for r in ref])
