"""Guarded generator of TFP-produced sampler fixtures -- the only route by which parity of the
Gibbs body with tensorflow-probability can ever be pinned (DESIGN.md "Oracle": until this has
been run somewhere and its output committed, per-draw and distributional parity with TFP stays
UNPINNED).

Where `tensorflow_probability` AND the reference package import (e.g. a machine with
`pip install tfp-causalimpact`; neither exists in the build container or on the GPU box):

    python tests/golden/make_tfp_fixtures.py [--out tests/golden/tfp] [--seeds 8]
                                             [--substrate tensorflow|numpy|jax]

(`--substrate numpy` / `jax` alias TFP's TensorFlow-free substrates under the names the reference
imports, so a machine with `pip install tfp-nightly tfp-causalimpact` and NO TensorFlow can
produce the fixtures; best effort, reports what is missing otherwise.)

it runs the REFERENCE's own hot path -- `causalimpact.causalimpact_lib._train_causalimpact_sts`,
i.e. `gibbs_sampler.fit_with_gibbs_sampling(...)` exactly as called at
`causalimpact/causalimpact_lib.py:365-388` with the priors of `:398-500` -- on three seeded
recipes and writes, per recipe, one small JSON of posterior SUMMARIES (no reference code, no draws):
means, standard deviations and 2.5 / 97.5 % quantiles of sigma_obs and sigma_level, mean and
inclusion frequency of every weight, mean / quantiles of the post-period average of the
posterior-predictive trajectories, for each of `--seeds` sampler seeds (so that the consumer
knows the Monte-Carlo spread), together with the exact inputs (y, X, periods) it used.
`tests/test_tfp_fixtures.py` then checks the oracle (CPU) and the HIP path (GPU) against them
within that spread; while the directory holds no fixture the test SKIPS with
"parity with TFP unpinned".

Recipes (the `north_star` configs that the reference itself can run):
  datacsv  the reference's testdata/data.csv with y[1,3,7] = NaN, pre = rows 0..19
           (causalimpact_lib_test.py:204-220, 242-271), 100 + 100 iterations
  cfg1     quickstart: T=100, 1 covariate (docs/quickstart.ipynb:279-298), 900 draws
  cfg2     T=1000, 10 covariates, 1000 draws (LocalLevel: the reference has no slope)
Nothing here is imported by the product, the tests' hot paths or the GPU box.
"""
import argparse
import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def _alias_substrate(substrate: str):
  """Runs the reference on a TensorFlow-FREE substrate of TFP (`pip install tfp-nightly[jax]` or
  just `tfp-nightly` + numpy): `tensorflow_probability.substrates.{numpy,jax}` re-exports the whole
  library -- including `experimental.sts_gibbs.gibbs_sampler` and `sts` -- over a backend that
  mimics the `tf` API (`tensorflow_probability.python.internal.backend.{numpy,jax}`).  The
  reference imports the plain names `tensorflow` / `tensorflow_probability` (and four
  `tensorflow_probability.python...` sub-modules, causalimpact_lib.py:31-35), so those names are
  aliased in sys.modules BEFORE it is imported.  Best effort: returns an error string when the
  installed TFP lacks a piece (nothing is written then)."""
  import importlib  # pylint: disable=import-outside-toplevel
  try:
    sub = importlib.import_module(f"tensorflow_probability.substrates.{substrate}")
    backend = importlib.import_module(f"tensorflow_probability.python.internal.backend.{substrate}")
  except ImportError as e:
    return f"tensorflow_probability.substrates.{substrate} is not importable here ({e})"
  tf_like = getattr(backend, "v2", backend)
  sys.modules["tensorflow"] = tf_like
  sys.modules["tensorflow_probability"] = sub
  base = f"tensorflow_probability.substrates.{substrate}"
  for tail in ("experimental.distributions", "experimental.sts_gibbs.gibbs_sampler",
               "experimental.sts_gibbs", "internal.prefer_static", "sts", "distributions",
               "bijectors"):
    try:
      mod = importlib.import_module(base + "." + tail)
    except ImportError as e:
      return f"the {substrate} substrate lacks {tail} ({e})"
    sys.modules["tensorflow_probability.python." + tail] = mod
  return None


def _import_reference(substrate: str = "tensorflow"):
  """The pip-installed reference, never this repository's drop-in package of the same name."""
  if substrate != "tensorflow":
    why = _alias_substrate(substrate)
    if why:
      return None, why
  try:
    import tensorflow_probability  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
  except ImportError:
    return None, "tensorflow_probability is not importable here"
  for p in list(sys.path):
    if os.path.abspath(p or ".").startswith(ROOT):
      sys.path.remove(p)
  try:
    import causalimpact  # pylint: disable=import-outside-toplevel
    from causalimpact import causalimpact_lib  # pylint: disable=import-outside-toplevel
  except ImportError as e:
    return None, f"the reference package is not importable here ({e})"
  except Exception as e:  # pylint: disable=broad-except
    return None, f"importing the reference on the {substrate} substrate failed ({type(e).__name__}: {e})"
  if os.path.abspath(causalimpact.__file__).startswith(ROOT):
    return None, "`import causalimpact` resolved to this repository's drop-in, not the reference"
  if not hasattr(causalimpact_lib, "_train_causalimpact_sts"):
    return None, "the imported causalimpact has no _train_causalimpact_sts"
  return causalimpact, None


def _recipes():
  sys.path.insert(0, os.path.join(ROOT, "tfp-causalimpact_amd", "causalimpact"))
  import _synthetic as syn  # pylint: disable=import-outside-toplevel  (numpy only)
  sys.path.pop(0)
  out = {}
  df = pd.read_csv(os.path.join(HERE, "ref_testdata", "data.csv"))
  df = df.set_index(pd.to_datetime(df["t"])).drop(columns=["t"])
  df.loc[df.index[[1, 3, 7]], "y"] = np.nan
  out["datacsv"] = dict(df=df, pre=(df.index[0], df.index[19]), post=(df.index[20], df.index[-1]),
                        num_results=100, num_warmup=100)
  rng = np.random.default_rng(20210614)
  n = 100
  x = np.zeros(n)
  x[0] = rng.normal()
  for t in range(1, n):
    x[t] = 0.999 * x[t - 1] + rng.normal()
  y = 1.2 * (100.0 + x) + rng.normal(size=n)
  y[72:] += 10.0
  d1 = pd.DataFrame({"y": y, "x1": 100.0 + x}, index=pd.date_range("2021-06-14", periods=n, freq="D"))
  out["cfg1"] = dict(df=d1, pre=(d1.index[0], d1.index[70]), post=(d1.index[71], d1.index[-1]),
                     num_results=900, num_warmup=100)
  yy, XX = syn.make_raw_series(1000, 10, 2024)
  d2 = pd.DataFrame(np.column_stack([yy, XX]), columns=["y"] + [f"x{j}" for j in range(10)])
  out["cfg2"] = dict(df=d2, pre=(0, 699), post=(700, 999), num_results=1000, num_warmup=112)
  return out


def _summaries(samples, trajectories, post_rows):
  def q(v):
    v = np.asarray(v, np.float64)
    lo, hi = np.quantile(v, [0.025, 0.975])
    return dict(mean=float(v.mean()), sd=float(v.std(ddof=1)), q025=float(lo), q975=float(hi))
  out = dict(sigma_obs=q(samples.observation_noise_scale), sigma_level=q(samples.level_scale))
  if samples.weights is not None:
    w = np.asarray(samples.weights, np.float64)
    out["weights_mean"] = [float(v) for v in w.mean(axis=0)]
    out["weights_inclusion"] = [float(v) for v in (w != 0).mean(axis=0)]
  tr = np.asarray(trajectories, np.float64)          # [draws, T] on the standardized scale
  out["post_mean_prediction"] = q(tr[:, post_rows].mean(axis=1))
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--out", default=os.path.join(HERE, "tfp"))
  ap.add_argument("--seeds", type=int, default=8)
  ap.add_argument("--substrate", choices=("tensorflow", "numpy", "jax"), default="tensorflow",
                  help="run TFP on this substrate; numpy / jax need no TensorFlow install")
  args = ap.parse_args()
  ci, why = _import_reference(args.substrate)
  if ci is None:
    print(f"make_tfp_fixtures: nothing written -- {why}.  Parity with TFP stays unpinned.")
    return 2
  import tensorflow as tf  # pylint: disable=import-outside-toplevel
  import tensorflow_probability as tfp  # pylint: disable=import-outside-toplevel
  lib = ci.causalimpact_lib
  os.makedirs(args.out, exist_ok=True)
  for name, r in _recipes().items():
    ci_data = ci.data.CausalImpactData(r["df"], r["pre"], r["post"], standardize_data=True,
                                       dtype=tf.float32)
    n_pre = int(ci_data.model_pre_data.shape[0])
    n_all = n_pre + int(ci_data.model_after_pre_data.shape[0])
    post_rows = np.arange(n_pre, n_all)
    per_seed = []
    for s in range(args.seeds):
      samples, _, trajectories = lib._train_causalimpact_sts(     # pylint: disable=protected-access
          ci_data=ci_data, prior_level_sd=0.01, seed=(0, s), num_results=r["num_results"],
          num_warmup_steps=r["num_warmup"], dtype=tf.float32, seasons=[])
      packed = lib.CausalImpactPosteriorSamples(
          observation_noise_scale=np.asarray(samples.observation_noise_scale),
          level_scale=np.asarray(samples.level_scale), level=np.asarray(samples.level),
          weights=np.asarray(samples.weights) if np.size(samples.weights) else None,
          seasonal_drift_scales=None, seasonal_levels=None)
      per_seed.append(_summaries(packed, np.asarray(trajectories), post_rows))
    fixture = dict(
        recipe=name, generator="tests/golden/make_tfp_fixtures.py",
        substrate=args.substrate,
        versions=dict(tensorflow=getattr(tf, "__version__", "substrate backend"),
                      tensorflow_probability=getattr(tfp, "__version__", "?"),
                      causalimpact=getattr(ci, "__version__", "?")),
        call="causalimpact_lib._train_causalimpact_sts (gibbs_sampler.fit_with_gibbs_sampling, "
             "causalimpact_lib.py:365-388)",
        num_results=r["num_results"], num_warmup=r["num_warmup"], prior_level_sd=0.01,
        pre=[str(v) for v in r["pre"]], post=[str(v) for v in r["post"]],
        columns=list(r["df"].columns), index=[str(i) for i in r["df"].index],
        data=[[None if not np.isfinite(v) else float(v) for v in row] for row in r["df"].values],
        seeds=[[0, s] for s in range(args.seeds)], per_seed=per_seed)
    with open(os.path.join(args.out, f"{name}.json"), "w") as f:
      json.dump(fixture, f)
    print("wrote", name)
  return 0


if __name__ == "__main__":
  sys.exit(main())
