"""Measures the BASELINE configs that are not the bench line (bench.py = cfg2) on one MI355X and
writes one JSON line per config (committed as profiles/r01_configs.json):
  cfg4: T=10000, 50 covariates + Seasonal(num_seasons=7), 1000 samples (1 and 8 chains)
  cfg5: 512 independent series, T=500, 5 covariates (what one GPU does with its 64-series share,
        and all 512 on one GPU)
Throughput = retained draws / Gibbs-kernel time (HIP events), inputs resident in HBM, plus the
end-to-end time of the batched API for cfg5."""
import json
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _model, _native
from causalimpact import _synthetic as syn


def cfg4(chains, S=1000):
  T, p = 10000, 50
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  spec = _model.series_params(y, mask, X, num_seasonal_blocks=1)
  counts, flg = _model.expand_seasons((ci.Seasons(num_seasons=7),), T)
  W = -(-S // 9)
  pb = _native.make_problem(T=T, P=X.shape[1], has_slope=0, num_seasons=counts, num_warmup=W,
                            num_results=S, num_chains=chains, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  nbytes = sess.algorithmic_bytes()
  sess.close()
  return {"config": "cfg4", "workload": "T=10000, 50 covariates (P=51), LocalLevel + Seasonal(7) + spike-slab",
          "chains": chains, "num_results": S, "num_warmup": W, "kernel_ms": ms,
          "us_per_iteration": ms / (W + S) * 1e3, "samples_per_s": chains * S / ms * 1e3,
          "algorithmic_GBps": nbytes / ms / 1e6}


def cfg5(B, S=1000):
  T, p = 500, 5
  frames = []
  for b in range(B):
    y, X = syn.make_raw_series(T, p, b)
    frames.append(np.column_stack([y, X]))
  values = np.stack(frames)
  opts = ci.InferenceOptions(num_results=S)
  ci.fit_causalimpact_batch(values[:2], (0, 349), (350, 499), seed=1, inference_options=opts)
  t0 = time.time()
  res = ci.fit_causalimpact_batch(values, (0, 349), (350, 499), seed=1, inference_options=opts)
  wall = time.time() - t0
  # kernel time of the same launch
  prep = ci.batch.prepare_batch(values, pd.RangeIndex(T), (0, 349), (350, 499))
  params = [_model.series_params(prep.y[b], prep.mask[b], prep.design[b]) for b in range(B)]
  W = opts.num_warmup_steps
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=1,
                            num_series=B, seed=(0, 1))
  sess = _native.Session(pb, prep.y, prep.mask, prep.design, None, _native.make_params(params))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  nbytes = sess.algorithmic_bytes()
  sess.close()
  return {"config": "cfg5", "workload": "independent series, T=500, 5 covariates (P=6), LocalLevel + spike-slab",
          "series": B, "num_results": S, "num_warmup": W, "kernel_ms": ms,
          "samples_per_s": B * S / ms * 1e3, "algorithmic_GBps": nbytes / ms / 1e6,
          "batched_api_wall_s": wall, "batched_api_samples_per_s": B * S / wall,
          "mean_abs_effect": float(res.summary.xs("average", level=1)["abs_effect"].mean())}


def extras():
  """Round-3 capability routes, timed once each (wall time of the C-ABI call)."""
  out = []
  # float64 compute at cfg2's size (sequential kernel, csrc/ci_gibbs64.h)
  T, p, W, S, C = 1000, 10, 112, 1000, 8
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
  spec = _model.series_params(y, mask, X, has_slope=True)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=W, num_results=S, num_chains=C,
                            seed=(0, 1))
  prm = _native.make_params([spec])
  _native.fit_gibbs_f64(_native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=1, num_results=2,
                                             seed=(0, 1)), y[None], mask[None], X[None], None, prm)
  t0 = time.time()
  _native.fit_gibbs_f64(pb, y[None], mask[None], X[None], None, prm,
                        want=("observation_noise_scale", "posterior_means"))
  dt = time.time() - t0
  kms = _native.fit_gibbs_f64_kernel_ms()
  out.append({"config": "cfg2 in float64", "kernel": "ci::gibbs64_trend_kernel (eight wavefronts per chain)",
              "chains": C, "wall_s": dt, "kernel_ms": kms, "us_per_iteration": kms / (W + S) * 1e3,
              "samples_per_s": C * S / kms * 1e3})
  # 100 covariates (regression block in the HBM workspace)
  T, p, W, S, C = 1000, 100, 50, 200, 8
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  spec = _model.series_params(y, mask, X)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=C,
                            seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  sess.run()
  ms = sess.run()
  out.append({"config": "T=1000, 100 covariates (P=101)", "kernel": sess.kernel_name(), "chains": C,
              "kernel_ms": ms, "us_per_iteration": ms / (W + S) * 1e3, "samples_per_s": C * S / ms * 1e3})
  sess.close()
  # HMC with a weekly block: the time-parallel scans (csrc/ci_wide_score.h) and, forced, the
  # sequential one-wavefront route (csrc/ci_score_seq.h); and a long trend-only series
  for label, T, seasons, slope, flags in (
      ("T=1000, 10 covariates + Seasonal(7), HMC", 1000, (ci.Seasons(num_seasons=7),), 0, 0),
      ("T=1000, 10 covariates + Seasonal(7), HMC, sequential route forced", 1000,
       (ci.Seasons(num_seasons=7),), 0, _native.FLAG_SEQUENTIAL_SEASONAL),
      ("T=8000, 10 covariates, local linear trend, HMC", 8000, (), 1, 0),
      ("T=8000, 10 covariates, local linear trend, HMC, sequential route forced", 8000, (), 1,
       _native.FLAG_SEQUENTIAL_SEASONAL)):
    p, W, S, C, NL = 10, 100, 100, 8, 15
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
    spec = _model.series_params(y, mask, X, has_slope=bool(slope), num_seasonal_blocks=len(seasons))
    counts, flg = _model.expand_seasons(seasons, T)
    pb = _native.make_problem(T=T, P=p + 1, has_slope=slope, num_seasons=counts, num_warmup=0,
                              num_results=1, seed=(0, 1), flags=flags)
    ll = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=8,
                               season_change=flg if seasons else None)
    hmc_ms, lat_ms = ll.hmc_run(num_chains=C, num_warmup=W, num_results=S, num_leapfrog=NL, seed=(0, 1))
    out.append({"config": label, "kernel": ll.kernel_name(), "chains": C,
                "kernel_ms": hmc_ms, "latents_ms": lat_ms,
                "us_per_leapfrog": hmc_ms * 1e3 / ((W + S) * NL),
                "samples_per_s": C * S / (hmc_ms + lat_ms) * 1e3})
    ll.close()
  return out


if __name__ == "__main__":
  which = sys.argv[1] if len(sys.argv) > 1 else "all"
  if which == "extras":
    for row in extras():
      print(json.dumps(row), flush=True)
    sys.exit(0)
  if which == "cfg4":        # counter passes: one short cfg4 fit per chain count
    runs = (lambda: cfg4(1, S=200), lambda: cfg4(8, S=200))
  else:
    runs = (lambda: cfg4(1), lambda: cfg4(8), lambda: cfg5(64), lambda: cfg5(512))
  for run in runs:
    print(json.dumps(run()), flush=True)
