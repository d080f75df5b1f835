"""Measures the BASELINE configs that are not the bench line (bench.py = cfg2) on one MI355X and
writes one JSON line per config (committed as profiles/r01_configs.json):
  cfg4: T=10000, 50 covariates + Seasonal(num_seasons=7), 1000 samples (1 and 8 chains)
  cfg5: 512 independent series, T=500, 5 covariates (what one GPU does with its 64-series share,
        and all 512 on one GPU)
Throughput = retained draws / Gibbs-kernel time (HIP events), inputs resident in HBM, plus the
end-to-end time of the batched API for cfg5."""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _model, _native
from causalimpact import _synthetic as syn

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md
REF_SEASONS = (ci.Seasons(num_seasons=4, num_steps_per_season=(2, 1, 1, 1)), ci.Seasons(num_seasons=7),
               ci.Seasons(num_seasons=6, num_steps_per_season=((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1))))


def cpu_baseline(y, mask, X, spec, iters, what):
  """The float64 oracle (oracle/ci_oracle.c, `make native`: -O3 -march=native -fopenmp, built on this
  host) on `iters` Gibbs iterations of the same series: one core, and one chain per core on all
  cores.  Test infrastructure used as the reported CPU figure ("port"), never as the product."""
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  _, flags = orc.native_lib()
  kw = dict(num_results=iters, num_warmup=0, seed=(0, 1))
  orc.fit_gibbs_chains_native(y, mask, X, spec, first_chain=0, n_chains=1, threads=1,
                              num_results=2, num_warmup=0, seed=(0, 1))
  t0 = time.perf_counter()
  orc.fit_gibbs_chains_native(y, mask, X, spec, first_chain=0, n_chains=1, threads=1, **kw)
  dt1 = time.perf_counter() - t0
  ncpu = os.cpu_count() or 1
  t0 = time.perf_counter()
  used = orc.fit_gibbs_chains_native(y, mask, X, spec, first_chain=100, n_chains=ncpu, threads=ncpu, **kw)
  dtn = time.perf_counter() - t0
  return {"kind": "port", "unit": "posterior samples/sec", "cores": 1, "value": iters / dt1,
          "us_per_iteration": dt1 / iters * 1e6,
          "all_cores": {"cores": used, "value": ncpu * iters / dtn,
                        "sample": f"{ncpu} chains x {iters} iterations, one chain per thread, {dtn:.1f} s"},
          "sample": f"1 chain x {iters} Gibbs iterations of {what}, float64 C restatement built here "
                    f"with `{flags}`, {dt1:.1f} s"}


def measured_traffic(which, kernel):
  """HBM bytes per launch of `kernel`, measured by two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE,
  each in its own child run of `run_configs.py <which>`; KiB -> bytes; FETCH_SIZE x 2 on gfx950 as the
  guide prescribes).  (bytes, note) or (None, why)."""
  prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
  if not os.path.exists(prof):
    return None, "rocprofv3 not found"
  tmp = tempfile.mkdtemp(prefix="ci_cfg_pmc_", dir="/tmp")
  per = {}
  try:
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
      out = os.path.join(tmp, counter)
      cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "cfg", "--",
             sys.executable, os.path.abspath(__file__), which]
      r = subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdin=subprocess.DEVNULL,
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=False)
      files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
      if r.returncode != 0 or not files:
        return None, f"rocprofv3 --pmc {counter} pass failed (rc {r.returncode})"
      tot, ids = 0.0, set()
      with open(files[0]) as f:
        for row in csv.DictReader(f):
          if row.get("Counter_Name") == counter and kernel in row.get("Kernel_Name", ""):
            tot += float(row["Counter_Value"])
            ids.add(row["Dispatch_Id"])
      if not ids:
        return None, f"no {kernel} dispatch in the {counter} pass"
      per[counter] = tot / len(ids)
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  return (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"


def roofline(nbytes, ms, traffic=None, traffic_note=None):
  ach = nbytes / ms / 1e6
  r = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
       "traffic": traffic}
  if traffic is not None:
    r["traffic_over_algorithmic"] = traffic / nbytes
  if traffic_note:
    r["traffic_source"] = traffic_note
  return r


def _seasonal_fit(T, p, seasons, chains, S, W, flags=0, data_seed=0):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, data_seed)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  spec = _model.series_params(y, mask, X, num_seasonal_blocks=len(seasons))
  counts, flg = _model.expand_seasons(seasons, T)
  pb = _native.make_problem(T=T, P=X.shape[1], has_slope=0, num_seasons=counts, num_warmup=W,
                            num_results=S, num_chains=chains, seed=(0, 1), flags=flags)
  sess = _native.Session(pb, y[None], mask[None], X[None], flg if seasons else None, _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  nbytes, name = sess.algorithmic_bytes(), sess.kernel_name()
  sess.close()
  return ms, nbytes, name, (y, mask, X, seasons)


def _oracle_spec(y, mask, X, seasons):
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  return orc.default_spec(y, mask, X, seasons=tuple(
      (int(s.num_seasons), s.num_steps_per_season) for s in seasons))


def cfg4(chains, S=1000, with_cpu=False, with_traffic=False):
  T, p = 10000, 50
  W = -(-S // 9)
  ms, nbytes, name, data = _seasonal_fit(T, p, (ci.Seasons(num_seasons=7),), chains, S, W)
  row = {"config": "cfg4", "workload": "T=10000, 50 covariates (P=51), LocalLevel + Seasonal(7) + spike-slab",
         "kernel": name, "chains": chains, "num_results": S, "num_warmup": W, "kernel_ms": ms,
         "us_per_iteration": ms / (W + S) * 1e3, "samples_per_s": chains * S / ms * 1e3,
         "algorithmic_GBps": nbytes / ms / 1e6}
  traffic, note = (None, None)
  if with_traffic:
    # counter passes over THIS launch (same chains, same S) in child runs of `run_configs.py cfg4_c<chains>_s<S>`
    traffic, note = measured_traffic(f"cfg4_c{chains}_s{S}", "gibbs_wide")
  row["roofline"] = roofline(nbytes, ms, traffic, note)
  if with_cpu:
    y, mask, X, seasons = data
    row["cpu_baseline"] = cpu_baseline(y, mask, X, _oracle_spec(y, mask, X, seasons), 200, "the cfg4 series")
    row["gpu_over_one_core_per_chain"] = row["cpu_baseline"]["us_per_iteration"] / row["us_per_iteration"]
  return row


def general_seasonal(chains=8, S=100, with_cpu=True):
  """The reference's own multi-block test model (4 + 7 + 6 seasons, causalimpact_lib_test.py:738-752)
  at cfg4's size: the time-parallel cluster kernel (csrc/ci_seasonal_tp.h), the sequential
  one-wavefront kernel it replaces as the default route, and the oracle on the host."""
  T, p = 10000, 50
  W = 10
  out = []
  for label, flags in (("time-parallel (default)", 0), ("sequential (CI_FLAG_SEQUENTIAL_SEASONAL)", _native.FLAG_SEQUENTIAL_SEASONAL)):
    ms, nbytes, name, data = _seasonal_fit(T, p, REF_SEASONS, chains, S, W, flags=flags)
    out.append({"config": "T=10000, 50 covariates + Seasons 4/7/6 (D = 18)", "route": label, "kernel": name,
                "chains": chains, "num_results": S, "num_warmup": W, "kernel_ms": ms,
                "us_per_iteration": ms / (W + S) * 1e3, "samples_per_s": chains * S / ms * 1e3,
                "roofline": roofline(nbytes, ms)})
  if not with_cpu:
    return out
  y, mask, X, seasons = data
  cpu = cpu_baseline(y, mask, X, _oracle_spec(y, mask, X, seasons), 60, "this series")
  for row in out:
    row["cpu_baseline"] = cpu
    row["gpu_over_one_core_per_chain"] = cpu["us_per_iteration"] / row["us_per_iteration"]
  return out


def cfg5(B, S=1000, with_cpu=False):
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
  nbytes, name = sess.algorithmic_bytes(), sess.kernel_name()
  sess.close()
  row = {"config": "cfg5", "workload": "independent series, T=500, 5 covariates (P=6), LocalLevel + spike-slab",
         "kernel": name, "series": B, "num_results": S, "num_warmup": W, "kernel_ms": ms,
         "samples_per_s": B * S / ms * 1e3, "algorithmic_GBps": nbytes / ms / 1e6,
         "roofline": roofline(nbytes, ms),
         "batched_api_wall_s": wall, "batched_api_samples_per_s": B * S / wall,
         "mean_abs_effect": float(res.summary.xs("average", level=1)["abs_effect"].mean())}
  if with_cpu:
    # the host figure: the oracle on ONE of the series (they all cost the same), one series per core
    from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
    cpu = cpu_baseline(prep.y[0], prep.mask[0], prep.design[0],
                       orc.default_spec(prep.y[0], prep.mask[0], prep.design[0]), 4000,
                       "one series of the batch")
    row["cpu_baseline"] = cpu
    row["gpu_over_all_host_cores"] = row["samples_per_s"] / cpu["all_cores"]["value"]
  return row


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
  # 100 covariates (regression block in the HBM workspace): time-parallel kernel with the
  # workgroup-wide draw (round 5), the sequential route, and the oracle on the host
  T, p, W, S, C = 1000, 100, 50, 200, 8
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  spec = _model.series_params(y, mask, X)
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  cpu = cpu_baseline(y, mask, X, orc.default_spec(y, mask, X), 1500, "this series")
  for flags in (0, _native.FLAG_SEQUENTIAL_SEASONAL):
    pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=C,
                              seed=(0, 1), flags=flags)
    sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
    sess.run()
    ms = sess.run()
    out.append({"config": "T=1000, 100 covariates (P=101)", "kernel": sess.kernel_name(), "chains": C,
                "kernel_ms": ms, "us_per_iteration": ms / (W + S) * 1e3, "samples_per_s": C * S / ms * 1e3,
                "roofline": roofline(sess.algorithmic_bytes(), ms), "cpu_baseline": cpu,
                "gpu_over_one_core_per_chain": cpu["us_per_iteration"] / (ms / (W + S) * 1e3)})
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
  if which in ("general", "general_gpu"):     # (general_gpu: the counter / trace passes, no CPU leg)
    for row in general_seasonal(with_cpu=which == "general"):
      print(json.dumps(row), flush=True)
    sys.exit(0)
  if which == "cfg4":        # counter passes: one short cfg4 fit per chain count
    runs = (lambda: cfg4(1, S=200), lambda: cfg4(8, S=200))
  elif which.startswith("cfg4_c"):   # counter passes of one launch shape: cfg4_c<chains>_s<S>
    cc, ss = which[len("cfg4_c"):].split("_s")
    runs = (lambda: cfg4(int(cc), S=int(ss)),)
  elif which == "cfg4_full":   # counter passes at BASELINE's own size: 8 chains x (112 + 1000) iterations
    runs = (lambda: cfg4(8),)
  else:
    runs = (lambda: cfg4(1, with_traffic=True), lambda: cfg4(8, with_cpu=True, with_traffic=True),
            lambda: cfg4(32, with_traffic=True),      # 32 chains x 8 CUs: the whole chip
            lambda: cfg5(64, with_cpu=True), lambda: cfg5(512, with_cpu=True))
  for run in runs:
    print(json.dumps(run()), flush=True)
  if which == "all":
    for row in general_seasonal():
      print(json.dumps(row), flush=True)
