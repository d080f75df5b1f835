#!/usr/bin/env python3
"""bench.py -- posterior samples/sec of the posterior-inference hot path on MI355X.

Default workload (BASELINE.json configs[1], "cfg2"): T=1000, 10 covariates (+intercept => P=11),
LocalLinearTrend + spike-and-slab regression, 1000 retained Gibbs draws x 8 chains per GPU,
W = ceil(1000/9) = 112 warm-up iterations, float32.  `--sampler hmc` runs configs[2] ("cfg3"):
the same series, windowed-adaptive HMC (15 leapfrog steps, 500 warm-up iterations, 1000 retained
draws), 8 chains per GPU = 64 chains on 8 GPUs, plus the latent-path / predictive draw of every
retained sample.  One "step" = one complete fit (all iterations of all chains of this rank).
Chains are independent, so ranks shard them with no data-path collective (weak scaling: 8 chains
per GPU); RCCL -- bound through the C-ABI (ci_comm_*), no PyTorch in this process -- is used only
for the barrier / max-over-ranks timing and, after the timed region, to gather per-chain blocks
from HBM and all-reduce the diagnostics' partial sums.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--sampler gibbs|hmc]
      N > 1 without a launcher: bench.py starts its own N ranks (one process per GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
      (RANK / LOCAL_RANK / WORLD_SIZE from the environment; torch itself is never imported)
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")):
  if p not in sys.path:
    sys.path.insert(0, p)

from causalimpact import _model  # noqa: E402
from causalimpact import _native  # noqa: E402
from causalimpact import _synthetic as syn  # noqa: E402

CFG = dict(T=1000, covariates=10, has_slope=1, num_results=1000, num_warmup=112,
           chains_per_gpu=8, data_seed=2024, seed=(0, 20240927),
           hmc_warmup=500, hmc_leapfrog=15, chunk_draws=32)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _cpu_chain(sampler, y, mask, X, spec, S, W, chain):
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  if sampler == "hmc":
    orc.fit_hmc(y, mask, X, spec, num_results=S, num_warmup=W, num_leapfrog=CFG["hmc_leapfrog"],
                seed=CFG["seed"], chain=chain)
  else:
    orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=CFG["seed"], chain=chain,
                  want=("obs_scale", "level_scale", "slope_scale", "weights", "level", "slope",
                        "pred_mean", "trajectories"))
  return 0


def _cpu_worker(sampler, y, mask, X, spec, S, W, first_chain, go, seconds, done):
  """One host core: whole chains until `seconds` after the common start; reports how many."""
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  orc.lib()
  go.wait()
  t0 = time.perf_counter()
  n = 0
  while time.perf_counter() - t0 < seconds:
    _cpu_chain(sampler, y, mask, X, spec, S, W, first_chain + n)
    n += 1
  done.put((n, time.perf_counter() - t0))


def _cpu_baseline(sampler, y, mask, X, min_seconds=10.0):
  """Times the float64 CPU restatement (oracle/, kind="port") on the host's cores.

  Gibbs (the headline): the build BASELINE.md section 2 names -- `-O3 -march=native`, one thread
  per chain, OpenMP across chains -- compiled on THIS host (oracle/Makefile `native`; the flags
  are echoed in `sample`).  1 core: whole chains of the bench workload until >= min_seconds of CPU
  work; all cores: rounds of os.cpu_count() chains (one per core) until the same budget.
  HMC (--sampler hmc): the checker build, one process per core.
  The oracle is the checker here, never the product path."""
  import multiprocessing as mp  # pylint: disable=import-outside-toplevel
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  spec = orc.default_spec(y, mask, X, has_slope=bool(CFG["has_slope"]))
  S = CFG["num_results"]
  W = CFG["hmc_warmup"] if sampler == "hmc" else CFG["num_warmup"]
  ncpu = os.cpu_count() or 1
  if sampler == "gibbs":
    _, flags = orc.native_lib()
    kw = dict(num_results=S, num_warmup=W, seed=CFG["seed"])
    orc.fit_gibbs_chains_native(y, mask, X, spec, first_chain=0, n_chains=1, threads=1, **kw)  # warm
    t0 = time.perf_counter()
    chains = 0
    while time.perf_counter() - t0 < min_seconds:
      orc.fit_gibbs_chains_native(y, mask, X, spec, first_chain=chains, n_chains=1, threads=1, **kw)
      chains += 1
    dt1 = time.perf_counter() - t0
    what = f"({W}+{S}) Gibbs iterations"
    res = {"value": chains * S / dt1, "unit": "posterior samples/sec", "cores": 1, "kind": "port",
           "sample": f"{chains} chains x {what} of the bench series, float64 C restatement "
                     f"(oracle/ci_oracle.c built on this host with `{flags}`), {dt1:.1f} s"}
    if ncpu > 1:
      t0 = time.perf_counter()
      n_tot, used = 0, ncpu
      while time.perf_counter() - t0 < min_seconds:
        used = orc.fit_gibbs_chains_native(y, mask, X, spec, first_chain=1000 + n_tot,
                                           n_chains=ncpu, threads=ncpu, **kw)
        n_tot += ncpu
      dtn = time.perf_counter() - t0
      res["all_cores"] = {"value": n_tot * S / dtn, "cores": used,
                          "sample": f"{n_tot} chains, OpenMP over chains on {used} threads "
                                    f"(one chain per thread), same build, {dtn:.1f} s"}
    return res
  orc.lib()
  t0 = time.perf_counter()
  chains = 0
  while time.perf_counter() - t0 < min_seconds:
    _cpu_chain(sampler, y, mask, X, spec, S, W, chains)
    chains += 1
  dt1 = time.perf_counter() - t0
  one = chains * S / dt1
  what = f"({W}+{S}) x {CFG['hmc_leapfrog']}-leapfrog HMC iterations + latent draws"
  res = {"value": one, "unit": "posterior samples/sec", "cores": 1, "kind": "port",
         "sample": f"{chains} chains x {what} of the bench series, float64 C restatement "
                   f"(oracle/ci_oracle.c, checker build -O2 -fno-fast-math), {dt1:.1f} s"}
  if ncpu > 1:
    ctx = mp.get_context("fork")
    go, done = ctx.Event(), ctx.Queue()
    procs = [ctx.Process(target=_cpu_worker, args=(sampler, y, mask, X, spec, S, W, 1000 + 4096 * i,
                                                   go, min_seconds, done)) for i in range(ncpu)]
    for pr in procs:
      pr.start()
    time.sleep(1.0)                              # workers load the library before the start signal
    go.set()
    got = [done.get() for _ in procs]
    for pr in procs:
      pr.join()
    n_tot = sum(n for n, _ in got)
    dtn = max(t for _, t in got)
    res["all_cores"] = {"value": n_tot * S / dtn, "cores": ncpu,
                        "sample": f"{n_tot} chains over {ncpu} processes (one per host core), "
                                  f"{dtn:.1f} s"}
  return res


def _pmc_traffic(sampler):
  """HBM bytes per launch from the COMMITTED rocprofv3 PMC passes (profiles/*pmc.json, written by
  tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command) and the
  file it came from -- the fallback of _measure_traffic."""
  import glob  # pylint: disable=import-outside-toplevel
  pat = "r[0-9][0-9]_cfg3_pmc.json" if sampler == "hmc" else "r[0-9][0-9]_pmc.json"
  # the newest round's file first (profiles/rNN_*: NN sorts by round)
  for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pat)), reverse=True):
    name = os.path.basename(path)
    if os.path.exists(path):
      with open(path) as f:
        return json.load(f).get("hbm_bytes_per_launch"), "profiles/" + name + " (committed; not measured in this run)"
  return None, None


def _measure_traffic(sampler, timeout_s=150):
  """HBM bytes per launch MEASURED NOW: two rocprofv3 counter passes over this very command with
  a short step count -- FETCH_SIZE and WRITE_SIZE in their own runs, no trace domain next to
  them, summed over the kernels of one fit and corrected as /opt/skills/guides/MI355X_MICROARCH.md
  prescribes (both counters in KiB; on gfx950 FETCH_SIZE reports half of a wide streaming read).
  A counter pass cannot run inside the timed process, so it runs after the timed region as a
  child.  Returns (bytes, source) or (None, why) -- the caller then falls back to the committed
  figure."""
  import csv  # pylint: disable=import-outside-toplevel
  import glob  # pylint: disable=import-outside-toplevel
  import shutil  # pylint: disable=import-outside-toplevel
  import signal  # pylint: disable=import-outside-toplevel
  import tempfile  # pylint: disable=import-outside-toplevel
  prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
  if not os.path.exists(prof):
    return None, "rocprofv3 not found"
  kernels = (("hmc_kernel", "latents_kernel", "hmc_mean_kernel", "hmc_unpack_kernel")
             if sampler == "hmc" else ("gibbs_kernel",))
  steps = 1 if sampler == "hmc" else 3
  tmp = tempfile.mkdtemp(prefix="ci_bench_pmc_", dir="/tmp")
  env = dict(os.environ, TMPDIR="/tmp", CI_BENCH_INNER="1")
  per_launch = {}
  try:
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
      out_dir = os.path.join(tmp, counter)
      cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", out_dir, "-o", "bench", "--",
             sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-pmc",
             "--sampler", sampler, "--steps", str(steps), "--warmup", "1"]
      p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           start_new_session=True)
      try:
        p.wait(timeout=timeout_s)
      except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        p.wait()
        return None, f"rocprofv3 --pmc {counter} pass exceeded {timeout_s} s"
      files = glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)
      if p.returncode != 0 or not files:
        return None, f"rocprofv3 --pmc {counter} pass failed (rc {p.returncode})"
      # KiB per dispatch, summed over the kernels of one fit, averaged over the gibbs/hmc launches
      per = {}
      main_ids = set()
      with open(files[0]) as f:
        for row in csv.DictReader(f):
          name = row.get("Kernel_Name", "")
          if row.get("Counter_Name") != counter or not any(k in name for k in kernels):
            continue
          per[name] = per.get(name, 0.0) + float(row["Counter_Value"])
          if kernels[0] in name:
            main_ids.add(row["Dispatch_Id"])
      if not main_ids:
        return None, f"no {kernels[0]} dispatch in the {counter} pass"
      per_launch[counter] = sum(per.values()) / len(main_ids)
  finally:
    shutil.rmtree(tmp, ignore_errors=True)
  total = 2.0 * per_launch["FETCH_SIZE"] * 1024.0 + per_launch["WRITE_SIZE"] * 1024.0
  return total, ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate "
                 f"passes of {steps + 1} fits each, KiB -> bytes, FETCH_SIZE x2 (gfx950 wide reads)")


class _GibbsFit:
  """cfg2: the persistent Gibbs kernel, inputs and outputs resident in HBM."""

  def __init__(self, y, mask, X, C, rank, device):
    spec = _model.series_params(y, mask, X, prior_level_sd=0.01, has_slope=bool(CFG["has_slope"]))
    pb = _native.make_problem(T=CFG["T"], P=X.shape[1], has_slope=CFG["has_slope"],
                              num_warmup=CFG["num_warmup"], num_results=CFG["num_results"],
                              num_chains=C, chain_offset=rank * C, seed=CFG["seed"], device=device)
    self.sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
    self.workload = ("cfg2: T=1000, 10 covariates (P=11), LocalLinearTrend + spike-and-slab "
                     "regression, Gibbs, W=112, S=1000")

  def run(self):
    return {"kernel": self.sess.run()}

  def fetch(self):
    res = self.sess.fetch()
    return {k: res[k][0] for k in ("observation_noise_scale", "level_scale", "posterior_means")}

  def run_and_fetch(self):
    """One fit with every result array delivered to (pinned) host memory: the copies are queued
    chunk by chunk while the kernel is still sampling (ci_session_run_streamed)."""
    if not hasattr(self, "_into"):
      self.sess.run_streamed()                      # allocates the pinned result buffers once
      self._into = self.sess._last_streamed         # pylint: disable=protected-access
    ms, _ = self.sess.run_streamed(into=self._into, chunk_draws=getattr(self, "chunk_draws", 125))
    return {"kernel": ms}

  def last_small(self):
    res = self._into[1]
    return {k: res[k][0] for k in ("observation_noise_scale", "level_scale", "posterior_means")}

  def note(self, C):
    return ("one persistent workgroup per chain: %d of 256 CUs busy; the fit is bound by the "
            "(W+S)-long sequential Gibbs dependency, not by HBM (DESIGN.md)" % C)


class _HmcFit:
  """cfg3: the on-device HMC chain + the latent / predictive pass."""

  def __init__(self, y, mask, X, C, rank, device):
    spec = _model.series_params(y, mask, X, prior_level_sd=0.01, has_slope=bool(CFG["has_slope"]))
    pb = _native.make_problem(T=CFG["T"], P=X.shape[1], has_slope=CFG["has_slope"], num_warmup=0,
                              num_results=1, seed=CFG["seed"], device=device)
    self.sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=8)
    self.kw = dict(num_chains=C, chain_offset=rank * C, num_warmup=CFG["hmc_warmup"],
                   num_results=CFG["num_results"], num_leapfrog=CFG["hmc_leapfrog"], seed=CFG["seed"])
    self.workload = ("cfg3: T=1000, 10 covariates (P=11), LocalLinearTrend + Gaussian-slab "
                     "regression, windowed-adaptive HMC, %d leapfrogs, W=%d, S=1000, latent path + "
                     "predictive trajectory per draw" % (CFG["hmc_leapfrog"], CFG["hmc_warmup"]))

  def run(self):
    hmc_ms, lat_ms = self.sess.hmc_run(**self.kw)
    return {"kernel": hmc_ms, "latents": lat_ms}

  def fetch(self):
    _, _, _, res = self.sess.hmc_fetch()
    return {k: res[k][0] for k in ("observation_noise_scale", "level_scale", "posterior_means")}

  def run_and_fetch(self):
    ms = self.run()
    self._last = self.fetch()
    return ms

  def last_small(self):
    return self._last

  def note(self, C):
    return ("one persistent workgroup per chain (%d of 256 CUs) runs (W+S) x leapfrog dependent "
            "score evaluations; the latent / predictive pass (one workgroup per draw) fills the "
            "chip; `achieved` counts the fit's algorithmic bytes over BOTH kernels' time" % C)


def _roof(nbytes, ms):
  ach = nbytes / (ms * 1e-3) / 1e9
  return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes}


def _other_configs(device):
  """The BASELINE configs that are not the headline, one short measurement each AFTER the timed
  region (round-5 review: only cfg2 was ever timed by the driver).  Kernel time by HIP events
  around one launch (after one warm launch), inputs resident in HBM; the same workloads as
  tools/run_configs.py (whose lines add traffic counters and CPU baselines)."""
  import causalimpact as ci  # pylint: disable=import-outside-toplevel
  rows = []

  def guarded(label, fn):
    try:
      rows.extend(fn())
    except Exception as e:  # pylint: disable=broad-except
      rows.append({"config": label, "error": f"{type(e).__name__}: {e}"})

  def cfg3():
    y, mask, X, _ = syn.make_sampler_inputs(CFG["T"], CFG["covariates"], CFG["data_seed"])
    fit = _HmcFit(y, mask, X, CFG["chains_per_gpu"], 0, device)
    fit.run()
    k = fit.run()
    nbytes, name = fit.sess.algorithmic_bytes(), fit.sess.kernel_name()
    fit.sess.close()
    ms = k["kernel"] + k["latents"]
    n_leap = (CFG["hmc_warmup"] + CFG["num_results"]) * CFG["hmc_leapfrog"]
    return [{"config": "cfg3", "workload": fit.workload + " (this GPU's 8 of the 64 chains)", "kernel": name,
             "chains": CFG["chains_per_gpu"], "kernel_ms": k["kernel"], "latents_kernel_ms": k["latents"],
             "us_per_leapfrog": k["kernel"] * 1e3 / n_leap,
             "samples_per_s": CFG["chains_per_gpu"] * CFG["num_results"] / ms * 1e3,
             "roofline": _roof(nbytes, ms)}]

  def cfg4():
    T, p, S = 10000, 50, 1000
    W = -(-S // 9)
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
    y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
    spec = _model.series_params(y, mask, X, num_seasonal_blocks=1)
    counts, flg = _model.expand_seasons((ci.Seasons(num_seasons=7),), T)
    out = []
    for chains in (8, 32):
      pb = _native.make_problem(T=T, P=X.shape[1], has_slope=0, num_seasons=counts, num_warmup=W,
                                num_results=S, num_chains=chains, seed=(0, 1), device=device)
      sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
      sess.run()
      ms = sess.run()
      out.append({"config": "cfg4", "workload": "T=10000, 50 covariates (P=51), LocalLevel + Seasonal(7) + "
                  "spike-and-slab regression, Gibbs, W=%d, S=%d" % (W, S), "kernel": sess.kernel_name(),
                  "chains": chains, "kernel_ms": ms, "us_per_gibbs_iteration": ms / (W + S) * 1e3,
                  "samples_per_s": chains * S / ms * 1e3, "roofline": _roof(sess.algorithmic_bytes(), ms)})
      sess.close()
    return out

  def cfg5():
    import pandas as pd  # pylint: disable=import-outside-toplevel
    T, p, S = 500, 5, 1000
    out = []
    values = np.stack([np.column_stack(syn.make_raw_series(T, p, b)) for b in range(512)])
    prep = ci.batch.prepare_batch(values, pd.RangeIndex(T), (0, 349), (350, 499))
    params = [_model.series_params(prep.y[b], prep.mask[b], prep.design[b]) for b in range(512)]
    W = ci.InferenceOptions(num_results=S).num_warmup_steps
    for B in (64, 512):
      pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=1,
                                num_series=B, seed=(0, 1), device=device)
      sess = _native.Session(pb, prep.y[:B], prep.mask[:B], prep.design[:B], None,
                             _native.make_params(params[:B]))
      sess.run()
      ms = sess.run()
      out.append({"config": "cfg5", "workload": "independent series, T=500, 5 covariates (P=6), LocalLevel + "
                  "spike-and-slab regression, Gibbs, W=%d, S=%d%s" % (
                      W, S, " (one GPU's share of the 512 on 8 GPUs)" if B == 64 else " (all 512 on one GPU)"),
                  "kernel": sess.kernel_name(), "series": B, "kernel_ms": ms,
                  "samples_per_s": B * S / ms * 1e3, "roofline": _roof(sess.algorithmic_bytes(), ms)})
      sess.close()
    return out

  def capability_routes():
    """Two routes outside BASELINE's five configs that rounds 5-6 built kernels for: 100 covariates
    (the BIGP build of the trend + one-block kernel) and the reference's own multi-block test model
    (causalimpact_lib_test.py:738-752: Seasons 4 / 7 / 6) at cfg4's size."""
    out = []
    T, p, W, S, C = 1000, 100, 50, 200, 8
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
    pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=C,
                              seed=(0, 1), device=device)
    sess = _native.Session(pb, y[None], mask[None], X[None], None,
                           _native.make_params([_model.series_params(y, mask, X)]))
    sess.run()
    ms = sess.run()
    out.append({"config": "T=1000, 100 covariates (P=101), LocalLevel + spike-and-slab regression, Gibbs",
                "kernel": sess.kernel_name(), "chains": C, "kernel_ms": ms,
                "us_per_gibbs_iteration": ms / (W + S) * 1e3, "samples_per_s": C * S / ms * 1e3,
                "roofline": _roof(sess.algorithmic_bytes(), ms)})
    sess.close()
    seasons = (ci.Seasons(num_seasons=4, num_steps_per_season=(2, 1, 1, 1)), ci.Seasons(num_seasons=7),
               ci.Seasons(num_seasons=6, num_steps_per_season=((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1))))
    T, p, W, S, C = 10000, 50, 12, 100, 8
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
    y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
    counts, flg = _model.expand_seasons(seasons, T)
    pb = _native.make_problem(T=T, P=X.shape[1], has_slope=0, num_seasons=counts, num_warmup=W, num_results=S,
                              num_chains=C, seed=(0, 1), device=device)
    sess = _native.Session(pb, y[None], mask[None], X[None], flg,
                           _native.make_params([_model.series_params(y, mask, X, num_seasonal_blocks=3)]))
    sess.run()
    ms = sess.run()
    out.append({"config": "T=10000, 50 covariates + Seasons 4/7/6 (state of 18), Gibbs", "kernel": sess.kernel_name(),
                "chains": C, "kernel_ms": ms, "us_per_gibbs_iteration": ms / (W + S) * 1e3,
                "samples_per_s": C * S / ms * 1e3, "roofline": _roof(sess.algorithmic_bytes(), ms)})
    sess.close()
    return out

  guarded("cfg3", cfg3)
  guarded("cfg4", cfg4)
  guarded("cfg5", cfg5)
  guarded("capability routes", capability_routes)
  return rows


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=None)
  ap.add_argument("--warmup", type=int, default=None)
  ap.add_argument("--sampler", choices=("gibbs", "hmc"), default="gibbs")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-pmc", action="store_true",
                  help="skip the two rocprofv3 counter passes that measure roofline.traffic")
  ap.add_argument("--no-other-configs", action="store_true",
                  help="skip the short cfg3 / cfg4 / cfg5 measurements after the timed region")
  ap.add_argument("--chains-per-gpu", type=int, default=CFG["chains_per_gpu"])
  ap.add_argument("--chunk-draws", type=int, default=CFG["chunk_draws"],
                  help="retained draws per device-to-host copy of the streamed fetch")
  args = ap.parse_args()
  if args.steps is None:
    args.steps = 20 if args.sampler == "gibbs" else 5
  if args.warmup is None:
    args.warmup = 3 if args.sampler == "gibbs" else 1

  # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one
  # process per GPU); rank 0 prints the JSON line.  Under torch.distributed.run (or any launcher
  # that exports RANK / LOCAL_RANK / WORLD_SIZE) this process IS a rank.
  from causalimpact import _comm  # pylint: disable=import-outside-toplevel
  launched = "WORLD_SIZE" in os.environ
  if not launched:
    rc = _comm.self_launch(args.gpus)
    if rc is not None:
      raise SystemExit(rc)
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
  if launched and os.environ.get("CI_COMM_SPAWNED") != "1" and os.environ.get("CI_COMM_DEVICES"):
    # an external launcher (torch.distributed.run) on a box with fewer GPUs than ranks: the same
    # rank -> device map the self-launcher takes (tests: two ranks of the driver's command line
    # sharing GPU 0 over the host transport)
    devs = [int(d) for d in os.environ["CI_COMM_DEVICES"].split(",")]
    if len(devs) != world:
      raise SystemExit(f"CI_COMM_DEVICES names {len(devs)} devices for {world} ranks")
    local_rank = devs[rank]
  # RCCL through the C-ABI (ci_comm_*): no PyTorch in this process.  CI_BENCH_FORCE_DIST=1
  # exercises the same path with a single rank (1-GPU box).
  # connect(): the host transport comes up first, RCCL is joined under a deadline and proven with
  # one all-reduce; if any rank cannot join, ALL ranks carry on over the host transport and the
  # reason lands in config.collectives -- an N-GPU run cannot hang or fail silently in set-up.
  comm = None
  if world > 1 or os.environ.get("CI_BENCH_FORCE_DIST") == "1":
    comm = _comm.connect(rank, world, local_rank)
    if comm.ranks_seen != world:
      raise SystemExit(f"rank {rank}: communicator sees {comm.ranks_seen} of {world} ranks")

  C = args.chains_per_gpu
  y, mask, X, _ = syn.make_sampler_inputs(CFG["T"], CFG["covariates"], CFG["data_seed"])
  fit = (_HmcFit if args.sampler == "hmc" else _GibbsFit)(y, mask, X, C, rank, local_rank)
  fit.chunk_draws = args.chunk_draws

  def sync():
    if comm is not None:
      comm.barrier()
    _native.device_synchronize(local_rank)

  # One step = one complete fit with every result array DELIVERED to (pinned) host memory: the
  # metric of SURVEY.md section 8(d) includes the device-to-host copy of the results; inputs are
  # resident in HBM.  The copies run in the shadow of the persistent kernel (run_streamed).
  for _ in range(args.warmup):
    fit.run_and_fetch()
  sync()
  t0 = time.perf_counter()
  kernel_ms = []
  for _ in range(args.steps):
    kernel_ms.append(fit.run_and_fetch())
  sync()
  dt = time.perf_counter() - t0
  if comm is not None:
    dt = float(comm.all_reduce([dt], _comm.MAX)[0])

  # ---- after the timed region: kernel-only rate, chain gather + diagnostics
  t1 = time.perf_counter()
  n_k = 3
  for _ in range(n_k):
    fit.run()
  dt_kernel = (time.perf_counter() - t1) / n_k
  local = fit.last_small()
  # chain gather + split-R-hat / ESS across ranks: all-gather of the blocks still resident in HBM
  # (ncclAllGather on the sessions' device buffers) + ONE all-reduce of the partial sums
  from causalimpact import _distributed  # pylint: disable=import-outside-toplevel

  def resident(key):
    if comm is None:
      return None
    g = comm.session_all_gather(fit.sess, key)             # [world, 1, C, ...]
    return g.reshape((world * C,) + g.shape[3:])

  comb = _distributed.fit_sharded(lambda first, count: local, world * C,
                                  gather_keys=("posterior_means",), comm=comm, resident=resident)
  rhat = comb["split_rhat"]
  ess = {"bulk": comb["ess_bulk"], "tail": comb["ess_tail"]}
  assert comb["posterior_means"].shape == (world * C, CFG["T"])

  samples_per_step = world * C * CFG["num_results"]
  value = samples_per_step * args.steps / dt
  k_ms = float(np.mean([sum(k.values()) for k in kernel_ms]))
  alg_bytes = fit.sess.algorithmic_bytes()
  achieved = alg_bytes / (k_ms * 1e-3) / 1e9
  traffic, traffic_src = None, None
  if rank == 0 and world == 1 and not args.no_pmc and os.environ.get("CI_BENCH_INNER") != "1":
    traffic, why = _measure_traffic(args.sampler)
    traffic_src = why
    if traffic is None:
      traffic, src = _pmc_traffic(args.sampler)
      traffic_src = f"{src}; in-run measurement unavailable: {why}"
  else:
    traffic, traffic_src = _pmc_traffic(args.sampler)
  roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
          "kernel": fit.sess.kernel_name(),
          "kernel_ms": float(np.mean([k["kernel"] for k in kernel_ms])),
          "algorithmic_bytes_per_launch": alg_bytes, "note": fit.note(C)}
  if args.sampler == "hmc":
    roof["latents_kernel_ms"] = float(np.mean([k["latents"] for k in kernel_ms]))
    n_leap = (CFG["hmc_warmup"] + CFG["num_results"]) * CFG["hmc_leapfrog"]
    roof["us_per_leapfrog"] = roof["kernel_ms"] * 1e3 / n_leap
  else:
    roof["us_per_gibbs_iteration"] = roof["kernel_ms"] * 1e3 / (CFG["num_warmup"] + CFG["num_results"])
  out = {
      "metric": "posterior samples/sec (T=1000, 10 covariates)",
      "value": value, "unit": "posterior samples/sec", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
      "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "config": {"workload": fit.workload, "sampler": args.sampler,
                 "chains_per_gpu": C, "chains_total": world * C,
                 "timed_region": "fit + delivery of every result array to pinned host memory",
                 "parallelism": f"chains sharded over {world} GPU(s), no data-path collective",
                 "launcher": ("bench.py spawned its own ranks (one process per GPU)"
                              if os.environ.get("CI_COMM_SPAWNED") == "1" else
                              ("external (RANK/WORLD_SIZE in the environment)" if launched else
                               "single process")),
                 "collectives": (f"{comm.transport} via ci_comm_* (C-ABI), ranks_seen="
                                 f"{comm.ranks_seen}" if comm is not None else "none"),
                 "ranks_seen": comm.ranks_seen if comm is not None else 1},
      "roofline": roof,
      "kernel_only_value": C * CFG["num_results"] / dt_kernel,
      "split_rhat": rhat, "ess": ess,
  }
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out["cpu_baseline"] = _cpu_baseline(args.sampler, y, mask, X)
  elif rank == 0:
    # not `null` without comment (round-4 review): the object says why there is no value here
    out["cpu_baseline"] = {"value": None, "unit": "posterior samples/sec", "cores": 0, "kind": "port",
                           "sample": ("not measured in this run: the CPU leg is timed on rank 0 of the "
                                      "N=1 run only (--no-cpu-baseline / N > 1); see the N=1 line")}
    out["config"]["n_gt_1_note"] = ("cpu_baseline is timed at N=1 only; roofline.traffic on this "
                                    "line is the committed single-GPU counter figure (per GPU), not "
                                    "measured in this run")
  if (rank == 0 and world == 1 and comm is None and args.sampler == "gibbs" and not args.no_other_configs
      and os.environ.get("CI_BENCH_INNER") != "1"):
    # the other BASELINE configs, after the timed region (the headline value / config stay cfg2)
    fit.sess.close()
    fit.sess = None
    out["other_configs"] = _other_configs(local_rank)
  hard_exit = False
  if comm is not None:
    comm.barrier()
    hard_exit = comm.hard_exit
  if hard_exit:
    #  say so in the line: the timed region ran beside an abandoned RCCL join
    out["config"]["rccl_join_abandoned"] = ("an RCCL join that missed its deadline may still have been "
                                            "blocked (and its kernel spinning on the GPU) while this run "
                                            "was timed over the host transport")
  if fit.sess is not None:
    fit.sess.close()
  if comm is not None:
    comm.close()
  # librccl prints a version banner through C stdio (buffered when stdout is a pipe): flush it now
  # so that the JSON line is the LAST line of this job's output
  import ctypes  # pylint: disable=import-outside-toplevel
  ctypes.CDLL(None).fflush(None)
  if rank == 0:
    print(json.dumps(out), flush=True)
  if hard_exit:
    # an abandoned RCCL attempt may still be blocked inside librccl on a helper thread: do not let
    # interpreter shutdown wait for it
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)


if __name__ == "__main__":
  main()
