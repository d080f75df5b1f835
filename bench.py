#!/usr/bin/env python3
"""bench.py -- posterior samples/sec of the Gibbs hot path on MI355X.

Workload (BASELINE.json configs[1], "cfg2"): T=1000, 10 covariates (+intercept => P=11),
LocalLinearTrend + spike-and-slab regression, 1000 retained Gibbs draws x 8 chains per GPU,
W = ceil(1000/9) = 112 warm-up iterations, float32.  One "step" = one complete fit
(all W+S iterations of all chains of this rank).  Chains are independent, so ranks shard
them with no data-path collective (weak scaling: 8 chains per GPU); RCCL is used only
after the timed region, to gather per-chain moments for the split-R-hat diagnostic.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import math
import os
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
           chains_per_gpu=8, data_seed=2024, seed=(0, 20240927))
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def _cpu_baseline(y, mask, X, min_seconds=10.0):
  """Times the float64 CPU restatement (oracle/, kind="port") on host cores.

  1 core: the cfg2 workload itself (8 chains x 1112 iterations), repeated until
  >= min_seconds of CPU work.  All cores: the same chains spread over os.cpu_count()
  processes.  The oracle is the checker here, never the product path.
  """
  import multiprocessing as mp  # pylint: disable=import-outside-toplevel
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  spec = orc.default_spec(y, mask, X, has_slope=bool(CFG["has_slope"]))
  S, W = CFG["num_results"], CFG["num_warmup"]
  want = ("obs_scale", "level_scale", "slope_scale", "weights", "level", "slope", "pred_mean",
          "trajectories")
  orc.lib()
  t0 = time.perf_counter()
  chains = 0
  while time.perf_counter() - t0 < min_seconds:
    orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=CFG["seed"], chain=chains,
                  want=want)
    chains += 1
  dt1 = time.perf_counter() - t0
  one = chains * S / dt1
  ncpu = os.cpu_count() or 1
  res = {"value": one, "unit": "posterior samples/sec", "cores": 1, "kind": "port",
         "sample": f"{chains} chains x ({W}+{S}) Gibbs iterations of the cfg2 series, float64 C "
                   f"restatement (oracle/ci_oracle.c), {dt1:.1f} s"}
  if ncpu > 1:
    # one process per core (capped at 64 so the sample stays ~10-20 s), 4 chains each
    nproc, per = min(ncpu, 64), 4
    ctx = mp.get_context("fork")
    with ctx.Pool(nproc) as pool:
      pool.map(_cpu_noop, range(nproc))          # start the workers before timing
      t1 = time.perf_counter()
      pool.starmap(_cpu_chain, [(y, mask, X, spec, S, W, 1000 + i) for i in range(nproc * per)],
                   chunksize=per)
      dtn = time.perf_counter() - t1
    res["all_cores"] = {"value": nproc * per * S / dtn, "cores": nproc,
                        "sample": f"{nproc * per} chains over {nproc} processes "
                                  f"(host has {ncpu} cores), {dtn:.1f} s"}
  return res


def _cpu_noop(_):
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  orc.lib()
  return 0


def _cpu_chain(y, mask, X, spec, S, W, chain):
  from oracle import ci_oracle as orc  # pylint: disable=import-outside-toplevel
  orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=CFG["seed"], chain=chain,
                want=("obs_scale", "level", "trajectories", "weights"))
  return 0


def _pmc_traffic():
  """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*pmc.json, written
  by tools/pmc_summary.py from separate FETCH_SIZE / WRITE_SIZE runs of this same command)."""
  path = os.path.join(ROOT, "profiles", "r01_pmc.json")
  if not os.path.exists(path):
    return None
  with open(path) as f:
    return json.load(f).get("hbm_bytes_per_launch")


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=3)
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--chains-per-gpu", type=int, default=CFG["chains_per_gpu"])
  args = ap.parse_args()

  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
  dist = None
  # CI_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (1-GPU smoke test)
  if world > 1 or os.environ.get("CI_BENCH_FORCE_DIST") == "1":
    import torch  # pylint: disable=import-outside-toplevel
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    torch.cuda.set_device(local_rank)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

  T, C = CFG["T"], args.chains_per_gpu
  y, mask, X, _ = syn.make_sampler_inputs(T, CFG["covariates"], CFG["data_seed"])
  spec = _model.series_params(y, mask, X, prior_level_sd=0.01, has_slope=bool(CFG["has_slope"]))
  pb = _native.make_problem(T=T, P=X.shape[1], has_slope=CFG["has_slope"],
                            num_warmup=CFG["num_warmup"], num_results=CFG["num_results"],
                            num_chains=C, chain_offset=rank * C, seed=CFG["seed"],
                            device=local_rank)
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))

  def sync():
    if dist is not None:
      import torch  # pylint: disable=import-outside-toplevel
      dist.barrier()
      torch.cuda.synchronize()

  for _ in range(args.warmup):
    sess.run()
  sync()
  t0 = time.perf_counter()
  kernel_ms = []
  for _ in range(args.steps):
    kernel_ms.append(sess.run())      # run() waits for the fit's stream
  sync()
  dt = time.perf_counter() - t0
  if dist is not None:
    import torch  # pylint: disable=import-outside-toplevel
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())

  # ---- after the timed region: PCIe-inclusive rate, chain gather + diagnostics (RCCL)
  t1 = time.perf_counter()
  sess.run()
  res = sess.fetch()
  dt_pcie = time.perf_counter() - t1
  # chain gather + split-R-hat across ranks (RCCL all-gather / all-reduce; no-op at N=1)
  from causalimpact import _distributed  # pylint: disable=import-outside-toplevel
  local = {k: res[k][0] for k in ("observation_noise_scale", "level_scale", "posterior_means")}
  comb = _distributed.fit_sharded(lambda first, count: local, world * C,
                                  gather_keys=("posterior_means",),
                                  device="cuda" if dist is not None else None)
  rhat = comb["split_rhat"]

  samples_per_step = world * C * CFG["num_results"]
  value = samples_per_step * args.steps / dt
  k_ms = float(np.mean(kernel_ms))
  alg_bytes = sess.algorithmic_bytes()
  achieved = alg_bytes / (k_ms * 1e-3) / 1e9
  out = {
      "metric": "posterior samples/sec (T=1000, 10 covariates)",
      "value": value, "unit": "posterior samples/sec", "n_gpus": world, "steps": args.steps,
      "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
      "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
      "config": {"workload": "cfg2: T=1000, 10 covariates (P=11), LocalLinearTrend + "
                             "spike-and-slab regression, Gibbs, W=112, S=1000",
                 "chains_per_gpu": C, "chains_total": world * C,
                 "parallelism": f"chains sharded over {world} GPU(s), no data-path collective"},
      "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic(),
                   "kernel": "ci::gibbs_kernel<2,4,1,false>", "kernel_ms": k_ms,
                   "algorithmic_bytes_per_launch": alg_bytes,
                   "note": "one workgroup per chain: 8 of 256 CUs busy; the fit is bound by the "
                           "(W+S)-long sequential Gibbs dependency, not by HBM (DESIGN.md)"},
      "pcie_inclusive_value": C * CFG["num_results"] / dt_pcie,
      "split_rhat": rhat,
  }
  if rank == 0 and world == 1 and not args.no_cpu_baseline:
    out["cpu_baseline"] = _cpu_baseline(y, mask, X)
  elif rank == 0:
    out["cpu_baseline"] = None
  sess.close()
  if rank == 0:
    print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
