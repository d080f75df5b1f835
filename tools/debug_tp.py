"""Diagnostic for the time-parallel general seasonal kernel (csrc/ci_seasonal_tp.h): per-output
maximum deviation from the float64 oracle over the first draws, for a ladder of models, on one
workgroup per chain and on a cluster; plus the time per Gibbs iteration of the big cases.

  python tools/debug_tp.py [quick]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _model, _native            # noqa: E402
from causalimpact import _synthetic as syn          # noqa: E402
from oracle import ci_oracle as orc                 # noqa: E402

REF = ((4, (2, 1, 1, 1)), (7, 1), (6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1))))


def run(T, p, slope, seasons, flags, S=3, W=0, C=1, oracle=True, label="", prof=False):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  mask = mask.copy()
  mask[[2, 3, 40, T // 2]] = True
  spec = orc.default_spec(y, mask, X, has_slope=bool(slope), seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=slope, num_seasons=counts, num_warmup=W,
                            num_results=S, num_chains=C, seed=(2, 6), flags=flags)
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None],
                         flg if seasons else None, _native.make_params([spec]))
  name = sess.kernel_name()
  ms = sess.run()
  got = sess.fetch()
  line = f"{label:28s} T={T:6d} P={spec['P']:3d} {name:44s} {ms / (W + S) * 1e3:9.1f} us/iter"
  if prof and "tp_kernel" in name:
    sess.profile(True)
    sess.run()
    cyc = sess.profile(False)
    names = ["targets", "serial+wait", "emit+normals", "sim", "build", "fwd final", "filter+M", "bwd scan", "r+recon",
             "fwd intra", "fwd barrier", "fwd cluster"]
    line += "\n      cycles/iter: " + "  ".join(f"{n}: {cyc[20 + i] / (W + S):.0f}" for i, n in enumerate(names))
    if spec["P"] > 52:
      reg = ["build + sweep-in", "order + proposals + flips", "active set + block", "Cholesky", "solve + weights"]
      line += "\n      regression draw of the whole workgroup: " + "  ".join(f"{n}: {cyc[9 + i] / (W + S):.0f}" for i, n in enumerate(reg))
  sess.close()
  if oracle:
    w = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=(2, 6))
    pairs = [("level", "level"), ("seasonal_levels", "seasonal"), ("weights", "weights"),
             ("observation_noise_scale", "obs_scale"), ("level_scale", "level_scale"),
             ("seasonal_drift_scales", "drift_scales"), ("posterior_trajectories", "trajectories")]
    devs = []
    for gk, wk in pairs:
      if gk in got and wk in w and np.size(w[wk]):
        g = np.asarray(got[gk][0, 0], np.float64)
        d = np.abs(g - np.asarray(w[wk], np.float64).reshape(g.shape))
        per_draw = d.reshape(d.shape[0], -1).max(axis=1)
        devs.append(f"{gk.split('_')[0][:6]}:" + "/".join(f"{v:.1e}" for v in per_draw))
    line += "  " + " ".join(devs)
  print(line, flush=True)


if __name__ == "__main__":
  quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
  NC, SEQ = _native.FLAG_NO_CLUSTER, _native.FLAG_SEQUENTIAL_SEASONAL
  for flags, lab in ((SEQ, "seq"), (NC, "tp x1"), (0, "tp cluster")):
    run(256, 0, 0, (), flags, label=lab + " trend") if False else None
    run(400, 60, 0, (), flags, label=lab + " trend P=61")
    run(300, 0, 0, ((12, 1),), flags, label=lab + " 12 seasons")
    run(300, 0, 1, ((12, 1),), flags, label=lab + " 12 seasons + slope")
    run(300, 3, 0, REF, flags, label=lab + " 4+7+6")
    run(2500, 12, 1, REF, flags, label=lab + " 4+7+6 slope")
    run(1200, 8, 0, ((24, 1), (7, 24)), flags, label=lab + " 24+7")
  if not quick:
    for flags, lab in ((0, "tp cluster"), (NC, "tp x1"), (SEQ, "seq")):
      run(10000, 50, 0, REF, flags, S=6, W=2, C=8, oracle=False, label=lab + " cfg4-size 8 chains", prof=True)
      run(2500, 12, 1, REF, flags, S=6, W=2, C=1, oracle=False, label=lab + " 4+7+6 slope", prof=True)
    run(10000, 50, 0, REF, 0, S=3, label="tp cluster cfg4-size")
    run(10000, 50, 0, ((7, 1),), 0, S=30, W=10, C=8, oracle=False, label="wide kernel cfg4")
    run(10000, 50, 0, ((7, 1),), _native.FLAG_CLUSTER_SEASONAL, S=30, W=10, C=8, oracle=False, label="tp kernel cfg4", prof=True)
    run(10000, 50, 0, ((7, 1),), _native.FLAG_CLUSTER_SEASONAL, S=3, label="tp kernel cfg4 vs oracle")
    run(1000, 100, 0, (), 0, S=10, W=2, C=8, oracle=False, label="tp trend P=101", prof=True)
    run(1000, 100, 0, (), SEQ, S=10, W=2, C=8, oracle=False, label="seq trend P=101")
