#!/usr/bin/env python3
"""Per-phase shader-clock budget of the time-parallel seasonal kernel (gibbs_wide_kernel) on
BASELINE cfg4 (T=10000, 50 covariates, weekly block).   python tools/profile_wide.py [T] [p] [chains]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
import causalimpact as ci  # noqa: E402
from causalimpact import _model, _native  # noqa: E402
from causalimpact import _synthetic as syn  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 50
C = int(sys.argv[3]) if len(sys.argv) > 3 else 8
y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
spec = _model.series_params(y, mask, X, num_seasonal_blocks=1)
counts, flg = _model.expand_seasons((ci.Seasons(num_seasons=7),), T)
W, S = 20, 100
pb = _native.make_problem(T=T, P=X.shape[1], has_slope=0, num_seasons=counts, num_warmup=W,
                          num_results=S, num_chains=C, seed=(0, 1))
sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
sess.run()
ms = min(sess.run() for _ in range(2))
sess.profile(True)
sess.run()
cyc = sess.profile(False)
n = W + S
names = {0: "(1) targets, X'targets, sums", 1: "(2) serial + regression block", 9: "   scale draws (wave 0)",
         10: "   regression block (workgroup)", 4: "     build A", 5: "     sweeps of the active set",
         6: "     flip proposals + sweeps", 7: "     Cholesky, solve, weights", 2: "(3) emit", 3: "(4) X w, residual",
         20: "dk A  prior simulation, in-wave scan", 21: "dk B  hand-over, x+ / y~, chunk elements",
         22: "dk B  filter scan inside the wave", 23: "dk C  hand-over + scan of the 16 wave totals",
         24: "dk C  prefix, local filter: gains", 25: "dk C  backward chunk maps",
         26: "dk D  backward scans + hand-over", 27: "dk D  r through the chunk",
         28: "dk D  forward reconstruction", 29: "dk    end of the draw: arrival", 16: "dk    barrier after A", 17: "dk    barrier after B", 18: "dk    barrier after C",
         19: "(DK worker) waits for the latents flag / other stamps", 8: "(DK worker) its segments of X'targets",
         14: "(DK worker) waits for main's serial section", 15: "(DK worker) its share of the emission",
         13: "(DK worker) its share of X w"}
print(f"T={T} P={pb.P} chains={C}: {ms * 1e3 / n:.1f} us per iteration ({ms:.1f} ms per launch)")
tot = 0
for k in sorted(names):
  print(f"  [{k:2d}] {names[k]:36s} {cyc[k] / n:10.0f} cycles / iteration")
  tot += cyc[k]
print(f"  (main's phases [0-10] and the draw's [20-29] are stamped by different workgroups when the cluster has 16: they overlap)")
print(f"  sum {tot / n:.0f} cycles / iteration")
w = sess.fetch(["weights"])["weights"]
print("  active features per draw:", float((w != 0).sum(axis=-1).mean()))
sess.close()
