#!/usr/bin/env python3
"""Kernel time of the seasonal-model Gibbs kernel on the reference's seasonality test shape
(T=300, Seasons 4/7/6 => 18-component state) and on a weekly model with covariates."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _model, _native  # noqa: E402
from causalimpact import _synthetic as syn  # noqa: E402

CASES = [
    ("ref seasonality test", 300, 0, ((4, (2, 1, 1, 1)), (7, 1), (6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1))))),
    ("weekly + 5 covariates", 1000, 5, ((7, 1),)),
    ("4 + 7 + 6 seasons, long", 10000, 0, ((4, 1), (7, 1), (6, 1))),
]
for name, T, p, seasons in CASES:
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 5)
  spec = _model.series_params(y, mask, X, num_seasonal_blocks=len(seasons))
  counts, flags = _model.expand_seasons(seasons, T)
  W, S, C = (20, 100, 8) if T < 5000 else (2, 10, 8)
  pb = _native.make_problem(T=T, P=0 if X is None else X.shape[1], has_slope=0, num_seasons=counts,
                            num_warmup=W, num_results=S, num_chains=C, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None], flags,
                         _native.make_params([spec]))
  sess.run()
  ms = np.mean([sess.run() for _ in range(3)])
  sess.profile(True)
  sess.run()
  cyc = sess.profile(False)
  names = ["targets+sums", "serial", "resid+normals", "pass0 sim", "P1 init", "pass1 filter",
           "pass2 backward", "pass3 reconstruct"]
  print("   " + "  ".join(f"{n}: {cyc[20 + i] / (W + S) / T:.0f}" for i, n in enumerate(names)),
        "(cycles per step per iteration)")
  per_it = ms * 1e3 / (W + S)
  print(f"{name}: T={T} P={pb.P} D_full={1 + sum(counts)}: {ms:.2f} ms per launch, {per_it:.1f} us/iteration, "
        f"{per_it * 2400 / T:.0f} cycles/step/iteration")
  sess.close()
