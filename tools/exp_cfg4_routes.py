#!/usr/bin/env python3
"""cfg4 (T=10000, 50 covariates, weekly block) through each route that can run it: the
trend + one-block kernel (ci_wide.h) at several cluster sizes / chain counts and the general
time-parallel kernel (ci_seasonal_tp.h, CI_FLAG_CLUSTER_SEASONAL).  us per Gibbs iteration."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
import causalimpact as ci  # noqa: E402
from causalimpact import _model, _native  # noqa: E402
from causalimpact import _synthetic as syn  # noqa: E402

T, p = 10000, 50
y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
spec = _model.series_params(y, mask, X, num_seasonal_blocks=1)
counts, flg = _model.expand_seasons((ci.Seasons(num_seasons=7),), T)
W, S = 20, 100


def run(C, flags, label):
  pb = _native.make_problem(T=T, P=X.shape[1], has_slope=0, num_seasons=counts, num_warmup=W,
                            num_results=S, num_chains=C, seed=(0, 1), flags=flags)
  sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  print(f"{label:40s} chains={C:3d}: {ms * 1e3 / (W + S):8.1f} us / iteration   {sess.kernel_name()}", flush=True)
  sess.close()


for C in (1, 8, 32, 64):
  run(C, 0, "wide (default cluster)")
run(8, _native.FLAG_NO_CLUSTER, "wide, one CU per chain")
for C in (1, 8):
  run(C, _native.FLAG_CLUSTER_SEASONAL, "general time-parallel kernel")
