"""Kernel time of the sequential seasonal kernel on the reference's 4+7+6-season test model
(causalimpact_lib_test.py:740-752) at BASELINE cfg4's size (T=10000, 50 covariates)."""
import sys, time
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import numpy as np
from causalimpact import _native, _model
from causalimpact import _synthetic as syn
SEAS = ((4, (2, 1, 1, 1)), (7, 1), (6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1))))
for T, p in ((300, 0), (1000, 10), (10000, 50)):
  y, mask, X, _ = syn.make_sampler_inputs(T, max(p, 1), 7)
  if p == 0: X = None
  spec = _model.series_params(y, mask, X, num_seasonal_blocks=3)
  counts, flg = _model.expand_seasons(SEAS, T)
  W, S, C = 2, 10, 8
  pb = _native.make_problem(T=T, P=0 if X is None else X.shape[1], has_slope=0, num_seasons=counts,
                            num_warmup=W, num_results=S, num_chains=C, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None], flg, _native.make_params([spec]))
  sess.run(); ms = sess.run()
  print(f"T={T} P={pb.P} D_full=18 {sess.kernel_name()}: {ms / (W + S) * 1e3:.0f} us per Gibbs iteration "
        f"({ms / (W + S) * 1e3 * 2400 / T:.0f} cycles per step), 8 chains")
  sess.close()
