"""Per-iteration time of the register-resident Gibbs kernels as the number of covariates grows
(T = 1000, local linear trend, 8 chains)."""
import sys
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
T, W, S, C = 1000, 50, 200, 8
for p in (0, 5, 10, 15, 16, 20, 30, 40, 51):
  y, mask, X, _ = syn.make_sampler_inputs(T, max(p, 1), 2024)
  X = X[:, :p + 1] if p else None
  spec = _model.series_params(y, mask, X, has_slope=True)
  P = p + 1 if p else 0
  pb = _native.make_problem(T=T, P=P, has_slope=1, num_warmup=W, num_results=S, num_chains=C, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None], None, _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  print(f"P={P}: {sess.kernel_name()} {ms / (W + S) * 1e3:.2f} us per iteration", flush=True)
  sess.close()
