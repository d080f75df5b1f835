"""Where the float64 kernel's time goes: fits with / without covariates, short / long series."""
import sys
import time

import numpy as np

sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
from causalimpact import _model, _native
from causalimpact import _synthetic as syn

for T, p, slope in ((1000, 10, 1), (1000, 0, 1), (250, 10, 1), (250, 0, 1), (1000, 10, 0), (4000, 10, 1)):
  W, S, C = 20, 100, 8
  y, mask, X, _ = syn.make_sampler_inputs(T, max(p, 1), 2024)
  X = X[:, :p + 1] if p else None
  spec = _model.series_params(y, mask, X, has_slope=bool(slope))
  prm = _native.make_params([spec])
  P = p + 1 if p else 0
  pb = _native.make_problem(T=T, P=P, has_slope=slope, num_warmup=W, num_results=S, num_chains=C, seed=(0, 1))
  Xb = None if X is None else X[None]
  _native.fit_gibbs_f64(pb, y[None], mask[None], Xb, None, prm, want=("observation_noise_scale",))
  t0 = time.time()
  _native.fit_gibbs_f64(pb, y[None], mask[None], Xb, None, prm, want=("observation_noise_scale",))
  dt = time.time() - t0
  print(f"T={T} P={P} slope={slope}: {dt / (W + S) * 1e6:.1f} us / iteration", flush=True)
