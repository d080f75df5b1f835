"""Per-iteration time of the Gibbs kernels as the series grows (10 covariates, local linear trend, 8 chains)."""
import sys
import numpy as np
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
W, S, C, p = 20, 80, 8, 10
for T in (100, 256, 512, 1000, 2048, 4096, 4100, 8192, 16384, 65536):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
  spec = _model.series_params(y, mask, X, has_slope=True)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=W, num_results=S, num_chains=C, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  print(f"T={T}: {sess.kernel_name()} {ms / (W + S) * 1e3:.2f} us per iteration", flush=True)
  sess.close()
