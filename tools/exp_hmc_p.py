"""Per-leapfrog time of the trend HMC kernel as the number of covariates grows (T = 1000, 8 chains)."""
import sys
import numpy as np
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
T, W, S, C, NL = 1000, 30, 30, 8, 15
for p in (10, 20, 30, 35, 40, 51):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
  spec = _model.series_params(y, mask, X, has_slope=True)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=0, num_results=1, seed=(0, 1))
  ll = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=8)
  ll.hmc_run(num_chains=C, num_warmup=W, num_results=S, num_leapfrog=NL, seed=(0, 1))
  hmc_ms, lat_ms = ll.hmc_run(num_chains=C, num_warmup=W, num_results=S, num_leapfrog=NL, seed=(0, 1))
  print(f"P={p + 1}: {ll.kernel_name()} {hmc_ms * 1e3 / ((W + S) * NL):.2f} us per leapfrog", flush=True)
  ll.close()
