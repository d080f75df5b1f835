"""Per-iteration time of the time-parallel seasonal kernel (trend + weekly block) over T and P."""
import sys
import numpy as np
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
W, S, C = 10, 40, 8
for T, p in ((1000, 0), (1000, 10), (1000, 30), (1000, 51), (4000, 10), (10000, 10), (10000, 51)):
  y, mask, X, _ = syn.make_sampler_inputs(T, max(p, 1), 2024)
  X = X[:, :p + 1] if p else None
  spec = _model.series_params(y, mask, X, num_seasonal_blocks=1)
  counts, flg = _model.expand_seasons((ci.Seasons(num_seasons=7),), T)
  pb = _native.make_problem(T=T, P=p + 1 if p else 0, has_slope=0, num_seasons=counts, num_warmup=W, num_results=S,
                            num_chains=C, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None], flg, _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  print(f"T={T} P={pb.P}: {sess.kernel_name()} {ms / (W + S) * 1e3:.1f} us per iteration", flush=True)
  sess.close()
