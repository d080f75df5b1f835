"""Experiment: are gibbs_kernel8 and gibbs_kernel<.,.,1> bit-identical? (prints per shape)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
import numpy as np
from causalimpact import _native, _model
from causalimpact import _synthetic as syn

for (T, p, slope, B, C, n) in [(1000, 10, 1, 1, 3, 400), (500, 5, 0, 4, 2, 300), (100, 1, 0, 1, 2, 300),
                               (300, 15, 1, 1, 2, 300), (2000, 8, 1, 1, 2, 200), (4000, 3, 0, 1, 2, 100)]:
  ys, ms, Xs, sp = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 60 + b)
    ys.append(y); ms.append(mask); Xs.append(X)
    sp.append(_model.series_params(y, mask, X, has_slope=bool(slope)))
  out = {}
  for flags in (0, _native.FLAG_FOUR_WAVES):
    pb = _native.make_problem(T=T, P=p + 1, has_slope=slope, num_warmup=0, num_results=n,
                              num_chains=C, num_series=B, seed=(6, 2), flags=flags)
    s = _native.Session(pb, np.stack(ys), np.stack(ms), np.stack(Xs), None, _native.make_params(sp))
    out[flags] = (s.kernel_name(), s.run(), s.fetch())
    s.close()
  a, b4 = out[0][2], out[4][2]
  rep = []
  for k in ("observation_noise_scale", "level_scale", "weights", "level", "slope", "posterior_trajectories", "posterior_means"):
    d = a[k] != b4[k]
    if d.any():
      # first iteration that differs
      ax = np.argwhere(d)
      first = ax[:, 2].min() if ax.shape[1] > 2 else -1
      rep.append(f"{k}: {int(d.sum())} differ (first it {first}, max {np.abs(a[k]-b4[k]).max():.3g})")
  print(f"T={T} p={p} slope={slope} B={B} C={C}: {out[0][0]} {out[0][1]:.2f} ms vs {out[4][0]} {out[4][1]:.2f} ms ->",
        "BIT-EQUAL" if not rep else "; ".join(rep), flush=True)
