import sys
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import numpy as np
from causalimpact import _native, _model
from causalimpact import _synthetic as syn
T, p = 1000, 10
y, mask, X, _ = syn.make_sampler_inputs(T, p, 60)
spec = _model.series_params(y, mask, X, has_slope=True)
out = {}
for flags in (0, _native.FLAG_FOUR_WAVES):
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=0, num_results=8, num_chains=1, seed=(6, 2), flags=flags)
  out[flags] = _native.fit_gibbs(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
a, b = out[0], out[4]
for s in range(8):
  print(s, "obs", a["observation_noise_scale"][0,0,s], b["observation_noise_scale"][0,0,s],
        "lvl", a["level_scale"][0,0,s] - b["level_scale"][0,0,s],
        "w max diff", np.abs(a["weights"][0,0,s] - b["weights"][0,0,s]).max(),
        "level max diff", np.abs(a["level"][0,0,s] - b["level"][0,0,s]).max(),
        "traj", np.abs(a["posterior_trajectories"][0,0,s] - b["posterior_trajectories"][0,0,s]).max())
  print("   w5", a["weights"][0,0,s], "\n   w4", b["weights"][0,0,s])
