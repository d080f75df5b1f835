import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _native
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc
T, p, slope, S = 1000, 10, 1, 6
y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
spec = orc.default_spec(y, mask, X, has_slope=bool(slope))
pb = _native.make_problem(T=T, P=spec["P"], has_slope=slope, num_warmup=0, num_results=S, seed=(5, 9))
got = _native.fit_gibbs(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
want = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=0, seed=(5, 9))
np.set_printoptions(precision=5, linewidth=200, suppress=True)
for s in range(S):
  print("it", s, "obs", got["observation_noise_scale"][0, 0, s], want["obs_scale"][s],
        "lvl", got["level_scale"][0, 0, s], want["level_scale"][s])
  print("  gpu w", got["weights"][0, 0, s])
  print("  orc w", want["weights"][s])
