import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _native, _model
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc
T, p, slope, seasons = 60, 0, 0, ((3, 1),)
y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
rng = np.random.default_rng(0)
y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0) + 0.1 * rng.normal(size=T)
spec = orc.default_spec(y, mask, X, has_slope=bool(slope), seasons=seasons)
counts, flags = _model.expand_seasons(seasons, T)
pb = _native.make_problem(T=T, P=spec["P"], has_slope=slope, num_seasons=counts, num_warmup=0, num_results=2, seed=(2, 6))
got = _native.fit_gibbs(pb, y[None], mask[None], None, flags, _native.make_params([spec]))
w = orc.fit_gibbs(y, mask, X, spec, num_results=2, num_warmup=0, seed=(2, 6))
np.set_printoptions(precision=4, linewidth=220, suppress=True)
print("init loc", spec["init_level_loc"], "y[:6]", y[:6])
print("gpu lev ", got["level"][0,0,0,:12]); print("orc lev ", w["level"][0,:12])
print("gpu seas", got["seasonal_levels"][0,0,0,:12,0]); print("orc seas", w["seasonal"][0,:12,0])
print("gpu sum ", (got["level"][0,0,0]+got["seasonal_levels"][0,0,0,:,0])[:12]); print("orc sum ", (w["level"][0]+w["seasonal"][0,:,0])[:12])
print("gpu obs/lvl/drift", got["observation_noise_scale"][0,0], got["level_scale"][0,0], got["seasonal_drift_scales"][0,0].ravel())
print("orc obs/lvl/drift", w["obs_scale"], w["level_scale"], w["drift_scales"].ravel())
