"""Experiment: how long do device HMC (float32 scans) and the float64 oracle stay together over a
full windowed warm-up at T=1000?  Prints step sizes and the agreement of retained draws."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
import numpy as np
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc

T, p = 1000, 10
y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
for prior, slope in (("slab", True), ("horseshoe", False)):
  spec = _model.series_params(y, mask, X, has_slope=slope)
  ospec = orc.default_spec(y, mask, X, has_slope=slope)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=slope, num_warmup=0, num_results=1, seed=(3, 4))
  sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=8)
  for (W, S, NL) in ((200, 10, 4), (200, 10, 15), (150, 5, 2)):
    sess.hmc_run(num_chains=2, num_warmup=W, num_results=S, num_leapfrog=NL, seed=(3, 4), prior=prior)
    draws, acc, eps, arrs = sess.hmc_fetch()
    for c in range(2):
      want = orc.fit_hmc(y, mask, X, ospec, num_results=S, num_warmup=W, num_leapfrog=NL, seed=(3, 4),
                         chain=c, prior=prior)
      rel = np.abs(draws[c] - want["draws"]) / (np.abs(want["draws"]) + 5e-3)
      print(f"{prior} slope={slope} W={W} S={S} NL={NL} chain {c}: eps dev {eps[c]:.5f} oracle "
            f"{want['step_size']:.5f} (rel {abs(eps[c]/want['step_size']-1):.2e}); acc {acc[c]:.3f} vs "
            f"{want['accept_rate']:.3f}; max rel diff per retained draw: "
            + " ".join(f"{v:.1e}" for v in rel.max(axis=1)), flush=True)
  sess.close()
