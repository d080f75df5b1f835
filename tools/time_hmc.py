import sys, time
import numpy as np, pandas as pd
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _synthetic as syn
T, p = 1000, 10
y, X = syn.make_raw_series(T, p, 0)
df = pd.DataFrame(np.column_stack([y, X]), columns=["y"] + [f"x{j}" for j in range(p)])
pre, post = (0, 699), (700, 999)
for S, C in ((200, 8), (1000, 8)):
  opts = ci.InferenceOptions(num_results=S, num_chains=C, sampler="hmc")
  t0 = time.time()
  an = ci.fit_causalimpact(df, pre, post, seed=1, inference_options=opts,
                           model_options=ci.ModelOptions(local_linear_trend=True))
  dt = time.time() - t0
  print(f"HMC S={S} C={C}: {dt:.2f} s -> {S*C/dt:.0f} samples/s; rhat {an.diagnostics['split_rhat']}")
  print(an.summary[["abs_effect", "abs_effect_lower", "abs_effect_upper"]])

# sampler alone (device chain + latent / trajectory draws), no host post-processing
from causalimpact import _hmc, _model
ys, mask, Xs, _ = syn.make_sampler_inputs(T, p, 0)
spec = _model.series_params(ys, mask, Xs, has_slope=True)
for C in (8, 64):
  _hmc.fit_hmc(ys, mask, Xs, spec, has_slope=True, num_results=50, num_warmup=10, num_chains=2, seed=1)
  t0 = time.time()
  out = _hmc.fit_hmc(ys, mask, Xs, spec, has_slope=True, num_results=1000, num_warmup=112,
                     num_chains=C, seed=1)
  dt = time.time() - t0
  print(f"fit_hmc alone C={C}: {dt:.3f} s -> {1000*C/dt:.0f} samples/s; accept {out['hmc_accept_rate'].mean():.2f}"
        f" eps {out['hmc_step_size'].mean():.3f}")
