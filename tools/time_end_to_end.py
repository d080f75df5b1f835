"""Where fit_causalimpact's wall time goes once the sampler is fast."""
import sys, time, cProfile, pstats, io
import numpy as np, pandas as pd
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _synthetic as syn

T, p = 1000, 10
y, X = syn.make_raw_series(T, p, 0)
df = pd.DataFrame(np.column_stack([y, X]), columns=["y"] + [f"x{j}" for j in range(p)])
pre, post = (0, int(0.7 * T) - 1), (int(0.7 * T), T - 1)
opts = ci.InferenceOptions(num_results=1000, num_chains=8)
mo = ci.ModelOptions(local_linear_trend=True)
for rep in range(2):
  t0 = time.time()
  an = ci.fit_causalimpact(df, pre, post, seed=1, inference_options=opts, model_options=mo)
  print(f"fit_causalimpact total {time.time() - t0:.3f} s")
pr = cProfile.Profile(); pr.enable()
an = ci.fit_causalimpact(df, pre, post, seed=1, inference_options=opts, model_options=mo)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:6000])
print(ci.summary(an))
