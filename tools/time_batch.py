import sys, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _synthetic as syn
B, T, p = 512, 500, 5
values = np.stack([np.column_stack(syn.make_raw_series(T, p, b)) for b in range(B)])
opts = ci.InferenceOptions(num_results=1000)
ci.fit_causalimpact_batch(values[:2], (0, 349), (350, 499), seed=1, inference_options=opts)
t0 = time.time(); res = ci.fit_causalimpact_batch(values, (0, 349), (350, 499), seed=1, inference_options=opts); print("wall", time.time() - t0)
pr = cProfile.Profile(); pr.enable()
res = ci.fit_causalimpact_batch(values, (0, 349), (350, 499), seed=1, inference_options=opts)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25); print(s.getvalue()[:5000])
