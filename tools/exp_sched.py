"""Experiment: kernel time of cfg2 for helper-wave schedule words (CI_DBG)."""
import os, sys, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
y, mask, X, _ = syn.make_sampler_inputs(1000, 10, 2024)
spec = _model.series_params(y, mask, X, has_slope=True)
pb = _native.make_problem(T=1000, P=11, has_slope=1, num_warmup=112, num_results=1000, num_chains=8, seed=(0, 1))
sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
def word(p3, p4, pa, c3, c4, ca): return (p3 << 4) | (p4 << 6) | (pa << 8) | (c3 << 10) | (c4 << 12) | (ca << 14)
res = []
for p3, p4, pa, c3, c4, ca in itertools.product((2, 3), (0, 1, 2), (0, 1), (1, 2, 3), (0, 1, 2, 3), (0,)):
  os.environ["CI_SCHED_WORD"] = str(word(p3, p4, pa, c3, c4, ca))
  sess.run()
  ms = float(np.median([sess.run() for _ in range(3)]))
  res.append((ms, (p3, p4, pa, c3, c4, ca)))
res.sort()
for ms, w in res[:15]: print(f"{ms:.3f} ms  p3,p4,pa,c3,c4,ca = {w}")
print("...")
for ms, w in res[-3:]: print(f"{ms:.3f} ms  {w}")
os.environ["CI_SCHED_WORD"] = "0"
print("default", float(np.median([sess.run() for _ in range(5)])))
