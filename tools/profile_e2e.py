"""Host profile of fit_causalimpact end to end at cfg2's size (tools/profile_e2e.py [n])."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tfp-causalimpact_amd"))
import causalimpact as ci  # noqa: E402


def make(T=1000, p=10, seed=0):
  rng = np.random.default_rng(seed)
  X = rng.normal(size=(T, p))
  y = 1.2 * X[:, 0] + np.cumsum(rng.normal(scale=0.05, size=T)) + rng.normal(scale=0.3, size=T)
  y[700:] += 0.5
  df = pd.DataFrame(np.column_stack([y, X]), columns=["y"] + [f"x{i}" for i in range(p)])
  return df


def run(df):
  return ci.fit_causalimpact(df, pre_period=(0, 699), post_period=(700, 999),
                             model_options=ci.ModelOptions(local_linear_trend=True),
                             inference_options=ci.InferenceOptions(num_results=1000, num_warmup_steps=112, num_chains=8),
                             seed=1)


if __name__ == "__main__":
  df = make()
  run(df)
  ts = []
  for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    t0 = time.perf_counter(); run(df); ts.append(time.perf_counter() - t0)
  print("fit_causalimpact wall ms: min %.2f median %.2f" % (min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3))
  pr = cProfile.Profile()
  pr.enable()
  for _ in range(5):
    run(df)
  pr.disable()
  s = io.StringIO()
  pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(60)
  print(s.getvalue()[:9000])
