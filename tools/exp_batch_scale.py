"""Kernel time of the batch Gibbs launch (cfg5's series: T = 500, 5 covariates) as the batch grows."""
import sys, time
import numpy as np
import pandas as pd
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import causalimpact as ci
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
T, p, S, W = 500, 5, 200, 23
base = np.stack([np.column_stack(syn.make_raw_series(T, p, b)) for b in range(512)])
for B in (256, 512, 1024, 2048, 4096):
  values = np.concatenate([base] * ((B + 511) // 512))[:B]
  prep = ci.batch.prepare_batch(values, pd.RangeIndex(T), (0, 349), (350, 499))
  params = [_model.series_params(prep.y[b], prep.mask[b], prep.design[b]) for b in range(B)]
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=1,
                            num_series=B, seed=(0, 1))
  sess = _native.Session(pb, prep.y, prep.mask, prep.design, None, _native.make_params(params))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  print(f"B={B}: {sess.kernel_name()} {ms:.2f} ms for {W + S} iterations -> {ms / (W + S) * 1e3:.2f} us per iteration of the batch, "
        f"{B * (W + S) / ms / 1e3:.1f}M series-iterations/s", flush=True)
  sess.close()
