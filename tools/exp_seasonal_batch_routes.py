"""Throughput of a LARGE batch of general-seasonal series (more chains than CUs) on the two routes:
the time-parallel kernel with one workgroup per chain (the route every launch size takes since
round 6, so that a series of a batch reproduces its single-series fit bit for bit) against the
one-wavefront sequential kernel (CI_FLAG_SEQUENTIAL_SEASONAL, which a caller may still ask for --
for the batch AND the single fit alike)."""
import sys
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import numpy as np
from causalimpact import _native, _model
from causalimpact import _synthetic as syn

SEAS = ((4, 1), (7, 4))
for T, p, B, C in ((500, 5, 512, 1), (500, 5, 128, 4), (2000, 10, 512, 1)):
  ys, ms, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 7 + b)
    ys.append(y); ms.append(mask); Xs.append(X)
    specs.append(_model.series_params(y, mask, X, num_seasonal_blocks=2))
  counts, flg = _model.expand_seasons(SEAS, T)
  W, S = 2, 10
  for flags, tag in ((0, "route by model (time-parallel)"), (_native.FLAG_SEQUENTIAL_SEASONAL, "sequential flag")):
    pb = _native.make_problem(T=T, P=Xs[0].shape[1], has_slope=0, num_seasons=counts, num_warmup=W, num_results=S,
                              num_chains=C, num_series=B, seed=(0, 1), flags=flags)
    sess = _native.Session(pb, np.stack(ys), np.stack(ms), np.stack(Xs), flg, _native.make_params(specs))
    sess.run(); t = sess.run()
    print(f"T={T} P={pb.P} B={B} C={C} {tag:32s} {sess.kernel_name():48s} {t:8.2f} ms per launch of {W + S} iterations "
          f"= {B * C * (W + S) / t:8.0f} chain-iterations / ms")
    sess.close()
