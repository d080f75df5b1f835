"""Experiment: first draw where the eight-wave and four-wave kernels differ, and where; also the eight-wave kernel under the schedule test modes (must not differ at all)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _native
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc
T, p, W, S = 500, 5, 0, 12
keys = ("level", "weights", "observation_noise_scale", "level_scale", "posterior_trajectories")
for b in (100, 511, 37, 5):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 1000 + b)
  spec = orc.default_spec(y, mask, X)
  out = {}
  for flags, dbg in ((0, "0"), (_native.FLAG_FOUR_WAVES, "0"), (1000, "1"), (2000, "2")):
    os.environ["CI_SCHED_WORD"] = dbg
    pb = _native.make_problem(T=T, P=6, has_slope=0, num_warmup=W, num_results=S, seed=(5, 12),
                              series_offset=b, flags=flags if flags < 1000 else 0)
    s1 = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
    s1.run()
    out[flags] = s1.fetch(want=keys)
    s1.close()
  for tag in (1000, 2000):
    for k in keys:
      if not np.array_equal(out[0][k], out[tag][k]):
        d = out[0][k][0, 0] != out[tag][k][0, 0]
        print(f"series {b}: eight-wave paced vs mode {tag // 1000}: {k} differs in draws {np.unique(np.nonzero(d.reshape(S, -1))[0])}")
  for s in range(S):
    msg = []
    for k in keys:
      a8, a4 = out[0][k][0, 0][s], out[4][k][0, 0][s]
      if not np.array_equal(a8, a4):
        idx = np.nonzero(np.atleast_1d(a8 != a4))[0]
        msg.append(f"{k}: {idx.size} differ at {idx[:12].tolist()}")
    if msg:
      print(f"series {b} draw {s}: " + "; ".join(msg))
print("done")
