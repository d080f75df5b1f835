"""Trend models with more than 52 design columns on the time-parallel kernel: time per Gibbs iteration
over a 250-iteration fit (8 chains) and the phase budget of the workgroup-wide regression draw
(tp_spike_slab_draw_big_wg, csrc/ci_seasonal_tp.h)."""
import sys
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
import numpy as np
from causalimpact import _native, _model
from causalimpact import _synthetic as syn

REG = ["build + sweep-in", "order + proposals + flips", "active set + block", "Cholesky", "solve + weights"]
for T, p in ((1000, 100), (1000, 60), (1000, 200), (4000, 100)):
  W, S, C = 50, 200, 8
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  spec = _model.series_params(y, mask, X)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=C, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  sess.run()
  ms = sess.run()
  sess.profile(True)
  sess.run()
  cyc = sess.profile(False)
  w = sess.fetch(want=("weights",))["weights"]
  print(f"T={T} P={p + 1} {sess.kernel_name()}: {ms / (W + S) * 1e3:.1f} us per iteration, "
        f"{(w != 0).mean() * (p + 1):.1f} columns in the model on average")
  s0 = 11 if "gibbs_wide" in sess.kernel_name() else 9      # the draw's five slots (ci_bigp.h: slot0)
  print("    regression draw, cycles per iteration: " +
        "  ".join(f"{n}: {cyc[s0 + i] / (W + S):.0f}" for i, n in enumerate(REG)))
  sess.close()
