#!/usr/bin/env python3
"""Per-phase shader-clock breakdown of the Gibbs kernel (block 0, thread 0) on cfg2.

    python tools/profile_phases.py [T] [covariates] [has_slope] [chains]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
from causalimpact import _model, _native  # noqa: E402
from causalimpact import _synthetic as syn  # noqa: E402

SLOTS = ["partial sums + reduce", "serial section (wave 0) total", "emit", "residual X w",
         "dk: normals + prior-sim scan", "dk: filter elements + scan", "dk: local Kalman pass",
         "dk: backward scan + fix-up", "serial: gather + scale draws", "serial: build + sweep-in",
         "serial: flips", "serial: gamma + active set", "serial: chol + weights",
         "  (0a) X'targets partials + DPP sums", "  (0b) boundary exchange + increments", "  (0a1) loop top + targets", "  (0a2) X reads + fma", "  (8a) gather partial sums",
         "  (8b) gamma_wave4"]


def main():
  T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
  p = int(sys.argv[2]) if len(sys.argv) > 2 else 10
  slope = int(sys.argv[3]) if len(sys.argv) > 3 else 1
  C = int(sys.argv[4]) if len(sys.argv) > 4 else 8
  W, S = 112, 1000
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
  spec = _model.series_params(y, mask, X, has_slope=bool(slope))
  pb = _native.make_problem(T=T, P=0 if X is None else X.shape[1], has_slope=slope, num_warmup=W,
                            num_results=S, num_chains=C, seed=(0, 1))
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None], None,
                         _native.make_params([spec]))
  sess.run()
  plain = np.mean([sess.run() for _ in range(3)])
  sess.profile(True)
  ms = sess.run()
  cyc = sess.profile(False)
  n_it = W + S
  print(f"T={T} P={pb.P} slope={slope} chains={C}: kernel {plain:.2f} ms plain, {ms:.2f} ms profiled; "
        f"{plain * 1e3 / n_it:.2f} us/iteration")
  top = cyc[:8].sum()
  for i, name in enumerate(SLOTS):
    print(f"  [{i:2d}] {name:34s} {cyc[i] / n_it:9.0f} cyc/iter  {100.0 * cyc[i] / top:5.1f} %")
  print(f"  total of phases 0-7: {top / n_it:.0f} cyc/iter")
  if "kernel5" in sess.kernel_name():
    print("  five-wave kernel: [1] = wait at (B3) for the regression wave, [2] = overlap work "
          "(emit, normals) before it")
    for i, name in ((16, "regression wave: idle until (B2)"), (17, "regression wave: serial section"),
                    (18, "regression wave: wait at (B3)"), (19, "regression wave: precompute + dk barriers"),
                    (20, "  serial: gather, scale draws, stores"), (21, "  serial: right-hand-side replay"),
                    (22, "  serial: flip proposals"), (23, "  serial: sigma^2, weights replay")):
      print(f"  [{i:2d}] {name:42s} {cyc[i] / n_it:9.0f} cyc/iter")


if __name__ == "__main__":
  main()
