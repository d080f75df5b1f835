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
         "dk: normals + prior-sim scan", "dk: forward scan of the means", "dk: local means pass + adjoint chunk",
         "dk: fix-up", "serial: gather + scale draws", "serial: build + sweep-in",
         "serial: flips", "serial: gamma + active set", "serial: chol + weights",
         "  (0a) X'targets partials + DPP sums", "  (0b) boundary exchange + increments", "  (0a1) loop top + targets", "  (0a2) X reads + fma", "  (8a) gather partial sums",
         "  (8b) gamma_wave4"]


def main():
  T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
  p = int(sys.argv[2]) if len(sys.argv) > 2 else 10
  slope = int(sys.argv[3]) if len(sys.argv) > 3 else 1
  C = int(sys.argv[4]) if len(sys.argv) > 4 else 8
  flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0      # 4 = CI_FLAG_FOUR_WAVES
  W, S = 112, 1000
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
  spec = _model.series_params(y, mask, X, has_slope=bool(slope))
  pb = _native.make_problem(T=T, P=0 if X is None else X.shape[1], has_slope=slope, num_warmup=W,
                            num_results=S, num_chains=C, seed=(0, 1), flags=flags)
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
  print(f"  kernel: {sess.kernel_name()}")
  top = cyc[:8].sum() + cyc[24:29].sum()
  for i, name in enumerate(SLOTS):
    print(f"  [{i:2d}] {name:34s} {cyc[i] / n_it:9.0f} cyc/iter  {100.0 * cyc[i] / top:5.1f} %")
  for i, name in ((24, "dk: matrix chunk (A, C, J)"), (25, "dk: matrix scan"), (26, "dk: matrix local pass"),
                  (27, "dk: forward chunk of the means"), (28, "dk: backward scan")):
    print(f"  [{i:2d}] {name:34s} {cyc[i] / n_it:9.0f} cyc/iter  {100.0 * cyc[i] / top:5.1f} %")
  print(f"  total of phases 0-7, 24-28: {top / n_it:.0f} cyc/iter")
  if pb.P > 16 and "kernel8" not in sess.kernel_name():
    if pb.P + 1 <= 32:
      print("  17-31 columns: wave 0 draws alone inside the serial section ([1], [8]): no breakdown")
    print("  workgroup-wide regression block (thread 0): [9] build, [10] sweep-in, [11] flips, "
          "[12] Cholesky + weights")
    print(f"       evaluation rounds per iteration {(cyc[30] + cyc[31]) / n_it:.2f}, accepted flips "
          f"{cyc[31] / n_it:.2f}")
  if "kernel8" in sess.kernel_name():
    print("  eight-wave kernel, time thread 0:")
    for i, name in ((0, "targets .. (B2)"), (2, "window: normals from LDS, emission (+ its normals), scales, prior scan"),
                    (1, "wait at (Bs) for sigma^2_obs"), (24, "matrix chunk + in-wave scan"),
                    (25, "(B3) + cross-wave + local covariance pass"), (3, "X w, residual, prior path"),
                    (27, "forward chunk + in-wave scan"), (5, "(B4) + cross-wave"),
                    (6, "local means + adjoint chunk + in-wave scan"), (28, "(B5) + cross-wave"),
                    (7, "fix-up")):
      print(f"  [{i:2d}] {name:52s} {cyc[i] / n_it:9.0f} cyc/iter")
    t_sum = sum(cyc[i] for i in (0, 2, 1, 24, 25, 3, 27, 5, 6, 28, 7))
    print(f"       time thread 0 total                                  {t_sum / n_it:9.0f} cyc/iter")
    print("  regression wave (lane 0):")
    for i, name in ((16, "(B1) + its two features + (B2)"), (20, "gather"), (21, "right-hand-side replay"),
                    (22, "flips + sigma^2_obs"), (18, "wait at (Bs)"), (23, "weights replay + scale draws + outputs"),
                    (19, "precompute steps (pure, from (Bs) on)"), (17, "its wait at (B3)"), (29, "its waits at (B4) (B5)")):
      print(f"  [{i:2d}] {name:52s} {cyc[i] / n_it:9.0f} cyc/iter")
    print(f"       iterations on the replay route: {cyc[14]} of {n_it}")
    print("  randomness wave 5 (lane 0):")
    for i, name in ((30, "work"), (31, "waits at barriers")):
      print(f"  [{i:2d}] {name:52s} {cyc[i] / n_it:9.0f} cyc/iter")


if __name__ == "__main__":
  main()
