"""Phase cycles of one HMC leapfrog step on cfg3's shape (csrc/ci_hmc.h, $CI_HMC_PROF): s_memtime on
thread 0 of chain 0, accumulated over every leapfrog of the fit and divided by their number.

  python tools/exp_hmc_phases.py [T] [p]
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r"""
import sys
sys.path[:0] = [%r, %r]
from causalimpact import _model, _native
from causalimpact import _synthetic as syn
T, p, W, S, NL = %d, %d, 100, 200, 15
y, mask, X, _ = syn.make_sampler_inputs(T, p, 12)
spec = _model.series_params(y, mask, X, has_slope=True)
pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=0, num_results=1, seed=(3, 4))
sess = _native.LogLikSession(pb, _native.make_params([spec]), y, mask, X, max_evals=8)
ms = sess.hmc_run(num_chains=8, num_warmup=W, num_results=S, num_leapfrog=NL, seed=(3, 4), prior="slab")
print("kernel_ms", ms, "leapfrogs", (W + S) * NL)
sess.close()
"""

NAMES = {20: "score: y, mask", 21: "score: filter elements (chunk)", 22: "score: X'e dots",
         15: "prior: horseshoe sum", 16: "prior: per-coordinate terms", 17: "prior: wave sum",
         0: "position update + device layout (wave 0)", 1: "barrier: position ready",
         8: "score: residual y - X w", 5: "score: filter elements + scan",
         9: "score: local filter pass", 10: "score: log-likelihood terms, (r, N) maps, chunk compose",
         11: "score: (r, N) suffix scan", 12: "score: backward pass over the owned steps",
         13: "score: X'e dots + reduce-scatter", 14: "score: barrier + float64 block sums",
         2: "barrier: score ready", 3: "prior terms (wave 0)", 4: "momentum half step",
         7: "per iteration: momentum draw, Metropolis test, adaptation, output"}


def main():
  T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
  p = int(sys.argv[2]) if len(sys.argv) > 2 else 10
  env = dict(os.environ, CI_HMC_PROF="1")
  r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, os.path.join(ROOT, "tfp-causalimpact_amd"), T, p)],
                     env=env, capture_output=True, text=True)
  print(r.stdout.strip())
  m = re.search(r"ci hmc prof:(.*)", r.stderr)
  if not m:
    print(r.stderr[-2000:])
    raise SystemExit("no profile line")
  cyc = [int(v) for v in m.group(1).split()]
  n = int(re.search(r"leapfrogs (\d+)", r.stdout).group(1))
  tot = 0
  for k in (0, 1, 20, 8, 21, 5, 9, 10, 11, 12, 22, 13, 14, 2, 15, 16, 17, 3, 4, 7):
    print(f"  [{k:2d}] {NAMES[k]:60s} {cyc[k] / n:9.0f} cycles / leapfrog")
    tot += cyc[k]
  print(f"  total {tot / n:.0f} cycles / leapfrog")


if __name__ == "__main__":
  main()
