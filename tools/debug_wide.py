"""Per-draw comparison of the time-parallel seasonal kernel (ci_wide.h) with the oracle."""
import sys, time
import numpy as np
sys.path.insert(0, "tfp-causalimpact_amd"); sys.path.insert(0, ".")
from causalimpact import _native, _model
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc

def run(T, p, has_slope, seasons, S=3, flags=0, seed=(2, 6)):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  rng = np.random.default_rng(0)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0) + 0.1 * rng.normal(size=T)
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_seasons=counts,
                            num_warmup=0, num_results=S, seed=seed, flags=flags)
  t0 = time.time()
  got = _native.fit_gibbs(pb, y[None], mask[None], None if X is None else X[None], flg,
                          _native.make_params([spec]))
  t1 = time.time()
  w = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=0, seed=seed)
  t2 = time.time()
  def err(a, b): return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
  print(f"T={T} p={p} slope={has_slope} seasons={seasons} flags={flags}: gpu {t1-t0:.2f}s oracle {t2-t1:.2f}s")
  for s in range(S):
    print(f"   draw {s}: level {err(got['level'][0,0,s], w['level'][s]):.2e}"
          f" seas {err(got['seasonal_levels'][0,0,s], w['seasonal'][s]):.2e}"
          f" traj {err(got['posterior_trajectories'][0,0,s], w['trajectories'][s]):.2e}"
          f" obs {got['observation_noise_scale'][0,0,s]:.5f}/{w['obs_scale'][s]:.5f}"
          f" lvl {got['level_scale'][0,0,s]:.5f}/{w['level_scale'][s]:.5f}"
          f" drift {got['seasonal_drift_scales'][0,0,s,0]:.6f}/{w['drift_scales'][s,0]:.6f}"
          + (f" w {err(got['weights'][0,0,s], w['weights'][s]):.2e}" if spec['P'] else "")
          + (f" slope {err(got['slope'][0,0,s], w['slope'][s]):.2e}" if has_slope else ""))

def perf(T, p, has_slope, C=8, W=12, S=100):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=((7, 1),))
  counts, flg = _model.expand_seasons(((7, 1),), T)
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_seasons=counts,
                            num_warmup=W, num_results=S, num_chains=C, seed=(1, 2))
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None], flg,
                         _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  sess.profile(True); sess.run(); cyc = sess.profile(False)
  it = W + S
  names = {0: "sums", 1: "serial(rest)", 9: "ss sweep-in (P<=16) | serial pre", 10: "ss flips (P<=16) | block regression (P>16)", 11: "ss gather", 12: "ss weights",
           2: "emit", 3: "Xw", 20: "P elem+scan", 21: "x+ / F elem", 22: "F scan", 23: "filter",
           24: "r elem", 25: "r scan", 26: "draw+stats"}
  print(f"perf T={T} P={spec['P']} slope={has_slope} C={C}: {ms:.1f} ms per launch, {ms/it*1e3:.0f} us/iteration,"
        f" {C*S/ms*1e3:.0f} samples/s")
  print("   kcycles/iteration: " + "  ".join(f"{n}: {cyc[i]/it/1e3:.1f}" for i, n in names.items()))

def perf_flagship(T, p, has_slope=0, C=8, W=12, S=100):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope))
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_warmup=W, num_results=S,
                            num_chains=C, seed=(1, 2))
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  sess.run()
  ms = min(sess.run() for _ in range(2))
  out = sess.fetch(["weights"])
  sess.profile(True); sess.run(); cyc = sess.profile(False)
  it = W + S
  print(f"flagship T={T} P={spec['P']}: {ms/it*1e3:.0f} us/iteration; mean #nonzero weights "
        f"{(out['weights'] != 0).sum(-1).mean():.1f}")
  print("   kcycles/iteration: " + "  ".join(f"{i}: {cyc[i]/it/1e3:.1f}" for i in range(19) if cyc[i]))


if len(sys.argv) > 1 and sys.argv[1] == "ss":
  perf_flagship(1000, 50)
  perf_flagship(1000, 20)


if __name__ == "__main__":
  if len(sys.argv) > 1 and sys.argv[1] == "ss":
    sys.exit(0)
  if len(sys.argv) > 1 and sys.argv[1] == "perf":
    perf(1000, 5, 0); perf(1000, 10, 1); perf(10000, 50, 0, S=40, W=4); perf(10000, 0, 0, S=40, W=4)
    run(10000, 50, 0, ((7, 1),), S=2)
    sys.exit(0)
  run(120, 0, 0, ((7, 1),))
  run(120, 4, 1, ((7, 1),))
  run(300, 2, 0, ((7, 2),))
  run(1030, 20, 0, ((7, 1),))
  run(3000, 3, 1, ((7, 1),), S=2)
