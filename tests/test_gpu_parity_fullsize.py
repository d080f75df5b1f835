"""Long-run posterior parity at BASELINE's full sizes: the HIP Gibbs path (float32, through the
C-ABI) against the float64 oracle (oracle/ci_oracle.c) on the same seeded inputs.

north_star: "posterior mean/CI within 1 %".  Each summary is compared within
    max(1 % of the oracle VALUE, 4 Monte-Carlo standard errors)
where the standard error is COMPUTED from the spread of per-chain summaries on both sides (not
guessed), and the table of (device, oracle, difference, MC s.e., difference relative to the value,
band as per cent of the value) is written to gpurun_out/parity_fullsize_<cfg>.json for DESIGN.md.
Round 6: the reference of the 1 % figure is the value itself for every scale parameter -- the
floor of 0.05 that rounds 2-5 applied to all quantities is kept for regression weights and
inclusion frequencies only -- and enough chains are run that sigma_obs and, at cfg2, sigma_level
reach 4 s.e. < 1 % and are ASSERTED at 1 %; the quantities that cannot (sigma_slope, sigma_drift,
the 95 % interval ends of slowly mixing scales, the post-period average of the predictive DRAWS,
which has a Monte-Carlo error of several per cent whatever the sampler) are listed with the band
they were held to under `_summary.mc_limited`.  The prediction is ALSO compared
through `posterior_means` (the Rao-Blackwellised predictor, no observation noise): path-wise over
all T steps, on independent replicates, in units of the outcome's standard deviation
(`_compare_paths`).  The device chains reuse the oracle's Philox streams (chain ids 0..C-1), so
oracle and device chains with the same id are the same chain up to float32 round-off; chains with
different ids are independent replicates.
"""
import concurrent.futures
import json
import os

import numpy as np
import pytest

from causalimpact import _native
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_chain(args):
  y, mask, X, spec, S, W, seed, chain, post = args
  r = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=seed, chain=chain,
                    want=("obs_scale", "level_scale", "slope_scale", "drift_scales", "weights",
                          "trajectories", "pred_mean"))
  out = _chain_summaries(r["obs_scale"], r["level_scale"], r["slope_scale"], r["weights"],
                         r["trajectories"][:, post].mean(axis=1), r["pred_mean"][post].mean(),
                         r.get("drift_scales"))
  out["_pred_mean_path"] = np.asarray(r["pred_mean"], np.float64)
  return out


def _chain_summaries(obs, lvl, slp, w, eff, cf, drift=None):
  """Per-chain posterior summaries: means and the 95 % interval ends of the scalars, mean and
  inclusion frequency of every weight, mean / interval of the post-period average prediction."""
  out = {}
  for name, v in (("sigma_obs", obs), ("sigma_level", lvl), ("sigma_slope", slp),
                  ("post_mean_prediction", eff)):
    v = np.asarray(v, np.float64)
    if np.all(v == 0):
      continue
    out[name + ".mean"] = v.mean()
    out[name + ".q025"], out[name + ".q975"] = np.quantile(v, [0.025, 0.975])
  if drift is not None and np.size(drift):
    out["sigma_drift.mean"] = float(np.mean(drift))
  out["counterfactual.mean"] = float(cf)
  w = np.asarray(w, np.float64)
  for j in range(w.shape[1]):
    out[f"w{j}.mean"] = w[:, j].mean()
    out[f"w{j}.incl"] = (w[:, j] != 0).mean()
  return out


def _compare_paths(tag, dev_pm, orc_pm, pre_end, outcome_sd=1.0):
  """posterior_means (mean over draws of level + X w + seasonal: no predictive noise) of
  INDEPENDENT chains, [chains, T] a side: the difference of the pooled paths, step by step, against
  max(1 % of the outcome's s.d., 4 s.e._t) with s.e._t from the spread over chains.  Over the
  pre-period (observed data: posterior s.d. of the predictor ~ 0.05) the 1 % figure must hold
  outright at >= 99 % of the steps; the whole table goes to gpurun_out/."""
  d, o = np.asarray(dev_pm, np.float64), np.asarray(orc_pm, np.float64)
  diff = d.mean(axis=0) - o.mean(axis=0)
  se = np.sqrt(d.var(axis=0, ddof=1) / d.shape[0] + o.var(axis=0, ddof=1) / o.shape[0])
  one = 0.01 * outcome_sd
  allowed = np.maximum(one, 4.0 * se)
  rows = dict(max_abs_diff_pre=float(np.abs(diff[:pre_end]).max()),
              max_abs_diff_post=float(np.abs(diff[pre_end:]).max()),
              max_se_pre=float(se[:pre_end].max()), max_se_post=float(se[pre_end:].max()),
              frac_within_1pct_pre=float((np.abs(diff[:pre_end]) <= one).mean()),
              frac_within_1pct_post=float((np.abs(diff[pre_end:]) <= one).mean()),
              frac_within_band=float((np.abs(diff) <= allowed).mean()), outcome_sd=outcome_sd)
  with open(os.path.join(ROOT, "gpurun_out", f"parity_fullsize_{tag}_paths.json"), "w") as f:
    json.dump(rows, f, indent=1)
  assert rows["frac_within_band"] >= 0.995, rows          # 4 s.e. over T steps: a few may exceed
  assert rows["frac_within_1pct_pre"] >= 0.99, rows
  return rows


def _compare(tag, dev_chains, orc_chains, weight_floor=0.05, must_reach_1pct=()):
  """Every summary against `max(1 % of |oracle value|, 4 Monte-Carlo s.e.)`.  The reference of the
  1 % figure is the ORACLE VALUE ITSELF for every scale parameter and prediction summary (round-5
  review: a floor of 0.05 on a sigma_level of 0.004 made the "1 %" band 13 % wide); only the
  regression weights and inclusion frequencies -- most of them exactly or nearly zero -- keep an
  absolute floor (`weight_floor`, in units of the standardised outcome).  `rel` is the difference
  relative to the oracle value.  Quantities whose Monte-Carlo error cannot reach 1 % with the chains
  run are LISTED (`_summary.mc_limited`, with the band they were actually held to), not hidden;
  those named in `must_reach_1pct` must have 4 s.e. < 1 % AND meet the 1 % figure."""
  keys = sorted(dev_chains[0])
  rows, bad = {}, []
  for k in keys:
    if k.startswith("_"):
      continue
    d = np.array([c[k] for c in dev_chains])
    o = np.array([c[k] for c in orc_chains])
    se = float(np.sqrt(d.var(ddof=1) / d.size + o.var(ddof=1) / o.size))
    diff = float(d.mean() - o.mean())
    val = abs(float(o.mean()))
    ref = max(val, weight_floor) if k.startswith("w") else val
    allowed = max(0.01 * ref, 4.0 * se)
    rows[k] = dict(device=float(d.mean()), oracle=float(o.mean()), diff=diff, mc_se=se,
                   rel=(diff / val if val > 0 else None), ref=ref, allowed=allowed,
                   band_pct_of_value=(100.0 * allowed / val if val > 0 else None),
                   mc_limited=bool(4.0 * se > 0.01 * ref))
    if abs(diff) > allowed:
      bad.append((k, rows[k]))
  head = [k for k in rows if not k.startswith("w")]
  rows["_summary"] = dict(
      chains=dict(device=len(dev_chains), oracle=len(orc_chains)),
      within_1pct_of_value=[k for k in head if rows[k]["rel"] is not None and abs(rows[k]["rel"]) < 0.01],
      mc_limited={k: rows[k]["band_pct_of_value"] for k in head if rows[k]["mc_limited"]},
      worst_rel={k: rows[k]["rel"] for k in head})
  os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
  with open(os.path.join(ROOT, "gpurun_out", f"parity_fullsize_{tag}.json"), "w") as f:
    json.dump(rows, f, indent=1)
  assert not bad, bad
  for k in must_reach_1pct:
    assert not rows[k]["mc_limited"], ("not enough chains for a 1 % statement", k, rows[k])
    assert abs(rows[k]["rel"]) < 0.01, (k, rows[k])
  return rows


def test_cfg2_long_run_posterior_matches_oracle_within_one_percent_or_mc_error():
  """BASELINE cfg2: T=1000, P=11, LocalLinearTrend + spike-and-slab, W=112, S=1000.  400 device
  chains (ids 0..399, one launch) and 200 float64 oracle chains (ids 0..199; ~0.2 s each, on the
  host cores).  Two comparisons:
    * INDEPENDENT replicates -- device chains 200..399 against oracle chains 0..199 (disjoint
      random streams).  200 chains a side are what brings 4 s.e. of sigma_level (a slowly mixing
      scale of ~0.01; 128 a side leave 1.17 %) below 1 % OF ITS OWN VALUE: sigma_obs, sigma_level
      and sigma_slope must meet the 1 % figure outright (`must_reach_1pct`); the interval ends of
      the slowly mixing scales and the
      post-period average of the predictive draws (Monte-Carlo error of a 300-step forecast) are
      held to 4 s.e. and listed with that band in the table -- and the prediction is pinned
      through the path of `posterior_means` instead (`_compare_paths`: 1 % of the outcome's s.d.
      over the pre-period);
    * the SAME chain ids 0..7 on both sides: float32 kernel vs float64 oracle on one random
      stream, i.e. pure arithmetic drift over 1112 iterations."""
  T, p, W, S, C, CO = 1000, 10, 112, 1000, 400, 200
  seed = (0, 20240927)
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 2024)
  spec = orc.default_spec(y, mask, X, has_slope=True)
  post = slice(int(0.7 * T), T)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=W, num_results=S, num_chains=C,
                            seed=seed)
  g = _native.fit_gibbs(pb, y[None], mask[None], X[None], None, _native.make_params([spec]),
                        want=("observation_noise_scale", "level_scale", "slope_scale", "weights",
                              "posterior_trajectories", "posterior_means"))
  dev = [_chain_summaries(g["observation_noise_scale"][0, c], g["level_scale"][0, c],
                          g["slope_scale"][0, c], g["weights"][0, c],
                          g["posterior_trajectories"][0, c][:, post].mean(axis=1),
                          g["posterior_means"][0, c][post].mean()) for c in range(C)]
  with concurrent.futures.ProcessPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:
    ora = list(ex.map(_oracle_chain, [(y, mask, X, spec, S, W, seed, c, post) for c in range(CO)]))
  rows = _compare("cfg2", dev[CO:], ora,
                  must_reach_1pct=("sigma_obs.mean", "sigma_level.mean", "sigma_slope.mean"))
  # the prediction, path-wise, through posterior_means: device chains 200..399 vs oracle chains 0..199
  _compare_paths("cfg2", g["posterior_means"][0, CO:], np.stack([c["_pred_mean_path"] for c in ora]),
                 pre_end=int(0.7 * T), outcome_sd=float(np.nanstd(np.where(mask, np.nan, y))))
  # float32 drift over 1112 iterations: the SAME chains (ids 0..7) on both sides
  same = _compare("cfg2_same_chains", dev[:8], ora[:8])
  assert abs(same["sigma_obs.mean"]["rel"]) < 1e-3
  assert rows["sigma_obs.mean"]["oracle"] > 0


def test_cfg4_long_run_posterior_matches_oracle_within_one_percent_or_mc_error():
  """BASELINE cfg4: T=10000, 50 covariates (P=51) + Seasonal(num_seasons=7), time-parallel
  kernel over its HBM workspace; W=112, S=400 -- NOT the configuration's 1000 draws: the oracle
  costs ~4 ms per iteration, ~2 s per chain, run in parallel on the host's cores.  256 device chains
  in one launch against 128 oracle chains (ids 0..127): device chains 128..255 are INDEPENDENT
  replicates of the oracle's, device chains 0..127 the same random streams (float32-vs-float64 drift
  of one stream).  128 chains a side are what brings 4 s.e. of sigma_level (0.004, slowly mixing)
  below 1 % of its value: sigma_obs AND sigma_level must meet the 1 % figure outright; sigma_drift
  (0.001) is held to 4 s.e., listed with that band in `_summary.mc_limited`.  (The predictive draws
  are not downloaded at this chain count -- 4 GB --: the prediction is compared path-wise through
  `posterior_means`, which is what pins it.)"""
  T, p, W, S, C, CO = 10000, 50, 112, 400, 256, 128
  seed = (3, 1)
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  t = np.arange(T)
  pattern = np.array([0.8, 0.3, -0.2, -0.6, -0.4, 0.0, 0.1])
  y = np.where(mask, y, y + pattern[t % 7])
  spec = orc.default_spec(y, mask, X, seasons=((7, 1),))
  post = slice(int(0.7 * T), T)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_seasons=(7,), num_warmup=W,
                            num_results=S, num_chains=C, seed=seed)
  sc = np.stack(spec["season_change"])
  g = _native.fit_gibbs(pb, y[None], mask[None], X[None], sc, _native.make_params([spec]),
                        want=("observation_noise_scale", "level_scale", "slope_scale", "weights",
                              "posterior_means", "seasonal_drift_scales"))
  dev = [_chain_summaries(g["observation_noise_scale"][0, c], g["level_scale"][0, c],
                          g["slope_scale"][0, c], g["weights"][0, c], np.zeros(S),
                          g["posterior_means"][0, c][post].mean(),
                          g["seasonal_drift_scales"][0, c]) for c in range(C)]
  with concurrent.futures.ProcessPoolExecutor(max_workers=min(CO, os.cpu_count() or 1)) as ex:
    ora = list(ex.map(_oracle_chain, [(y, mask, X, spec, S, W, seed, c, post) for c in range(CO)]))
  _compare("cfg4", dev[CO:], ora, must_reach_1pct=("sigma_obs.mean", "sigma_level.mean"))   # independent replicates
  _compare("cfg4_same_chains", dev[:CO], ora)                              # drift of the same streams
  _compare_paths("cfg4", g["posterior_means"][0, CO:], np.stack([c["_pred_mean_path"] for c in ora]),
                 pre_end=int(0.7 * T), outcome_sd=float(np.nanstd(np.where(mask, np.nan, y))))


def test_general_seasonal_long_run_posterior_matches_oracle_within_one_percent_or_mc_error():
  """The reference's 4+7+6-season model (causalimpact_lib_test.py:738-752) at cfg4's size -- T=10000,
  50 covariates -- on the time-parallel cluster kernel of round 5 (csrc/ci_seasonal_tp.h: 128 chunks
  of 80 steps, wave-cooperative float32 scan elements): W=100, S=300 (the oracle costs ~10 ms per
  iteration: ~4 s per chain, in parallel on the host).  192 device chains in one launch (one workgroup
  per chain at that count) against 96 oracle chains: device chains 96..191 are independent replicates,
  device chains 0..95 the oracle's own random streams (the float32-vs-float64 drift of one stream over
  400 iterations).  The predictive draws are not downloaded at this chain count; the prediction is
  compared path-wise through `posterior_means`."""
  T, p, W, S, C, CO = 10000, 50, 100, 300, 192, 96
  seed = (3, 1)
  seasons = ((4, (2, 1, 1, 1)), (7, 1), (6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1))))
  from causalimpact import _model
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  t = np.arange(T)
  pattern = np.array([0.8, 0.3, -0.2, -0.6, -0.4, 0.0, 0.1])
  y = np.where(mask, y, y + pattern[t % 7])
  spec = orc.default_spec(y, mask, X, seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  post = slice(int(0.7 * T), T)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_seasons=counts, num_warmup=W,
                            num_results=S, num_chains=C, seed=seed)
  sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
  assert "gibbs_seasonal_tp_kernel" in sess.kernel_name()
  sess.run()
  g = sess.fetch(["observation_noise_scale", "level_scale", "slope_scale", "weights",
                  "posterior_means", "seasonal_drift_scales"])
  sess.close()
  dev = [_chain_summaries(g["observation_noise_scale"][0, c], g["level_scale"][0, c],
                          g["slope_scale"][0, c], g["weights"][0, c], np.zeros(S),
                          g["posterior_means"][0, c][post].mean(),
                          g["seasonal_drift_scales"][0, c]) for c in range(C)]
  with concurrent.futures.ProcessPoolExecutor(max_workers=min(CO, os.cpu_count() or 1)) as ex:
    ora = list(ex.map(_oracle_chain, [(y, mask, X, spec, S, W, seed, c, post) for c in range(CO)]))
  _compare("general_seasonal", dev[CO:], ora, must_reach_1pct=("sigma_obs.mean", "sigma_level.mean"))   # independent replicates
  _compare("general_seasonal_same_chains", dev[:CO], ora)                            # drift of the same streams
  _compare_paths("general_seasonal", g["posterior_means"][0, CO:],
                 np.stack([c["_pred_mean_path"] for c in ora]),
                 pre_end=int(0.7 * T), outcome_sd=float(np.nanstd(np.where(mask, np.nan, y))))


def test_hundred_covariates_long_run_posterior_matches_oracle_within_one_percent_or_mc_error():
  """T=1000, 100 covariates (P=101) on the BIGP build of the trend + one-block kernel (round 6:
  csrc/ci_wide.h + ci_bigp.h -- packed triangular sweeps, flat entry loops): W=50, S=350.  256 device
  chains in one launch against 128 oracle chains (0.55 ms per iteration each; 128 a side are what
  brings 4 s.e. of sigma_obs below 1 %): device chains 128..255 are independent replicates, device
  chains 0..127 the oracle's own random streams (float32 latents and float64 regression draw
  against the float64 oracle over 400 iterations)."""
  T, p, W, S, C, CO = 1000, 100, 50, 350, 256, 128
  seed = (5, 2)
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  spec = orc.default_spec(y, mask, X)
  post = slice(int(0.7 * T), T)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S, num_chains=C, seed=seed)
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  assert "bigp" in sess.kernel_name()
  sess.run()
  g = sess.fetch(["observation_noise_scale", "level_scale", "slope_scale", "weights",
                  "posterior_trajectories", "posterior_means"])
  sess.close()
  dev = [_chain_summaries(g["observation_noise_scale"][0, c], g["level_scale"][0, c],
                          g["slope_scale"][0, c], g["weights"][0, c],
                          g["posterior_trajectories"][0, c][:, post].mean(axis=1),
                          g["posterior_means"][0, c][post].mean()) for c in range(C)]
  with concurrent.futures.ProcessPoolExecutor(max_workers=min(64, os.cpu_count() or 1)) as ex:
    ora = list(ex.map(_oracle_chain, [(y, mask, X, spec, S, W, seed, c, post) for c in range(CO)]))
  _compare("p101", dev[CO:], ora, must_reach_1pct=("sigma_obs.mean",))      # independent replicates
  same = _compare("p101_same_chains", dev[:CO], ora)                        # drift of the same streams
  assert abs(same["sigma_obs.mean"]["rel"]) < 2e-3
  _compare_paths("p101", g["posterior_means"][0, CO:], np.stack([c["_pred_mean_path"] for c in ora]),
                 pre_end=int(0.7 * T), outcome_sd=float(np.nanstd(np.where(mask, np.nan, y))))
