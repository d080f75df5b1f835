"""Batched API: vectorised data preparation == per-series CausalImpactData
(CPU); a batch launch == B separate fit_causalimpact calls (GPU)."""
import numpy as np
import pandas as pd
import pytest

import causalimpact as ci
from causalimpact import batch
from causalimpact import causalimpact_lib as lib
from causalimpact import data as cid
from causalimpact import _synthetic as syn


def _frames(B, T, p, seed=0, dates=True):
  idx = pd.date_range("2021-01-04", periods=T, freq="D") if dates else pd.RangeIndex(T)
  out = []
  for b in range(B):
    y, X = syn.make_raw_series(T, p, seed + b, effect=5.0 + b)
    df = pd.DataFrame(np.column_stack([y, X]), index=idx,
                      columns=["y"] + [f"x{j}" for j in range(p)])
    out.append(df)
  out[1].iloc[[3, 17], 0] = np.nan          # missing pre-period outcomes
  if p:
    out[2].iloc[:, 1] = 7.0                 # constant covariate: left unscaled
  return out


@pytest.mark.parametrize("standardize", [True, False])
def test_prepare_batch_equals_causal_impact_data(standardize):
  T, p = 90, 2
  frames = _frames(4, T, p)
  idx = frames[0].index
  pre, post = (idx[5], idx[59]), (idx[63], idx[84])       # rows before the pre-period, a gap, a tail
  values = np.stack([f.to_numpy(float) for f in frames])
  prep = batch.prepare_batch(values, idx, pre, post, standardize)
  for b, f in enumerate(frames):
    one = cid.CausalImpactData(f, pre, post, standardize_data=standardize, dtype=np.float64)
    n_pre = len(one.outcome_ts.time_series)
    assert prep.num_pre == n_pre and prep.y.shape[1] == n_pre + one.num_steps_forecast
    np.testing.assert_allclose(prep.y[b, :n_pre], one.outcome_ts.time_series, rtol=1e-12,
                               equal_nan=True)
    assert prep.mask[b, n_pre:].all() and np.isnan(prep.y[b, n_pre:]).all()
    np.testing.assert_array_equal(prep.mask[b, :n_pre], one.outcome_ts.is_missing)
    np.testing.assert_allclose(prep.design[b], one.feature_ts.values, rtol=1e-12)
    if standardize:
      np.testing.assert_allclose(prep.outcome_mean[b], one.outcome_scaler.mean_, rtol=1e-13)
      np.testing.assert_allclose(prep.outcome_sd[b], one.outcome_scaler.stddev_, rtol=1e-13)
  assert list(idx[prep.model_rows]) == list(idx[5:])


def test_prepare_batch_rejects_what_the_reference_rejects():
  T = 40
  v = np.random.default_rng(0).normal(size=(2, T, 2))
  idx = pd.RangeIndex(T)
  bad = v.copy(); bad[1, :, 0] = 3.0
  with pytest.raises(ValueError, match="cannot be constant"):
    batch.prepare_batch(bad, idx, (0, 19), (20, 39))
  bad = v.copy(); bad[0, 4, 1] = np.nan
  with pytest.raises(ValueError, match="cannot have any missing values"):
    batch.prepare_batch(bad, idx, (0, 19), (20, 39))
  with pytest.raises(ValueError):
    batch.prepare_batch(v, idx, (0, 25), (20, 39))           # overlapping periods


@pytest.mark.gpu
@pytest.mark.parametrize("p,dates", [(2, True), (0, False)])
def test_batch_fit_equals_separate_fits(p, dates):
  T, B = 100, 5
  frames = _frames(B, T, p, dates=dates)
  idx = frames[0].index
  pre, post = (idx[0], idx[69]), (idx[72], idx[95])
  opts = ci.InferenceOptions(num_results=150, num_chains=2)
  got = ci.fit_causalimpact_batch(frames, pre, post, alpha=0.1, seed=5, inference_options=opts,
                                  names=[f"geo{b}" for b in range(B)], shared_streams=True)
  assert len(got) == B and got.summary.shape == (2 * B, 15)
  for b, f in enumerate(frames):
    one = ci.fit_causalimpact(f, pre, post, alpha=0.1, seed=5, inference_options=opts)
    np.testing.assert_allclose(got.summary.loc[f"geo{b}"].to_numpy(float),
                               one.summary.to_numpy(float), rtol=2e-5, atol=1e-7)
    mine = got[b]
    assert list(mine.series.columns) == list(one.series.columns)
    num = [c for c in one.series.columns if one.series[c].dtype.kind == "f"]
    np.testing.assert_allclose(mine.series[num].to_numpy(float), one.series[num].to_numpy(float),
                               rtol=2e-5, atol=1e-6, equal_nan=True)
    assert ci.summary(mine) == ci.summary(one)
  assert set(got.diagnostics) == {"split_rhat", "ess_bulk", "ess_tail"}
  # default: per-series streams.  Series 0 still equals its single fit; the others are the same
  # posterior seen through different random numbers; identical series no longer give identical
  # draws; and the result does not depend on how the batch is split over launches.
  ind = ci.fit_causalimpact_batch(frames, pre, post, alpha=0.1, seed=5, inference_options=opts,
                                  names=[f"geo{b}" for b in range(B)])
  np.testing.assert_allclose(ind.summary.loc["geo0"].to_numpy(float),
                             got.summary.loc["geo0"].to_numpy(float), rtol=2e-5, atol=1e-7)
  a = ind.summary.loc["geo3"].to_numpy(float)
  b_ = got.summary.loc["geo3"].to_numpy(float)
  assert not np.allclose(a, b_, rtol=1e-6)
  assert a[0, 0] == b_[0, 0]                                          # actual
  sd = float(got.summary.loc["geo3"]["predicted_sd"]["average"])
  assert abs(a[0, 1] - b_[0, 1]) < 3.0 * sd                           # predicted: same posterior
  twins = ci.fit_causalimpact_batch([frames[1], frames[1]], pre, post, alpha=0.1, seed=5,
                                    inference_options=opts)
  assert not np.allclose(twins.summary.loc[0].to_numpy(float), twins.summary.loc[1].to_numpy(float),
                         rtol=1e-6)


@pytest.mark.gpu
def test_batch_accepts_an_array():
  T, B = 80, 3
  v = np.stack([f.to_numpy(float) for f in _frames(B, T, 1, seed=9)])
  got = ci.fit_causalimpact_batch(v, (0, 55), (56, 79), seed=1,
                                  inference_options=ci.InferenceOptions(num_results=50))
  assert got.summary.index.get_level_values(0).unique().tolist() == [0, 1, 2]
  assert np.isfinite(got.summary["abs_effect"].to_numpy()).all()
  assert got.diagnostics is None


@pytest.mark.gpu
def test_batch_with_a_weekly_seasonal_block_equals_separate_fits():
  T, B = 140, 3
  frames = _frames(B, T, 1, seed=30)
  idx = frames[0].index
  for b, f in enumerate(frames):
    f["y"] += 3.0 * np.sin(2 * np.pi * (np.arange(T) + b) / 7.0)
  pre, post = (idx[0], idx[97]), (idx[98], idx[-1])
  kw = dict(seed=8, inference_options=ci.InferenceOptions(num_results=120),
            model_options=ci.ModelOptions(seasons=[ci.Seasons(num_seasons=7)]))
  got = ci.fit_causalimpact_batch(frames, pre, post, shared_streams=True, **kw)
  for b, f in enumerate(frames):
    one = ci.fit_causalimpact(f, pre, post, **kw)
    np.testing.assert_allclose(got.summary.loc[b].to_numpy(float), one.summary.to_numpy(float),
                               rtol=2e-5, atol=1e-7)


@pytest.mark.gpu
def test_batch_summary_bands_equal_the_host_quantiles_also_when_totals_straddle_zero():
  """The batch summary interpolates device order statistics of the per-draw totals (monotone maps
  for the mean / relative-effect bands); series whose predicted totals change sign take the
  host sort.  Either way the frame equals the single-series one, which sorts on the host."""
  T, B = 90, 4
  frames = _frames(B, T, 1, seed=3)
  frames[1]["y"] -= frames[1]["y"].iloc[:60].mean()                    # predicted totals around 0
  frames[2] = -frames[2]                                                # negative totals
  idx = frames[0].index
  pre, post = (idx[0], idx[59]), (idx[60], idx[89])
  opts = ci.InferenceOptions(num_results=200, num_chains=2)
  got = ci.fit_causalimpact_batch(frames, pre, post, seed=11, inference_options=opts,
                                  shared_streams=True)
  totals = got._dsum["per_draw"][:, 0]                                 # pylint: disable=protected-access
  assert totals[1].min() < 0 < totals[1].max() and totals[2].max() < 0 < totals[0].min()
  for b, f in enumerate(frames):
    one = ci.fit_causalimpact(f, pre, post, seed=11, inference_options=opts)
    np.testing.assert_allclose(got.summary.loc[b].to_numpy(float), one.summary.to_numpy(float),
                               rtol=2e-5, atol=1e-7)


def test_summary_bands_from_order_statistics_equal_numpy_quantiles():
  """Host arithmetic of the batch summary (no GPU): bands interpolated from order statistics of
  the per-draw totals -- mapped through the monotone transforms to means and relative effects --
  equal numpy's quantiles of the transformed draws, for totals of either sign and for series whose
  totals straddle zero (those take the sort).  Without device order statistics the same code
  sorts on the host."""
  from causalimpact import causalimpact_lib as lib
  T, B, N, alpha = 60, 5, 401, 0.08
  frames = _frames(B, T, 1, seed=4)
  idx = frames[0].index
  pre, post = (idx[0], idx[39]), (idx[42], idx[57])
  values = np.stack([f.to_numpy(float) for f in frames])
  prep = batch.prepare_batch(values, idx, pre, post, True)
  rng = np.random.default_rng(0)
  Tm = prep.y.shape[1]
  means = rng.normal(size=(B, Tm)).astype(np.float32)
  per_draw = rng.normal(size=(B, 2, N)) * 20.0
  per_draw[0, 0] += 3000.0                      # positive totals
  per_draw[1, 0] -= 3000.0                      # negative totals
  per_draw[2, 0] += 5.0                         # straddling zero
  per_draw[3, 0] = np.abs(per_draw[3, 0]) + 1.0
  per_draw[4, 0] += 800.0
  quantiles = (alpha / 2.0, 1.0 - alpha / 2.0)
  ranks = lib._summary_ranks(N, quantiles)                          # pylint: disable=protected-access
  order = np.sort(per_draw, axis=2)[:, :, ranks]
  names = [f"s{b}" for b in range(B)]
  cols = list(frames[0].columns)
  with_order = batch.CausalImpactBatchAnalysis(prep, names, alpha, means,
                                               dict(per_draw=per_draw, per_draw_order=order),
                                               ranks, cols, None).summary
  host_sort = batch.CausalImpactBatchAnalysis(prep, names, alpha, means, dict(per_draw=per_draw),
                                              ranks, cols, None).summary
  pd.testing.assert_frame_equal(with_order, host_sort)
  # reference: numpy quantiles of the transformed draws
  idxm = prep.index[prep.model_rows]
  in_post = np.asarray((idxm >= post[0]) & (idxm <= post[1]))
  obs = prep.values[:, prep.model_rows, 0].copy()
  obs[:, prep.num_pre:][:, ~in_post[prep.num_pre:]] = np.nan
  win = in_post & ~np.asarray(idxm < post[0])
  obs_sum = np.nansum(obs[:, win], axis=1)
  n_win, n_obs = int(win.sum()), np.sum(~np.isnan(obs[:, win]), axis=1)
  pred_sum, point_sum = per_draw[:, 0], per_draw[:, 1]
  rel = obs_sum[:, None] / pred_sum - 1.0
  for b in range(B):
    row_avg, row_cum = with_order.loc[(names[b], "average")], with_order.loc[(names[b], "cumulative")]
    lo, hi = np.quantile(pred_sum[b], quantiles)
    assert (row_cum["predicted_lower"], row_cum["predicted_upper"]) == (lo, hi)
    lo, hi = np.quantile(pred_sum[b] / n_win, quantiles)
    assert (row_avg["predicted_lower"], row_avg["predicted_upper"]) == (lo, hi)
    lo, hi = np.quantile(point_sum[b] / n_obs[b], quantiles)
    assert (row_avg["abs_effect_lower"], row_avg["abs_effect_upper"]) == (lo, hi)
    lo, hi = np.quantile(rel[b], quantiles)
    np.testing.assert_allclose([row_avg["rel_effect_lower"], row_avg["rel_effect_upper"]], [lo, hi],
                               rtol=1e-13)


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(dtype=np.float64), dict(standardize_data=False)])
def test_batched_fit_takes_float64_and_raw_scale_through_the_single_series_route(opts):
  """float64 compute and raw-scale outcomes (internal conditioning) exist on the single-series
  path: the batched API fits such batches series by series -- same container, same table, every
  series equal to fit_causalimpact on it alone with the same seed."""
  frames = _frames(3, 80, 1)
  idx = frames[0].index
  pre, post = (idx[0], idx[55]), (idx[56], idx[79])
  do = lib.DataOptions(**opts)
  io = lib.InferenceOptions(num_results=60, num_chains=2)
  got = batch.fit_causalimpact_batch(frames, pre, post, seed=3, data_options=do, inference_options=io,
                                     names=["a", "b", "c"])
  assert isinstance(got, batch.CausalImpactBatchAnalysis) and len(got) == 3
  assert got.summary.shape == (6, 15)
  for b, name in enumerate("abc"):
    one = lib.fit_causalimpact(frames[b], pre, post, seed=3, data_options=do, inference_options=io)
    np.testing.assert_array_equal(got.summary.loc[name].to_numpy(float), one.summary.to_numpy(float))
    pd.testing.assert_frame_equal(got[b].series, one.series)
    assert got[b].posterior_samples is None
  assert set(got.diagnostics) == {"split_rhat", "ess_bulk", "ess_tail"}


def test_per_series_batch_container_assembles_the_summary_table():
  summ = pd.DataFrame({"actual": [1.0, 2.0], "alpha": [0.05, 0.05]}, index=["average", "cumulative"])
  one = lib.CausalImpactAnalysis(pd.DataFrame({"x": [1.0]}), summ, None, None)
  two = lib.CausalImpactAnalysis(pd.DataFrame({"x": [2.0]}), summ * 2, None, None)
  got = batch.PerSeriesBatchAnalysis(["u", "v"], 0.05, [one, two])
  assert len(got) == 2 and got[1] is two and [a for a in got] == [one, two]
  assert list(got.summary.index) == [("u", "average"), ("u", "cumulative"), ("v", "average"), ("v", "cumulative")]
  assert got.summary.loc["v"]["actual"]["cumulative"] == 4.0
  assert got.diagnostics is None and got.diagnostics_of(-1) is None


def test_lazy_diagnostics_pickle_as_a_plain_dict():
  import pickle
  calls = []
  m = lib._LazyMapping(lambda: (calls.append(1), {"split_rhat": {"a": 1.0}, "num_chains": 2})[1])
  back = pickle.loads(pickle.dumps(m))
  assert back == {"split_rhat": {"a": 1.0}, "num_chains": 2} and isinstance(back, dict)
  assert calls == [1]
