"""Many independent series in one launch (BASELINE "batch of 512 independent series, T=500,
5 covariates").

`fit_causalimpact_batch` is the batched form of `fit_causalimpact` for B series that share an
index and the pre/post periods (the same calendar for many geos / products):
  * data preparation (data.py:105-137, standardize.py:42-55) is vectorised over series in numpy
    -- no per-series pandas;
  * one kernel launch per device runs all B x chains Gibbs fits (one workgroup per
    (series, chain)); series are sharded over `InferenceOptions.devices`, no collective;
  * the T x draws post-processing of every series runs on the device that holds the draws
    (csrc/ci_summary.h); only order statistics and per-draw totals come back;
  * the (B*2) x 15 summary table is assembled with numpy; a per-series `CausalImpactAnalysis`
    (14-column `series` frame) is built lazily on indexing.
Random streams are keyed by (series position in the batch, chain): Monte-Carlo errors are
independent across series; `shared_streams=True` keys them by chain only, which makes series b
of a batch equal `fit_causalimpact` on series b alone with the same seed, draw for draw.
"""
from __future__ import annotations

import concurrent.futures
import dataclasses
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd

from causalimpact import _model
from causalimpact import _native
from causalimpact import causalimpact_lib as lib
from causalimpact import data as cid
from causalimpact import indices
from causalimpact import standardize


@dataclasses.dataclass
class PreparedBatch:
  """What the sampler consumes for B series (the batched CausalImpactData)."""
  values: np.ndarray          # [B, T_all, 1+p] raw data
  index: pd.Index             # [T_all]
  pre_period: Tuple[Any, Any]
  post_period: Tuple[Any, Any]
  standardize_data: bool
  model_rows: np.ndarray      # positions (into index) of the T steps handed to the sampler
  num_pre: int
  y: np.ndarray               # [B, T] standardised outcome, NaN where missing / forecast
  mask: np.ndarray            # [B, T] bool
  design: Optional[np.ndarray]   # [B, T, P] standardised covariates + intercept, or None
  outcome_mean: np.ndarray    # [B] pre-period mean of the outcome (0 if not standardised)
  outcome_sd: np.ndarray      # [B] pre-period sd (ddof=1)     (1 if not standardised)


def prepare_batch(values: np.ndarray, index: pd.Index, pre_period, post_period,
                  standardize_data: bool = True) -> PreparedBatch:
  """Vectorised CausalImpactData.__init__ (data.py:77-137) for [B, T_all, 1+p] values whose
  first column is the outcome."""
  values = np.asarray(values, dtype=np.float64)
  if values.ndim != 3:
    raise ValueError("`values` must be [num_series, num_timesteps, 1 + num_covariates]")
  B, T_all, ncol = values.shape
  index = pd.Index(index)
  if len(index) != T_all:
    raise ValueError("`index` must have one entry per timestep")
  probe = pd.DataFrame({"y": np.zeros(T_all)}, index=index)
  pre, post = indices.parse_and_validate_date_data(data=probe, pre_period=pre_period,
                                                  post_period=post_period)
  outcome = values[:, :, 0]
  with np.errstate(invalid="ignore"):
    if np.any(np.nanstd(outcome, axis=1) == 0):
      raise ValueError("Input response cannot be constant.")
  if np.any(np.sum(~np.isnan(outcome), axis=1) < 3):
    raise ValueError("Input data must have at least 3 observations.")
  if ncol > 1 and np.isnan(values[:, :, 1:]).any():
    raise ValueError("Input data cannot have any missing values.")
  in_pre = np.asarray((index >= pre[0]) & (index <= pre[1]))
  after = np.asarray(index > pre[1])
  rows = np.concatenate([np.flatnonzero(in_pre), np.flatnonzero(after)])
  n_pre = int(in_pre.sum())
  model = values[:, rows, :]                               # [B, T, 1+p]
  if standardize_data:
    scaled, mu, sd = standardize.standardize_batch(model, n_pre)
    o_mu, o_sd = mu[:, 0].copy(), sd[:, 0].copy()
  else:
    scaled = model
    o_mu, o_sd = np.zeros(B), np.ones(B)
  y = scaled[:, :, 0].copy()
  y[:, n_pre:] = np.nan
  mask = np.isnan(y)
  design = None
  if ncol > 1:
    design = np.concatenate([scaled[:, :, 1:], np.ones((B, len(rows), 1))], axis=2)
  return PreparedBatch(values=values, index=index, pre_period=pre, post_period=post,
                       standardize_data=standardize_data, model_rows=rows, num_pre=n_pre, y=y,
                       mask=mask, design=design, outcome_mean=o_mu, outcome_sd=o_sd)


class CausalImpactBatchAnalysis:
  """Results for B series.  `summary`: DataFrame indexed by (series, average|cumulative) with the
  reference's 15 summary columns; `analysis[b]` / iteration: per-series CausalImpactAnalysis
  (the `series` frame is assembled on first access); `diagnostics`: split-R-hat / ESS per series
  when more than one chain was run."""

  def __init__(self, prepared, names, alpha, posterior_means, device_summary, ranks, columns,
               diagnostic_draws):
    self._prep, self._names, self.alpha = prepared, list(names), alpha
    self._means, self._dsum, self._ranks, self._columns = posterior_means, device_summary, ranks, columns
    # {key: [B, chains, draws]} of the scalars the diagnostics rank, or None for one chain.  The
    # diagnostics themselves (three rank / FFT passes per key per series) are computed on access:
    # for thousands of series they would otherwise cost more host time than the device fit.
    self._diag_draws = diagnostic_draws
    self._diag: Dict[int, Dict] = {}
    self._cache: Dict[int, lib.CausalImpactAnalysis] = {}
    self.summary = self._build_summary()

  def diagnostics_of(self, b: int):
    """{"split_rhat" | "ess_bulk" | "ess_tail": {key: value}} of series b (None for one chain)."""
    if self._diag_draws is None:
      return None
    b = range(len(self))[b]
    if b not in self._diag:
      d = {k: v[b] for k, v in self._diag_draws.items()}
      self._diag[b] = {
          "split_rhat": {k: lib.split_rhat(v) for k, v in d.items()},
          "ess_bulk": {k: lib.effective_sample_size(v, "bulk") for k, v in d.items()},
          "ess_tail": {k: lib.effective_sample_size(v, "tail") for k, v in d.items()}}
    return self._diag[b]

  @property
  def diagnostics(self):
    """{"split_rhat" | "ess_bulk" | "ess_tail": [per-series {key: value}]} for ALL series (computed
    now, O(B) host work); None when a single chain was run."""
    if self._diag_draws is None:
      return None
    per = [self.diagnostics_of(b) for b in range(len(self))]
    return {name: [p[name] for p in per] for name in ("split_rhat", "ess_bulk", "ess_tail")}

  def __len__(self):
    return len(self._names)

  def _request(self, b: int) -> Dict:
    p = self._prep
    idx = p.index[p.model_rows]
    in_post = np.asarray((idx >= p.post_period[0]) & (idx <= p.post_period[1]))
    obs = p.values[b, p.model_rows, 0].copy()
    obs[p.num_pre:][~in_post[p.num_pre:]] = np.nan        # gap / tail: predictions only
    flags = (~np.asarray(idx < p.post_period[0])).astype(np.uint8) | (in_post.astype(np.uint8) << 1)
    return dict(scale=float(p.outcome_sd[b]) if p.standardize_data else 1.0,
                shift=float(p.outcome_mean[b]) if p.standardize_data else 0.0,
                observed=obs, flags=flags, ranks=self._ranks,
                quantiles=(self.alpha / 2.0, 1.0 - self.alpha / 2.0))

  def _build_summary(self) -> pd.DataFrame:
    """The reference's 15 summary columns (causalimpact_lib.py:934-1093) for every series at
    once: the same numpy reductions as `_summary_rows`, along axis 1 of [B, draws] arrays."""
    p, B = self._prep, len(self)
    quantiles = (self.alpha / 2.0, 1.0 - self.alpha / 2.0)
    rq = self._request(0)
    win = (rq["flags"] & 2) != 0
    n_win = int(win.sum())
    idx = p.index[p.model_rows]
    in_post = np.asarray((idx >= p.post_period[0]) & (idx <= p.post_period[1]))
    obs = p.values[:, p.model_rows, 0].copy()
    obs[:, p.num_pre:][:, ~in_post[p.num_pre:]] = np.nan
    obs_w = obs[:, win]                                                    # [B, T_w]
    scale = p.outcome_sd if p.standardize_data else np.ones(B)
    shift = p.outcome_mean if p.standardize_data else np.zeros(B)
    post_mean = (self._means.astype(np.float64) * scale[:, None] + shift[:, None])[:, win]
    pred_sum, point_sum = self._dsum["per_draw"][:, 0], self._dsum["per_draw"][:, 1]   # [B, N]
    n_obs = np.sum(~np.isnan(obs_w), axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
      obs_mean, obs_sum = np.nanmean(obs_w, axis=1), np.nansum(obs_w, axis=1)
      pred_mean = pred_sum / n_win
      point_mean = np.where(n_obs[:, None] > 0, point_sum / np.maximum(n_obs, 1)[:, None], np.nan)
      rel = obs_sum[:, None] / pred_sum - 1.0

      # The bands interpolate order statistics of the per-draw totals; the device returned
      # those (per_draw_order), and every banded quantity is a monotone map of the totals, so
      # its order statistics are the mapped ones -- no [B, draws] sort on the host.
      ranks = list(self._ranks)
      N = pred_sum.shape[1]
      order = self._dsum.get("per_draw_order")
      if order is None:
        order = np.sort(self._dsum["per_draw"], axis=2)[:, :, ranks]
      (lo_a, hi_a, g_a), (lo_b, hi_b, g_b) = lib._quantile_ranks(N, quantiles)   # pylint: disable=protected-access
      lerp = lib._lerp_order_stats                                               # pylint: disable=protected-access

      def band_of(by_rank):
        return np.stack([lerp(by_rank, lo_a, hi_a, g_a), lerp(by_rank, lo_b, hi_b, g_b)])

      pred_o = {r: order[:, 0, i] for i, r in enumerate(ranks)}
      point_o = {r: order[:, 1, i] for i, r in enumerate(ranks)}
      need = (lo_a, hi_a, lo_b, hi_b)
      band_pred_sum, band_point_sum = band_of(pred_o), band_of(point_o)
      band_pred_mean = band_of({k: pred_o[k] / n_win for k in need})
      band_point_mean = band_of({k: np.where(n_obs > 0, point_o[k] / np.maximum(n_obs, 1), np.nan)
                                 for k in need})
      # rel = obs_sum / pred_sum - 1 is monotone in pred_sum on either side of zero: decreasing
      # when obs_sum > 0 (its k-th smallest comes from the (N-1-k)-th smallest total), else
      # increasing.  Series whose totals straddle zero take the sort.
      down = obs_sum > 0
      band_rel = band_of({k: np.where(down, obs_sum / pred_o[N - 1 - k], obs_sum / pred_o[k]) - 1.0
                          for k in need})
      straddle = ~((pred_sum.min(axis=1) > 0) | (pred_sum.max(axis=1) < 0))
      if straddle.any():
        band_rel[:, straddle] = np.quantile(rel[straddle], quantiles, axis=1)

      def sd(x):
        return np.std(x, axis=1, ddof=1)

      avg_pred, cum_pred = post_mean.mean(axis=1), post_mean.sum(axis=1)
      cols = {
          "actual": (obs_mean, obs_sum),
          "predicted": (avg_pred, cum_pred),
          "predicted_lower": (band_pred_mean[0], band_pred_sum[0]),
          "predicted_upper": (band_pred_mean[1], band_pred_sum[1]),
          "predicted_sd": (sd(pred_mean), sd(pred_sum)),
          "abs_effect": (obs_mean - avg_pred, obs_sum - cum_pred),
          "abs_effect_lower": (band_point_mean[0], band_point_sum[0]),
          "abs_effect_upper": (band_point_mean[1], band_point_sum[1]),
          "abs_effect_sd": (sd(point_mean), sd(point_sum)),
          "rel_effect": (rel.mean(axis=1),) * 2,
          "rel_effect_lower": (band_rel[0],) * 2,
          "rel_effect_upper": (band_rel[1],) * 2,
          "rel_effect_sd": (sd(rel),) * 2,
      }
    pool_le = ((obs_sum[:, None] <= pred_sum).sum(axis=1) + 1) / (pred_sum.shape[1] + 1)
    pool_ge = ((obs_sum[:, None] >= pred_sum).sum(axis=1) + 1) / (pred_sum.shape[1] + 1)
    p_value = np.minimum(pool_le, pool_ge)
    data = {k: np.stack(v, axis=1).reshape(-1) for k, v in cols.items()}      # (b, avg|cum) order
    data["p_value"] = np.repeat(p_value, 2)
    data["alpha"] = np.full(2 * B, self.alpha)
    index = pd.MultiIndex.from_product([self._names, ["average", "cumulative"]], names=["series", None])
    return pd.DataFrame(data, index=index)

  def __getitem__(self, b: int) -> lib.CausalImpactAnalysis:
    b = range(len(self))[b]
    if b not in self._cache:
      p = self._prep
      df = pd.DataFrame(p.values[b], index=p.index, columns=self._columns)
      ci_data = cid.CausalImpactData(df, p.pre_period, p.post_period,
                                     standardize_data=p.standardize_data)
      dsum = {k: v[b] for k, v in self._dsum.items()}
      rq = lib._device_summary_request(ci_data, self.alpha)   # pylint: disable=protected-access
      rq["ranks"] = self._ranks
      series, summary = lib._compute_impact_device(            # pylint: disable=protected-access
          self._means[b], dsum, rq, ci_data, self.alpha)
      self._cache[b] = lib.CausalImpactAnalysis(series, summary, None, self.diagnostics_of(b))
    return self._cache[b]

  def __iter__(self):
    return (self[b] for b in range(len(self)))


class PerSeriesBatchAnalysis(CausalImpactBatchAnalysis):
  """The same container over analyses that were fitted one series at a time (the routes the
  one-launch path does not have: float64 compute, raw-scale outcomes)."""

  def __init__(self, names, alpha, analyses):   # pylint: disable=super-init-not-called
    self._names, self.alpha = list(names), alpha
    self._cache = dict(enumerate(analyses))
    self._diag_draws = None
    self.summary = pd.concat([a.summary for a in analyses], keys=self._names, names=["series", None])

  def diagnostics_of(self, b: int):
    return self._cache[range(len(self))[b]].diagnostics

  @property
  def diagnostics(self):
    per = [self.diagnostics_of(b) for b in range(len(self))]
    if any(p is None for p in per):
      return None
    return {name: [p[name] for p in per] for name in ("split_rhat", "ess_bulk", "ess_tail")}

  def __getitem__(self, b: int) -> lib.CausalImpactAnalysis:
    return self._cache[range(len(self))[b]]


def fit_causalimpact_batch(data: Union[Sequence[pd.DataFrame], np.ndarray],
                           pre_period, post_period, alpha: float = 0.05, seed=None,
                           data_options: Optional[lib.DataOptions] = None,
                           model_options: Optional[lib.ModelOptions] = None,
                           inference_options: Optional[lib.InferenceOptions] = None,
                           index: Optional[pd.Index] = None,
                           names: Optional[Sequence[Any]] = None,
                           shared_streams: bool = False) -> CausalImpactBatchAnalysis:
  """`fit_causalimpact` for B series at once.

  data: a sequence of DataFrames with identical index and column layout (outcome first, or
  `DataOptions.outcome_column`), or an array [B, T, 1 + covariates] (outcome first) with
  `index` (default: 0..T-1).  Other arguments as `fit_causalimpact`.  Latent-state draws are not
  downloaded (B x chains x draws x T values); the per-series frames and the summary table are.

  Random streams: series b draws from streams keyed by (its position b in the batch, chain), so
  the Monte-Carlo errors of different series are independent (pooling effects over geos averages
  them out) and the result does not depend on how the batch is split over devices.  Series 0 of
  a batch equals `fit_causalimpact` on that series alone with the same seed.
  `shared_streams=True` keys the streams by chain only: EVERY series then reproduces its
  single-series fit draw for draw, at the price of perfectly correlated Monte-Carlo errors.
  "Equals" is bit for bit on every route: the kernel a series runs on is a function of its model
  and length alone (trend models, trend + one block of 2-7 seasons, and the general seasonal /
  more-than-52-covariate routes alike), never of the batch size or the device's CU count; the
  launch size only decides how many workgroups share one chain's work, which does not change the
  arithmetic (tests/test_gpu_gibbs.py, including a seasonal batch with more chains than CUs).
  `InferenceOptions.kernel_flags` (e.g. `_native.FLAG_SEQUENTIAL_SEASONAL`: 1.5-1.7x the throughput
  for batches of hundreds of short multi-block series) applies to the batch as to a single fit: give
  it to both when comparing them.

  `DataOptions.dtype=float64` and `standardize_data=False` batches are NOT one launch: they are
  fitted series by series on the single-series routes (float64 kernels / exact internal
  conditioning), i.e. B sequential fits on one device -- B times the cost of one fit, and
  `inference_options.devices` is not used to shard them.  Their streams are keyed per series in
  the same way (series b on the key of series id b) unless `shared_streams=True`.
  """
  data_options = data_options or lib.DataOptions()
  model_options = model_options or lib.ModelOptions()
  inference_options = inference_options or lib.InferenceOptions()
  if not 0 < alpha < 1:
    raise ValueError("`alpha` must be between 0 and 1.")
  if inference_options.sampler != "gibbs":
    raise NotImplementedError("batched fits use the Gibbs sampler")
  if isinstance(data, np.ndarray):
    values = np.asarray(data, np.float64)
    index = pd.RangeIndex(values.shape[1]) if index is None else pd.Index(index)
    columns = ["y"] + [f"x{j}" for j in range(values.shape[2] - 1)]
  else:
    frames = [pd.DataFrame(d) for d in data]
    if not frames:
      raise ValueError("`data` is empty")
    first = frames[0]
    oc = data_options.outcome_column if data_options.outcome_column is not None else first.columns[0]
    columns = [oc] + [c for c in first.columns if c != oc]
    for f in frames:
      if not f.index.equals(first.index) or list(f.columns) != list(first.columns):
        raise ValueError("all series of a batch must share the index and the columns")
    values = np.stack([f[columns].to_numpy(dtype=np.float64) for f in frames])
    index = first.index
  B = values.shape[0]
  names = list(range(B)) if names is None else list(names)
  if (cid._as_numpy_dtype(data_options.dtype) == np.float64  # pylint: disable=protected-access
      or not data_options.standardize_data):
    # float64 compute (csrc/ci_gibbs64.h) and raw-scale outcomes (their per-series internal
    # conditioning, causalimpact_lib._internal_conditioning) exist on the single-series path: the
    # batch is fitted series by series there -- same container, same summary table, every series
    # keyed like the one-launch path (series b on the Philox key of series id b,
    # ci_series_stream_key, so the Monte-Carlo errors of different series are independent; with
    # shared_streams=True every series equals `fit_causalimpact` on it alone with this seed).
    # Not the one-launch path: B sequential fits on one device (see the docstring).
    opts = dataclasses.replace(data_options, outcome_column=columns[0])
    analyses = []
    base_seed = lib._sanitize_seed(seed)   # pylint: disable=protected-access
    for b in range(B):
      seed_b = base_seed if shared_streams else _native.series_stream_key(base_seed, b)
      one = lib.fit_causalimpact(pd.DataFrame(values[b], index=index, columns=columns), pre_period,
                                 post_period, alpha=alpha, seed=seed_b, data_options=opts,
                                 model_options=model_options, inference_options=inference_options)
      analyses.append(dataclasses.replace(one, posterior_samples=None))   # (draws are not kept)
    return PerSeriesBatchAnalysis(names, alpha, analyses)
  prep = prepare_batch(values, index, pre_period, post_period, data_options.standardize_data)
  T = prep.y.shape[1]
  P = 0 if prep.design is None else prep.design.shape[2]
  num_seasons, season_change = _model.expand_seasons(model_options.seasons, T)
  # the sampler sees the outcome in DataOptions.dtype (data.py:121-128), priors included
  y_model = prep.y.astype(cid._as_numpy_dtype(data_options.dtype)).astype(np.float64)  # pylint: disable=protected-access
  with np.errstate(invalid="ignore"):
    pre_sd = np.nanstd(y_model[:, :prep.num_pre], axis=1, ddof=1)
  params = [_model.series_params(y_model[b], prep.mask[b],
                                 None if prep.design is None else prep.design[b],
                                 prior_level_sd=model_options.prior_level_sd,
                                 num_seasonal_blocks=len(num_seasons),
                                 has_slope=model_options.local_linear_trend,
                                 outcome_sd=float(pre_sd[b])) for b in range(B)]
  seed_pair = lib._sanitize_seed(seed)   # pylint: disable=protected-access
  C, S = inference_options.num_chains, inference_options.num_results
  ranks = lib._summary_ranks(C * S, (alpha / 2.0, 1.0 - alpha / 2.0))   # pylint: disable=protected-access
  idx = index[prep.model_rows]
  in_post = np.asarray((idx >= prep.post_period[0]) & (idx <= prep.post_period[1]))
  flags = (~np.asarray(idx < prep.post_period[0])).astype(np.uint8) | (in_post.astype(np.uint8) << 1)
  observed = values[:, prep.model_rows, 0].copy()
  observed[:, prep.num_pre:][:, ~in_post[prep.num_pre:]] = np.nan
  devs = list(inference_options.devices) if inference_options.devices else [0]
  shards = [s for s in np.array_split(np.arange(B), len(devs)) if len(s)]

  def run(dev, ids):
    pb = _native.make_problem(T=T, P=P, has_slope=model_options.local_linear_trend,
                              num_seasons=num_seasons, num_warmup=inference_options.num_warmup_steps,
                              num_results=S, num_chains=C, num_series=len(ids), seed=seed_pair,
                              device=dev, series_offset=int(ids[0]),
                              flags=int(getattr(inference_options, "kernel_flags", 0)) |
                              (_native.FLAG_SHARED_SERIES_STREAMS if shared_streams else 0))
    sess = _native.Session(pb, y_model[ids], prep.mask[ids],
                           None if prep.design is None else prep.design[ids], season_change,
                           _native.make_params([params[b] for b in ids]))
    try:
      sess.run()
      out = sess.fetch(["posterior_means", "observation_noise_scale", "level_scale"])
      dsum = sess.summarize(prep.outcome_sd[ids] if prep.standardize_data else 1.0,
                            prep.outcome_mean[ids] if prep.standardize_data else 0.0,
                            observed[ids], flags, ranks)
      if len(ids) == 1:
        dsum = {k: v[None] for k, v in dsum.items()}
    finally:
      sess.close()
    return out, dsum

  if len(shards) == 1:
    # one device: no worker thread (a fresh host thread pays the runtime's per-thread set-up,
    # ~30 ms, more than the fit of 512 series takes)
    results = [run(devs[0], shards[0])]
  else:
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(shards)) as pool:
      results = list(pool.map(lambda a: run(*a), zip(devs, shards)))
  means = np.concatenate([r[0]["posterior_means"].mean(axis=1) for r in results], axis=0)   # [B, T]
  dsum = {k: np.concatenate([r[1][k] for r in results], axis=0) for k in results[0][1]}
  diag_draws = None
  if C > 1:
    keys = ("observation_noise_scale", "level_scale")
    diag_draws = {k: np.concatenate([r[0][k] for r in results], axis=0) for k in keys}   # [B, C, S]
  return CausalImpactBatchAnalysis(prep, names, alpha, means, dsum, ranks, columns, diag_draws)
