"""fit_causalimpact() on MI355X: the reference's public surface over the HIP Gibbs kernel.

Drop-in for /root/reference/causalimpact/causalimpact_lib.py: same function / dataclass
names, argument meaning, result frames and error behaviour.  What changed underneath:

  * `_train_causalimpact_sts` (reference :503-606) no longer builds a TFP model and traces a
    tf.function; it packs plain arrays and calls the C-ABI (`_native.fit_gibbs`), which runs
    every Gibbs iteration of every chain inside one persistent HIP kernel.
  * tensors in the results are numpy arrays (subclass with a `.numpy()` method so code written
    against the reference keeps working).
  * extensions BASELINE.json asks for, all optional: `InferenceOptions.num_chains`,
    `InferenceOptions.devices`, `ModelOptions.local_linear_trend`.

Host-side post-processing (reference :635-1093) is re-implemented on numpy arrays and pinned
against the reference's own output by tests/test_golden_postprocessing.py.
"""
import collections.abc
import dataclasses
import math
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd

from causalimpact import _diagnostics
from causalimpact import _model
from causalimpact import _native
from causalimpact import data as cid
from causalimpact import posterior_processing
from causalimpact.indices import InputDateType
from causalimpact.indices import OutputDateType
from causalimpact.indices import OutputPeriodType

_SeedType = Union[int, Tuple[int, int], Sequence[int]]
_KEPT_AFTER_POST = ["observed", "posterior_mean", "posterior_lower", "posterior_upper"]


class Tensor(np.ndarray):
  """numpy array that also answers `.numpy()` like the reference's tf.Tensor results."""

  def numpy(self):
    return np.asarray(self)


def _tensor(a) -> Tensor:
  return np.asarray(a).view(Tensor)


class _LazyMapping(collections.abc.Mapping):
  """A read-only dict whose content is computed on first use (the convergence diagnostics cost
  as much host time as a third of the fit; most callers never read them)."""

  def __init__(self, make):
    self._make, self._value = make, None

  def _get(self):
    if self._value is None:
      self._value = dict(self._make())
      self._make = None
    return self._value

  def __getitem__(self, key):
    return self._get()[key]

  def __iter__(self):
    return iter(self._get())

  def __len__(self):
    return len(self._get())

  def __repr__(self):
    return repr(self._get())

  def __reduce__(self):
    # pickles (multiprocessing, joblib, caches) as the plain dict it stands for
    return (dict, (self._get(),))


@dataclasses.dataclass
class CausalImpactPosteriorSamples:
  """Draws of the model's latents (reference :44-58).  Leading axis = pooled draws
  (chains x num_results, chain-major)."""
  observation_noise_scale: np.ndarray           # [draws]
  level_scale: Optional[np.ndarray]             # [draws]
  level: Optional[np.ndarray]                   # [draws, T]
  weights: Optional[np.ndarray]                 # [draws, covariates + 1] or None
  seasonal_drift_scales: Optional[np.ndarray]   # [draws, K] or None
  seasonal_levels: Optional[np.ndarray]         # [draws, T, K]
  slope_scale: Optional[np.ndarray] = None      # [draws]      (local_linear_trend only)
  slope: Optional[np.ndarray] = None            # [draws, T]   (local_linear_trend only)


@dataclasses.dataclass
class CausalImpactAnalysis:
  """series / summary frames + posterior draws (reference :61-144; schemas SURVEY App. E)."""
  series: pd.DataFrame
  summary: pd.DataFrame
  posterior_samples: CausalImpactPosteriorSamples
  diagnostics: Optional[Dict[str, Any]] = None   # split-R-hat per scalar when num_chains > 1


@dataclasses.dataclass
class DataOptions:
  """reference :147-159.  dtype may be numpy / python float types (or anything with a
  `.name` of "float32"/"float64").  The Gibbs sampler computes in that type (float64: the kernels
  of csrc/ci_gibbs64.h, draw for draw equal to the float64 oracle); the HMC extension computes in
  float32 for either."""
  outcome_column: Optional[str] = None
  standardize_data: bool = True
  dtype: Any = np.float32


@dataclasses.dataclass(frozen=True)
class Seasons:
  """One seasonal effect (reference :162-180): int, per-season tuple or per-cycle table."""
  num_seasons: int
  num_steps_per_season: Union[int, Tuple[int, ...], Tuple[Tuple[int, ...], ...]] = 1


@dataclasses.dataclass
class ModelOptions:
  """reference :183-203 (+ local_linear_trend, the BASELINE cfg2 extension)."""
  prior_level_sd: float = 0.01
  seasons: List[Seasons] = dataclasses.field(default_factory=list)
  local_linear_trend: bool = False


@dataclasses.dataclass
class InferenceOptions:
  """reference :206-220 (+ num_chains / devices / sampler extensions; defaults reproduce the
  reference: one Gibbs chain)."""
  num_results: int = 900
  num_warmup_steps: Optional[int] = None
  num_chains: int = 1
  devices: Optional[Sequence[int]] = None
  sampler: str = "gibbs"          # "gibbs" (the reference's sampler) or "hmc" (extension, _hmc.py)
  hmc_init: str = "gibbs"         # HMC chains start at the Gibbs initial state, or ("vi") at draws
                                  # of a mean-field surrogate posterior (_vi.py), as tfp.sts.fit_with_hmc
  hmc_prior: str = "slab"         # HMC regression prior: "slab" (Gaussian slab of the reference's
                                  # spike-and-slab prior) or "horseshoe" (tfp.sts.SparseLinearRegression)
  # Quantiles / effect sums of the T x (chains * draws) predictive draws computed on the GPU that
  # holds them (csrc/ci_summary.h) instead of pandas on the host.  Single-device Gibbs fits only;
  # `False` keeps the reference's host arithmetic (and downloads the trajectories).
  summarize_on_device: bool = True
  # CI_FLAG_* bits handed to the C-ABI (include/causalimpact_amd.h), e.g. `_native.FLAG_NO_CLUSTER`
  # on a GPU shared with other jobs: the time-parallel seasonal kernel then runs one workgroup per
  # chain instead of spin-synchronised clusters of CUs (same draws, bit for bit).
  kernel_flags: int = 0

  def __post_init__(self):
    if self.num_warmup_steps is None:
      self.num_warmup_steps = math.ceil(self.num_results / 9)


def fit_causalimpact(data: pd.DataFrame,
                     pre_period: Tuple[InputDateType, InputDateType],
                     post_period: Tuple[InputDateType, InputDateType],
                     alpha: float = 0.05,
                     seed: Optional[_SeedType] = None,
                     data_options: Optional[DataOptions] = None,
                     model_options: Optional[ModelOptions] = None,
                     inference_options: Optional[InferenceOptions] = None,
                     **kwargs) -> CausalImpactAnalysis:
  """Fits the CausalImpact model and summarises the effect (reference :223-339)."""
  data_options = data_options if data_options is not None else DataOptions()
  model_options = model_options if model_options is not None else ModelOptions()
  inference_options = inference_options if inference_options is not None else InferenceOptions()
  experimental_model = kwargs.pop("experimental_model", None)
  kwargs.pop("experimental_tf_function_cache_key_addition", 0)   # no graph cache to key
  if kwargs:
    raise TypeError(f"Received unknown {kwargs=}")
  if experimental_model is not None:
    raise NotImplementedError(
        "experimental_model takes a tfp.sts.StructuralTimeSeries; this build has no TFP. Use "
        "ModelOptions(local_linear_trend=..., seasons=...) instead.")

  ci_data = cid.CausalImpactData(
      data=data, pre_period=pre_period, post_period=post_period,
      outcome_column=data_options.outcome_column,
      standardize_data=data_options.standardize_data, dtype=data_options.dtype)
  if not 0 < alpha < 1:
    raise ValueError("`alpha` must be between 0 and 1.")
  request = (_device_summary_request(ci_data, alpha) if inference_options.summarize_on_device
             else None)
  samples, posterior_means, posterior_trajectories, device_summary = _run_sampler(
      ci_data=ci_data, prior_level_sd=model_options.prior_level_sd, seed=seed,
      num_results=inference_options.num_results,
      num_warmup_steps=inference_options.num_warmup_steps, dtype=data_options.dtype,
      seasons=model_options.seasons, num_chains=inference_options.num_chains,
      devices=inference_options.devices, local_linear_trend=model_options.local_linear_trend,
      sampler=inference_options.sampler, summary_request=request,
      hmc_init=inference_options.hmc_init, hmc_prior=inference_options.hmc_prior,
      kernel_flags=inference_options.kernel_flags)
  # (draws pooled on the host -- several devices, float64, HMC -- were summarised inside
  #  _run_sampler, in the sampler's internal units)
  if device_summary is not None:
    series, summary = _compute_impact_device(posterior_means, device_summary, request, ci_data,
                                             alpha)
  else:
    series, summary = _compute_impact(posterior_means=posterior_means,
                                      posterior_trajectories=posterior_trajectories,
                                      ci_data=ci_data, alpha=alpha)
  has_weights = samples["weights"].shape[-1] > 0
  has_seasons = samples["seasonal_drift_scales"].shape[-1] > 0
  posterior = CausalImpactPosteriorSamples(
      observation_noise_scale=_tensor(samples["observation_noise_scale"]),
      level_scale=_tensor(samples["level_scale"]),
      level=_tensor(samples["level"]),
      weights=_tensor(samples["weights"]) if has_weights else None,                 # :330-331
      seasonal_drift_scales=(_tensor(samples["seasonal_drift_scales"])
                             if has_seasons else None),                              # :332-334
      seasonal_levels=_tensor(samples["seasonal_levels"]),                           # :312-322
      slope_scale=_tensor(samples["slope_scale"]) if model_options.local_linear_trend else None,
      slope=_tensor(samples["slope"]) if model_options.local_linear_trend else None)
  return CausalImpactAnalysis(series, summary, posterior, samples.get("diagnostics"))


def _sanitize_seed(seed: Optional[_SeedType]) -> Tuple[int, int]:
  """int s -> (0, s); pair -> pair; None -> fresh entropy (reference :535-543)."""
  if seed is None:
    return tuple(int(v) for v in np.random.SeedSequence().generate_state(2))
  if isinstance(seed, (int, np.integer)):
    return (0, int(seed) & 0xFFFFFFFF)
  pair = np.asarray(seed).reshape(-1)
  if pair.shape[0] != 2:
    raise ValueError(f"seed must be an int or a pair of ints, got {seed!r}")
  return (int(pair[0]) & 0xFFFFFFFF, int(pair[1]) & 0xFFFFFFFF)


def split_rhat(draws: np.ndarray) -> float:
  """Split-R-hat of [chains, draws] (Gelman et al. 2013); NaN for degenerate input."""
  return _diagnostics.split_rhat(draws)


def effective_sample_size(draws: np.ndarray, kind: str = "plain") -> float:
  """Effective sample size of [chains, draws] scalar draws: split chains, FFT autocovariances,
  Geyer's initial monotone positive sequence.  kind: "plain" (the draws as they are), "bulk"
  (rank-normalised) or "tail" (min over the 5 % / 95 % indicators) -- Vehtari et al. 2021.
  NaN for degenerate input.  (SURVEY.md 8(f) N3 -- absent upstream.)"""
  fn = {"plain": _diagnostics.ess_plain, "bulk": _diagnostics.ess_bulk,
        "tail": _diagnostics.ess_tail}[kind]
  return fn(draws)


def _train_causalimpact_sts(*,
                            ci_data: cid.CausalImpactData,
                            prior_level_sd,
                            seed: Optional[_SeedType],
                            num_results: int,
                            num_warmup_steps: int,
                            model=None,
                            dtype=np.float32,
                            seasons: Sequence[Seasons] = (),
                            experimental_tf_function_cache_key_addition: int = 0,
                            num_chains: int = 1,
                            devices: Optional[Sequence[int]] = None,
                            local_linear_trend: bool = False,
                            sampler: str = "gibbs"):
  """Runs the Gibbs sampler on the GPU(s) (reference :503-606).

  Returns (samples dict, posterior_means [T], posterior_trajectories [draws, T]); draws are
  pooled over chains, chain-major.  Chain c uses RNG stream c regardless of how chains are
  spread over `devices`, so results are invariant to the device count.
  """
  del experimental_tf_function_cache_key_addition
  return _run_sampler(ci_data=ci_data, prior_level_sd=prior_level_sd, seed=seed,
                      num_results=num_results, num_warmup_steps=num_warmup_steps, model=model,
                      dtype=dtype, seasons=seasons, num_chains=num_chains, devices=devices,
                      local_linear_trend=local_linear_trend, sampler=sampler)[:3]


def _outcome_float64(ci_data) -> np.ndarray:
  """The pre-period outcome the sampler models.  Standardised data: `outcome_ts.time_series`, i.e.
  the values rounded to DataOptions.dtype exactly as the reference hands them to its sampler
  (data.py:125-126).  Raw-scale data (`standardize_data=False`): the float64 source column -- the
  rounding to float32 must come AFTER the internal conditioning, not before it."""
  src = getattr(ci_data, "model_pre_data", None)
  if getattr(ci_data, "standardize_data", True) or src is None:
    return np.asarray(ci_data.outcome_ts.time_series, np.float64)
  return np.asarray(src[ci_data.outcome_column], np.float64)


def _internal_conditioning(ci_data) -> Tuple[float, float]:
  """(mu, s) of the affine map y -> (y - mu) / s applied to the outcome before it reaches the
  float32 kernels when the caller did NOT standardise (`standardize_data=False`); (0, 1) else.

  The default model is exactly equivariant under it: every prior and every initial value the
  reference builds is expressed in units of outcome_sd (:424-443, :467-474, :566-574), the level
  prior is centred on the first observation (:467-469), and the weights prior is conjugate (its
  covariance carries sigma^2_obs).  So with y' = (y - mu) / s the Markov chain on
  (level' = (level - mu) / s, slope / s, seasonal / s, weights / s, scales / s) driven by the SAME
  random numbers is the same chain; the draws are mapped back in float64.  What this buys: the
  float32 scans never see a large offset or a tiny scale (the reference's own TODO case,
  causalimpact_lib_test.py:679-682: no standardisation, y + 100 with noise 1e-4, where float32
  would resolve the noise with a dozen levels).  tests/test_conditioning.py checks the
  equivariance on the float64 oracle (1e-9) and the GPU tests run that TODO case."""
  if getattr(ci_data, "standardize_data", True):
    return 0.0, 1.0
  pre = _outcome_float64(ci_data)
  pre = pre[~np.isnan(pre)]
  if pre.size < 2:
    return 0.0, 1.0
  mu, sd = float(np.mean(pre)), float(np.std(pre, ddof=1))
  if not np.isfinite(sd) or sd <= 0.0:
    sd = 1.0
  return mu, sd


def _run_sampler(*, ci_data, prior_level_sd, seed, num_results, num_warmup_steps, model=None,
                 dtype=np.float32, seasons=(), num_chains=1, devices=None,
                 local_linear_trend=False, sampler="gibbs", summary_request=None,
                 hmc_init="gibbs", hmc_prior="slab", kernel_flags=0):
  """_train_causalimpact_sts plus, when `summary_request` is given (single device, Gibbs), the
  on-device summary of the predictive draws; the [draws, T] trajectories then stay in HBM and
  are returned as None."""
  if model is not None:
    raise NotImplementedError("custom tfp.sts models are not supported by the HIP path")
  seed_pair = _sanitize_seed(seed)
  np_dtype = cid._as_numpy_dtype(dtype)  # pylint: disable=protected-access
  # dtype (reference :159: the sampler runs in DataOptions.dtype).  float32 (the default): the
  # latency / time-parallel kernels, on an INTERNALLY CONDITIONED copy of a raw-scale outcome (see
  # `_internal_conditioning`: an exact reparametrisation that keeps every quantity the float32
  # scans touch O(1), mapped back in float64).  float64: the Gibbs sampler runs in float64 on the
  # sequential one-wavefront kernel (csrc/ci_gibbs64.h; draw for draw equal to the float64 oracle
  # to ~1e-9, tests/test_gpu_float64.py); slower -- it is the precision option.  The HMC
  # extension computes in float32 for either dtype.
  design = None if ci_data.feature_ts is None else np.asarray(ci_data.feature_ts.values,
                                                              dtype=np.float64)    # :545-546
  # Post-period handled as missing observations: forecasting == sampling (:548-562).
  n_after = ci_data.model_after_pre_data.shape[0]
  y = np.concatenate([_outcome_float64(ci_data), np.full(n_after, np.nan)])
  mask = np.concatenate([np.asarray(ci_data.outcome_ts.is_missing, bool),
                         np.ones(n_after, bool)])
  T = y.shape[0]
  cond_mu, cond_s = _internal_conditioning(ci_data)
  if (cond_mu, cond_s) != (0.0, 1.0):
    y = (y - cond_mu) / cond_s
  outcome_sd = float(np.nanstd(y[:T - n_after], ddof=1))
  num_seasons, season_change = _model.expand_seasons(seasons, T)
  params = _model.series_params(y, mask, design, prior_level_sd=prior_level_sd,
                                num_seasonal_blocks=len(num_seasons),
                                has_slope=local_linear_trend, outcome_sd=outcome_sd)
  params["weights_prior_scale"] = cond_s * cond_s     # Omega in the conditioned units (exact map)
  P = 0 if design is None else design.shape[1]
  if P > 0:
    # the spike-and-slab branch clips the VARIANCE at upper_bound = 1.2 sd, a scale (:442-443 as
    # the sampler reads it): var <= 1.2 sd  <=>  var' <= 1.2 sd / s^2 in the conditioned units
    params["obs_ub"] = params["obs_ub"] / cond_s
  K = len(num_seasons)

  if sampler not in ("gibbs", "hmc"):
    raise ValueError(f"sampler must be 'gibbs' or 'hmc', got {sampler!r}")
  devs = list(devices) if devices else [0]
  shares = np.array_split(np.arange(num_chains), len(devs))
  parts = []
  device_summary = None
  def run_on(dev, chain_ids):
    """One device's share of the chains (chain ids keep their global RNG streams)."""
    nonlocal device_summary
    if sampler == "hmc":
      from causalimpact import _hmc  # pylint: disable=import-outside-toplevel
      res = _hmc.fit_hmc(y, mask, design, params, has_slope=local_linear_trend,
                         num_results=num_results, num_warmup=num_warmup_steps,
                         num_chains=len(chain_ids), seed=seed_pair, device=dev,
                         chain_offset=int(chain_ids[0]), init=hmc_init, prior=hmc_prior,
                         horseshoe_scale=0.1 / cond_s, num_seasons=num_seasons,
                         season_change=season_change)
      return {k: v for k, v in res.items() if not k.startswith("hmc_")}
    pb = _native.make_problem(T=T, P=P, has_slope=local_linear_trend, num_seasons=num_seasons,
                              num_warmup=num_warmup_steps, num_results=num_results,
                              num_chains=len(chain_ids), chain_offset=int(chain_ids[0]),
                              seed=seed_pair, device=dev, flags=int(kernel_flags))
    if np_dtype == np.float64:
      # float64 compute (csrc/ci_gibbs64.h): the sequential kernel, every buffer float64
      return _native.fit_gibbs_f64(pb, y[None], mask[None], None if design is None else design[None],
                                   season_change, _native.make_params([params]))
    if summary_request is not None and len(devs) == 1:
      sess = _native.Session(pb, y[None], mask[None], None if design is None else design[None],
                             season_change, _native.make_params([params]))
      try:
        # the draws travel to (pinned) host memory while the sampler runs; the trajectories stay
        # on the device for the summary kernels
        _, part = sess.run_streamed(
            want=[k for k in _native._OUT_FIELDS  # pylint: disable=protected-access
                  if k != "posterior_trajectories" and (k != "slope" or local_linear_trend)],
            chunk_draws=32)
        summary_request["ranks"] = _summary_ranks(len(chain_ids) * num_results,
                                                  summary_request["quantiles"])
        # value = trajectory * scale + shift, the trajectory being in the internal units
        device_summary = sess.summarize(
            scale=np.asarray(summary_request["scale"], np.float64) * cond_s,
            shift=(np.asarray(summary_request["shift"], np.float64)
                   + cond_mu * np.asarray(summary_request["scale"], np.float64)),
            **{k: summary_request[k] for k in ("observed", "flags", "ranks")})
      finally:
        sess.close()
      return part
    return _native.fit_gibbs(pb, y[None], mask[None], None if design is None else design[None],
                             season_change, _native.make_params([params]))

  work = [(dev, ids) for dev, ids in zip(devs, shares) if len(ids)]
  if len(work) == 1:
    parts = [run_on(*work[0])]
  else:
    # the library calls are synchronous and release the GIL: one host thread per device runs
    # the shares concurrently (no collective: chains are independent)
    import concurrent.futures  # pylint: disable=import-outside-toplevel
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(work)) as pool_:
      parts = list(pool_.map(lambda a: run_on(*a), work))
  out = ({k: v[0] for k, v in parts[0].items()} if len(parts) == 1 else              # [C, ...]
         {k: np.concatenate([p[k][0] for p in parts], axis=0) for k in parts[0]})

  if (summary_request is not None and device_summary is None
      and "posterior_trajectories" in out):
    # draws pooled on the host (several devices, the float64 kernels, the HMC path): the summary
    # kernels read float32 trajectories, so they get them in the INTERNAL (conditioned) units --
    # O(1) values -- with the map to the caller's scale folded into (scale, shift) in float64, as
    # on the single-device path.  Summarising the rescaled draws would round the offset back in
    # (8e-6 steps at y ~ 100).
    tr = out["posterior_trajectories"]
    summary_request["ranks"] = _summary_ranks(tr.shape[0] * tr.shape[1], summary_request["quantiles"])
    device_summary = _native.summarize_draws(
        tr.reshape((tr.shape[0] * tr.shape[1],) + tr.shape[2:]),
        float(summary_request["scale"]) * cond_s,
        float(summary_request["shift"]) + cond_mu * float(summary_request["scale"]),
        summary_request["observed"], summary_request["flags"], summary_request["ranks"],
        device=devs[0])
  if (cond_mu, cond_s) != (0.0, 1.0):
    # back to the caller's scale, in float64: locations get the offset, everything else the scale
    out = {k: np.asarray(v, np.float64) * cond_s for k, v in out.items()}
    for k in ("level", "posterior_means", "posterior_trajectories"):
      if k in out:
        out[k] += cond_mu

  def pool(a):   # [C, S, ...] -> [C*S, ...]
    return a.reshape((a.shape[0] * a.shape[1],) + a.shape[2:]).astype(np_dtype, copy=False)

  samples = dict(
      observation_noise_scale=pool(out["observation_noise_scale"]),
      level_scale=pool(out["level_scale"]), slope_scale=pool(out["slope_scale"]),
      weights=pool(out["weights"]), level=pool(out["level"]),
      slope=pool(out["slope"]) if "slope" in out else None,
      seasonal_drift_scales=pool(out["seasonal_drift_scales"]),
      seasonal_levels=pool(out["seasonal_levels"]))
  assert samples["weights"].shape[-1] == P and samples["seasonal_levels"].shape[-1] == K
  if num_chains > 1:
    scalars = {k: out[k] for k in ("observation_noise_scale", "level_scale")}
    samples["diagnostics"] = _LazyMapping(
        lambda: dict(_diagnostics.summarize(scalars), num_chains=num_chains))
  # the predictive arrays feed the data-scale summaries: with internal conditioning they carry the
  # offset again and stay float64 (float32 would round 100.0001 to 8e-6 steps)
  pred_dtype = np.float64 if (cond_mu, cond_s) != (0.0, 1.0) else np_dtype
  posterior_means = out["posterior_means"].mean(axis=0).astype(pred_dtype, copy=False)    # :627
  posterior_trajectories = None
  if "posterior_trajectories" in out:                                                     # :631
    tr = out["posterior_trajectories"]
    posterior_trajectories = tr.reshape((tr.shape[0] * tr.shape[1],) + tr.shape[2:]).astype(
        pred_dtype, copy=False)
  return samples, posterior_means, posterior_trajectories, device_summary


# --------------------------------------------------------------------------------------
# impact post-processing (reference :635-1093) -- numpy inside, the reference's frames outside
# --------------------------------------------------------------------------------------
def _compute_impact(posterior_means, posterior_trajectories, ci_data: cid.CausalImpactData,
                    alpha: float = 0.05) -> Tuple[pd.DataFrame, pd.DataFrame]:
  """(series, summary) from the sampler's predictive draws (reference :635-705)."""
  if not 0 < alpha < 1:
    raise ValueError("`alpha` must be between 0 and 1.")
  observed_pre = ci_data.pre_data[ci_data.outcome_column]
  observed_post = ci_data.after_pre_data[ci_data.outcome_column]
  in_post = (observed_post.index >= ci_data.post_period[0]) & (observed_post.index <=
                                                              ci_data.post_period[1])
  observed_post = observed_post.loc[in_post]
  observed_full = pd.concat([observed_pre, observed_post], axis=0)
  quantiles = (alpha / 2.0, 1.0 - alpha / 2.0)
  trajectories, trajectory_summary = _sample_posterior_predictive(
      posterior_means=posterior_means, posterior_trajectories=posterior_trajectories,
      ci_data=ci_data, quantiles=quantiles)
  trajectory_dict = _compute_impact_trajectories(trajectories, observed_full,
                                                 treatment_start=ci_data.post_period[0])
  series = _compute_impact_estimates(posterior_trajectory_summary=trajectory_summary,
                                     trajectory_dict=trajectory_dict,
                                     observed_ts_full=observed_full, ci_data=ci_data,
                                     quantiles=quantiles)
  summary = _compute_summary(posterior_trajectory_summary=trajectory_summary,
                             trajectory_dict=trajectory_dict, observed_ts_post=observed_post,
                             post_period=ci_data.post_period, quantiles=quantiles, alpha=alpha)
  return series, summary


def _observed_series(ci_data: cid.CausalImpactData):
  observed_pre = ci_data.pre_data[ci_data.outcome_column]
  observed_post = ci_data.after_pre_data[ci_data.outcome_column]
  in_post = (observed_post.index >= ci_data.post_period[0]) & (observed_post.index <=
                                                              ci_data.post_period[1])
  observed_post = observed_post.loc[in_post]
  return observed_post, pd.concat([observed_pre, observed_post], axis=0)


def _quantile_ranks(num_draws: int, quantiles):
  """numpy's 'linear' quantile of n values: lerp(x_(lo), x_(lo+1), gamma) -- the order
  statistics it needs and the interpolation weight (numpy/lib/_function_base_impl.py
  _quantile: virtual index q (n - 1), previous = floor, next = previous + 1 clipped)."""
  out = []
  for q in quantiles:
    virtual = (num_draws - 1) * np.float64(q)
    lo = int(np.floor(virtual))
    if lo >= num_draws - 1:
      lo, hi, gamma = num_draws - 1, num_draws - 1, np.float64(0.0)
    else:
      hi, gamma = lo + 1, virtual - lo
    out.append((lo, hi, gamma))
  return out


def _summary_ranks(num_draws: int, quantiles) -> List[int]:
  """Order statistics the two quantiles need, and their mirror images (effect = observed - value
  reverses the order)."""
  qs = _quantile_ranks(num_draws, quantiles)
  return sorted({r for lo, hi, _ in qs for r in (lo, hi)} |
                {num_draws - 1 - r for lo, hi, _ in qs for r in (lo, hi)})


def _device_summary_request(ci_data: cid.CausalImpactData, alpha: float) -> Dict:
  """Arguments of ci_session_summarize for this analysis (see include/causalimpact_amd.h)."""
  _, observed_full = _observed_series(ci_data)
  idx = posterior_processing.model_index(ci_data)
  obs = observed_full.reindex(idx).to_numpy(dtype=np.float64)
  after_start = ~np.asarray(idx < ci_data.post_period[0])
  window = np.asarray((idx >= ci_data.post_period[0]) & (idx <= ci_data.post_period[1]))
  if ci_data.standardize_data:
    scale = float(np.ravel(ci_data.outcome_scaler.stddev_)[0])
    shift = float(np.ravel(ci_data.outcome_scaler.mean_)[0])
  else:
    scale, shift = 1.0, 0.0
  return dict(scale=scale, shift=shift, observed=obs,
              flags=after_start.astype(np.uint8) | (window.astype(np.uint8) << 1),
              quantiles=(alpha / 2.0, 1.0 - alpha / 2.0),
              ranks=None)   # filled in by _run_sampler once the number of draws is known


def _lerp_order_stats(order: Dict[int, np.ndarray], lo: int, hi: int, gamma) -> np.ndarray:
  """numpy's own interpolation of two order statistics (so results match np.quantile bit for
  bit): the quantile of the 2-point sample {x_(lo), x_(hi)} at `gamma` is lerp(x_lo, x_hi, gamma)."""
  pair = np.stack([order[lo], order[hi]])
  with np.errstate(invalid="ignore"):
    return np.quantile(pair, gamma, axis=0)


def _compute_impact_device(posterior_means, device_summary: Dict, request: Dict,
                           ci_data: cid.CausalImpactData, alpha: float):
  """(series, summary) from the on-device summary (csrc/ci_summary.h) -- same frames as
  _compute_impact, which stays the host reference of this arithmetic."""
  quantiles = (alpha / 2.0, 1.0 - alpha / 2.0)
  observed_post, observed_full = _observed_series(ci_data)
  idx = posterior_processing.model_index(ci_data)
  obs = request["observed"]
  ranks = list(request["ranks"])
  num_draws = device_summary["per_draw"].shape[1]
  value_order = {r: device_summary["value_order"][i] for i, r in enumerate(ranks)}
  cum_order = {r: device_summary["cum_order"][i] for i, r in enumerate(ranks)}
  (lo_a, hi_a, g_a), (lo_b, hi_b, g_b) = _quantile_ranks(num_draws, quantiles)

  def frame(prefix, lower, upper):
    return pd.DataFrame({prefix + "_lower": lower, prefix + "_upper": upper}, index=idx)

  means = posterior_processing.process_posterior_quantities(ci_data, posterior_means,
                                                            ["posterior_mean"])
  trajectory_summary = means.join(frame("posterior",
                                        _lerp_order_stats(value_order, lo_a, hi_a, g_a),
                                        _lerp_order_stats(value_order, lo_b, hi_b, g_b)))
  # point effect = -(value - observed) is decreasing in the value: its k-th smallest is the
  # (N-1-k)-th smallest value, mapped
  mirrored = {k: -(value_order[num_draws - 1 - k] - obs) for k in {lo_a, hi_a, lo_b, hi_b}}
  bands = {
      "point_effects": frame("point_effects", _lerp_order_stats(mirrored, lo_a, hi_a, g_a),
                             _lerp_order_stats(mirrored, lo_b, hi_b, g_b)),
      "cumulative_effects": frame("cumulative_effects",
                                  _lerp_order_stats(cum_order, lo_a, hi_a, g_a),
                                  _lerp_order_stats(cum_order, lo_b, hi_b, g_b)),
  }
  series = _compute_impact_estimates(posterior_trajectory_summary=trajectory_summary,
                                     trajectory_dict=None, observed_ts_full=observed_full,
                                     ci_data=ci_data, quantiles=quantiles, bands=bands)
  window = (request["flags"] & 2) != 0
  n_obs_window = int(np.sum(~np.isnan(obs[window])))
  pred_sum, point_sum = device_summary["per_draw"]
  with np.errstate(invalid="ignore", divide="ignore"):
    per_draw = dict(pred_mean=pred_sum / int(window.sum()), pred_sum=pred_sum,
                    point_mean_t=point_sum / n_obs_window if n_obs_window else
                    np.full_like(point_sum, np.nan),
                    point_sum_t=point_sum)
  summary = _compute_summary(posterior_trajectory_summary=trajectory_summary,
                             trajectory_dict=None, observed_ts_post=observed_post,
                             post_period=ci_data.post_period, quantiles=quantiles, alpha=alpha,
                             per_draw=per_draw)
  return series, summary


def _sample_posterior_predictive(posterior_means, posterior_trajectories,
                                 ci_data: cid.CausalImpactData, quantiles: Tuple[float, float]):
  """Data-scale trajectories (T x draws) and their mean/quantile summary (reference :708-767)."""
  if any((q < 0) | (q > 1) for q in quantiles):
    raise ValueError("All elements of `quantiles` must be in (0, 1). Got %s" % (quantiles,))
  if quantiles[0] > quantiles[1]:
    raise ValueError("`quantiles` must be sorted in ascending order. Got %s" % (quantiles,))
  means = posterior_processing.process_posterior_quantities(ci_data, posterior_means,
                                                            ["posterior_mean"])
  trajectories = _package_posterior_trajectories(posterior_trajectories, ci_data)
  bands = posterior_processing.calculate_trajectory_quantiles(trajectories, "posterior", quantiles)
  return trajectories, means.join(bands)


def _package_posterior_trajectories(posterior_trajectories,
                                    ci_data: cid.CausalImpactData) -> pd.DataFrame:
  """[draws, T] -> T x draws frame named sample_1..sample_n, unscaled (reference :770-790)."""
  names = [f"sample_{i + 1}" for i in range(np.shape(posterior_trajectories)[0])]
  return posterior_processing.process_posterior_quantities(ci_data, posterior_trajectories, names)


def _compute_impact_trajectories(posterior_trajectories: pd.DataFrame,
                                 observed_ts_full: pd.Series,
                                 treatment_start: OutputDateType) -> Dict[str, pd.DataFrame]:
  """Per-draw point effects (observed - predicted) and their running sum from the treatment
  start; NaN observations give NaN effects and are skipped by the running sum
  (reference :793-837)."""
  idx, cols = posterior_trajectories.index, posterior_trajectories.columns
  pred = posterior_trajectories.to_numpy(dtype=np.float64)
  obs = observed_ts_full.reindex(idx).to_numpy(dtype=np.float64)
  point = -(pred - obs[:, None])
  base = np.where(np.asarray(idx < treatment_start)[:, None], 0.0, point)
  holes = np.isnan(base)
  cumulative = np.cumsum(np.where(holes, 0.0, base), axis=0)
  cumulative[holes] = np.nan
  return {
      "predictions": posterior_trajectories,
      "point_effects": pd.DataFrame(point, index=idx, columns=cols),
      "cumulative_effects": pd.DataFrame(cumulative, index=idx, columns=cols),
  }


def _compute_impact_estimates(posterior_trajectory_summary: pd.DataFrame,
                              trajectory_dict: Dict[str, pd.DataFrame],
                              observed_ts_full: pd.Series, ci_data: cid.CausalImpactData,
                              quantiles: Tuple[float, float],
                              bands: Optional[Dict[str, pd.DataFrame]] = None) -> pd.DataFrame:
  """The 14-column `series` frame over the full input index (reference :840-931).  `bands`:
  precomputed quantile frames of the effect trajectories (on-device summary)."""
  if bands is None:
    bands = {k: posterior_processing.calculate_trajectory_quantiles(trajectory_dict[k], k,
                                                                    quantiles)
             for k in ("point_effects", "cumulative_effects")}
  idx = posterior_trajectory_summary.index
  obs_s = observed_ts_full.reindex(idx)
  obs = obs_s.to_numpy(dtype=np.float64)
  pm = posterior_trajectory_summary["posterior_mean"]
  if not (pm.dtype == np.float64 and obs_s.dtype == np.float64 and
          all(b[c].dtype == np.float64 for b in bands.values() for c in b.columns) and
          all(b.index.equals(idx) for b in bands.values())):
    return _compute_impact_estimates_frames(posterior_trajectory_summary, observed_ts_full,
                                            ci_data, bands)
  # float64 throughout (every ordinary call): the columns as arrays, ONE frame at the end -- the same
  # IEEE operations as the frame-by-frame version below (tests/test_golden_postprocessing.py pins
  # both against the reference's output), 4 ms -> 1 ms of host time per fit
  point_mean = obs - pm.to_numpy()
  base = np.where(idx < ci_data.post_period[0], 0.0, point_mean)
  hole = np.isnan(base)
  cum_mean = np.cumsum(np.where(hole, 0.0, base))          # skips NaN like the reference
  cum_mean[hole] = np.nan
  # between pre- and post-period, and after the post-period: predictions only (:899-907);
  # no observation => no effect (NaN, not 0) (:909-915)
  blank = np.asarray(((idx > ci_data.pre_period[1]) & (idx < ci_data.post_period[0])) |
                     (idx > ci_data.post_period[1])) | np.isnan(obs)

  def effect(v):
    v = np.array(v, dtype=np.float64)
    v[blank] = np.nan
    return v

  cols = {"observed": obs}
  for c in posterior_trajectory_summary.columns:
    cols[c] = posterior_trajectory_summary[c].to_numpy()
  cols["point_effects_mean"] = effect(point_mean)
  for c in bands["point_effects"].columns:
    cols[c] = effect(bands["point_effects"][c].to_numpy())
  cols["cumulative_effects_mean"] = effect(cum_mean)
  for c in bands["cumulative_effects"].columns:
    cols[c] = effect(bands["cumulative_effects"][c].to_numpy())
  for c in posterior_trajectory_summary.columns:           # any summary column beyond the kept ones
    if c not in _KEPT_AFTER_POST:
      cols[c] = effect(cols[c])
  if idx.equals(ci_data.data.index):                        # (the usual case; the input's own index object)
    frame = pd.DataFrame(cols, index=ci_data.data.index)
  else:
    frame = pd.DataFrame(cols, index=idx).reindex(ci_data.data.index, fill_value=np.nan)
  frame["observed"] = ci_data.data[ci_data.outcome_column]
  frame["pre_period_start"] = ci_data.pre_period[0]
  frame["pre_period_end"] = ci_data.pre_period[1]
  frame["post_period_start"] = ci_data.post_period[0]
  frame["post_period_end"] = ci_data.post_period[1]
  return frame


def _compute_impact_estimates_frames(posterior_trajectory_summary, observed_ts_full, ci_data, bands):
  """_compute_impact_estimates frame by frame (any dtypes / band indices)."""
  idx = posterior_trajectory_summary.index
  obs = observed_ts_full.reindex(idx)
  point_mean = obs - posterior_trajectory_summary["posterior_mean"]
  base = point_mean.where(~(idx < ci_data.post_period[0]), 0.0)
  cum_mean = base.cumsum()                                 # skips NaN like the reference
  frame = pd.concat([
      obs.rename("observed"), posterior_trajectory_summary,
      point_mean.rename("point_effects_mean"),
      bands["point_effects"],
      cum_mean.rename("cumulative_effects_mean"),
      bands["cumulative_effects"],
  ], axis=1)
  effect_cols = frame.columns.difference(_KEPT_AFTER_POST)
  # between pre- and post-period, and after the post-period: predictions only (:899-907)
  outside = (((frame.index > ci_data.pre_period[1]) & (frame.index < ci_data.post_period[0])) |
             (frame.index > ci_data.post_period[1]))
  frame.loc[outside, effect_cols] = np.nan
  # no observation => no effect (NaN, not 0) (:909-915)
  frame.loc[np.isnan(frame["observed"].to_numpy(dtype=np.float64)), effect_cols] = np.nan
  frame = frame.reindex(ci_data.data.index, fill_value=np.nan)
  frame["observed"] = ci_data.data[ci_data.outcome_column]
  frame["pre_period_start"] = ci_data.pre_period[0]
  frame["pre_period_end"] = ci_data.pre_period[1]
  frame["post_period_start"] = ci_data.post_period[0]
  frame["post_period_end"] = ci_data.post_period[1]
  return frame


def _summary_rows(post_mean, obs, pred_mean, pred_sum, point_mean_t, point_sum_t, quantiles):
  """The numbers of the `summary` frame (reference :966-1091) from the post-period posterior
  mean [T_w], observations [T_w] and per-draw window means / totals [draws]."""
  obs_mean, obs_sum = float(np.nanmean(obs)), float(np.nansum(obs))

  def sd(v):
    return float(np.std(v, ddof=1))

  def band(v):
    lo, hi = np.quantile(v, quantiles)
    return float(lo), float(hi)

  rel = obs_sum / pred_sum - 1.0
  avg_pred, cum_pred = float(post_mean.mean()), float(post_mean.sum())
  b_pm, b_ps, b_em, b_es, b_rel = (band(pred_mean), band(pred_sum), band(point_mean_t),
                                   band(point_sum_t), band(rel))     # one selection per vector
  rows = {
      "actual": (obs_mean, obs_sum),
      "predicted": (avg_pred, cum_pred),
      "predicted_lower": (b_pm[0], b_ps[0]),
      "predicted_upper": (b_pm[1], b_ps[1]),
      "predicted_sd": (sd(pred_mean), sd(pred_sum)),
      "abs_effect": (obs_mean - avg_pred, obs_sum - cum_pred),
      "abs_effect_lower": (b_em[0], b_es[0]),
      "abs_effect_upper": (b_em[1], b_es[1]),
      "abs_effect_sd": (sd(point_mean_t), sd(point_sum_t)),
      "rel_effect": (float(rel.mean()),) * 2,
      "rel_effect_lower": (b_rel[0],) * 2,
      "rel_effect_upper": (b_rel[1],) * 2,
      "rel_effect_sd": (sd(rel),) * 2,
  }
  # one-sided tail area of the observed total among the sampled totals, the observed total
  # included so that p stays in (0, 1)   (:1077-1091)
  pool = np.append(pred_sum, obs_sum)
  p_value = min(float((obs_sum <= pool).mean()), float((obs_sum >= pool).mean()))
  return rows, p_value


def _compute_summary(posterior_trajectory_summary: pd.DataFrame,
                     trajectory_dict: Dict[str, pd.DataFrame], observed_ts_post: pd.Series,
                     post_period: OutputPeriodType, quantiles: Tuple[float, float],
                     alpha: float, per_draw: Optional[Dict[str, np.ndarray]] = None) -> pd.DataFrame:
  """The 2 x 15 `summary` frame over the post-period (reference :934-1093).  `per_draw`:
  precomputed per-draw window means / totals (on-device summary)."""

  def window(frame):
    keep = (frame.index >= post_period[0]) & (frame.index <= post_period[1])
    return frame.loc[keep]

  post_mean = window(posterior_trajectory_summary)["posterior_mean"].to_numpy(dtype=np.float64)
  obs = observed_ts_post.to_numpy(dtype=np.float64)
  if per_draw is None:
    pred = window(trajectory_dict["predictions"]).to_numpy(dtype=np.float64)      # [T_post, draws]
    point = window(trajectory_dict["point_effects"]).to_numpy(dtype=np.float64)
    pred_mean, pred_sum = pred.mean(axis=0), pred.sum(axis=0)
    with np.errstate(invalid="ignore"):
      point_mean_t, point_sum_t = np.nanmean(point, axis=0), np.nansum(point, axis=0)
  else:
    pred_mean, pred_sum = per_draw["pred_mean"], per_draw["pred_sum"]
    point_mean_t, point_sum_t = per_draw["point_mean_t"], per_draw["point_sum_t"]
  rows, p_value = _summary_rows(post_mean, obs, pred_mean, pred_sum, point_mean_t, point_sum_t,
                                quantiles)
  summary = pd.DataFrame({k: {"average": v[0], "cumulative": v[1]} for k, v in rows.items()})
  summary["p_value"] = p_value
  summary["alpha"] = alpha
  return summary
