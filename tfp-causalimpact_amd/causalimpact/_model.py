"""Model constants handed to the device: priors, initial Gibbs state, season calendar.

Host-side mirror of
  * `_build_default_gibbs_model`      /root/reference/causalimpact/causalimpact_lib.py:398-500
  * the initial `GibbsSamplerState`   /root/reference/causalimpact/causalimpact_lib.py:563-581
  * `Seasons` semantics               /root/reference/causalimpact/causalimpact_lib.py:162-180
All constants are the reference's; each scales with the
outcome's standard deviation, so one dict of plain floats per series is produced
(== one `ci_series_params`, include/causalimpact_amd.h).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

# causalimpact_lib.py:424, :436-441, :472-473, :566
_LEVEL_PRIOR_SAMPLE_SIZE = 32.0
_OBS_PRIOR_WITH_COVARIATES = (25.0, 5.0)        # IG(25, 5 sd^2)
_OBS_PRIOR_NO_COVARIATES = (0.005, 0.005)       # IG(.005, .005 sd^2)
_OBS_UPPER_BOUND = 1.2
_DRIFT_PRIOR = (0.005, 5e-7)
_EXPECTED_R2 = 0.8
_EXPECTED_MODEL_SIZE = 3.0


def outcome_sd_of(y: np.ndarray, mask: np.ndarray) -> float:
  """np.nanstd(pre-period standardized outcome, ddof=1)  (causalimpact_lib.py:563-564)."""
  obs = np.asarray(y, np.float64)[~np.asarray(mask, bool)]
  return float(np.std(obs, ddof=1))


def season_change_flags(num_timesteps: int, num_seasons: int, num_steps_per_season) -> np.ndarray:
  """uint8[T]: 1 where step t is the last step of a season (the t -> t+1 transition rotates
  the seasonal effects).  An int applies to every season; a 1-D tuple gives per-season
  lengths; a 2-D tuple gives per-cycle per-season lengths and the whole table repeats
  (causalimpact_lib.py:170-177; tfp.sts.Seasonal's is_last_day_of_season)."""
  steps = np.asarray(num_steps_per_season, dtype=np.int64)
  if steps.ndim == 0:
    steps = np.full((num_seasons,), int(steps), dtype=np.int64)
  if steps.ndim > 2 or steps.shape[-1] != num_seasons:
    raise ValueError(
        f"num_steps_per_season must be an int, [num_seasons] or [num_cycles, num_seasons]; got "
        f"shape {steps.shape} for num_seasons={num_seasons}")
  lengths = steps.reshape(-1)
  if (lengths < 1).any():
    raise ValueError("every season must last at least one step")
  period = int(lengths.sum())
  table = np.zeros(period, dtype=np.uint8)
  table[np.cumsum(lengths) - 1] = 1
  return table[np.arange(num_timesteps) % period]


def series_params(y: np.ndarray, mask: np.ndarray, X: Optional[np.ndarray], *,
                  prior_level_sd: float = 0.01, num_seasonal_blocks: int = 0,
                  has_slope: bool = False, outcome_sd: Optional[float] = None,
                  prior_slope_sd: Optional[float] = None) -> Dict[str, float]:
  """Priors and initial state of one series as plain floats.

  y, mask: the masked outcome the sampler sees (pre-period values, everything after the
  pre-period masked, causalimpact_lib.py:548-562).  X: [T, P] design incl. intercept or None.
  """
  y = np.asarray(y, np.float64)
  mask = np.asarray(mask, bool)
  num_features = 0 if X is None else int(np.asarray(X).shape[-1])
  sd = outcome_sd_of(y, mask) if outcome_sd is None else float(outcome_sd)
  sigma_level = float(prior_level_sd) * sd                                   # :572
  half = _LEVEL_PRIOR_SAMPLE_SIZE / 2.0
  obs_prior = _OBS_PRIOR_WITH_COVARIATES if num_features else _OBS_PRIOR_NO_COVARIATES
  # Prior mean of the initial level is the first outcome value (:467-469).  The reference
  # feeds NaN there when y[0] is missing (undefined upstream); the first observed value is
  # used instead (DESIGN.md "Edge semantics").
  first = float(y[~mask][0]) if mask[0] else float(y[0])
  sigma_slope = (float(prior_slope_sd) if prior_slope_sd is not None else float(prior_level_sd)) * sd
  return dict(
      outcome_sd=sd,
      level_conc=half, level_scale=half * sigma_level ** 2, level_ub=sd,      # :424-432
      # LocalLinearTrend is not part of the reference's default model (:496); BASELINE cfg2
      # asks for it, so the slope variance gets the level's prior family.
      slope_conc=half, slope_scale=half * sigma_slope ** 2, slope_ub=sd,
      obs_conc=obs_prior[0], obs_scale=obs_prior[1] * sd * sd,                # :434-441
      obs_ub=_OBS_UPPER_BOUND * sd,                                           # :442-443
      drift_conc=_DRIFT_PRIOR[0], drift_scale=_DRIFT_PRIOR[1] * sd * sd, drift_ub=sd,  # :472-474
      nonzero_prob=min(1.0, _EXPECTED_MODEL_SIZE / num_features) if num_features else 1.0,  # :449
      init_level_loc=first, init_level_scale=sd,                              # :467-469
      init_slope_scale=sd, init_seasonal_scale=sd,                            # :489
      obs_scale0=(math.sqrt(1.0 - _EXPECTED_R2) * sd) if num_features else sd,  # :566-571
      level_scale0=sigma_level,                                               # :572
      slope_scale0=sigma_slope if has_slope else 0.0,                         # :373-374
      drift_scale0=[0.01 * sd] * int(num_seasonal_blocks),                    # :573-574
  )


def expand_seasons(seasons: Sequence, num_timesteps: int) -> Tuple[list, np.ndarray]:
  """[(num_seasons, num_steps_per_season)] -> (num_seasons list, uint8 [K, T] change flags)."""
  counts, flags = [], []
  for s in seasons:
    n = int(getattr(s, "num_seasons", s[0] if isinstance(s, (tuple, list)) else s))
    steps = getattr(s, "num_steps_per_season", s[1] if isinstance(s, (tuple, list)) else 1)
    if n < 2:
      raise ValueError("a seasonal effect needs at least 2 seasons")
    counts.append(n)
    flags.append(season_change_flags(num_timesteps, n, steps))
  arr = np.stack(flags) if flags else np.zeros((0, num_timesteps), np.uint8)
  return counts, arr
