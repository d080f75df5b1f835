"""On-device summarisation (csrc/ci_summary.h) against the host arithmetic
it replaces.  The device path orders its float64 operations like the numpy code and returns
order statistics (numpy's own interpolation is applied to them), so the two paths must agree
to round-off -- tolerance 1e-10 relative, far below any "float32 tolerance"."""
import numpy as np
import pandas as pd
import pytest

import causalimpact as ci
from causalimpact import _native, _model
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc
from test_golden_postprocessing import CASES, _load

pytestmark = pytest.mark.gpu


def _assert_frames_equal(a: pd.DataFrame, b: pd.DataFrame, rtol=1e-10):
  assert list(a.columns) == list(b.columns) and list(a.index) == list(b.index)
  for c in a.columns:
    if a[c].dtype.kind in "fc":
      np.testing.assert_allclose(a[c].to_numpy(float), b[c].to_numpy(float), rtol=rtol,
                                 atol=1e-12, equal_nan=True, err_msg=str(c))
    else:
      assert (a[c] == b[c]).all(), c


@pytest.mark.parametrize("name", CASES)
def test_device_summary_equals_host_postprocessing_on_reference_cases(name):
  """The reference's own edge cases (NaN outcomes, gap between periods, tail after the
  post-period, no covariates, integer index, unstandardised data)."""
  meta, _, df, pre, post = _load(name)
  kw = dict(alpha=meta["alpha"], seed=3,
            data_options=ci.DataOptions(standardize_data=meta["standardize"]))
  dev = ci.fit_causalimpact(df, pre, post, inference_options=ci.InferenceOptions(
      num_results=200, num_chains=3), **kw)
  host = ci.fit_causalimpact(df, pre, post, inference_options=ci.InferenceOptions(
      num_results=200, num_chains=3, summarize_on_device=False), **kw)
  _assert_frames_equal(dev.series, host.series)
  _assert_frames_equal(dev.summary, host.summary)
  assert ci.summary(dev) == ci.summary(host)


def test_order_statistics_and_running_sums_match_numpy():
  T, p, C, S = 257, 2, 5, 123          # N = 615 draws: not a multiple of anything
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 4)
  spec = orc.default_spec(y, mask, X)
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=0, num_warmup=20, num_results=S,
                            num_chains=C, seed=(9, 9))
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  sess.run()
  traj = sess.fetch(["posterior_trajectories"])["posterior_trajectories"].reshape(C * S, T)
  rng = np.random.default_rng(0)
  obs = rng.normal(size=T) * 3 + 10
  obs[[5, 200, 230]] = np.nan
  start, stop = 180, 240
  flags = (np.arange(T) >= start).astype(np.uint8) | (((np.arange(T) >= start) &
                                                       (np.arange(T) <= stop)).astype(np.uint8) << 1)
  ranks = [0, 7, 15, 307, 599, 606, 613, 614]
  scale, shift = 2.5, -1.25
  got = sess.summarize(scale, shift, obs, flags, ranks)
  sess.close()
  value = traj.astype(np.float64) * scale + shift                 # [N, T]
  np.testing.assert_array_equal(got["value_order"], np.sort(value, axis=0)[ranks])
  point = -(value - obs[None, :])
  base = np.where((np.arange(T) < start)[None, :], 0.0, point)
  holes = np.isnan(base)
  cum = np.cumsum(np.where(holes, 0.0, base), axis=1)
  cum[holes] = np.nan
  np.testing.assert_array_equal(got["cum_order"], np.sort(cum, axis=0)[ranks])
  win = (flags & 2) != 0
  np.testing.assert_array_equal(got["per_draw"][0], value.T[win].sum(axis=0))
  np.testing.assert_array_equal(got["per_draw"][1], np.nansum(point.T[win], axis=0))
  np.testing.assert_array_equal(got["per_draw_order"], np.sort(got["per_draw"], axis=1)[:, ranks])


def test_summarize_argument_errors():
  T = 50
  y, mask, X, _ = syn.make_sampler_inputs(T, 0, 1)
  spec = orc.default_spec(y, mask, X)
  pb = _native.make_problem(T=T, P=0, has_slope=0, num_warmup=2, num_results=8, num_chains=1)
  sess = _native.Session(pb, y[None], mask[None], None, None, _native.make_params([spec]))
  with pytest.raises(_native.NativeError, match="finished ci_session_run"):
    sess.summarize(1.0, 0.0, np.zeros(T), np.zeros(T, np.uint8), [0])
  sess.run()
  with pytest.raises(_native.NativeError, match="out of range"):
    sess.summarize(1.0, 0.0, np.zeros(T), np.zeros(T, np.uint8), [8])
  sess.close()


def test_host_pooled_draws_are_summarised_on_device_too():
  """Multi-device shares and the HMC path pool their draws on the host; `ci_summarize_draws`
  uploads them and gives the frames of the pure-host arithmetic."""
  import ref_pins_common as rp
  df = rp.create_test_data(8.0, 70, num_timesteps=100, seed=3)
  pre, post = (df.index[0], df.index[69]), (df.index[72], df.index[-2])
  for opts in (dict(num_results=120, num_chains=4, devices=[0, 0]),
               dict(num_results=60, num_warmup_steps=80, num_chains=3, sampler="hmc")):
    dev = ci.fit_causalimpact(df, pre, post, seed=2, inference_options=ci.InferenceOptions(**opts))
    host = ci.fit_causalimpact(df, pre, post, seed=2, inference_options=ci.InferenceOptions(
        summarize_on_device=False, **opts))
    _assert_frames_equal(dev.series, host.series)
    _assert_frames_equal(dev.summary, host.summary)


def test_float64_draws_are_summarised_without_a_float32_round_trip():
  """`ci_summarize_draws_f64` (round 5): float64 trajectories -- what `ci_fit_gibbs_f64` returns --
  go into the order statistics as they are.  Values that differ only beyond float32's 24 bits keep
  their order and their digits (the float32 entry point collapses them); the float64 fit through
  `fit_causalimpact(dtype=float64)` gives the frames of the host arithmetic."""
  N, T = 1500, 16
  rng = np.random.default_rng(5)
  traj = 1.0 + 1e-9 * rng.normal(size=(N, T))                    # spread far below float32 resolution
  obs = rng.normal(size=T)
  flags = np.zeros(T, np.uint8); flags[6:] = 1; flags[8:14] |= 2
  ranks = [0, 37, 749, 1462, 1499]
  got = _native.summarize_draws(traj, 2.0, 0.5, obs, flags, ranks)
  value = traj * 2.0 + 0.5
  np.testing.assert_array_equal(got["value_order"], np.sort(value, axis=0)[ranks])
  assert np.unique(got["value_order"][:, 0]).size == len(ranks)
  as32 = _native.summarize_draws(traj.astype(np.float32), 2.0, 0.5, obs, flags, ranks)
  assert np.unique(as32["value_order"][:, 0]).size == 1            # what the round trip loses
  import ref_pins_common as rp
  df = rp.create_test_data(8.0, 70, num_timesteps=100, seed=3)
  pre, post = (df.index[0], df.index[69]), (df.index[72], df.index[-2])
  opts = dict(num_results=100, num_chains=2)
  dev = ci.fit_causalimpact(df, pre, post, seed=2, data_options=ci.DataOptions(dtype=np.float64),
                            inference_options=ci.InferenceOptions(**opts))
  host = ci.fit_causalimpact(df, pre, post, seed=2, data_options=ci.DataOptions(dtype=np.float64),
                             inference_options=ci.InferenceOptions(summarize_on_device=False, **opts))
  _assert_frames_equal(dev.series, host.series)
  _assert_frames_equal(dev.summary, host.summary)


def _numpy_summary(traj, scale, shift, obs, flags, ranks):
  value = traj.astype(np.float64) * scale + shift                 # [N, T]
  point = -(value - obs[None, :])
  base = np.where(((flags & 1) == 0)[None, :], 0.0, point)
  holes = np.isnan(base)
  cum = np.cumsum(np.where(holes, 0.0, base), axis=1)
  cum[holes] = np.nan
  return np.sort(value, axis=0)[ranks], np.sort(cum, axis=0)[ranks]


@pytest.mark.parametrize("N", [1, 2, 615, 1024, 1025, 8000, 8192, 8193, 12345, 16384, 16385, 20000])
def test_select_is_exact_for_every_row_length_and_awkward_rows(N):
  """The order statistics are bit-equal to numpy's sort for every code path of the select
  (registers with 8 / 16 keys per thread, long rows from L2) and for rows built to stress it:
  constant rows, heavy ties, two-valued rows, one huge outlier, mixed signs and +-0, tiny spreads
  around a large mean (keys sharing 40+ leading bits), wide dynamic range."""
  T = 24
  rng = np.random.default_rng(N)
  traj = rng.normal(size=(N, T)).astype(np.float32)
  traj[:, 1] = 3.25                                               # constant row
  traj[:, 2] = rng.integers(0, 4, size=N)                         # heavy ties
  traj[:, 3] = np.where(rng.random(N) < 0.5, -1.0, 1.0)           # two values, mixed sign
  traj[:, 4] = rng.normal(size=N) * 1e-3 + 1000.0                 # tiny spread, large mean
  traj[:, 5] = rng.normal(size=N); traj[0, 5] = 1e30              # one outlier stretches the range
  traj[:, 6] = np.where(rng.random(N) < 0.5, -0.0, 0.0)           # +-0
  traj[:, 7] = np.exp(rng.normal(size=N) * 8)                     # 50 binades
  traj[:, 8] = -np.exp(rng.normal(size=N) * 8)
  traj[:, 9] = rng.normal(size=N) * 1e-30                         # near-denormal floats
  traj[:, 10] = np.round(rng.normal(size=N) * 3)                  # ties around zero, both signs
  traj[:, 11] = np.where(rng.random(N) < 0.999, 5.0, rng.normal(size=N))   # almost constant
  obs = rng.normal(size=T)
  obs[13] = np.nan
  flags = (np.arange(T) >= 6).astype(np.uint8) * 3
  ranks = sorted({0, N // 40, N // 2, (N - 1) // 2, N - 1 - N // 40, N - 1, min(N - 1, 1), max(0, N - 2)})
  got = _native.summarize_draws(traj, 1.0, 0.0, obs, flags, ranks)
  want_value, want_cum = _numpy_summary(traj, 1.0, 0.0, obs, flags, np.asarray(ranks))
  np.testing.assert_array_equal(got["value_order"], want_value)
  np.testing.assert_array_equal(got["cum_order"], want_cum)
  np.testing.assert_array_equal(got["per_draw_order"], np.sort(got["per_draw"], axis=1)[:, ranks])


def test_select_survives_infinities_and_random_shapes():
  """Rows with +-inf (non-finite rows leave the linear-bucket path for the radix select on the
  ordered keys), many random (N, T, ranks) shapes and value distributions: always numpy's sort."""
  rng = np.random.default_rng(123)
  for trial in range(24):
    N = int(rng.choice([1, 3, 64, 257, 1000, 4097, 8192, 9000, 16384, 17000]))
    T = int(rng.integers(1, 9))
    kind = trial % 6
    if kind == 0:
      traj = rng.normal(size=(N, T))
    elif kind == 1:
      traj = rng.standard_cauchy(size=(N, T))                      # heavy tails: outliers stretch the range
    elif kind == 2:
      traj = rng.integers(-3, 4, size=(N, T)).astype(float)        # ties
    elif kind == 3:
      traj = rng.normal(size=(N, T))
      traj[rng.integers(0, N, size=max(1, N // 50)), :] = np.inf   # +inf entries
      traj[rng.integers(0, N, size=max(1, N // 70)), 0] = -np.inf
    elif kind == 4:
      traj = np.exp(rng.normal(size=(N, T)) * 5) * rng.choice([-1.0, 1.0], size=(N, T))
    else:
      traj = np.full((N, T), 2.5) + (rng.random((N, T)) < 0.01)    # almost constant
    traj = traj.astype(np.float32)
    R = int(rng.integers(1, 9))
    ranks = sorted(int(r) for r in rng.integers(0, N, size=R))
    obs = np.zeros(T)
    flags = np.zeros(T, np.uint8)                                   # no effects: cum rows are zeros
    got = _native.summarize_draws(traj, 1.0, 0.0, obs, flags, ranks)
    want = np.sort(traj.astype(np.float64), axis=0)[ranks]
    np.testing.assert_array_equal(got["value_order"], want, err_msg=f"trial {trial} N={N} kind={kind}")
    np.testing.assert_array_equal(got["cum_order"], np.zeros_like(want))
