"""GPU parity of the full Gibbs fit (through the C-ABI) against the float64 CPU oracle.

Two kinds of checks:
  * per-draw: with the specified Philox stream the kernel and the oracle consume the same
    random numbers, so the first iterations agree to float32 accuracy (tolerance 5e-3 on
    O(1) quantities; discrete inclusion decisions must be identical);
  * posterior summaries over long runs agree within Monte-Carlo error.
"""
import numpy as np
import pytest

from causalimpact import _native
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc

pytestmark = pytest.mark.gpu


def _fit_both(T, p, has_slope, W, S, C=1, seed=(5, 9), data_seed=0, chain_offset=0):
  y, mask, X, _ = syn.make_sampler_inputs(T, p, data_seed)
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope))
  P = spec["P"]
  pb = _native.make_problem(T=T, P=P, has_slope=has_slope, num_warmup=W, num_results=S,
                            num_chains=C, chain_offset=chain_offset, seed=seed)
  got = _native.fit_gibbs(pb, y[None], mask[None], None if X is None else X[None], None,
                          _native.make_params([spec]))
  want = [orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=seed,
                        chain=chain_offset + c) for c in range(C)]
  return got, want, spec


@pytest.mark.parametrize("T,p,has_slope", [
    (100, 1, 0),    # reference default model, P=2 => every feature always included
    (100, 0, 0),    # no covariates: sigma_obs from the conjugate draw
    (300, 4, 1),    # P=5 => pi=0.6, inclusion flips, local linear trend
    (1000, 10, 1),  # BASELINE cfg2 shape
    (700, 10, 0),   # L=4 with padding, local level
    (2048, 10, 1),  # L=8 build of the eight-wave kernel (1024 < T <= 2048), design in LDS limit
    (1500, 12, 0),  # L=8, local level, 13 columns, ragged last chunk
    (4096, 10, 1),  # the register-resident regression block with the design streamed from L2
    (3001, 15, 0),  # ... 16 columns, T % 4 != 0 (scalar rows), ragged last chunk
    (500, 24, 0),   # P=25 > 16: LDS-resident regression block (in-place sweeps)
    (1000, 34, 1),  # P=35: the design no longer fits LDS and streams from L2, float4 rows
    (998, 33, 0),   # P=34, T % 4 != 0: the streamed design read row-scalar, ragged last chunk
    (5000, 3, 0),   # T > 4096: trend-only series on the time-parallel kernel (inert seasonal block)
    (9000, 2, 1),   # same with a local linear trend
    (40000, 1, 0),  # 157 steps per thread (the kernel's own limit is 65536 steps)
    (65536, 1, 0),  # ... and that limit: 256 steps per thread
    (400, 60, 0),   # P = 61 > 52: sequential kernel, regression block in the HBM workspace
    (300, 130, 1),  # P = 131: three rounds of 64 visiting positions, local linear trend
    (6000, 70, 0),  # big P and T > 4096 together (arrays over time in the workspace too)
    (700, 511, 0),  # P = 512: the largest design the C-ABI accepts (eight rounds of positions)
])
def test_first_iterations_match_oracle_per_draw(T, p, has_slope):
  S = 4
  got, want, spec = _fit_both(T, p, has_slope, W=0, S=S)
  w = want[0]
  P = spec["P"]
  for name, key in [("observation_noise_scale", "obs_scale"), ("level_scale", "level_scale")]:
    np.testing.assert_allclose(got[name][0, 0], w[key], rtol=5e-3, err_msg=name)
  if has_slope:
    np.testing.assert_allclose(got["slope_scale"][0, 0], w["slope_scale"], rtol=5e-3)
    np.testing.assert_allclose(got["slope"][0, 0], w["slope"], atol=5e-3)
  if P:
    np.testing.assert_array_equal(got["weights"][0, 0] != 0, w["weights"] != 0)
    np.testing.assert_allclose(got["weights"][0, 0], w["weights"], atol=5e-3)
  # float32 trend recursions over thousands of steps: tolerance relative to the path's range
  lev_tol = 5e-3 if T < 4096 else 5e-3 + 2e-3 * float(np.ptp(w["level"]))
  np.testing.assert_allclose(got["level"][0, 0], w["level"], atol=lev_tol)
  np.testing.assert_allclose(got["posterior_trajectories"][0, 0], w["trajectories"], atol=2 * lev_tol)
  np.testing.assert_allclose(got["posterior_means"][0, 0], w["pred_mean"], atol=lev_tol)


@pytest.mark.parametrize("T,p,has_slope", [
    (600, 30, 0),   # P = 31, n = 32: the largest matrix whose tiles fit one wavefront
    (500, 51, 1),   # P = 52, n = 53: the largest; 14 x 14 tiles over the workgroup, X from L2
    (400, 16, 1),   # P = 17, n = 18: the smallest
    (300, 31, 1),   # P = 32, n = 33: the first size on the whole workgroup
    (300, 46, 0),   # n = 48: a multiple of 16 (no padding rows behind the matrix)
    (300, 47, 1),   # n = 49
    (260, 19, 0),   # n = 21: tiles with three dead columns
])
def test_workgroup_wide_regression_block_follows_the_oracle_over_many_iterations(T, p, has_slope):
  """17-52 columns (spike_slab_draw_block): iterations WITHOUT an accepted flip take the weights
  from the recorded pivot rows of the register-tile sweeps, iterations with one from the explicit
  Cholesky route -- both occur within 40 iterations, and every draw of every iteration must be the
  oracle's: same inclusion pattern, weights, sigma_obs (float32 kernel vs float64 oracle on one
  random stream: the tolerance covers 40 iterations of drift)."""
  S = 40
  got, want, spec = _fit_both(T, p, has_slope, W=0, S=S)
  w = want[0]
  incl_dev, incl_orc = got["weights"][0, 0] != 0, w["weights"] != 0
  np.testing.assert_array_equal(incl_dev, incl_orc)
  changes = int((incl_orc[1:] != incl_orc[:-1]).any(axis=1).sum())
  assert 0 < changes < S - 1, changes            # both routes of the weights draw were taken
  np.testing.assert_allclose(got["weights"][0, 0], w["weights"], atol=2e-2)
  np.testing.assert_allclose(got["observation_noise_scale"][0, 0], w["obs_scale"], rtol=2e-2)
  np.testing.assert_allclose(got["posterior_means"][0, 0], w["pred_mean"], atol=2e-2)


def test_every_regression_block_size_from_17_to_52_columns_matches_the_oracle():
  """Every tile geometry of the register-resident sweeps (NB = 5 .. 14, one wavefront up to 31
  columns, the workgroup beyond) on a short ragged series: the first draws are the oracle's."""
  for p in range(16, 52):
    T = 90 + (p % 7)
    got, want, spec = _fit_both(T, p, p % 2, W=0, S=6)
    w = want[0]
    np.testing.assert_array_equal(got["weights"][0, 0] != 0, w["weights"] != 0, err_msg=f"p={p}")
    np.testing.assert_allclose(got["weights"][0, 0], w["weights"], atol=5e-3, err_msg=f"p={p}")
    np.testing.assert_allclose(got["observation_noise_scale"][0, 0], w["obs_scale"], rtol=5e-3,
                               err_msg=f"p={p}")
    assert np.isfinite(got["posterior_trajectories"]).all()


def test_chain_ids_do_not_depend_on_launch_split():
  # chains 0..3 in one call == chains {0,1} and {2,3} in two calls (multi-GPU sharding rule)
  T, p = 200, 3
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 3)
  spec = orc.default_spec(y, mask, X)
  params = _native.make_params([spec])
  def run(C, off):
    pb = _native.make_problem(T=T, P=spec["P"], has_slope=0, num_warmup=3, num_results=5,
                              num_chains=C, chain_offset=off, seed=(1, 1))
    return _native.fit_gibbs(pb, y[None], mask[None], X[None], None, params)
  whole, lo, hi = run(4, 0), run(2, 0), run(2, 2)
  for k in ("level", "weights", "observation_noise_scale", "posterior_trajectories"):
    np.testing.assert_array_equal(whole[k][0, :2], lo[k][0])
    np.testing.assert_array_equal(whole[k][0, 2:], hi[k][0])
  assert not np.array_equal(whole["level"][0, 0], whole["level"][0, 1])


def test_posterior_summaries_match_oracle_long_run():
  T, p, W, S, C = 200, 4, 150, 600, 4
  got, want, spec = _fit_both(T, p, 1, W, S, C=C, seed=(21, 4), data_seed=2)
  def pooled(key):
    return np.concatenate([w[key] for w in want], axis=0)
  for name, key in [("observation_noise_scale", "obs_scale"), ("level_scale", "level_scale"),
                    ("slope_scale", "slope_scale")]:
    g, o = got[name][0].ravel(), pooled(key)
    # north_star: posterior mean within 1 %; 3 % here covers MC error of 2400 correlated draws
    assert abs(g.mean() / o.mean() - 1) < 0.03, (name, g.mean(), o.mean())
  gw = got["weights"][0].reshape(-1, spec["P"])
  ow = pooled("weights")
  np.testing.assert_allclose(gw.mean(0), ow.mean(0), atol=0.02)
  np.testing.assert_allclose((gw != 0).mean(0), (ow != 0).mean(0), atol=0.06)
  gm = got["posterior_means"][0].mean(0)
  om = np.mean([w["pred_mean"] for w in want], axis=0)
  np.testing.assert_allclose(gm, om, atol=0.03)
  gt = got["posterior_trajectories"][0].reshape(-1, T)
  ot = pooled("trajectories")
  post = slice(int(0.7 * T), T)
  np.testing.assert_allclose(np.quantile(gt[:, post].mean(1), [0.025, 0.975]),
                             np.quantile(ot[:, post].mean(1), [0.025, 0.975]), atol=0.03)


def test_invariants_pinned_by_reference_tests():
  # causalimpact_lib_test.py:335-338: no NaNs, sigma_obs <= 1.2, sigma_level <= 1 (sd = 1)
  got, _, spec = _fit_both(91, 2, 0, W=100, S=10, seed=(1, 1))
  assert np.isfinite(got["level"]).all() and np.isfinite(got["weights"]).all()
  assert (got["observation_noise_scale"] <= 1.2 * spec["outcome_sd"] + 1e-6).all()
  assert (got["level_scale"] <= spec["outcome_sd"] + 1e-6).all()
  # :376-379  P <= 3 => no exactly-zero weights
  assert (got["weights"] == 0).sum() == 0


SEQ = _native.FLAG_SEQUENTIAL_SEASONAL


@pytest.mark.parametrize("T,p,has_slope,seasons,flags", [
    (60, 0, 0, ((3, 1),), 0),                           # smallest: one block, no regression
    (90, 1, 0, ((4, (2, 1, 1, 1)), (2, 3)), 0),         # per-season lengths + an n=2 block
    (120, 4, 1, ((7, 1),), SEQ),                        # weekly effect + trend + flips (P=5)
    (300, 0, 0, ((4, (2, 1, 1, 1)), (7, 1), (6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1)))), 0),
    # one weekly block: the time-parallel kernel (ci_wide.h)
    (120, 4, 1, ((7, 1),), 0),                          # same case as above, other kernel
    (120, 0, 0, ((7, 1),), 0),                          # no regression
    (300, 2, 0, ((7, 2),), 0),                          # two steps per season, P=3 (pi=1)
    (1030, 20, 0, ((7, 1),), 0),                        # P=21: LDS regression block, padded chunks
    (777, 6, 1, ((7, (1, 2, 1, 1, 1, 1, 3)),), 0),      # ragged seasons, trend
    (37, 0, 1, ((7, 1),), 0),                           # shorter than one chunk row, T % 4 != 0
    (64, 2, 0, ((7, 30),), 0),                          # long seasons: two changes in the series
    (200, 2, 1, ((4, 3),), 0),                          # quarterly-type block, trend (d = 5)
    (150, 0, 0, ((2, 1),), 0),                          # n = 2: a single free effect
    (330, 5, 0, ((5, 1),), 0), (96, 1, 1, ((3, 2),), 0), (240, 3, 0, ((6, (1, 1, 2, 1, 1, 3)),), 0),
    (350, 66, 0, ((7, 1),), 0),                         # P = 67 > 52 with a weekly block
    # the sequential kernel's wider covariance rows (8, 16, 32 registers more per lane)
    (200, 2, 0, ((12, 1),), 0),                         # monthly-type block: D = 13
    (260, 0, 1, ((24, 1),), 0),                         # hour-of-day block + trend: D = 26
    (300, 3, 0, ((52, 1),), 0),                         # week-of-year block: D = 53
    (250, 0, 1, ((62, 1),), 0),                         # D = 64: the widest state the kernel holds
    (280, 1, 0, ((24, 1), (7, 24)), 0),                 # hour-of-day and day-of-week: D = 32
    (280, 90, 1, ((4, 2), (7, 1)), 0),                  # P = 91, trend + two blocks
    # (general block lists with a state <= 32 take the time-parallel kernel of ci_seasonal_tp.h by
    # default since round 5; the sequential kernel keeps these cases through the flag)
    (90, 1, 0, ((4, (2, 1, 1, 1)), (2, 3)), SEQ),
    (300, 0, 0, ((4, (2, 1, 1, 1)), (7, 1), (6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1)))), SEQ),
    (260, 0, 1, ((24, 1),), SEQ),
    (280, 90, 1, ((4, 2), (7, 1)), SEQ),
])
def test_seasonal_first_iterations_match_oracle_per_draw(T, p, has_slope, seasons, flags):
  """Seasonal kernels vs the oracle's (n-1)-dimensional Durbin-Koopman draw, same random
  numbers: the one-wavefront sequential kernel (full n-effect state, fast state smoother) and
  the time-parallel kernel (chunked Sarkka scan in the oracle's coordinates)."""
  from causalimpact import _model
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  rng = np.random.default_rng(0)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0) + 0.1 * rng.normal(size=T)
  if T >= 100:                                          # missing outcomes inside the pre-period
    mask = mask.copy()
    mask[[2, 3, 40, T // 2]] = True
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  S, K = 4, len(seasons)
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_seasons=counts,
                            num_warmup=0, num_results=S, seed=(2, 6), flags=flags)
  got = _native.fit_gibbs(pb, y[None], mask[None], None if X is None else X[None], flg,
                          _native.make_params([spec]))
  w = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=0, seed=(2, 6))
  np.testing.assert_allclose(got["level"][0, 0], w["level"], atol=5e-3)
  np.testing.assert_allclose(got["seasonal_levels"][0, 0], w["seasonal"], atol=5e-3)
  np.testing.assert_allclose(got["seasonal_drift_scales"][0, 0], w["drift_scales"], rtol=2e-2)
  np.testing.assert_allclose(got["observation_noise_scale"][0, 0], w["obs_scale"], rtol=5e-3)
  np.testing.assert_allclose(got["level_scale"][0, 0], w["level_scale"], rtol=5e-3)
  if has_slope:
    np.testing.assert_allclose(got["slope"][0, 0], w["slope"], atol=5e-3)
  if spec["P"]:
    np.testing.assert_allclose(got["weights"][0, 0], w["weights"], atol=5e-3)
  np.testing.assert_allclose(got["posterior_trajectories"][0, 0], w["trajectories"], atol=1e-2)
  np.testing.assert_allclose(got["posterior_means"][0, 0], w["pred_mean"], atol=5e-3)
  assert got["seasonal_levels"].shape == (1, 1, S, T, K)


def test_time_parallel_and_sequential_seasonal_kernels_sample_the_same_posterior():
  """Beyond the first draws (where the two kernels agree per random number): long runs of the
  time-parallel kernel and of the sequential one give the same posterior summaries."""
  from causalimpact import _model
  T, p, seasons, S, C = 400, 3, ((7, 1),), 400, 4
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 21)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  spec = orc.default_spec(y, mask, X, has_slope=False, seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  res = {}
  for name, flags in (("wide", 0), ("seq", SEQ)):
    pb = _native.make_problem(T=T, P=spec["P"], has_slope=0, num_seasons=counts, num_warmup=100,
                              num_results=S, num_chains=C, seed=(8, 1), flags=flags)
    res[name] = _native.fit_gibbs(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
  w, q = res["wide"], res["seq"]
  for key, tol in (("observation_noise_scale", 0.02), ("level_scale", 0.05),
                   ("seasonal_drift_scales", 0.10)):
    np.testing.assert_allclose(w[key].mean(), q[key].mean(), rtol=tol, err_msg=key)
  np.testing.assert_allclose(w["weights"].mean(axis=(0, 1, 2)), q["weights"].mean(axis=(0, 1, 2)),
                             atol=0.02)
  np.testing.assert_allclose(w["posterior_means"].mean(axis=(0, 1)),
                             q["posterior_means"].mean(axis=(0, 1)), atol=0.05)
  np.testing.assert_allclose(w["seasonal_levels"].mean(axis=(0, 1, 2))[:, 0],
                             q["seasonal_levels"].mean(axis=(0, 1, 2))[:, 0], atol=0.03)


def test_baseline_cfg4_shape_matches_oracle_per_draw():
  """BASELINE 'T=10000, 50 covariates + Seasonal(num_seasons=7)' at full size: the first
  draws agree with the oracle per random number (level / seasonal effect to 1e-3, weights and
  inclusion pattern exactly)."""
  from causalimpact import _model
  T, p, seasons, S = 10000, 50, ((7, 1),), 2
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 11)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  spec = orc.default_spec(y, mask, X, has_slope=False, seasons=seasons)
  assert spec["P"] == 51
  counts, flg = _model.expand_seasons(seasons, T)
  pb = _native.make_problem(T=T, P=51, has_slope=0, num_seasons=counts, num_warmup=0,
                            num_results=S, seed=(4, 4))
  got = _native.fit_gibbs(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
  w = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=0, seed=(4, 4))
  np.testing.assert_array_equal(got["weights"][0, 0] != 0, w["weights"] != 0)
  np.testing.assert_allclose(got["weights"][0, 0], w["weights"], atol=1e-3)
  np.testing.assert_allclose(got["level"][0, 0], w["level"], atol=1e-3)
  np.testing.assert_allclose(got["seasonal_levels"][0, 0], w["seasonal"], atol=1e-3)
  np.testing.assert_allclose(got["seasonal_drift_scales"][0, 0], w["drift_scales"], rtol=2e-2)
  np.testing.assert_allclose(got["observation_noise_scale"][0, 0], w["obs_scale"], rtol=5e-3)
  np.testing.assert_allclose(got["posterior_trajectories"][0, 0], w["trajectories"], atol=5e-3)


def test_batch_of_series_equals_single_series_runs():
  """BASELINE cfg5 shape in miniature: B independent series in one launch (one workgroup per
  (series, chain)) give exactly what B separate launches give.  Random streams are keyed by
  (global series id, chain): a single-series launch with series_offset = b reproduces series b
  of the batch, i.e. the result does not depend on how a batch is split over launches / GPUs;
  CI_FLAG_SHARED_SERIES_STREAMS keys them by chain only (series b == a plain single fit)."""
  T, p, B, C = 120, 3, 5, 2
  ys, masks, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 100 + b)
    ys.append(y); masks.append(mask); Xs.append(X)
    specs.append(orc.default_spec(y, mask, X))
  P = specs[0]["P"]
  pb = _native.make_problem(T=T, P=P, has_slope=0, num_warmup=4, num_results=6, num_chains=C,
                            num_series=B, seed=(8, 8))
  batch = _native.fit_gibbs(pb, np.stack(ys), np.stack(masks), np.stack(Xs), None,
                            _native.make_params(specs))
  for b in range(B):
    pb1 = _native.make_problem(T=T, P=P, has_slope=0, num_warmup=4, num_results=6, num_chains=C,
                               num_series=1, seed=(8, 8), series_offset=b)
    one = _native.fit_gibbs(pb1, ys[b][None], masks[b][None], Xs[b][None], None,
                            _native.make_params([specs[b]]))
    for k in ("level", "weights", "observation_noise_scale", "posterior_trajectories",
              "posterior_means"):
      np.testing.assert_array_equal(batch[k][b], one[k][0], err_msg=f"{k} series {b}")
  assert not np.array_equal(batch["level"][0], batch["level"][1])
  # identical series: different draws by default, identical draws with shared streams
  twin = dict(y=np.stack([ys[1], ys[1]]), m=np.stack([masks[1], masks[1]]), X=np.stack([Xs[1], Xs[1]]))
  for flags, same in ((0, False), (_native.FLAG_SHARED_SERIES_STREAMS, True)):
    pb2 = _native.make_problem(T=T, P=P, has_slope=0, num_warmup=4, num_results=6, num_chains=C,
                               num_series=2, seed=(8, 8), flags=flags)
    tw = _native.fit_gibbs(pb2, twin["y"], twin["m"], twin["X"], None,
                           _native.make_params([specs[1], specs[1]]))
    assert np.array_equal(tw["level"][0], tw["level"][1]) == same
  # shared streams: series 1 of that launch is the plain single-series fit of the same data
  pb3 = _native.make_problem(T=T, P=P, has_slope=0, num_warmup=4, num_results=6, num_chains=C,
                             seed=(8, 8))
  plain = _native.fit_gibbs(pb3, ys[1][None], masks[1][None], Xs[1][None], None,
                            _native.make_params([specs[1]]))
  np.testing.assert_array_equal(tw["level"][1], plain["level"][0])


@pytest.mark.gpu
def test_seasonal_batch_larger_than_the_device_equals_single_series_runs():
  """The route of a seasonal model is a function of (T, D, P) alone: a batch with MORE chains than
  the device has CUs (66 series x 4 chains = 264 workgroups) runs every series on the kernel a
  single-series launch picks, so series b of the batch reproduces the single launch with
  series_offset = b bit for bit (round-5 advisor: the route used to flip to the sequential kernel
  at B*C > CU count, and the two agree only up to float summation order)."""
  from causalimpact import _model
  T, p, B, C, seasons = 130, 2, 66, 4, ((5, 1), (3, 5))
  counts, flg = _model.expand_seasons(seasons, T)
  ys, masks, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 300 + b)
    y = y + 0.5 * np.sin(2 * np.pi * np.arange(T) / 5.0)
    ys.append(y); masks.append(mask); Xs.append(X)
    specs.append(orc.default_spec(y, mask, X, has_slope=True, seasons=seasons))
  P = specs[0]["P"]
  kw = dict(T=T, P=P, has_slope=1, num_seasons=counts, num_warmup=2, num_results=4, num_chains=C, seed=(5, 9))
  sess = _native.Session(_native.make_problem(num_series=B, **kw), np.stack(ys), np.stack(masks), np.stack(Xs),
                         flg, _native.make_params(specs))
  name = sess.kernel_name()
  assert "gibbs_seasonal_tp_kernel" in name
  sess.run()
  batch = sess.fetch()
  sess.close()
  for b in (0, 37, B - 1):
    one_s = _native.Session(_native.make_problem(num_series=1, series_offset=b, **kw), ys[b][None], masks[b][None],
                            Xs[b][None], flg, _native.make_params([specs[b]]))
    assert one_s.kernel_name().split(" ")[0] == name.split(" ")[0]
    one_s.run()
    one = one_s.fetch()
    one_s.close()
    for k in ("level", "weights", "observation_noise_scale", "seasonal_levels", "posterior_trajectories"):
      np.testing.assert_array_equal(batch[k][b], one[k][0], err_msg=f"{k} series {b}")


def test_baseline_cfg5_full_size_batch_equals_single_series_runs_and_oracle():
  """BASELINE 'batch of 512 independent series, T=500, 5 covariates' at full size (fewer
  iterations): one launch of 512 workgroups; spot-checked series equal their own single-series
  launch bit for bit and the oracle per draw."""
  T, p, B, W, S = 500, 5, 512, 3, 5
  ys, masks, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 1000 + b)
    ys.append(y); masks.append(mask); Xs.append(X)
    specs.append(orc.default_spec(y, mask, X))
  pb = _native.make_problem(T=T, P=6, has_slope=0, num_warmup=W, num_results=S, num_series=B,
                            seed=(5, 12))
  batch = _native.fit_gibbs(pb, np.stack(ys), np.stack(masks), np.stack(Xs), None,
                            _native.make_params(specs),
                            want=("level", "weights", "observation_noise_scale"))
  assert np.isfinite(batch["level"]).all()
  for b in (0, 255, 511):
    # the 512-workgroup launch runs the four-wavefront build, a single series the eight-wavefront
    # latency build: the SAME bits (the library is compiled with -ffp-contract=on, so shared
    # source rounds identically in both kernels)
    pb1 = _native.make_problem(T=T, P=6, has_slope=0, num_warmup=W, num_results=S, seed=(5, 12),
                               series_offset=b)
    s1 = _native.Session(pb1, ys[b][None], masks[b][None], Xs[b][None], None,
                         _native.make_params([specs[b]]))
    assert "gibbs_kernel8" in s1.kernel_name()
    s1.run()
    one = s1.fetch(want=("level", "weights", "observation_noise_scale"))
    s1.close()
    for k in ("level", "weights", "observation_noise_scale"):
      np.testing.assert_array_equal(batch[k][b], one[k][0], err_msg=f"{k} series {b}")
    # the stream of series b: both key words carry the mixed series id (ci_series_stream_key;
    # counter word 3 = the chain id)
    w = orc.fit_gibbs(ys[b], masks[b], Xs[b], specs[b], num_results=S, num_warmup=W,
                      seed=_native.series_stream_key((5, 12), b), chain=0)
    np.testing.assert_allclose(batch["level"][b, 0], w["level"], atol=5e-3)
    np.testing.assert_allclose(batch["weights"][b, 0], w["weights"], atol=5e-3)


def test_fit_causalimpact_over_several_device_shares_equals_one_launch():
  """`InferenceOptions.devices` shards chains over devices, one host thread each.  With the
  single test GPU listed twice the two shares run concurrently on it; pooled draws must be the
  ones of a single launch (chain ids keep their RNG streams)."""
  import pandas as pd
  import causalimpact as ci
  y, X = syn.make_raw_series(160, 2, 3)
  df = pd.DataFrame(np.column_stack([y, X]), columns=["y", "x0", "x1"])
  kw = dict(seed=11, model_options=ci.ModelOptions(local_linear_trend=True))
  one = ci.fit_causalimpact(df, (0, 109), (110, 159), inference_options=ci.InferenceOptions(
      num_results=60, num_chains=4, summarize_on_device=False), **kw)
  two = ci.fit_causalimpact(df, (0, 109), (110, 159), inference_options=ci.InferenceOptions(
      num_results=60, num_chains=4, devices=[0, 0]), **kw)
  np.testing.assert_array_equal(one.posterior_samples.level, two.posterior_samples.level)
  np.testing.assert_array_equal(one.posterior_samples.weights, two.posterior_samples.weights)
  np.testing.assert_allclose(one.summary.to_numpy(float), two.summary.to_numpy(float), rtol=1e-12)


@pytest.mark.parametrize("pinned,chunk", [(True, 7), (True, 1000), (False, 50)])
def test_streamed_fetch_equals_run_then_fetch(pinned, chunk):
  """ci_session_run_streamed: results are copied to the host in chunks WHILE the persistent kernel
  is still sampling (the workgroup publishes its progress after a system-scope release).  Every
  byte must equal what run() + fetch() return, for pinned and pageable destinations, chunk sizes
  that do and do not divide S, several series and chains (uneven progress between workgroups)."""
  T, p, B, C, W, S = 1000, 10, 3, 5, 20, 333
  ys, masks, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 40 + b)
    ys.append(y); masks.append(mask); Xs.append(X)
    specs.append(orc.default_spec(y, mask, X, has_slope=True))
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=W, num_results=S, num_chains=C,
                            num_series=B, seed=(2, 9))
  sess = _native.Session(pb, np.stack(ys), np.stack(masks), np.stack(Xs), None,
                         _native.make_params(specs))
  sess.run()
  want = sess.fetch()
  for _ in range(2):                      # the second call re-uses recycled pinned buffers
    ms, got = sess.run_streamed(chunk_draws=chunk, pinned=pinned)
    assert ms > 0
    for k, v in want.items():
      np.testing.assert_array_equal(got[k], v, err_msg=k)
  sess.close()


def test_streamed_fetch_of_a_seasonal_model_equals_run_then_fetch():
  """Models on the seasonal kernels do not publish progress: same chunked copies after the
  kernel has finished, same bytes."""
  T, p, S = 700, 3, 40
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 3)
  spec = orc.default_spec(y, mask, X, seasons=((7, 1),))
  pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_seasons=(7,), num_warmup=5,
                            num_results=S, num_chains=2, seed=(2, 9))
  sess = _native.Session(pb, y[None], mask[None], X[None], np.stack(spec["season_change"]),
                         _native.make_params([spec]))
  sess.run()
  want = sess.fetch()
  _, got = sess.run_streamed(chunk_draws=16)
  for k, v in want.items():
    np.testing.assert_array_equal(got[k], v, err_msg=k)
  sess.close()


@pytest.mark.parametrize("T,p,has_slope,B,C", [(1000, 10, 1, 1, 3), (500, 5, 0, 4, 2), (100, 1, 0, 1, 2),
                                               (300, 15, 1, 1, 2),
                                               (2000, 10, 1, 1, 2),    # L = 8 (1024 < T <= 2048)
                                               (1300, 6, 0, 2, 2),
                                               # L = 16, the design streamed from L2 by both kernels:
                                               (4096, 10, 1, 1, 2),    # float4 rows
                                               (4094, 7, 0, 1, 2)])    # T % 4 != 0: guarded scalar reads
                                                                       # (rolled in the four-wave build)
def test_eight_wave_latency_kernel_equals_the_four_wave_kernel(T, p, has_slope, B, C):
  """gibbs_kernel8 (a dedicated regression wavefront that sweeps the next iteration's matrix
  during the Durbin-Koopman draw and replays the recorded multipliers on the new right-hand
  side, csrc/ci_kernels8.h; three more wavefronts produce every random number one iteration ahead) against gibbs_kernel<D, L, 1> (CI_FLAG_FOUR_WAVES).  Same operations in
  the same order on the same random numbers, and -- the library being compiled with
  -ffp-contract=on, where fusion is a property of the source expression, not of the inlining
  context -- the same roundings: EVERY output of every draw is bit-identical.  Covers warm-up with
  accepted inclusion flips (fall-back route), P <= 3 (all features always in), several series and
  chains."""
  ys, masks, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 60 + b)
    ys.append(y); masks.append(mask); Xs.append(X)
    specs.append(orc.default_spec(y, mask, X, has_slope=bool(has_slope)))
  out = {}
  for flags in (0, _native.FLAG_FOUR_WAVES):
    pb = _native.make_problem(T=T, P=p + 1, has_slope=has_slope, num_warmup=0, num_results=120,
                              num_chains=C, num_series=B, seed=(6, 2), flags=flags)
    sess = _native.Session(pb, np.stack(ys), np.stack(masks), np.stack(Xs), None,
                           _native.make_params(specs))
    out[flags] = (sess.kernel_name(), sess.run(), sess.fetch())
    sess.close()
  assert "gibbs_kernel8" in out[0][0] and "gibbs_kernel<" in out[_native.FLAG_FOUR_WAVES][0]
  five, four = out[0][2], out[_native.FLAG_FOUR_WAVES][2]
  for k, v in four.items():
    np.testing.assert_array_equal(five[k], v, err_msg=k)
  w = five["weights"]
  assert np.isfinite(w).all() and (w != 0).any()
  if p >= 5:   # inclusion patterns do change during the run: the fall-back route is exercised
    incl = (w != 0)
    assert (incl[:, :, 1:] != incl[:, :, :-1]).any()


def test_series_and_chain_ids_beyond_16_bits_have_their_own_streams():
  """Series ids enter the Philox key (mixed into both words), chain ids the counter word: neither is
  limited to 16 bits any more (rounds 1-3 packed both into one counter word).  Series 70001 with
  chain ids 66000, 66001, against the oracle on the same stream."""
  T, p, W, S = 160, 3, 4, 6
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 31)
  spec = orc.default_spec(y, mask, X, has_slope=True)
  pb = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=W, num_results=S, num_chains=2,
                            chain_offset=66000, seed=(9, 77), series_offset=70001)
  got = _native.fit_gibbs(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  for c in range(2):
    want = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=W, seed=_native.series_stream_key((9, 77), 70001),
                         chain=66000 + c)
    np.testing.assert_array_equal(got["weights"][0, c] != 0, want["weights"] != 0)
    np.testing.assert_allclose(got["level"][0, c], want["level"], atol=5e-3)
    np.testing.assert_allclose(got["observation_noise_scale"][0, c], want["obs_scale"], rtol=5e-3)
  # and it is a different stream from series 0 / chain 0
  pb0 = _native.make_problem(T=T, P=p + 1, has_slope=1, num_warmup=W, num_results=S, num_chains=1,
                             seed=(9, 77))
  base = _native.fit_gibbs(pb0, y[None], mask[None], X[None], None, _native.make_params([spec]))
  assert np.abs(base["level"][0, 0] - got["level"][0, 0]).max() > 1e-3


def test_batches_under_different_seeds_never_share_a_stream():
  """with the series id XORed into seed1, series b under seed s drew from the
  stream of series b ^ s ^ s' under seed s' -- a batch under seed 1 was a permutation of the same
  batch under seed 0.  The id is now mixed into BOTH key words by bijections that fix 0: the keys
  of 4096 series under seeds 0..7 are all distinct, series 0 keeps the plain seeds, and a device
  fit of series 1 under seed 0 differs from a single-series fit under seed 1."""
  keys = set()
  for s in range(8):
    assert _native.series_stream_key(s, 0) == (0, s)
    for b in range(4096):
      keys.add(_native.series_stream_key(s, b))
  assert len(keys) == 8 * 4096
  T, p = 120, 2
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 3)
  spec = orc.default_spec(y, mask, X, has_slope=False)
  def fit(seed, series_offset):
    pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=2, num_results=3, num_chains=1,
                              seed=seed, series_offset=series_offset)
    return _native.fit_gibbs(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  a, b = fit(0, 1), fit(1, 0)
  assert np.abs(a["level"] - b["level"]).max() > 1e-3


@pytest.mark.parametrize("T,p,has_slope", [(1000, 10, 1), (500, 5, 0), (180, 2, 1)])
def test_eight_wave_kernel_does_not_depend_on_the_schedule_of_its_helper_waves(T, p, has_slope, monkeypatch):
  """The regression wave's precompute and the randomness waves' rounds are background work placed
  between the workgroup barriers by a schedule word (csrc/ci_kernels8.h SCHED_DEFAULT).  Whatever
  the schedule -- everything as early as possible ($CI_SCHED_WORD=1), everything after the last barrier
  (2), or other quotas -- every output of every draw must be bit-identical: a difference would mean
  a buffer is read before it is complete or overwritten while still in use."""
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 11)
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope))
  pb = _native.make_problem(T=T, P=p + 1, has_slope=has_slope, num_warmup=10, num_results=150,
                            num_chains=2, seed=(8, 1))
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  assert "gibbs_kernel8" in sess.kernel_name()
  out = {}
  for word in ("0", "1", "2", str((3 << 4) | (3 << 6) | (1 << 8) | (3 << 10) | (1 << 12) | (1 << 14)),
               str((0 << 4) | (0 << 6) | (3 << 8) | (0 << 10) | (3 << 12) | (0 << 14))):
    monkeypatch.setenv("CI_SCHED_WORD", word)
    sess.run()
    out[word] = sess.fetch()
  sess.close()
  for word, got in out.items():
    for k, v in out["0"].items():
      np.testing.assert_array_equal(got[k], v, err_msg=f"schedule {word}: {k}")


def test_a_batch_split_over_launches_of_any_size_gives_the_bits_of_one_launch():
  """identical outputs regardless of the number of GPUs.  512 series in ONE
  launch (more workgroups than CUs: the four-wavefront throughput build) against the same batch as
  8 launches of 64 series (`series_offset`; every chain has a CU: the eight-wavefront latency
  build) -- what 8 GPUs would run -- and against an uneven 300 + 212 split that straddles the
  CU count: all draws bit-equal."""
  T, p, B, W, S = 200, 5, 512, 4, 6
  ys, masks, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 3000 + b)
    ys.append(y); masks.append(mask); Xs.append(X)
    specs.append(orc.default_spec(y, mask, X))
  ys, masks, Xs = np.stack(ys), np.stack(masks), np.stack(Xs)
  want_keys = ("level", "weights", "observation_noise_scale", "level_scale", "posterior_trajectories")

  def launch(first, count):
    pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_warmup=W, num_results=S,
                              num_series=count, series_offset=first, seed=(8, 1))
    sess = _native.Session(pb, ys[first:first + count], masks[first:first + count],
                           Xs[first:first + count], None,
                           _native.make_params(specs[first:first + count]))
    name = sess.kernel_name()
    sess.run()
    out = sess.fetch(want=want_keys)
    sess.close()
    return name, out

  name_all, whole = launch(0, B)
  assert "gibbs_kernel<" in name_all
  names = set()
  for parts in ([(64 * g, 64) for g in range(8)], [(0, 300), (300, 212)]):
    for first, count in parts:
      name, part = launch(first, count)
      names.add(name.split("<")[0])
      for k in want_keys:
        np.testing.assert_array_equal(part[k], whole[k][first:first + count],
                                      err_msg=f"{k} series {first}..{first + count}")
  assert names == {"ci::gibbs_kernel8", "ci::gibbs_kernel"}       # both builds took part


REF_SEASONS = ((4, (2, 1, 1, 1)), (7, 1), (6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1))))
WS = _native.FLAG_SEASONAL_WORKSPACE


@pytest.mark.parametrize("T,p,has_slope,seasons,flags", [
    (300, 3, 0, REF_SEASONS, WS),            # the reference's 4+7+6 test model, arrays in HBM
    (300, 20, 0, REF_SEASONS, SEQ),          # P = 21 > 16: LDS one-wavefront regression block
    (2500, 12, 1, REF_SEASONS, SEQ),         # beyond the LDS bound of round 1 (T <~ 890 at D = 18)
    (5000, 30, 0, ((12, 30),), SEQ),         # a 12-season block (not on ci_wide.h's kernel), P = 31
    # the same models on the TIME-PARALLEL kernel (ci_seasonal_tp.h, the default route since round 5)
    (300, 20, 0, REF_SEASONS, 0),            # 32 chunks of 12 steps, LDS regression block on wave 0
    (2500, 12, 1, REF_SEASONS, 0),           # D = 19 with a slope: 128 chunks
    (5000, 30, 0, ((12, 30),), 0),           # two changes per chunk at most
    (10000, 50, 0, REF_SEASONS, 0),          # BASELINE cfg4's size with the reference's 4+7+6 model
    (1200, 8, 1, ((24, 1), (7, 24)), 0),     # hour-of-day + day-of-week + trend: D = 33 > 32 -> sequential
    (1200, 8, 0, ((24, 1), (7, 24)), 0),     # ... without the slope: D = 32, four register chunks
    (900, 70, 0, REF_SEASONS, 0),            # P = 71 > 52: the workspace regression block on wave 0
    # every register width of the time-parallel kernel (8..32 columns in steps of 4) has a case:
    (112, 3, 0, REF_SEASONS, 0),             # the shortest series on this route (4 chunks of 28 steps)
    (130, 2, 1, ((5, 1), (3, 5)), 0),        # D = 10: 12 columns
    (400, 4, 1, ((12, 1), (7, 12)), 0),      # D = 21: 24 columns
    (400, 4, 1, ((24, 1),), 0),              # D = 26: 28 columns
])
def test_general_seasonal_models_beyond_the_lds_bound_match_oracle_per_draw(T, p, has_slope, seasons,
                                                                           flags):
  """Models other than trend + one 2..7-season block -- the reference's own 4+7+6-season model
  (causalimpact_lib_test.py:740-752) with covariates, at lengths round 1 rejected -- on both
  routes: the one-wavefront sequential kernel (arrays over time in the per-chain HBM workspace,
  LDS-resident regression block for P > 16) and the time-parallel kernel of round 5 (chunks of
  the series on the wavefronts of a cluster of workgroups, wave-cooperative scan elements)."""
  from causalimpact import _model
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 7)
  rng = np.random.default_rng(0)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0) + 0.1 * rng.normal(size=T)
  mask = mask.copy()
  mask[[2, 3, 40, T // 2]] = True
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  S, K = 3, len(seasons)
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_seasons=counts,
                            num_warmup=0, num_results=S, seed=(2, 6), flags=flags)
  sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
  dfull = 1 + has_slope + sum(c for c in counts)
  if flags == 0 and dfull <= 32:
    assert "gibbs_seasonal_tp_kernel" in sess.kernel_name()
  else:
    assert "gibbs_seasonal_kernel" in sess.kernel_name()
    if flags & WS or T > 1000:
      assert "<true," in sess.kernel_name()
  sess.run()
  got = sess.fetch()
  sess.close()
  w = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=0, seed=(2, 6))
  tol = 5e-3 + 2e-3 * float(np.ptp(w["level"]))
  np.testing.assert_allclose(got["level"][0, 0], w["level"], atol=tol)
  np.testing.assert_allclose(got["seasonal_levels"][0, 0], w["seasonal"], atol=tol)
  np.testing.assert_allclose(got["seasonal_drift_scales"][0, 0], w["drift_scales"], rtol=2e-2)
  np.testing.assert_allclose(got["observation_noise_scale"][0, 0], w["obs_scale"], rtol=5e-3)
  np.testing.assert_allclose(got["level_scale"][0, 0], w["level_scale"], rtol=5e-3)
  np.testing.assert_array_equal(got["weights"][0, 0] != 0, w["weights"] != 0)
  np.testing.assert_allclose(got["weights"][0, 0], w["weights"], atol=5e-3)
  np.testing.assert_allclose(got["posterior_trajectories"][0, 0], w["trajectories"], atol=2 * tol)
  assert got["seasonal_levels"].shape == (1, 1, S, T, K)


def test_seasonal_kernel_with_arrays_in_hbm_equals_the_lds_variant():
  """Same kernel source, arrays over time in LDS vs in the HBM workspace: identical draws."""
  from causalimpact import _model
  T, p = 200, 4
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 9)
  spec = orc.default_spec(y, mask, X, seasons=REF_SEASONS)
  counts, flg = _model.expand_seasons(REF_SEASONS, T)
  out = []
  for flags in (SEQ, SEQ | WS):
    pb = _native.make_problem(T=T, P=p + 1, has_slope=0, num_seasons=counts, num_warmup=5,
                              num_results=20, num_chains=2, seed=(3, 3), flags=flags)
    out.append(_native.fit_gibbs(pb, y[None], mask[None], X[None], flg, _native.make_params([spec])))
  for k, v in out[0].items():
    np.testing.assert_array_equal(out[1][k], v, err_msg=k)


def test_clusters_of_workgroups_give_the_same_bits_as_one_workgroup_per_chain():
  """Time-parallel seasonal kernel: a chain's X~'targets / emission / X w phases are shared by
  16, 8, 4 or 2 workgroups when the launch leaves CUs idle (ci_wide.h "clusters"), and the
  Durbin-Koopman draw runs on the first 8, 4, 4 or 2 of them (ci_wide_quad.h: eight virtual
  workgroups of 64 quads of lanes, however many real ones carry them).  The reductions run over
  fixed segments and the draw's chunk grid is fixed by T, so every cluster size -- chosen from the
  number of chains -- and the single-workgroup kernel produce identical draws."""
  from causalimpact import _model
  T, p, seasons, W, S = 4800, 20, ((7, 1),), 3, 6            # 5 segments of 1024 steps
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 5)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  mask = mask.copy()
  mask[[5, 1000, 2047, 2048]] = True
  spec = orc.default_spec(y, mask, X, has_slope=False, seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)

  def fit(C, flags):
    pb = _native.make_problem(T=T, P=spec["P"], has_slope=0, num_seasons=counts, num_warmup=W,
                              num_results=S, num_chains=C, seed=(4, 2), flags=flags)
    return _native.fit_gibbs(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))

  one = fit(2, _native.FLAG_NO_CLUSTER)
  sixteen = fit(2, 0)                                         # 8 x 16 workgroups <= 256 CUs
  eight = fit(20, 0)                                          # 24 x 16 > 256 >= 24 x 8
  four = fit(40, 0)                                           # 40 x 8 > 256 >= 40 x 4
  two = fit(72, 0)                                            # 72 x 4 > 256 >= 72 x 2
  # a cluster that does not assemble (one helper "never scheduled"): main notices and runs alone
  lonely = fit(2, _native.FLAG_TEST_DROP_HELPER)
  for key in one:
    np.testing.assert_array_equal(lonely[key], one[key], err_msg=key)
    np.testing.assert_array_equal(sixteen[key], one[key], err_msg=key)
    np.testing.assert_array_equal(eight[key][:, :2], one[key], err_msg=key)
    np.testing.assert_array_equal(four[key][:, :2], one[key], err_msg=key)
    np.testing.assert_array_equal(two[key][:, :2], one[key], err_msg=key)
  assert np.isfinite(one["posterior_trajectories"]).all()
  assert (one["weights"] == 0).any() and (one["weights"] != 0).any()


def test_concurrent_cluster_launches_on_one_gpu_neither_hang_nor_change_the_draws():
  """Two device shares of a weekly-seasonal fit run concurrently on ONE GPU (`devices=[0, 0]`):
  together their clusters ask for more workgroups than there are CUs, so some clusters cannot
  assemble and fall back to one workgroup per chain (ci_wide.h, cl_assemble).  Nothing may
  hang and the pooled draws must be those of the single launch."""
  import pandas as pd
  import causalimpact as ci
  T, p = 2000, 20
  y, X = syn.make_raw_series(T, p, 3)
  y = y + 2.0 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  df = pd.DataFrame(np.column_stack([y, X]), columns=["y"] + [f"x{j}" for j in range(p)])
  kw = dict(seed=5, model_options=ci.ModelOptions(seasons=[ci.Seasons(num_seasons=7)]))
  opts = dict(num_results=40, num_warmup_steps=10, num_chains=48)
  one = ci.fit_causalimpact(df, (0, 1399), (1400, T - 1),
                            inference_options=ci.InferenceOptions(devices=[0], **opts), **kw)
  for _ in range(3):
    two = ci.fit_causalimpact(df, (0, 1399), (1400, T - 1),
                              inference_options=ci.InferenceOptions(devices=[0, 0], **opts), **kw)
    for f in ("observation_noise_scale", "level_scale", "level", "weights", "seasonal_levels"):
      np.testing.assert_array_equal(getattr(two.posterior_samples, f),
                                    getattr(one.posterior_samples, f), err_msg=f)


def test_cluster_handshakes_under_contention_give_the_bits_of_one_workgroup_per_chain():
  """Round-5 advisor (high): on one XCD the cluster handshakes publish data with a relaxed flag, so
  every wave must have drained its stores to the shared L2 before the arrival -- a missing drain
  shows only under load, as silently different draws.  Both cluster kernels (trend + weekly block:
  ci_wide.h, sixteen workgroups per chain with the draw's rows in LDS; a two-block model:
  ci_seasonal_tp.h) run ten times each WHILE another thread keeps a batch of 160 series running on
  its own stream (L2 / HBM / CU contention; when the cluster cannot assemble the fall-back runs --
  same bits either way), and every run must equal the one-workgroup-per-chain launch bit for bit."""
  import threading
  from causalimpact import _model
  stop = threading.Event()

  def contention():
    T, p, B = 500, 5, 160
    ys, ms, Xs, specs = [], [], [], []
    for b in range(B):
      y, mask, X, _ = syn.make_sampler_inputs(T, p, 900 + b)
      ys.append(y); ms.append(mask); Xs.append(X)
      specs.append(orc.default_spec(y, mask, X))
    pb = _native.make_problem(T=T, P=specs[0]["P"], has_slope=0, num_warmup=10, num_results=200,
                              num_chains=1, num_series=B, seed=(1, 1))
    sess = _native.Session(pb, np.stack(ys), np.stack(ms), np.stack(Xs), None, _native.make_params(specs))
    while not stop.is_set():
      sess.run()
    sess.close()

  th = threading.Thread(target=contention)
  th.start()
  try:
    for T, p, seasons in ((4800, 20, ((7, 1),)), (1300, 6, ((4, 1), (7, 4)))):
      y, mask, X, _ = syn.make_sampler_inputs(T, p, 5)
      y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
      spec = orc.default_spec(y, mask, X, has_slope=False, seasons=seasons)
      counts, flg = _model.expand_seasons(seasons, T)

      def session(flags):
        pb = _native.make_problem(T=T, P=spec["P"], has_slope=0, num_seasons=counts, num_warmup=3,
                                  num_results=12, num_chains=2, seed=(4, 2), flags=flags)
        return _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))

      ref_s = session(_native.FLAG_NO_CLUSTER)
      ref_s.run()
      ref = ref_s.fetch()
      ref_s.close()
      sess = session(0)
      for rep in range(10):
        sess.run()
        got = sess.fetch()
        for key in ref:
          np.testing.assert_array_equal(got[key], ref[key], err_msg=f"{sess.kernel_name()} {key} run {rep}")
      sess.close()
  finally:
    stop.set()
    th.join()


def test_fit_causalimpact_with_more_than_52_covariates():
  """The reference has no cap on the number of covariates (causalimpact_lib.py:445-453).  80
  control series, 3 of them carrying the signal: the drop-in call runs (sequential kernel,
  regression block in the HBM workspace), recovers the effect and keeps the model sparse."""
  import pandas as pd
  import causalimpact as ci
  rng = np.random.default_rng(5)
  n, p = 200, 80
  X = rng.normal(size=(n, p)).cumsum(axis=0) * 0.1 + rng.normal(size=(n, p))
  y = 1.5 * X[:, 3] - 2.0 * X[:, 40] + 1.2 * X[:, 77] + 0.3 * rng.normal(size=n)
  y[140:] += 4.0
  df = pd.DataFrame(np.column_stack([y, X]), columns=["y"] + [f"x{j}" for j in range(p)])
  res = ci.fit_causalimpact(df, (0, 139), (140, 199), seed=3,
                            inference_options=ci.InferenceOptions(num_results=300, num_chains=2))
  w = res.posterior_samples.weights
  assert w.shape == (600, p + 1)
  incl = (w != 0).mean(axis=0)
  assert incl[[3, 40, 77]].min() > 0.9 and np.delete(incl, [3, 40, 77, p]).mean() < 0.15
  np.testing.assert_allclose(res.summary.loc["average", "abs_effect"], 4.0, atol=0.5)


@pytest.mark.gpu
@pytest.mark.parametrize("T,p,has_slope", [(3, 0, 0), (4, 1, 0), (5, 0, 1), (9, 2, 1), (6, 0, 0)])
def test_shortest_series_match_the_oracle(T, p, has_slope):
  """The C-ABI's lower bound T = 3 and its neighbours: one step per thread, most threads idle, the
  last step missing (the forecast)."""
  y, mask, X, _ = syn.make_sampler_inputs(12, max(p, 1), 3)
  y, mask = y[:T].copy(), np.zeros(T, bool)
  mask[-1] = True
  X = X[:T, :p + 1] if p else None
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope))
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_warmup=0, num_results=4,
                            seed=(1, 2))
  got = _native.fit_gibbs(pb, y[None], mask[None], None if X is None else X[None], None,
                          _native.make_params([spec]))
  w = orc.fit_gibbs(y, mask, X, spec, num_results=4, num_warmup=0, seed=(1, 2))
  np.testing.assert_allclose(got["level"][0, 0], w["level"], atol=1e-4)
  np.testing.assert_allclose(got["observation_noise_scale"][0, 0], w["obs_scale"], rtol=1e-4)
  np.testing.assert_allclose(got["posterior_trajectories"][0, 0], w["trajectories"], atol=1e-4)


def _tp_fit(T, p, has_slope, seasons, flags, W, S, C=1, seed=(2, 6), data_seed=7):
  from causalimpact import _model
  y, mask, X, _ = syn.make_sampler_inputs(T, p, data_seed)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  mask = mask.copy()
  mask[[2, 3, 40, T // 2]] = True
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_seasons=counts, num_warmup=W,
                            num_results=S, num_chains=C, seed=seed, flags=flags)
  sess = _native.Session(pb, y[None], mask[None], None if X is None else X[None], flg if seasons else None,
                         _native.make_params([spec]))
  name = sess.kernel_name()
  sess.run()
  got = sess.fetch()
  sess.close()
  return name, got, (y, mask, X, spec)


def test_time_parallel_general_seasonal_kernel_with_a_dropped_helper_gives_the_same_bits():
  """ci_seasonal_tp.h: if a workgroup of a chain's cluster is never scheduled, the first workgroup
  notices while assembling the cluster and runs every chunk itself -- same chunks, same arithmetic,
  same bits (CI_FLAG_TEST_DROP_HELPER makes the last workgroup leave at once)."""
  args = dict(T=3000, p=6, has_slope=1, seasons=REF_SEASONS, W=2, S=4, C=2)
  name, whole, _ = _tp_fit(flags=0, **args)
  assert "gibbs_seasonal_tp_kernel" in name and not name.endswith("x1")
  _, alone, _ = _tp_fit(flags=_native.FLAG_TEST_DROP_HELPER, **args)
  for k, v in whole.items():
    np.testing.assert_array_equal(alone[k], v, err_msg=k)


def test_time_parallel_general_seasonal_kernel_gives_the_same_bits_on_any_cluster_size():
  """The chunk grid is fixed by the series (T alone); the launch only decides how many real
  workgroups share the chunks -- one per chain (CI_FLAG_NO_CLUSTER), a cluster of 32 for a single
  chain, a cluster of 4 per chain in a launch of 40 chains: the same chunks, the same scan trees,
  the same arithmetic, hence the same bits for the same (seed, chain) -- and the oracle's draws to
  float32 accuracy."""
  args = dict(T=2000, p=4, has_slope=0, seasons=REF_SEASONS, W=0, S=3)
  n1, one, (y, mask, X, spec) = _tp_fit(flags=_native.FLAG_NO_CLUSTER, **args)
  n2, many, _ = _tp_fit(flags=0, **args)
  n3, crowd, _ = _tp_fit(flags=0, C=40, **args)
  assert n1.endswith("x1") and "tp_kernel" in n2 and not n2.endswith("x1")
  assert "tp_kernel" in n3 and n3.split("x")[-1] != n2.split("x")[-1]        # another cluster size
  w = orc.fit_gibbs(y, mask, X, spec, num_results=3, num_warmup=0, seed=(2, 6))
  np.testing.assert_array_equal(one["weights"][0, 0] != 0, w["weights"] != 0)
  np.testing.assert_allclose(one["level"][0, 0], w["level"], atol=5e-3)
  np.testing.assert_allclose(one["seasonal_levels"][0, 0], w["seasonal"], atol=5e-3)
  np.testing.assert_allclose(one["seasonal_drift_scales"][0, 0], w["drift_scales"], rtol=2e-2)
  np.testing.assert_allclose(one["observation_noise_scale"][0, 0], w["obs_scale"], rtol=5e-3)
  np.testing.assert_allclose(one["posterior_trajectories"][0, 0], w["trajectories"], atol=1e-2)
  for k, v in one.items():
    np.testing.assert_array_equal(many[k], v, err_msg=k)
    np.testing.assert_array_equal(crowd[k][:, :1], v, err_msg=k)


def test_time_parallel_and_sequential_general_seasonal_kernels_sample_the_same_posterior():
  """Long runs (beyond the draws where the kernels agree per random number): the time-parallel
  kernel and the sequential one give the same posterior summaries on the reference's 4+7+6 model."""
  args = dict(T=600, p=3, has_slope=0, seasons=REF_SEASONS, W=100, S=300, C=4, seed=(8, 1))
  nt, tp, _ = _tp_fit(flags=0, **args)
  ns, sq, _ = _tp_fit(flags=SEQ, **args)
  assert "tp_kernel" in nt and "gibbs_seasonal_kernel" in ns
  for key, tol in (("observation_noise_scale", 0.02), ("level_scale", 0.06)):
    np.testing.assert_allclose(tp[key].mean(), sq[key].mean(), rtol=tol, err_msg=key)
  np.testing.assert_allclose(tp["weights"].mean(axis=(0, 1, 2)), sq["weights"].mean(axis=(0, 1, 2)), atol=0.02)
  np.testing.assert_allclose(tp["posterior_means"].mean(axis=(0, 1)), sq["posterior_means"].mean(axis=(0, 1)),
                             atol=0.05)
  np.testing.assert_allclose(tp["seasonal_levels"].mean(axis=(0, 1, 2)), sq["seasonal_levels"].mean(axis=(0, 1, 2)),
                             atol=0.04)


def test_trend_models_with_more_than_52_covariates_agree_on_their_three_routes():
  """Trend + 53..~150 columns runs on the BIGP build of the trend + one-block kernel by default
  (round 6: ci_wide.h with the regression draw of ci_bigp.h; test_first_iterations_...: 61 and 131
  columns against the oracle); the general time-parallel kernel (CI_FLAG_CLUSTER_SEASONAL; the route
  of wider designs: 512 columns in that test) and the sequential one-wavefront kernel stay
  reachable and equal per draw."""
  T, p = 400, 60
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
  spec = orc.default_spec(y, mask, X, has_slope=False)
  out = {}
  for flags in (0, _native.FLAG_CLUSTER_SEASONAL, SEQ):
    pb = _native.make_problem(T=T, P=spec["P"], has_slope=0, num_warmup=0, num_results=4, seed=(5, 9), flags=flags)
    sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
    out[flags] = (sess.kernel_name(), sess.run(), sess.fetch())
    sess.close()
  assert "gibbs_wide_kernel" in out[0][0] and "bigp" in out[0][0]
  assert "tp_kernel" in out[_native.FLAG_CLUSTER_SEASONAL][0] and "gibbs_seasonal_kernel" in out[SEQ][0]
  for other in (_native.FLAG_CLUSTER_SEASONAL, SEQ):
    np.testing.assert_array_equal(out[0][2]["weights"] != 0, out[other][2]["weights"] != 0)
    np.testing.assert_allclose(out[0][2]["level"], out[other][2]["level"], atol=2e-3)
    np.testing.assert_allclose(out[0][2]["weights"], out[other][2]["weights"], atol=2e-3)


@pytest.mark.parametrize("T,p,has_slope,seasons", [
    (1000, 100, 0, ()),            # the P = 101 case of the round-5 review, trend only
    (600, 70, 1, ((7, 1),)),       # local linear trend + weekly block, 71 columns
    (2048, 139, 0, ()),            # close to the widest design whose matrix and index table fit in LDS
    (500, 60, 0, ((4, 1),)),       # a four-season block (every block size 2..7 has a BIGP build)
])
def test_wide_kernel_with_more_than_52_columns_matches_the_oracle_and_any_cluster_size(T, p, has_slope, seasons):
  """The BIGP builds of ci_wide.h: per draw against the float64 oracle, and the same bits from one
  workgroup per chain (CI_FLAG_NO_CLUSTER), the default cluster and a launch of 40 chains (smaller
  clusters)."""
  from causalimpact import _model
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 11)
  if seasons:
    y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  spec = orc.default_spec(y, mask, X, has_slope=bool(has_slope), seasons=seasons)
  counts, flg = (_model.expand_seasons(seasons, T) if seasons else ((), None))
  S = 3

  def fit(C, flags):
    pb = _native.make_problem(T=T, P=spec["P"], has_slope=has_slope, num_seasons=counts, num_warmup=0,
                              num_results=S, num_chains=C, seed=(2, 6), flags=flags)
    sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
    name = sess.kernel_name()
    sess.run()
    got = sess.fetch()
    sess.close()
    return name, got

  name, one = fit(1, _native.FLAG_NO_CLUSTER)
  assert "gibbs_wide_kernel" in name and "bigp" in name
  _, many = fit(1, 0)
  _, crowd = fit(40, 0)
  for k, v in one.items():
    np.testing.assert_array_equal(many[k], v, err_msg=k)
    np.testing.assert_array_equal(crowd[k][:, :1], v, err_msg=k)
  w = orc.fit_gibbs(y, mask, X, spec, num_results=S, num_warmup=0, seed=(2, 6))
  np.testing.assert_array_equal(one["weights"][0, 0] != 0, w["weights"] != 0)
  np.testing.assert_allclose(one["level"][0, 0], w["level"], atol=5e-3)
  np.testing.assert_allclose(one["weights"][0, 0], w["weights"], atol=5e-3)
  np.testing.assert_allclose(one["observation_noise_scale"][0, 0], w["obs_scale"], rtol=5e-3)
  np.testing.assert_allclose(one["posterior_trajectories"][0, 0], w["trajectories"], atol=1e-2)
  if seasons:
    np.testing.assert_allclose(one["seasonal_levels"][0, 0], w["seasonal"], atol=5e-3)


def test_wide_kernel_with_more_than_52_columns_on_a_ragged_batch():
  """BIGP builds off the beaten path: a length that is no multiple of 4 (no clusters, scalar reads),
  53 columns (the first width beyond the LDS block), the first observation missing, and a batch of
  three series x two chains whose series reproduce their single-series launches bit for bit; per
  draw against the oracle."""
  T, p, B, C, S = 1001, 52, 3, 2, 3
  ys, masks, Xs, specs = [], [], [], []
  for b in range(B):
    y, mask, X, _ = syn.make_sampler_inputs(T, p, 40 + b)
    mask = mask.copy()
    mask[0] = True
    mask[[17, 500]] = True
    ys.append(y); masks.append(mask); Xs.append(X)
    specs.append(orc.default_spec(y, mask, X, has_slope=True))
  P = specs[0]["P"]
  assert P == 53
  kw = dict(T=T, P=P, has_slope=1, num_warmup=0, num_results=S, num_chains=C, seed=(3, 1))
  sess = _native.Session(_native.make_problem(num_series=B, **kw), np.stack(ys), np.stack(masks), np.stack(Xs), None,
                         _native.make_params(specs))
  assert "bigp" in sess.kernel_name()
  sess.run()
  batch = sess.fetch()
  sess.close()
  for b in range(B):
    one_s = _native.Session(_native.make_problem(num_series=1, series_offset=b, **kw), ys[b][None], masks[b][None],
                            Xs[b][None], None, _native.make_params([specs[b]]))
    one_s.run()
    one = one_s.fetch()
    one_s.close()
    for k in ("level", "slope", "weights", "observation_noise_scale", "posterior_trajectories"):
      np.testing.assert_array_equal(batch[k][b], one[k][0], err_msg=f"{k} series {b}")
  # (series 0 of a batch draws from the plain single-series streams: the oracle's)
  w = orc.fit_gibbs(ys[0], masks[0], Xs[0], specs[0], num_results=S, num_warmup=0, seed=(3, 1), chain=1)
  np.testing.assert_array_equal(batch["weights"][0, 1] != 0, w["weights"] != 0)
  np.testing.assert_allclose(batch["level"][0, 1], w["level"], atol=5e-3)
  np.testing.assert_allclose(batch["slope"][0, 1], w["slope"], atol=5e-3)
  np.testing.assert_allclose(batch["weights"][0, 1], w["weights"], atol=5e-3)
  np.testing.assert_allclose(batch["posterior_trajectories"][0, 1], w["trajectories"], atol=1e-2)


def test_cfg4_clusters_equal_one_workgroup_per_chain_over_a_long_run():
  """BASELINE cfg4's shape (T=10000, 50 covariates, weekly block), 150 iterations, 8 chains: the
  cluster of sixteen (the draw on eight workgroups with its rows in LDS, the matrix swept ahead and
  imported during the draw, randomness drawn ahead) against one workgroup per chain, bit for bit in
  every output -- a hand-off that is wrong once in a thousand iterations shows here."""
  from causalimpact import _model
  T, p, seasons, W, S, C = 10000, 50, ((7, 1),), 30, 120, 8
  y, mask, X, _ = syn.make_sampler_inputs(T, p, 0)
  y = y + 0.8 * np.sin(2 * np.pi * np.arange(T) / 7.0)
  spec = orc.default_spec(y, mask, X, has_slope=False, seasons=seasons)
  counts, flg = _model.expand_seasons(seasons, T)
  want = ("observation_noise_scale", "level_scale", "seasonal_drift_scales", "weights", "posterior_means")
  out = {}
  for flags in (_native.FLAG_NO_CLUSTER, 0):
    pb = _native.make_problem(T=T, P=spec["P"], has_slope=0, num_seasons=counts, num_warmup=W, num_results=S,
                              num_chains=C, seed=(9, 9), flags=flags)
    sess = _native.Session(pb, y[None], mask[None], X[None], flg, _native.make_params([spec]))
    sess.run()
    out[flags] = sess.fetch(want)
    sess.close()
  for k in want:
    np.testing.assert_array_equal(out[0][k], out[_native.FLAG_NO_CLUSTER][k], err_msg=k)
  assert np.isfinite(out[0]["posterior_means"]).all()
