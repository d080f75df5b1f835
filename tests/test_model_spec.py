"""The product's model constants (causalimpact/_model.py) equal the oracle's independent
restatement (oracle/ci_oracle.py) of causalimpact_lib.py:398-500, :563-581."""
import numpy as np
import pytest

from causalimpact import _model
from causalimpact import _synthetic as syn
from oracle import ci_oracle as orc


@pytest.mark.parametrize("p,has_slope,seasons", [
    (0, False, ()), (1, False, ()), (10, True, ()), (3, False, ((7, 1), (4, (2, 1, 1, 1)))),
])
def test_series_params_match_oracle_spec(p, has_slope, seasons):
  y, mask, X, _ = syn.make_sampler_inputs(120, p, seed=p + 1)
  mask[[0, 4]] = True   # missing first observation: both sides use the first observed value
  mine = _model.series_params(y, mask, X, prior_level_sd=0.1, has_slope=has_slope,
                              num_seasonal_blocks=len(seasons))
  ref = orc.default_spec(y, mask, X, prior_level_sd=0.1, has_slope=has_slope, seasons=seasons)
  for k, v in mine.items():
    if k == "drift_scale0":
      np.testing.assert_allclose(v, ref[k], rtol=1e-14)
    else:
      np.testing.assert_allclose(v, ref[k], rtol=1e-14, err_msg=k)
  counts, flags = _model.expand_seasons(seasons, 120)
  assert counts == ref["num_seasons"]
  for k in range(len(seasons)):
    np.testing.assert_array_equal(flags[k], ref["season_change"][k])


def test_season_calendar_shapes():
  # int: every season lasts s steps; changes at s-1, 2s-1, ...
  f = _model.season_change_flags(10, 3, 2)
  np.testing.assert_array_equal(f, [0, 1, 0, 1, 0, 1, 0, 1, 0, 1])
  # per-season lengths (2,1,1,1): cycle of 5 with changes after steps 1,2,3,4
  f = _model.season_change_flags(10, 4, (2, 1, 1, 1))
  np.testing.assert_array_equal(f, [0, 1, 1, 1, 1, 0, 1, 1, 1, 1])
  # per-cycle table wraps after both cycles (causalimpact_lib_test.py:748-750)
  f = _model.season_change_flags(16, 6, ((2, 2, 1, 1, 1, 1), (2, 2, 1, 1, 1, 1)))
  np.testing.assert_array_equal(f[:8], [0, 1, 0, 1, 1, 1, 1, 1])
  np.testing.assert_array_equal(f[8:], f[:8])
  with pytest.raises(ValueError):
    _model.season_change_flags(10, 3, (1, 2))


def test_effective_sample_size_on_known_processes():
  from causalimpact import causalimpact_lib as lib
  rng = np.random.default_rng(0)
  iid = rng.normal(size=(4, 2000))
  assert 0.8 * 8000 < lib.effective_sample_size(iid) < 1.25 * 8000
  phi = 0.9
  ar = np.zeros((4, 4000))
  e = rng.normal(size=ar.shape)
  for t in range(1, ar.shape[1]):
    ar[:, t] = phi * ar[:, t - 1] + e[:, t]
  want = ar.size * (1 - phi) / (1 + phi)
  assert 0.6 * want < lib.effective_sample_size(ar) < 1.6 * want
  assert np.isnan(lib.effective_sample_size(np.zeros((2, 6))))
  # chains stuck at different levels: R-hat large, ESS tiny
  stuck = rng.normal(size=(4, 500)) * 0.01 + np.arange(4)[:, None]
  assert lib.effective_sample_size(stuck) < 20 and lib.split_rhat(stuck) > 5


def test_bulk_and_tail_ess_and_additivity_of_the_partial_sums():
  """Vehtari et al. (2021) bulk / tail ESS, written as partial sums that add over ranks
  (`_diagnostics`): iid draws give ~N for both; an AR(1) gives the textbook bulk value; a
  heavy-tailed parameter (where the plain estimator's variance does not exist) still gets a
  finite, sensible bulk ESS; summing the partial sums of two disjoint chain blocks reproduces
  the all-chains value exactly (what the all-reduce in `_distributed.fit_sharded` relies on)."""
  from causalimpact import _diagnostics as dg
  from causalimpact import causalimpact_lib as lib
  rng = np.random.default_rng(1)
  iid = rng.normal(size=(4, 2000))
  assert 0.8 * 8000 < lib.effective_sample_size(iid, "bulk") < 1.25 * 8000
  assert 0.6 * 8000 < lib.effective_sample_size(iid, "tail") < 1.4 * 8000
  phi = 0.9
  ar = np.zeros((4, 4000))
  e = rng.normal(size=ar.shape)
  for t in range(1, ar.shape[1]):
    ar[:, t] = phi * ar[:, t - 1] + e[:, t]
  want = ar.size * (1 - phi) / (1 + phi)
  assert 0.6 * want < lib.effective_sample_size(ar, "bulk") < 1.6 * want
  assert lib.effective_sample_size(ar, "tail") < ar.size
  cauchy = rng.standard_cauchy(size=(4, 2000))
  assert 0.7 * 8000 < lib.effective_sample_size(cauchy, "bulk") < 1.3 * 8000
  assert np.isnan(lib.effective_sample_size(np.zeros((2, 50)), "bulk"))
  # additivity: chains {0,1} + chains {2,3} == chains {0..3}
  whole = dg.bulk_partial(ar, ar)
  a, b = dg.bulk_partial(ar[:2], ar), dg.bulk_partial(ar[2:], ar)
  summed = dg.unpack(dg.pack(a) + dg.pack(b))
  np.testing.assert_allclose(dg.ess_from_sums(summed), dg.ess_from_sums(whole), rtol=1e-12)
  np.testing.assert_allclose(dg.rhat_from_sums(dg.unpack(dg.pack(dg.partial_sums(dg.split_chains(ar[:2]))) +
                                                          dg.pack(dg.partial_sums(dg.split_chains(ar[2:]))))),
                             lib.split_rhat(ar), rtol=1e-12)


def test_diagnostics_mapping_is_computed_on_first_use_and_behaves_like_a_dict():
  """CausalImpactAnalysis.diagnostics (num_chains > 1) is a read-only mapping filled in when it is
  first read: the convergence statistics cost a third of a fit's host time and most callers never
  look at them."""
  from causalimpact import causalimpact_lib as lib
  calls = []

  def make():
    calls.append(1)
    return {"split_rhat": {"observation_noise_scale": 1.01}, "num_chains": 4}

  d = lib._LazyMapping(make)      # pylint: disable=protected-access
  assert not calls
  assert d["num_chains"] == 4 and calls == [1]
  assert set(d) == {"split_rhat", "num_chains"} and len(d) == 2 and "split_rhat" in d
  assert dict(d)["split_rhat"]["observation_noise_scale"] == 1.01
  assert "split_rhat" in repr(d)
  assert calls == [1]                                  # computed once
  with pytest.raises(TypeError):
    d["x"] = 1                                         # read-only
