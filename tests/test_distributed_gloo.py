"""world_size-2 gloo test of the chain-sharding path (runs on CPU).

The sampler itself needs a GPU, so each rank's local fit is a deterministic stand-in keyed
by the GLOBAL chain id (exactly the property the HIP kernel's Philox stream provides and
tests/test_gpu_gibbs.py::test_chain_ids_do_not_depend_on_launch_split checks on hardware).
What is tested here is the distributed logic: block partition, gather order, padding of
uneven blocks, the moment all-reduce and the resulting split-R-hat.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_fit(first, count, S=40, T=7):
  assert count > 0, "local_fit must not be called for an empty block"
  out = {"posterior_trajectories": np.zeros((count, S, T), np.float32),
         "posterior_means": np.zeros((count, T), np.float32),
         "observation_noise_scale": np.zeros((count, S), np.float32),
         "level_scale": np.zeros((count, S), np.float32)}
  for i in range(count):
    rng = np.random.default_rng(1000 + first + i)           # keyed by global chain id
    out["posterior_trajectories"][i] = rng.normal(size=(S, T))
    out["posterior_means"][i] = rng.normal(size=T)
    out["observation_noise_scale"][i] = 0.3 + 0.01 * rng.normal(size=S)
    out["level_scale"][i] = 0.01 + 0.001 * rng.normal(size=S) + 0.002 * (first + i)
  return out


def _worker(rank, world, port, num_chains, q):
  sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
  import torch.distributed as dist
  from causalimpact import _distributed as d
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  sys.path.insert(0, os.path.join(ROOT, "tests"))
  import _torch_comm
  res = d.fit_sharded(_fake_fit, num_chains, comm=_torch_comm.from_initialized_group())
  q.put((rank, res["posterior_trajectories"], res["posterior_means"],
         {k: res[k] for k in ("split_rhat", "ess_bulk", "ess_tail")}))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("num_chains", [4, 5, 1])
def test_two_ranks_equal_one_process(num_chains):
  """num_chains = 1 leaves rank 1 with an EMPTY block: it must not call the fit (a zero-chain
  problem is invalid) and must still take part in every collective."""
  import multiprocessing as mp
  sys.path[:0] = [os.path.join(ROOT, "tfp-causalimpact_amd")]
  from causalimpact import _distributed as d
  single = d.fit_sharded(_fake_fit, num_chains)
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, num_chains, q)) for r in range(2)]
  for p in procs:
    p.start()
  got = [q.get(timeout=180) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for _, traj, means, diag in got:
    np.testing.assert_array_equal(traj, single["posterior_trajectories"])
    np.testing.assert_array_equal(means, single["posterior_means"])
    for name in ("split_rhat", "ess_bulk", "ess_tail"):
      for k, v in single[name].items():
        np.testing.assert_allclose(diag[name][k], v, rtol=1e-9, err_msg=f"{name}.{k}")
  if num_chains > 1:
    # the level_scale stand-in drifts with the chain id, so its R-hat must flag it
    assert single["split_rhat"]["level_scale"] > 1.2
    assert single["split_rhat"]["observation_noise_scale"] < 1.1
    assert single["ess_bulk"]["observation_noise_scale"] > 0.5 * num_chains * 40
    assert single["ess_tail"]["observation_noise_scale"] > 0.2 * num_chains * 40
    assert single["ess_bulk"]["level_scale"] < 0.2 * num_chains * 40


def test_chain_blocks_partition_exactly():
  sys.path[:0] = [os.path.join(ROOT, "tfp-causalimpact_amd")]
  from causalimpact import _distributed as d
  for n in (1, 7, 8, 64):
    for w in (1, 2, 3, 8):
      blocks = [d.chain_block(n, r, w) for r in range(w)]
      ids = [c for f, k in blocks for c in range(f, f + k)]
      assert ids == list(range(n))
      assert max(k for _, k in blocks) - min(k for _, k in blocks) <= 1


def _gpu_worker(rank, world, port, num_chains, q):
  """One rank of the real thing: its share of the chains runs on GPU 0 through the C-ABI
  (`chain_offset` = first chain id), then the collectives of `fit_sharded` pool the results."""
  sys.path[:0] = [ROOT, os.path.join(ROOT, "tfp-causalimpact_amd")]
  import torch.distributed as dist
  from causalimpact import _distributed as d, _model, _native
  from causalimpact import _synthetic as syn
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  y, mask, X, _ = syn.make_sampler_inputs(300, 3, 5)
  spec = _model.series_params(y, mask, X, has_slope=True)

  def local_fit(first, count):
    pb = _native.make_problem(T=300, P=4, has_slope=1, num_warmup=20, num_results=60,
                              num_chains=count, chain_offset=first, seed=(4, 4), device=0)
    out = _native.fit_gibbs(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
    return {k: v[0] for k, v in out.items()}

  sys.path.insert(0, os.path.join(ROOT, "tests"))
  import _torch_comm
  res = d.fit_sharded(local_fit, num_chains, comm=_torch_comm.from_initialized_group())
  q.put((rank, res["posterior_trajectories"], res["posterior_means"],
         {k: res[k] for k in ("split_rhat", "ess_bulk", "ess_tail")}))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_with_real_gpu_shares_equal_one_launch():
  """The collective path on hardware: two processes, each running its block of chains on the
  GPU through `_native.fit_gibbs`, pooled by `fit_sharded` (gloo here: two ranks cannot share
  one device under RCCL; bench.py's N > 1 path uses backend "nccl", and its single-rank RCCL run
  is logged in profiles/r02_bench_force_dist_rccl_1rank.*).  The pooled draws must be bit-equal
  to ONE launch of all chains."""
  import multiprocessing as mp        # (not torch.multiprocessing: keep torch out of THIS process)
  sys.path[:0] = [os.path.join(ROOT, "tfp-causalimpact_amd")]
  from causalimpact import _distributed as d, _model, _native
  from causalimpact import _synthetic as syn
  num_chains = 5
  y, mask, X, _ = syn.make_sampler_inputs(300, 3, 5)
  spec = _model.series_params(y, mask, X, has_slope=True)
  pb = _native.make_problem(T=300, P=4, has_slope=1, num_warmup=20, num_results=60,
                            num_chains=num_chains, seed=(4, 4), device=0)
  one = {k: v[0] for k, v in _native.fit_gibbs(pb, y[None], mask[None], X[None], None,
                                               _native.make_params([spec])).items()}
  single = d.fit_sharded(lambda first, count: one, num_chains)
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, num_chains, q)) for r in range(2)]
  for p in procs:
    p.start()
  got = [q.get(timeout=300) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  for _, traj, means, diag in got:
    np.testing.assert_array_equal(traj, one["posterior_trajectories"])
    np.testing.assert_array_equal(means, one["posterior_means"])
    for name in ("split_rhat", "ess_bulk", "ess_tail"):
      for k, v in single[name].items():
        np.testing.assert_allclose(diag[name][k], v, rtol=1e-9, err_msg=f"{name}.{k}")
