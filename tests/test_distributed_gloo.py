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
  res = d.fit_sharded(_fake_fit, num_chains)
  q.put((rank, res["posterior_trajectories"], res["posterior_means"],
         {k: res[k] for k in ("split_rhat", "ess_bulk", "ess_tail")}))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("num_chains", [4, 5, 1])
def test_two_ranks_equal_one_process(num_chains):
  """num_chains = 1 leaves rank 1 with an EMPTY block: it must not call the fit (a zero-chain
  problem is invalid) and must still take part in every collective."""
  import torch.multiprocessing as mp
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
