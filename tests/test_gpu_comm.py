"""Multi-rank runs through the C-ABI collectives (`ci_comm_*`) on hardware -- no PyTorch.

A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device: "Duplicate GPU
detected"), so:
  * RCCL is exercised with one rank (communicator creation from the file rendezvous, barrier,
    all-reduce, all-gather of the session's DEVICE-RESIDENT arrays == ci_session_fetch);
  * two ranks SHARING GPU 0 run real fits of their chain blocks and pool them over the host
    transport (same entry points, staged through shared memory): bit-equal to ONE launch of all
    chains; they are started by `_comm.spawn_ranks`, the launcher behind `bench.py --gpus N`;
  * asking RCCL for two ranks on one device must fail with a clean error, not hang -- and through
    `_comm.connect` (what bench.py uses) it must FALL BACK to the host transport, labelled;
  * on a box with >= 2 GPUs (the driver's round-end box may be one) two tests run REAL RCCL with
    two ranks: pooled results bit-equal to one launch, and `bench.py --gpus 2` end to end.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tfp-causalimpact_amd")
pytestmark = pytest.mark.gpu

_RANK = r"""
import json, os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np
from causalimpact import _comm, _distributed as d, _model, _native
from causalimpact import _synthetic as syn
comm = _comm.connect()          # host control plane first, RCCL under a deadline, agreed fallback
num_chains = %(num_chains)d
y, mask, X, _ = syn.make_sampler_inputs(300, 3, 5)
spec = _model.series_params(y, mask, X, has_slope=True)
holder = {}

def local_fit(first, count):
  pb = _native.make_problem(T=300, P=4, has_slope=1, num_warmup=20, num_results=60,
                            num_chains=count, chain_offset=first, seed=(4, 4), device=comm.device)
  sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
  sess.run()
  holder["sess"] = sess
  return {k: v[0] for k, v in sess.fetch().items()}

def resident(key):
  g = comm.session_all_gather(holder["sess"], key)       # [world, 1, C, ...] from HBM
  return g.reshape((-1,) + g.shape[3:])

res = d.fit_sharded(local_fit, num_chains, comm=comm, resident=resident if %(even)d else None)
np.savez(os.path.join(%(out)r, "rank%%d.npz" %% comm.rank), traj=res["posterior_trajectories"],
         means=res["posterior_means"], rhat=res["split_rhat"]["level_scale"],
         ess=res["ess_bulk"]["observation_noise_scale"], seen=comm.ranks_seen,
         transport=comm.transport, kind=comm.kind)
comm.barrier()
hard = comm.hard_exit
comm.close()
assert "torch" not in sys.modules
if hard:
  sys.stdout.flush()
  os._exit(0)
"""


def _one_launch(num_chains):
  sys.path[:0] = [PKG]
  from causalimpact import _distributed as d, _model, _native
  from causalimpact import _synthetic as syn
  y, mask, X, _ = syn.make_sampler_inputs(300, 3, 5)
  spec = _model.series_params(y, mask, X, has_slope=True)
  pb = _native.make_problem(T=300, P=4, has_slope=1, num_warmup=20, num_results=60,
                            num_chains=num_chains, seed=(4, 4), device=0)
  one = {k: v[0] for k, v in _native.fit_gibbs(pb, y[None], mask[None], X[None], None,
                                               _native.make_params([spec])).items()}
  return one, d.fit_sharded(lambda first, count: one, num_chains)


@pytest.mark.parametrize("num_chains", [4, 5])
def test_two_ranks_sharing_gpu0_equal_one_launch(tmp_path, num_chains):
  """Even blocks gather straight from the sessions' device buffers (`ci_comm_session_all_gather`),
  uneven ones pad and gather the host copies; both must reproduce the single launch bit for bit."""
  from causalimpact import _comm
  one, single = _one_launch(num_chains)
  script = tmp_path / "rank.py"
  script.write_text(_RANK % dict(root=ROOT, pkg=PKG, num_chains=num_chains, out=str(tmp_path),
                                 even=int(num_chains % 2 == 0)))
  codes = _comm.spawn_ranks(2, [sys.executable, str(script)], transport="host", devices=[0, 0],
                            timeout=300)
  assert codes == [0, 0]
  for transport, kind in _check_ranks(tmp_path, one, single):
    assert transport == "host" and kind == "host"


def _check_ranks(tmp_path, one, single, world=2):
  out = []
  for r in range(world):
    got = np.load(tmp_path / f"rank{r}.npz")
    assert int(got["seen"]) == world
    np.testing.assert_array_equal(got["traj"], one["posterior_trajectories"])
    np.testing.assert_array_equal(got["means"], one["posterior_means"])
    np.testing.assert_allclose(got["rhat"], single["split_rhat"]["level_scale"], rtol=1e-9)
    np.testing.assert_allclose(got["ess"], single["ess_bulk"]["observation_noise_scale"], rtol=1e-9)
    out.append((str(got["transport"]), str(got["kind"])))
  return out


def _device_count():
  sys.path[:0] = [PKG]
  from causalimpact import _native
  return _native.device_count()


@pytest.mark.parametrize("num_chains", [4, 5])
def test_rccl_with_real_ranks_on_two_gpus_equals_one_launch(tmp_path, num_chains):
  """Armed for any box with >= 2 GPUs (skips on one): two ranks, one per GPU, over REAL RCCL --
  even blocks gathered straight from HBM with ncclAllGather, uneven ones padded -- bit-equal to ONE
  launch of all chains on GPU 0, and the transport must really be RCCL (no silent fallback)."""
  if _device_count() < 2:
    pytest.skip("needs two GPUs")
  from causalimpact import _comm
  one, single = _one_launch(num_chains)
  script = tmp_path / "rank.py"
  script.write_text(_RANK % dict(root=ROOT, pkg=PKG, num_chains=num_chains, out=str(tmp_path),
                                 even=int(num_chains % 2 == 0)))
  codes = _comm.spawn_ranks(2, [sys.executable, str(script)], transport="rccl", devices=[0, 1],
                            timeout=600)
  assert codes == [0, 0]
  for transport, kind in _check_ranks(tmp_path, one, single):
    assert transport == "rccl" and kind == "rccl", transport


def test_bench_gpus_2_over_rccl_on_two_gpus():
  """`python bench.py --gpus 2` as the driver runs it, on a box that has the GPUs: RCCL must carry
  the collectives, both ranks must be seen, the whole-job value must be about twice one GPU's."""
  if _device_count() < 2:
    pytest.skip("needs two GPUs")
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK",
                                                          "CI_COMM_TRANSPORT", "CI_COMM_DEVICES")}
  base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2",
          "--no-cpu-baseline", "--no-pmc"]
  two = json.loads(subprocess.run(base + ["--gpus", "2"], env=env, capture_output=True, text=True,
                                  timeout=900, check=True).stdout.strip().splitlines()[-1])
  one = json.loads(subprocess.run(base, env=env, capture_output=True, text=True, timeout=900,
                                  check=True).stdout.strip().splitlines()[-1])
  assert two["n_gpus"] == 2 and two["config"]["ranks_seen"] == 2
  assert two["config"]["collectives"].startswith("rccl"), two["config"]["collectives"]
  assert 1.5 < two["value"] / one["value"] < 2.3


def test_connect_falls_back_to_the_host_transport_when_rccl_refuses(tmp_path):
  """Two ranks asked to use RCCL on ONE device: RCCL refuses ("Duplicate GPU detected").  `connect`
  must not hang and must not fail: every rank drops the attempt, the collectives run over the
  host transport, the reason is in `transport`, and the pooled result still equals one launch."""
  from causalimpact import _comm
  one, single = _one_launch(4)
  script = tmp_path / "rank.py"
  script.write_text(_RANK % dict(root=ROOT, pkg=PKG, num_chains=4, out=str(tmp_path), even=1))
  env = dict(os.environ, CI_COMM_INIT_TIMEOUT_S="60")
  codes = _comm.spawn_ranks(2, [sys.executable, str(script)], env=env, transport="rccl",
                            devices=[0, 0], timeout=600)
  assert codes == [0, 0]
  for transport, kind in _check_ranks(tmp_path, one, single):
    assert kind == "host" and transport.startswith("host (rccl failed:"), transport


_RCCL1 = r"""
import sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np
from causalimpact import _comm, _model, _native
from causalimpact import _synthetic as syn
y, mask, X, _ = syn.make_sampler_inputs(200, 2, 1)
spec = _model.series_params(y, mask, X, has_slope=False)
pb = _native.make_problem(T=200, P=3, has_slope=0, num_warmup=5, num_results=30, num_chains=3,
                          seed=(1, 2), device=0)
sess = _native.Session(pb, y[None], mask[None], X[None], None, _native.make_params([spec]))
sess.run()
want = sess.fetch()
comm = _comm.Comm(0, 1, device=0, transport="rccl", path=%(rdzv)r)
assert comm.ranks_seen == 1 and comm.world == 1
comm.barrier()
np.testing.assert_array_equal(comm.all_reduce([1.5, -2.0]), [1.5, -2.0])
np.testing.assert_array_equal(comm.all_reduce([1.5, -2.0], _comm.MAX), [1.5, -2.0])
a = np.arange(12, dtype=np.float64).reshape(3, 4)
np.testing.assert_array_equal(comm.all_gather(a), a[None])
for key in ("posterior_trajectories", "level", "weights", "posterior_means",
            "observation_noise_scale"):
  np.testing.assert_array_equal(comm.session_all_gather(sess, key), want[key][None], err_msg=key)
comm.close()
sess.close()
assert "torch" not in sys.modules
print("rccl single rank ok")
"""


def test_rccl_single_rank_gathers_device_resident_arrays(tmp_path):
  """In its own process, as in production: a process that has imported PyTorch carries the ROCm
  runtime bundled with the torch wheel, and librccl resolved there does not see the devices of
  the runtime libcausalimpact_amd.so is linked to -- the product path never imports torch."""
  script = tmp_path / "rccl1.py"
  script.write_text(_RCCL1 % dict(root=ROOT, pkg=PKG, rdzv=str(tmp_path / "rdzv")))
  out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
  assert out.returncode == 0, out.stderr[-2000:]
  assert "rccl single rank ok" in out.stdout


_DUP = r"""
import sys
sys.path[:0] = [%(root)r, %(pkg)r]
from causalimpact import _comm, _native
try:
  _comm.Comm.from_env()
except _native.NativeError as e:
  print("NativeError:", e)
  raise SystemExit(3)
raise SystemExit(0)
"""


def test_rccl_refuses_two_ranks_on_one_device_with_an_error(tmp_path):
  from causalimpact import _comm
  script = tmp_path / "dup.py"
  script.write_text(_DUP % dict(root=ROOT, pkg=PKG))
  env = dict(os.environ, NCCL_DEBUG="WARN")
  codes = _comm.spawn_ranks(2, [sys.executable, str(script)], env=env, transport="rccl",
                            devices=[0, 0], timeout=240)
  assert all(c != 0 for c in codes), codes           # an error on every rank, no hang


def test_bench_single_rank_through_the_rccl_path_matches_plain_bench():
  """bench.py with CI_BENCH_FORCE_DIST=1: the N > 1 code path (C-ABI communicator, barrier,
  max-reduce of the time, gather from HBM, diagnostics all-reduce) with one rank."""
  env = dict(os.environ, CI_BENCH_FORCE_DIST="1")
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "3",
         "--no-cpu-baseline", "--no-pmc"]
  # the JSON line must be the LAST line of the output (librccl's banner is flushed before it)
  forced = json.loads(subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600,
                                     check=True).stdout.strip().splitlines()[-1])
  plain = json.loads(subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                                    check=True).stdout.strip().splitlines()[-1])
  assert "ranks_seen=1" in forced["config"]["collectives"]
  assert forced["config"]["collectives"].startswith("rccl"), forced["config"]["collectives"]
  assert plain["config"]["collectives"] == "none"
  assert forced["n_gpus"] == plain["n_gpus"] == 1
  assert forced["split_rhat"] == plain["split_rhat"] and forced["ess"] == plain["ess"]
  # same work per step either way; a loose band: 8 steps of 9 ms on a box that is not ours alone
  assert abs(forced["value"] / plain["value"] - 1.0) < 0.3


def test_bench_gpus_2_starts_its_own_ranks_and_reports_the_whole_job():
  """`python bench.py --gpus 2` with no launcher around it (the form the driver uses): bench.py
  spawns its two ranks, they meet through the C-ABI communicator, rank 0 prints one JSON line for
  the whole job.  On this one-GPU box both ranks share GPU 0 over the host transport
  (CI_COMM_TRANSPORT / CI_COMM_DEVICES); on a node with 2+ GPUs the same command uses RCCL."""
  env = dict(os.environ, CI_COMM_TRANSPORT="host", CI_COMM_DEVICES="0,0")
  env.pop("WORLD_SIZE", None)
  env.pop("RANK", None)
  cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"]
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, check=True).stdout
  line = json.loads(out.strip().splitlines()[-1])
  assert line["n_gpus"] == 2 and line["scaling"] == "weak"
  assert line["config"]["chains_total"] == 2 * line["config"]["chains_per_gpu"]
  assert "ranks_seen=2" in line["config"]["collectives"]
  assert "spawned its own ranks" in line["config"]["launcher"]
  assert line["value"] > 0 and set(line["split_rhat"]) == {"observation_noise_scale", "level_scale"}
  assert "cpu_baseline" not in line or line.get("cpu_baseline") is None or line["n_gpus"] == 2


def test_bench_under_torch_distributed_run_the_drivers_command_line():
  """The driver's N > 1 form, verbatim: `python -m torch.distributed.run --nnodes=1
  --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...`.  The launcher
  only exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*: the ranks never import torch, they meet
  through the rendezvous file derived from that environment and the C-ABI communicator.  On this
  one-GPU box CI_COMM_DEVICES folds both ranks onto GPU 0 and RCCL (which refuses two ranks on one
  device) hands over to the host transport, labelled as such."""
  pytest.importorskip("torch")
  import socket
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CI_COMM_TRANSPORT")}
  env.update(CI_COMM_DEVICES="0,0", CI_COMM_INIT_TIMEOUT_S="60")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
         "--gpus", "2", "--steps", "2", "--warmup", "1"]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]                     # rank 0 alone prints the line
  line = json.loads(lines[0])
  assert line["n_gpus"] == 2 and line["config"]["chains_total"] == 2 * line["config"]["chains_per_gpu"]
  assert line["config"]["ranks_seen"] == 2
  assert "external" in line["config"]["launcher"]
  assert line["config"]["collectives"].startswith(("host", "rccl"))
  assert line["value"] > 0 and line["cpu_baseline"]["value"] is None

