"""The PyTorch-free collectives of the C-ABI (`ci_comm_*`, causalimpact/_comm.py) on CPU.

Two (and three) processes over the HOST transport -- the shared-memory twin of the RCCL transport
that bench.py / `fit_sharded` use on GPUs; same entry points, same results -- exercise: the file
rendezvous, all-gather, all-reduce (sum / max, identical bits on every rank), payloads larger than
one staging slot, `fit_sharded` with uneven and empty chain blocks equal to one process, and the
self-launcher (`spawn_ranks`).  No torch import anywhere in these processes.
"""
import multiprocessing as mp
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "tfp-causalimpact_amd")


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


def _worker(rank, world, path, num_chains, q):
  sys.path[:0] = [ROOT, PKG]
  from causalimpact import _comm, _distributed as d
  comm = _comm.Comm(rank, world, device=0, transport="host", path=path)
  assert comm.ranks_seen == world
  a = np.arange(6, dtype=np.float32).reshape(2, 3) + 10 * rank
  g = comm.all_gather(a)
  big = np.full(3_000_000, rank + 1, np.float32)            # 12 MB: three staging slots
  gb = comm.all_gather(big)
  s = comm.all_reduce(np.array([0.1 * (rank + 1), 1.0, -rank]), _comm.SUM)
  m = comm.all_reduce(np.array([float(rank), -float(rank)]), _comm.MAX)
  e = comm.all_gather(np.zeros((0, 4), np.float64))          # empty payloads are legal
  res = d.fit_sharded(_fake_fit, num_chains, comm=comm)
  comm.barrier()
  q.put((rank, g, [float(gb[r].min()) for r in range(world)] + [float(gb[r].max()) for r in range(world)],
         s, m, e.shape, res["posterior_trajectories"], res["posterior_means"],
         {k: res[k] for k in ("split_rhat", "ess_bulk", "ess_tail")}))
  comm.close()
  assert "torch" not in sys.modules


@pytest.mark.parametrize("world,num_chains", [(2, 4), (2, 5), (2, 1), (3, 7)])
def test_host_transport_ranks_equal_one_process(world, num_chains):
  sys.path[:0] = [PKG]
  from causalimpact import _distributed as d
  single = d.fit_sharded(_fake_fit, num_chains)
  fd, path = tempfile.mkstemp(prefix="ci_comm_test_")
  os.close(fd)
  os.unlink(path)
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  procs = [ctx.Process(target=_worker, args=(r, world, path, num_chains, q)) for r in range(world)]
  for p in procs:
    p.start()
  got = [q.get(timeout=180) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert not os.path.exists(path)                            # rank 0 removed the rendezvous file
  want_sum = np.array([sum(0.1 * (r + 1) for r in range(world)), float(world),
                       -float(sum(range(world)))])
  ref_sum = None
  for rank, g, big, s, m, eshape, traj, means, diag in got:
    for r in range(world):
      np.testing.assert_array_equal(g[r], np.arange(6, dtype=np.float32).reshape(2, 3) + 10 * r)
      assert big[r] == r + 1 and big[world + r] == r + 1
    np.testing.assert_allclose(s, want_sum, rtol=1e-15)
    ref_sum = s if ref_sum is None else ref_sum
    np.testing.assert_array_equal(s, ref_sum)                # the same BITS on every rank
    np.testing.assert_array_equal(m, [world - 1.0, 0.0])
    assert eshape == (world, 0, 4)
    np.testing.assert_array_equal(traj, single["posterior_trajectories"])
    np.testing.assert_array_equal(means, single["posterior_means"])
    for name in ("split_rhat", "ess_bulk", "ess_tail"):
      for k, v in single[name].items():
        np.testing.assert_allclose(diag[name][k], v, rtol=1e-9, err_msg=f"{name}.{k}")


_RANK_SCRIPT = r"""
import os, sys
sys.path[:0] = [%r, %r]
import numpy as np
from causalimpact import _comm
c = _comm.Comm.from_env()
tot = c.all_reduce(np.array([c.rank + 1.0]))
if c.rank == 0:
  print("ranks_seen=%%d world=%%d sum=%%g device=%%d" %% (c.ranks_seen, c.world, tot[0], c.device))
c.close()
assert "torch" not in sys.modules
"""


def test_spawn_ranks_launches_one_process_per_rank(tmp_path):
  """`spawn_ranks` is what `python bench.py --gpus N` uses when no launcher started it."""
  script = tmp_path / "rank.py"
  script.write_text(_RANK_SCRIPT % (ROOT, PKG))
  launcher = tmp_path / "launch.py"
  launcher.write_text(
      "import sys\nsys.path[:0] = [%r, %r]\nfrom causalimpact import _comm\n"
      "codes = _comm.spawn_ranks(3, [sys.executable, %r], transport='host')\n"
      "assert codes == [0, 0, 0], codes\n" % (ROOT, PKG, str(script)))
  out = subprocess.run([sys.executable, str(launcher)], capture_output=True, text=True, timeout=180)
  assert out.returncode == 0, out.stderr
  assert out.stdout.strip() == "ranks_seen=3 world=3 sum=6 device=0"


def test_create_rejects_bad_ranks_before_any_transport_work():
  sys.path[:0] = [PKG]
  import ctypes as C
  from causalimpact import _comm, _native
  L = _comm._bind(_native.load())   # pylint: disable=protected-access
  h = C.c_void_p()
  buf = (C.c_uint8 * _comm.ID_BYTES)()
  assert L.ci_comm_create(1, buf, 2, 2, 0, C.byref(h)) != 0
  assert b"rank" in L.ci_last_error()
  assert L.ci_comm_create(7, buf, 0, 1, 0, C.byref(h)) != 0
  assert b"transport" in L.ci_last_error()


_CONNECT = r"""
import os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np
from causalimpact import _comm
c = _comm.connect()
assert c.ranks_seen == c.world == 2
s = c.all_reduce([1.0 + c.rank])
g = c.all_gather(np.array([c.rank], np.float32))
assert float(s[0]) == 3.0 and g.ravel().tolist() == [0.0, 1.0]
with open(os.path.join(%(out)r, "rank%%d.txt" %% c.rank), "w") as f:
  f.write(c.kind + "|" + c.transport)
c.barrier()
hard = c.hard_exit
c.close()
if hard:
  os._exit(0)
"""


def test_connect_falls_back_to_host_when_rccl_cannot_start(tmp_path):
  """No GPU here: librccl cannot even draw a unique id.  `_comm.connect` (bench.py's entry) must
  neither raise nor hang: all ranks agree to carry on over the host transport and say why."""
  sys.path[:0] = [PKG]
  from causalimpact import _comm
  script = tmp_path / "connect.py"
  script.write_text(_CONNECT % dict(root=ROOT, pkg=PKG, out=str(tmp_path)))
  env = dict(os.environ, CI_COMM_INIT_TIMEOUT_S="60")
  codes = _comm.spawn_ranks(2, [sys.executable, str(script)], env=env, transport="rccl", timeout=300)
  assert codes == [0, 0]
  for r in range(2):
    kind, transport = (tmp_path / f"rank{r}.txt").read_text().split("|", 1)
    assert kind == "host" and transport.startswith("host (rccl failed:"), transport


def test_second_communicator_of_a_process_uses_its_own_rendezvous_file(monkeypatch):
  sys.path[:0] = [PKG]
  from causalimpact import _comm
  monkeypatch.setenv("CI_COMM_RDZV", "/tmp/ci_rdzv_x")
  monkeypatch.setattr(_comm, "_SEQ", [0])
  assert _comm.rendezvous_path() == "/tmp/ci_rdzv_x"
  assert _comm.rendezvous_path() == "/tmp/ci_rdzv_x.1"
  assert _comm.rendezvous_path() == "/tmp/ci_rdzv_x.2"


def _nonce_worker(rank, world, path, nonce, q):
  sys.path[:0] = [ROOT, PKG]
  os.environ["CI_COMM_NONCE"] = nonce
  from causalimpact import _comm
  comm = _comm.Comm(rank, world, device=0, transport="host", path=path)
  s = comm.all_reduce([1.0])
  q.put((rank, float(s[0])))
  comm.close()


def test_a_young_leftover_of_another_launch_is_not_taken_for_this_launchs_id(tmp_path):
  """a rendezvous file younger than the staleness window, left at the same path by
  a crashed run, was accepted by a rank that polled before rank 0 replaced it.  The file now ends
  with the launch nonce: rank 1 ignores the leftover (128 bytes of a dead id + another nonce) and
  attaches only to what rank 0 of ITS launch publishes."""
  sys.path[:0] = [PKG]
  from causalimpact import _comm
  path = str(tmp_path / "rdzv")
  with open(path, "wb") as f:                       # the leftover: a well-formed file, foreign nonce
    f.write(bytes(range(128)) + b"\x07" * _comm.NONCE_BYTES)
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  late = ctx.Process(target=_nonce_worker, args=(1, 2, path, "this-launch", q))
  late.start()                                      # polls the leftover for a while ...
  import time
  time.sleep(1.0)
  assert late.is_alive() and q.empty()              # ... and has not attached to anything
  first = ctx.Process(target=_nonce_worker, args=(0, 2, path, "this-launch", q))
  first.start()
  got = sorted(q.get(timeout=120) for _ in range(2))
  first.join(60); late.join(60)
  assert got == [(0, 2.0), (1, 2.0)]


def test_a_collective_whose_peer_died_times_out_with_an_error(tmp_path):
  """Every collective is bounded per communicator (ci_comm_set_timeout): rank 1 attaches and then
  exits without taking part; rank 0's barrier must return an error after ~1 s, not hang."""
  script = tmp_path / "die.py"
  script.write_text(r"""
import os, sys, time
sys.path[:0] = [%r, %r]
from causalimpact import _comm, _native
c = _comm.Comm(int(os.environ["RANK"]), 2, device=0, transport="host")
if c.rank == 1:
  os._exit(0)                      # dies after the set-up barrier, before the next collective
c.set_timeout(1.0)
t0 = time.monotonic()
try:
  c.all_reduce([1.0])
except _native.NativeError as e:
  dt = time.monotonic() - t0
  assert "timed out" in str(e) and dt < 30.0, (str(e), dt)
  print("bounded", flush=True)
  os._exit(0)
raise SystemExit("the collective returned although a rank was dead")
""" % (ROOT, PKG))
  sys.path[:0] = [PKG]
  from causalimpact import _comm
  launcher = tmp_path / "launch.py"
  launcher.write_text(
      "import sys\nsys.path[:0] = [%r, %r]\nfrom causalimpact import _comm\n"
      "codes = _comm.spawn_ranks(2, [sys.executable, %r], transport='host', timeout=120)\n"
      "assert codes == [0, 0], codes\n" % (ROOT, PKG, str(script)))
  out = subprocess.run([sys.executable, str(launcher)], capture_output=True, text=True, timeout=180)
  assert out.returncode == 0, out.stderr + out.stdout
  assert "bounded" in out.stdout
