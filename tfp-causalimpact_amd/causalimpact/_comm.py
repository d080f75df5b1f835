"""Collectives for chain gather / diagnostics through the C-ABI (`ci_comm_*`) -- no PyTorch.

The reference runs one chain in one process and never communicates; here
independent chains are sharded over the GPUs of one node, one rank (process) per GPU, and the
only communication happens AFTER the fit: an all-gather of result blocks that are still resident
in HBM and one small all-reduce of the diagnostics' partial sums.

  transport "rccl"  librccl behind `ci_comm_*` (ncclAllGather / ncclAllReduce over xGMI);
  transport "host"  a shared-memory segment on one node: ranks that share a GPU (RCCL refuses two
                    ranks on one device) and GPU-less tests.

Rendezvous: rank 0 asks the library for the 128-byte unique id and publishes it in a file (atomic
rename); the other ranks poll for it.  The path is `$CI_COMM_RDZV` (set by `spawn_ranks`) or is
derived from MASTER_PORT and the launcher's pid, so ranks started by `torch.distributed.run` (which
this module does not import) find each other too.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from causalimpact import _native

TRANSPORTS = {"rccl": 0, "host": 1}       # == CI_COMM_RCCL / CI_COMM_HOST
ID_BYTES = 128                            # == CI_COMM_ID_BYTES
SUM, MAX = 0, 1                           # == CI_COMM_SUM / CI_COMM_MAX
FIELDS = {name: i for i, name in enumerate(_native._OUT_FIELDS)}   # == CI_FIELD_*


def _bind(L):
  if getattr(L, "_ci_comm_bound", False):
    return L
  L.ci_comm_unique_id.argtypes = [C.c_int32, C.c_void_p]
  L.ci_comm_create.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                               C.POINTER(C.c_void_p)]
  L.ci_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                             C.POINTER(C.c_int32)]
  L.ci_comm_barrier.argtypes = [C.c_void_p]
  L.ci_comm_all_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
  L.ci_comm_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
  L.ci_comm_session_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
  L.ci_comm_ll_session_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
  L.ci_comm_destroy.argtypes = [C.c_void_p]
  L._ci_comm_bound = True
  return L


def rendezvous_path() -> str:
  """Where rank 0 publishes the unique id."""
  p = os.environ.get("CI_COMM_RDZV")
  if p:
    return p
  port = os.environ.get("MASTER_PORT", "0")
  run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
  return os.path.join(tempfile.gettempdir(), f"ci_comm_{os.getuid()}_{port}_{run}_{os.getppid()}")


def _exchange_id(rank: int, transport: int, path: str, timeout: float = 300.0) -> bytes:
  L = _bind(_native.load())
  if rank == 0:
    buf = (C.c_uint8 * ID_BYTES)()
    _native._check(L.ci_comm_unique_id(transport, buf))
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
      f.write(bytes(buf))
    os.replace(tmp, path)
    return bytes(buf)
  t0 = time.monotonic()
  while True:
    try:
      with open(path, "rb") as f:
        b = f.read()
      if len(b) == ID_BYTES:
        return b
    except FileNotFoundError:
      pass
    if time.monotonic() - t0 > timeout:
      raise _native.NativeError(f"rank {rank}: no unique id at {path} after {timeout:.0f} s")
    time.sleep(0.01)


class Comm:
  """One rank of a communicator (`ci_comm`).  All methods are collective."""

  def __init__(self, rank: int, world: int, device: int = 0, transport: str = "rccl",
               path: Optional[str] = None):
    if transport not in TRANSPORTS:
      raise ValueError(f"transport must be one of {sorted(TRANSPORTS)}, got {transport!r}")
    self._lib = _bind(_native.load())
    self.rank, self.world, self.device, self.transport = int(rank), int(world), int(device), transport
    self._path = path or rendezvous_path()
    uid = _exchange_id(self.rank, TRANSPORTS[transport], self._path)
    self._h = C.c_void_p()
    buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
    _native._check(self._lib.ci_comm_create(TRANSPORTS[transport], buf, self.rank, self.world,
                                            self.device, C.byref(self._h)))
    seen = C.c_int32(0)
    _native._check(self._lib.ci_comm_info(self._h, None, None, C.byref(seen)))
    self.ranks_seen = int(seen.value)
    self.barrier()                      # every rank has read the id: rank 0 may remove the file
    if self.rank == 0:
      try:
        os.unlink(self._path)
      except OSError:
        pass

  @classmethod
  def from_env(cls, transport: Optional[str] = None) -> "Comm":
    """RANK / LOCAL_RANK / WORLD_SIZE as `spawn_ranks` and torch.distributed.run export them."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = int(os.environ.get("LOCAL_RANK", str(rank)))
    transport = transport or os.environ.get("CI_COMM_TRANSPORT", "rccl")
    return cls(rank, world, device, transport)

  def barrier(self):
    _native._check(self._lib.ci_comm_barrier(self._h))

  def all_reduce(self, values, op: int = SUM) -> np.ndarray:
    """float64 values reduced over the ranks (same bits on every rank)."""
    v = np.array(values, dtype=np.float64, copy=True, order="C")
    _native._check(self._lib.ci_comm_all_reduce(self._h, v.ctypes.data, int(v.size), int(op)))
    return v

  def all_gather(self, a: np.ndarray) -> np.ndarray:
    """[world, *a.shape]: every rank's array (equal shapes and dtypes), rank order."""
    a = np.ascontiguousarray(a)
    out = np.empty((self.world,) + a.shape, a.dtype)
    _native._check(self._lib.ci_comm_all_gather(self._h, a.ctypes.data if a.size else None,
                                                out.ctypes.data if out.size else None,
                                                int(a.nbytes)))
    return out

  def session_all_gather(self, session, field: str) -> np.ndarray:
    """Every rank's device-resident result array `field` of a finished `Session` /
    `LogLikSession` HMC fit -> [world, B, C, ...] on the host, gathered from HBM."""
    pb = session.pb if hasattr(session, "pb") else None
    if pb is not None:
      shp = _native.output_shapes(pb)[field]
      fn = self._lib.ci_comm_session_all_gather
    else:
      Cn, S = session._hmc_shape       # pylint: disable=protected-access
      hp = _native.make_problem(T=session.T, P=session.P, has_slope=session.D == 2,
                                num_seasons=getattr(session, "num_seasons", ()), num_warmup=0,
                                num_results=S, num_chains=Cn)
      shp = _native.output_shapes(hp)[field]
      fn = self._lib.ci_comm_ll_session_all_gather
    out = np.empty((self.world,) + tuple(shp), np.float32)
    _native._check(fn(self._h, session._h, FIELDS[field], out.ctypes.data if out.size else None))
    return out

  def close(self):
    if getattr(self, "_h", None):
      self._lib.ci_comm_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


def spawn_ranks(world: int, argv: Sequence[str], env: Optional[Dict[str, str]] = None,
                transport: str = "rccl", devices: Optional[Sequence[int]] = None,
                timeout: Optional[float] = None) -> List[int]:
  """Starts `world` copies of `argv` (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE /
  CI_COMM_RDZV / CI_COMM_TRANSPORT in their environment), waits for them and returns their exit
  codes.  stdout / stderr are inherited, so whatever rank 0 prints is this process's output.
  `devices[r]` is rank r's device ordinal (default r)."""
  fd, path = tempfile.mkstemp(prefix="ci_comm_rdzv_")
  os.close(fd)
  os.unlink(path)
  procs = []
  for r in range(world):
    e = dict(os.environ if env is None else env)
    e.update(RANK=str(r), WORLD_SIZE=str(world), CI_COMM_RDZV=path, CI_COMM_TRANSPORT=transport,
             LOCAL_RANK=str(r if devices is None else devices[r]), CI_COMM_SPAWNED="1")
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs.append(subprocess.Popen(list(argv), env=e))
  # wait for all ranks; when one dies the others would wait for it in a collective: give them a
  # short grace period, then stop them (the caller sees the non-zero exit codes)
  deadline = None if timeout is None else time.monotonic() + timeout
  failed_at = None
  while any(p.poll() is None for p in procs):
    now = time.monotonic()
    if failed_at is None and any(p.poll() not in (None, 0) for p in procs):
      failed_at = now
    if (failed_at is not None and now - failed_at > 5.0) or (deadline is not None and now > deadline):
      for p in procs:
        if p.poll() is None:
          p.kill()
      break
    time.sleep(0.02)
  codes = [p.wait() for p in procs]
  try:
    os.unlink(path)
  except OSError:
    pass
  return codes


def self_launch(world: int) -> Optional[int]:
  """`python script.py --gpus N` started by hand or by a driver WITHOUT a launcher: re-executes
  this very command line once per GPU and returns the worst exit code; returns None inside a rank
  (WORLD_SIZE already set) or when world == 1, i.e. when the caller should just carry on."""
  if world <= 1 or "WORLD_SIZE" in os.environ:
    return None
  # CI_COMM_TRANSPORT / CI_COMM_DEVICES (e.g. "host" and "0,0": two ranks sharing GPU 0) let a
  # one-GPU box run the N-rank path; the defaults are RCCL and rank r on device r
  transport = os.environ.get("CI_COMM_TRANSPORT", "rccl")
  devices = os.environ.get("CI_COMM_DEVICES")
  devs = [int(d) for d in devices.split(",")] if devices else None
  if devs is not None and len(devs) != world:
    raise ValueError(f"CI_COMM_DEVICES names {len(devs)} devices for {world} ranks")
  codes = spawn_ranks(world, [sys.executable] + sys.argv, transport=transport, devices=devs)
  return max((abs(c) for c in codes), default=0)
