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
this module does not import) find each other too; the n-th communicator a process builds uses
`<path>.<n>`, rank 0 clears a leftover before publishing and readers ignore files older than an
hour, so neither a second communicator nor a crashed earlier run can hand out a stale id.

`connect()` is what multi-GPU callers (bench.py) use: it ALWAYS builds the host transport first
(the control plane), then tries RCCL under a deadline, lets the ranks agree on the outcome and --
if any rank could not join the RCCL communicator -- carries on over the host transport with the
reason in `Comm.transport` ("host (rccl failed: ...)").  An N-GPU run can therefore not fail
silently or hang in communicator set-up.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import hashlib
import sys
import tempfile
import threading
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
  L.ci_comm_set_timeout.argtypes = [C.c_void_p, C.c_double]
  L.ci_comm_all_reduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32]
  L.ci_comm_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
  L.ci_comm_session_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
  L.ci_comm_ll_session_all_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
  L.ci_comm_destroy.argtypes = [C.c_void_p]
  L._ci_comm_bound = True
  return L


_SEQ = [0]                                # communicators this process has built from the default path
STALE_SECONDS = 3600.0                    # a rendezvous file older than this belongs to an earlier run


def rendezvous_path() -> str:
  """Where rank 0 publishes the unique id of the NEXT communicator this process builds.  Every rank
  builds its communicators in the same order, so the running number keeps a second communicator
  off the first one's file (rank 0 unlinks a file only after a barrier; a faster rank could
  otherwise read the previous id)."""
  p = os.environ.get("CI_COMM_RDZV")
  if not p:
    port = os.environ.get("MASTER_PORT", "0")
    run = os.environ.get("TORCHELASTIC_RUN_ID", "none")
    restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
    p = os.path.join(tempfile.gettempdir(),
                     f"ci_comm_{os.getuid()}_{port}_{run}_{restart}_{os.getppid()}")
  seq = _SEQ[0]
  _SEQ[0] += 1
  return p if seq == 0 else f"{p}.{seq}"


NONCE_BYTES = 16


def launch_nonce() -> bytes:
  """What ties a rendezvous file to THIS launch: `spawn_ranks` hands every rank a fresh random
  $CI_COMM_NONCE; under another launcher it is derived from the launcher's pid and run id.  A
  leftover of a crashed earlier run at the same path carries another nonce and is ignored however
  young it is (the one-hour staleness rule alone let a non-zero rank that polled
  before rank 0's os.replace attach to a dead segment)."""
  src = os.environ.get("CI_COMM_NONCE")
  if not src:
    # An explicit $CI_COMM_RDZV is a rendezvous the caller arranged for ranks that may have
    # DIFFERENT parents (two shells, one scheduler task per rank): the parent's pid must not enter
    # the nonce there (such ranks never accepted rank 0's file).  Such launchers
    # should export a fresh $CI_COMM_NONCE per launch; without one a rerun at the same path is told
    # from a crashed run's leftover only by the file's age.
    who = "rdzv:" + os.environ["CI_COMM_RDZV"] if os.environ.get("CI_COMM_RDZV") else str(os.getppid())
    src = ":".join([who, os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                    os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                    os.environ.get("MASTER_PORT", "0")])
  return hashlib.sha256(src.encode()).digest()[:NONCE_BYTES]


def _exchange_id(rank: int, transport: int, path: str, timeout: float = 300.0) -> bytes:
  L = _bind(_native.load())
  nonce = launch_nonce()
  if rank == 0:
    buf = (C.c_uint8 * ID_BYTES)()
    _native._check(L.ci_comm_unique_id(transport, buf))
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "wb") as f:
      f.write(bytes(buf) + nonce)
    os.replace(tmp, path)                 # atomically replaces a leftover of a crashed run, too
    return bytes(buf)
  t0 = time.monotonic()
  why = "no file"
  while True:
    try:
      with open(path, "rb") as f:
        b = f.read()
        fresh = time.time() - os.fstat(f.fileno()).st_mtime < STALE_SECONDS
      if len(b) == ID_BYTES + NONCE_BYTES and fresh and b[ID_BYTES:] == nonce:
        return b[:ID_BYTES]
      if len(b) == ID_BYTES + NONCE_BYTES:
        why = ("a file is there but carries another launch nonce (a leftover of an earlier run, or "
               "rank 0 was started with a different $CI_COMM_NONCE / parent / MASTER_PORT: export the "
               "same $CI_COMM_NONCE to every rank)" if fresh else "the file there is stale")
    except FileNotFoundError:
      pass
    if time.monotonic() - t0 > timeout:
      raise _native.NativeError(f"rank {rank}: no unique id at {path} after {timeout:.0f} s ({why})")
    time.sleep(0.01)


class Comm:
  """One rank of a communicator (`ci_comm`).  All methods are collective."""

  def __init__(self, rank: int, world: int, device: int = 0, transport: str = "rccl",
               path: Optional[str] = None, uid: Optional[bytes] = None):
    """`uid`: the 128-byte id when the caller has already distributed it (`connect` sends the
    RCCL id over the host communicator); otherwise rank 0 publishes it in the file `path`."""
    if transport not in TRANSPORTS:
      raise ValueError(f"transport must be one of {sorted(TRANSPORTS)}, got {transport!r}")
    self._lib = _bind(_native.load())
    self.rank, self.world, self.device, self.transport = int(rank), int(world), int(device), transport
    self.kind = transport               # "rccl" | "host": what actually carries the collectives
    self._path = None
    if uid is None:
      self._path = path or rendezvous_path()
      uid = _exchange_id(self.rank, TRANSPORTS[transport], self._path)
    self._h = C.c_void_p()
    buf = (C.c_uint8 * ID_BYTES).from_buffer_copy(uid)
    _native._check(self._lib.ci_comm_create(TRANSPORTS[transport], buf, self.rank, self.world,
                                            self.device, C.byref(self._h)))
    seen = C.c_int32(0)
    _native._check(self._lib.ci_comm_info(self._h, None, None, C.byref(seen)))
    self.ranks_seen = int(seen.value)
    if self.ranks_seen != self.world:
      raise _native.NativeError(f"rank {self.rank}: communicator sees {self.ranks_seen} ranks, "
                                f"world size is {self.world}")
    if self._path is not None:
      self.barrier()                    # every rank has read the id: rank 0 may remove the file
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

  @property
  def hard_exit(self) -> bool:
    """True when an abandoned RCCL attempt may still be blocked inside the library on a helper
    thread: the process should leave through os._exit after flushing its output."""
    return getattr(self, "_hard_exit", False)

  def barrier(self):
    _native._check(self._lib.ci_comm_barrier(self._h))

  def set_timeout(self, seconds: float):
    """Bound of one collective on this communicator (0: $CI_COMM_TIMEOUT_S, default 300 s)."""
    _native._check(self._lib.ci_comm_set_timeout(self._h, float(seconds)))

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
    # zeros: a field the model does not have (the slope of a local-level fit) has no device buffer
    # and reads as zeros, as `Session.fetch` returns it
    out = np.zeros((self.world,) + tuple(shp), np.float32)
    _native._check(fn(self._h, session._h, FIELDS[field], out.ctypes.data if out.size else None))
    return out

  def close(self):
    if getattr(self, "_h", None):
      self._lib.ci_comm_destroy(self._h)
      self._h = C.c_void_p()
    ctl = getattr(self, "ctl", None)      # the control plane `connect` built next to an RCCL comm
    if ctl is not None:
      self.ctl = None
      ctl.close()

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass


def _with_deadline(fn, seconds: float):
  """Runs fn() on a helper thread (ctypes releases the GIL inside the library).  Returns
  (result, error text or None, finished): a call still blocked after `seconds` is abandoned
  (daemon thread) and reported as not finished."""
  box = {}

  def run():
    try:
      box["res"] = fn()
    except BaseException as e:  # pylint: disable=broad-except
      box["err"] = f"{type(e).__name__}: {e}"

  t = threading.Thread(target=run, daemon=True)
  t.start()
  t.join(seconds)
  if t.is_alive():
    return None, f"no answer within {seconds:.0f} s", False
  return box.get("res"), box.get("err"), True


def connect(rank: Optional[int] = None, world: Optional[int] = None, device: Optional[int] = None,
            transport: Optional[str] = None, path: Optional[str] = None,
            init_timeout: Optional[float] = None) -> Comm:
  """The communicator of a multi-GPU run, built so that set-up can neither hang nor fail silently.

  1. the HOST transport (shared memory, always available on one node) comes up first;
  2. for transport "rccl": rank 0 draws the RCCL id and sends it over (1); every rank joins the
     RCCL communicator under a deadline ($CI_COMM_INIT_TIMEOUT_S, default 90 s) and proves it with
     one all-reduce; the ranks agree on the outcome over (1);
  3. everybody in -> the RCCL communicator is returned (`transport == "rccl"`); anybody out ->
     every rank drops its RCCL attempt and the host communicator is returned with
     `transport == "host (rccl failed: <first reason>)"`.
  rank / world / device default to RANK / WORLD_SIZE / LOCAL_RANK."""
  rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
  world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
  device = int(os.environ.get("LOCAL_RANK", str(rank))) if device is None else int(device)
  transport = transport or os.environ.get("CI_COMM_TRANSPORT", "rccl")
  if transport not in TRANSPORTS:
    raise ValueError(f"transport must be one of {sorted(TRANSPORTS)}, got {transport!r}")
  if init_timeout is None:
    init_timeout = float(os.environ.get("CI_COMM_INIT_TIMEOUT_S", "90"))
  ctl = Comm(rank, world, device, "host", path)
  if transport == "host":
    return ctl
  L = _bind(_native.load())
  # ---- the id, from rank 0 over the control plane
  uid, why = bytes(ID_BYTES), ""
  if rank == 0:
    buf = (C.c_uint8 * ID_BYTES)()
    # (_check on the helper thread: the library's error string is thread-local)
    _, err, _ = _with_deadline(lambda: _native._check(L.ci_comm_unique_id(TRANSPORTS["rccl"], buf)),
                               init_timeout)
    if err is None:
      uid = bytes(buf)
    else:
      why = err
  msg = np.zeros(ID_BYTES + 1, np.uint8)
  if rank == 0:
    msg[:ID_BYTES] = np.frombuffer(uid, np.uint8)
    msg[ID_BYTES] = 0 if why else 1
  got = ctl.all_gather(msg)[0]
  have_id = bool(got[ID_BYTES])
  uid = got[:ID_BYTES].tobytes()
  # ---- join + one proving all-reduce, under the deadline
  data, finished = None, True
  if have_id:
    def join():
      c = Comm(rank, world, device, "rccl", uid=uid)
      # the proving all-reduce may not outlive the deadline of the join itself: a join abandoned
      # after init_timeout would otherwise keep its kernel spinning on the GPU for up to
      # $CI_COMM_TIMEOUT_S while the fall-back run is being timed
      c.set_timeout(init_timeout)
      s = c.all_reduce([1.0])
      if int(round(float(s[0]))) != world:
        raise _native.NativeError(f"proving all-reduce returned {float(s[0])}, expected {world}")
      c.set_timeout(0.0)
      return c
    data, err, finished = _with_deadline(join, init_timeout)
    if err is not None:
      why, data = err, None
  # ---- agree: everybody in, or nobody
  text = np.zeros(200, np.uint8)
  b = (f"rank {rank}: {why}" if why else "").encode()[:200]
  text[:len(b)] = np.frombuffer(b, np.uint8)
  verdict = ctl.all_gather(np.concatenate([[np.uint8(0 if data is not None else 1)], text]))
  if not verdict[:, 0].any():
    data.ctl = ctl                      # keeps the control plane alive next to the data plane
    return data
  first = int(np.argmax(verdict[:, 0] != 0))
  reason = bytes(verdict[first, 1:]).rstrip(b"\0").decode(errors="replace") or f"rank {first} could not join"
  if data is not None:
    data.close()
  ctl.transport = f"host (rccl failed: {reason})"
  ctl._hard_exit = not finished         # pylint: disable=protected-access
  return ctl


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
  nonce = os.urandom(16).hex()            # ties the rendezvous files to this launch (launch_nonce)
  procs = []
  for r in range(world):
    e = dict(os.environ if env is None else env)
    e.update(RANK=str(r), WORLD_SIZE=str(world), CI_COMM_RDZV=path, CI_COMM_TRANSPORT=transport,
             CI_COMM_NONCE=nonce,
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
