"""Test-only adapter: a `torch.distributed` process group (gloo on CPU) behind the interface of
`causalimpact._comm.Comm` (rank / world / all_gather / all_reduce), so that the world_size-2 CPU
tests can drive `_distributed.fit_sharded` without the C-ABI transport.  The product package does
not import PyTorch anywhere (round-5 review: this class used to live in `_distributed.py`)."""
import numpy as np


class TorchComm:
  """`torch.distributed` (gloo in the CPU tests) behind the interface of `_comm.Comm`."""

  def __init__(self, dist, torch, group, device):
    self._dist, self._torch, self._group = dist, torch, group
    self._dev = torch.device(device) if device else torch.device("cpu")
    self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

  def all_gather(self, a: np.ndarray) -> np.ndarray:
    mine = self._torch.from_numpy(np.ascontiguousarray(a)).to(self._dev)
    parts = [self._torch.empty_like(mine) for _ in range(self.world)]
    self._dist.all_gather(parts, mine, group=self._group)
    return np.stack([p.cpu().numpy() for p in parts])

  def all_reduce(self, values, op: int = 0) -> np.ndarray:
    t = self._torch.from_numpy(np.array(values, dtype=np.float64, copy=True)).to(self._dev)
    red = self._dist.ReduceOp.MAX if op == 1 else self._dist.ReduceOp.SUM
    self._dist.all_reduce(t, op=red, group=self._group)
    return t.cpu().numpy()


def from_initialized_group(group=None, device=None):
  import torch  # pylint: disable=import-outside-toplevel
  import torch.distributed as dist  # pylint: disable=import-outside-toplevel
  assert dist.is_available() and dist.is_initialized()
  return TorchComm(dist, torch, group, device)
