"""Chain sharding over the GPUs of one node (one process per GPU, torch.distributed).

The reference runs one chain in one process; chains are
independent given the data, so ranks take contiguous blocks of chain ids with NO collective
on the data path.  Chain c always uses RNG stream c (`chain_offset`), hence the pooled
result is identical for any world size.  RCCL (backend "nccl" on ROCm) / gloo is used only
after sampling: an all-gather of the per-chain draws needed for pooled summaries and an
all-reduce of per-chain moments for split-R-hat.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np


def chain_block(num_chains: int, rank: int, world_size: int) -> Tuple[int, int]:
  """(first chain id, count) of `rank`'s contiguous block; blocks differ by at most one."""
  base, extra = divmod(num_chains, world_size)
  count = base + (1 if rank < extra else 0)
  first = rank * base + min(rank, extra)
  return first, count


def _moments(draws: np.ndarray) -> np.ndarray:
  """[C_local, S] -> [C_local, 4]: split-half means and variances per chain."""
  half = draws.shape[1] // 2
  a, b = draws[:, :half], draws[:, half:2 * half]
  return np.stack([a.mean(1), b.mean(1), a.var(1, ddof=1), b.var(1, ddof=1)], axis=1)


def split_rhat_from_moments(moments: np.ndarray, half: int) -> float:
  """moments: [C_total, 4] as produced by _moments; Gelman et al. (2013) split-R-hat."""
  means = np.concatenate([moments[:, 0], moments[:, 1]])
  variances = np.concatenate([moments[:, 2], moments[:, 3]])
  within = variances.mean()
  between = half * means.var(ddof=1)
  return float(np.sqrt(((half - 1) / half * within + between / half) / within))


def fit_sharded(local_fit: Callable[[int, int], Dict[str, np.ndarray]], num_chains: int,
                gather_keys: Sequence[str] = ("posterior_trajectories", "posterior_means"),
                rhat_keys: Sequence[str] = ("observation_noise_scale", "level_scale"),
                group=None, device: Optional[str] = None) -> Dict[str, object]:
  """Runs `local_fit(first_chain, count)` on every rank and combines the results.

  local_fit returns arrays with a leading chain axis [count, ...] (e.g. the [0] slice of
  `_native.fit_gibbs` outputs for one series).  Returns on every rank:
    {key: [num_chains, ...] for key in gather_keys, "split_rhat": {key: float}}.
  Works without an initialised process group (world size 1).
  """
  try:
    import torch
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
  except ImportError:   # torch is plumbing only; a single process needs none of it
    live = False
  rank = dist.get_rank(group) if live else 0
  world = dist.get_world_size(group) if live else 1
  first, count = chain_block(num_chains, rank, world)
  local = local_fit(first, count)
  out: Dict[str, object] = {}
  half = None
  if not live:
    for k in gather_keys:
      out[k] = local[k]
    mom = {k: _moments(np.asarray(local[k], np.float64)) for k in rhat_keys}
    half = np.asarray(local[rhat_keys[0]]).shape[1] // 2 if rhat_keys else 0
  else:
    dev = torch.device(device) if device else torch.device("cpu")
    counts = [chain_block(num_chains, r, world)[1] for r in range(world)]
    cmax = max(counts)
    for k in gather_keys:
      a = np.ascontiguousarray(local[k], dtype=np.float32)
      pad = np.zeros((cmax,) + a.shape[1:], np.float32)
      pad[:count] = a
      mine = torch.from_numpy(pad).to(dev)
      parts = [torch.empty_like(mine) for _ in range(world)]
      dist.all_gather(parts, mine, group=group)          # RCCL all-gather of the chain blocks
      out[k] = np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)], axis=0)
    mom = {}
    for k in rhat_keys:
      d = np.asarray(local[k], np.float64)
      half = d.shape[1] // 2
      slot = torch.zeros((num_chains, 4), dtype=torch.float64, device=dev)
      slot[first:first + count] = torch.from_numpy(_moments(d)).to(dev)
      dist.all_reduce(slot, op=dist.ReduceOp.SUM, group=group)   # small all-reduce of moments
      mom[k] = slot.cpu().numpy()
  out["split_rhat"] = {k: split_rhat_from_moments(mom[k], half) for k in rhat_keys}
  return out
