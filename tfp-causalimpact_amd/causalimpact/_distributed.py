"""Chain sharding over the GPUs of one node (one process per GPU, torch.distributed).

The reference runs one chain in one process; chains are
independent given the data, so ranks take contiguous blocks of chain ids with NO collective
on the data path.  Chain c always uses RNG stream c (`chain_offset`), hence the pooled
result is identical for any world size.  RCCL (backend "nccl" on ROCm) / gloo is used only
after sampling:
  * an all-gather of the per-chain blocks needed for pooled summaries (`gather_keys`) and of
    the scalar parameters the diagnostics rank (`rhat_keys`: [chains, draws] floats);
  * ONE all-reduce(sum) of the diagnostics' partial sums -- per-split-chain means and
    autocovariance sums of the rank-normalised draws and of the tail indicators
    (`_diagnostics.partial_sums`) -- from which every rank forms split-R-hat, bulk ESS and
    tail ESS of ALL chains.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import numpy as np

from causalimpact import _diagnostics as dg


def chain_block(num_chains: int, rank: int, world_size: int) -> Tuple[int, int]:
  """(first chain id, count) of `rank`'s contiguous block; blocks differ by at most one."""
  base, extra = divmod(num_chains, world_size)
  count = base + (1 if rank < extra else 0)
  first = rank * base + min(rank, extra)
  return first, count


def fit_sharded(local_fit: Callable[[int, int], Dict[str, np.ndarray]], num_chains: int,
                gather_keys: Sequence[str] = ("posterior_trajectories", "posterior_means"),
                rhat_keys: Sequence[str] = ("observation_noise_scale", "level_scale"),
                group=None, device: Optional[str] = None) -> Dict[str, object]:
  """Runs `local_fit(first_chain, count)` on every rank and combines the results.

  local_fit returns arrays with a leading chain axis [count, ...] (e.g. the [0] slice of
  `_native.fit_gibbs` outputs for one series).  It is NOT called on a rank whose block is empty
  (more ranks than chains): that rank contributes zero-length blocks to the collectives.
  Returns on every rank:
    {key: [num_chains, ...] for key in gather_keys,
     "split_rhat" / "ess_bulk" / "ess_tail": {key: float for key in rhat_keys}}.
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
  if num_chains < 1:
    raise ValueError(f"num_chains must be >= 1, got {num_chains}")
  first, count = chain_block(num_chains, rank, world)
  local = local_fit(first, count) if count > 0 else None
  out: Dict[str, object] = {}

  if not live:
    for k in gather_keys:
      out[k] = local[k]
    scal = {k: np.asarray(local[k], np.float64) for k in rhat_keys}
    sums = {k: _local_sums(scal[k], scal[k]) for k in rhat_keys}
  else:
    dev = torch.device(device) if device else torch.device("cpu")
    counts = [chain_block(num_chains, r, world)[1] for r in range(world)]
    cmax = max(counts)

    def gather(a: Optional[np.ndarray], tail_shape, dtype) -> np.ndarray:
      """all-gather of [count, *tail] blocks (padded to the largest block); RCCL on GPUs."""
      pad = np.zeros((cmax,) + tuple(tail_shape), dtype)
      if count > 0:
        pad[:count] = a
      mine = torch.from_numpy(pad).to(dev)
      parts = [torch.empty_like(mine) for _ in range(world)]
      dist.all_gather(parts, mine, group=group)
      return np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)], axis=0)

    # tail shapes must be known on ranks with an empty block too: agree on them first
    shapes = _agree_on_shapes(local, list(gather_keys) + list(rhat_keys), dist, group, torch, dev)
    for k in gather_keys:
      a = None if local is None else np.ascontiguousarray(local[k], dtype=np.float32)
      out[k] = gather(a, shapes[k], np.float32)
    sums = {}
    for k in rhat_keys:
      a = None if local is None else np.ascontiguousarray(local[k], dtype=np.float64)
      pooled = gather(a, shapes[k], np.float64)                  # [num_chains, S] scalars
      mine = pooled[first:first + count]
      packed = np.stack([dg.pack(p) for p in _local_sums(mine, pooled)])       # [4, n + 3]
      t = torch.from_numpy(packed).to(dev)
      dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)       # small all-reduce of the sums
      sums[k] = [dg.unpack(v) for v in t.cpu().numpy()]
  out["split_rhat"] = {k: dg.rhat_from_sums(sums[k][0]) for k in rhat_keys}
  out["ess_bulk"] = {k: dg.ess_from_sums(sums[k][1]) for k in rhat_keys}
  out["ess_tail"] = {k: float(np.nanmin([dg.ess_from_sums(sums[k][2]), dg.ess_from_sums(sums[k][3])]))
                     for k in rhat_keys}
  return out


def _local_sums(mine: np.ndarray, pooled: np.ndarray):
  """The four additive statistics of this rank's chains: plain (R-hat), rank-normalised (bulk
  ESS), and the two tail indicators."""
  lo, hi = dg.tail_partials(mine, pooled)
  return [dg.partial_sums(dg.split_chains(mine)), dg.bulk_partial(mine, pooled), lo, hi]


def _agree_on_shapes(local, keys, dist, group, torch, dev):
  """Per-key trailing shapes, taken from any rank that ran a fit (max-reduced; ranks with an
  empty block contribute zeros)."""
  max_nd = 4
  t = torch.zeros((len(keys), max_nd + 1), dtype=torch.int64, device=dev)
  if local is not None:
    for i, k in enumerate(keys):
      shp = np.asarray(local[k]).shape[1:]
      if len(shp) > max_nd:
        raise ValueError(f"{k}: at most {max_nd} trailing dimensions are supported")
      t[i, 0] = len(shp)
      for j, v in enumerate(shp):
        t[i, 1 + j] = int(v)
  dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
  t = t.cpu().numpy()
  return {k: tuple(int(v) for v in t[i, 1:1 + int(t[i, 0])]) for i, k in enumerate(keys)}
