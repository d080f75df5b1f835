"""Chain sharding over the GPUs of one node (one process per GPU).

The reference runs one chain in one process; chains are
independent given the data, so ranks take contiguous blocks of chain ids with NO collective
on the data path.  Chain c always uses RNG stream c (`chain_offset`), hence the pooled
result is identical for any world size.  Collectives run only after sampling -- through
`_comm.Comm` (librccl or a shared-memory host transport behind the C-ABI; no PyTorch) or any object
with its `rank` / `world` / `all_gather` / `all_reduce` interface (the CPU tests wrap a gloo
process group that way, tests/_torch_comm.py -- that adapter is test code, not part of this package):
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
                comm=None,
                resident: Optional[Callable[[str], Optional[np.ndarray]]] = None
                ) -> Dict[str, object]:
  """Runs `local_fit(first_chain, count)` on every rank and combines the results.

  local_fit returns arrays with a leading chain axis [count, ...] (e.g. the [0] slice of
  `_native.fit_gibbs` outputs for one series).  It is NOT called on a rank whose block is empty
  (more ranks than chains): that rank contributes zero-length blocks to the collectives.
  `comm`: a `_comm.Comm` (RCCL over xGMI or the host transport, through the C-ABI -- the product
  path, no PyTorch) or an object with the same interface; without one the call is a single process.
  `resident(key)` (optional, equal blocks only): returns the gathered [world, count, ...] array
  of `key` straight from the ranks' device-resident sessions (`Comm.session_all_gather`), or
  None to fall back to gathering the host copy in `local_fit`'s result.
  Returns on every rank:
    {key: [num_chains, ...] for key in gather_keys,
     "split_rhat" / "ess_bulk" / "ess_tail": {key: float for key in rhat_keys}}.
  """
  rank = comm.rank if comm is not None else 0
  world = comm.world if comm is not None else 1
  if num_chains < 1:
    raise ValueError(f"num_chains must be >= 1, got {num_chains}")
  first, count = chain_block(num_chains, rank, world)
  local = local_fit(first, count) if count > 0 else None
  out: Dict[str, object] = {}

  if comm is None:
    for k in gather_keys:
      out[k] = local[k]
    scal = {k: np.asarray(local[k], np.float64) for k in rhat_keys}
    sums = {k: _local_sums(scal[k], scal[k]) for k in rhat_keys}
  else:
    counts = [chain_block(num_chains, r, world)[1] for r in range(world)]
    cmax = max(counts)
    even = min(counts) == cmax

    def gather(key: str, tail_shape, dtype) -> np.ndarray:
      """all-gather of [count, *tail] blocks (padded to the largest block)."""
      if even and resident is not None:
        got = resident(key)
        if got is not None:
          return np.asarray(got, dtype).reshape((num_chains,) + tuple(tail_shape))
      pad = np.zeros((cmax,) + tuple(tail_shape), dtype)
      if count > 0:
        pad[:count] = np.asarray(local[key], dtype)
      parts = comm.all_gather(pad)
      return np.concatenate([parts[r][:c] for r, c in enumerate(counts)], axis=0)

    # tail shapes must be known on ranks with an empty block too: agree on them first
    shapes = _agree_on_shapes(local, list(gather_keys) + list(rhat_keys), comm)
    for k in gather_keys:
      out[k] = gather(k, shapes[k], np.float32)
    sums = {}
    for k in rhat_keys:
      pooled = gather(k, shapes[k], np.float64)                  # [num_chains, S] scalars
      mine = pooled[first:first + count]
      packed = np.stack([dg.pack(p) for p in _local_sums(mine, pooled)])       # [4, n + 3]
      red = comm.all_reduce(packed).reshape(packed.shape)         # small all-reduce of the sums
      sums[k] = [dg.unpack(v) for v in red]
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


def _agree_on_shapes(local, keys, comm):
  """Per-key trailing shapes, taken from any rank that ran a fit (max-reduced; ranks with an
  empty block contribute zeros)."""
  max_nd = 4
  t = np.zeros((len(keys), max_nd + 1), np.float64)
  if local is not None:
    for i, k in enumerate(keys):
      shp = np.asarray(local[k]).shape[1:]
      if len(shp) > max_nd:
        raise ValueError(f"{k}: at most {max_nd} trailing dimensions are supported")
      t[i, 0] = len(shp)
      for j, v in enumerate(shp):
        t[i, 1 + j] = int(v)
  t = comm.all_reduce(t, 1).reshape(t.shape)       # MAX (exact: small integers in float64)
  return {k: tuple(int(v) for v in t[i, 1:1 + int(t[i, 0])]) for i, k in enumerate(keys)}
