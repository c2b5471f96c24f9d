"""
Sharding of independent lattices (chains) over ranks -- one process per GPU, no data-path collective.

The Fourier-method GP path evaluates independent lattices: the reference runs one CPU process per chain
(cmdstan `chains=`); here a batch of chains is split over the ranks of a `torch.distributed` job and every
rank evaluates its slice with one batched launch sequence. Nothing is exchanged while evaluating. The only
collectives are bookkeeping around it: collecting the per-chain scalars (`gather_chain_values`) and the
max-over-ranks clock of `bench.py` (`max_over_ranks`). Works on any backend (`nccl` = RCCL on the GPUs,
`gloo` in the CPU tests).
"""
from __future__ import annotations

import torch


def shard_bounds(total: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous slice [lo, hi) of `total` chains owned by `rank`; sizes differ by at most one, the first
    `total % world` ranks hold the longer slices, ranks beyond `total` hold an empty slice."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"invalid rank {rank} for world size {world}")
    if total < 0:
        raise ValueError(f"negative number of chains {total}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(total: int, world: int) -> list[int]:
    return [hi - lo for lo, hi in (shard_bounds(total, world, r) for r in range(world))]


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def world_and_rank() -> tuple[int, int]:
    dist = _dist()
    return (dist.get_world_size(), dist.get_rank()) if dist else (1, 0)


def local_shard(x: torch.Tensor, total: int | None = None) -> torch.Tensor:
    """Slice of the leading (chain) axis this rank owns (a view, no copy)."""
    world, rank = world_and_rank()
    lo, hi = shard_bounds(x.shape[0] if total is None else total, world, rank)
    return x[lo:hi]


def gather_chain_values(local: torch.Tensor, total: int) -> torch.Tensor:
    """All ranks receive the per-chain values (leading axis = chain) of all `total` chains in chain order.
    `local` is this rank's slice as laid out by `shard_bounds` (ragged slices are padded for the exchange)."""
    dist = _dist()
    if dist is None:
        if local.shape[0] != total:
            raise ValueError(f"single process holds {local.shape[0]} of {total} chains")
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = shard_sizes(total, world)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} chains, its shard has {sizes[rank]}")
    width = max(sizes) if sizes else 0
    padded = local.new_zeros((width,) + tuple(local.shape[1:]))
    padded[: local.shape[0]] = local
    pieces = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(pieces, padded)
    return torch.cat([p[:s] for p, s in zip(pieces, sizes)], dim=0)


def max_over_ranks(seconds: float, device=None) -> float:
    """The job's clock: the slowest rank's elapsed time (bench.py contract)."""
    dist = _dist()
    if dist is None:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sharded_eval(fn, *chain_tensors: torch.Tensor, shared=()):
    """Evaluate `fn(*local_slices, *shared)` on this rank's chains and gather its per-chain FIRST output.

    `fn` returns a tensor or a tuple whose first element has the chain axis leading (e.g. the lpdf of
    `gp_rfft2_lpdf_vjp`); the other outputs (gradients) stay local to the rank that owns the chains, as
    they do for the reference's one-process-per-chain runs. Returns (all_values, local_outputs)."""
    total = chain_tensors[0].shape[0]
    for t in chain_tensors:
        if t.shape[0] != total:
            raise ValueError("chain tensors disagree on the number of chains")
    outs = fn(*[local_shard(t) for t in chain_tensors], *shared)
    first = outs[0] if isinstance(outs, (tuple, list)) else outs
    return gather_chain_values(first, total), outs
