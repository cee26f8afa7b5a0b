"""Multi-GPU decoding: independent songs shard across ranks, one gather of the decoded paths.

The reference has no distributed code at all (every script asserts a single GPU, e.g.
tonet/for_paper.py:144-149); songs are independent, so the only exchange is the final gather of
``states [B/G, T] int32`` + ``loglik [B/G]`` to rank 0.  One process per GPU,
``torch.distributed`` (backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for the tests).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_songs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: the first ``n_songs % world`` ranks get one extra song."""
    if world < 1 or not (0 <= rank < world) or n_songs < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_songs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_paths(states_local: torch.Tensor, loglik_local: torch.Tensor, n_songs: int, dst: int = 0,
                 group=None) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Gather per-rank ``states [b_r, T]`` / ``loglik [b_r]`` (block partition of ``n_songs``) to ``dst``.

    Ragged shards are padded to the largest shard so that a single fixed-size gather carries
    everything (15.4 MB per GPU at 128 x 30000 int32: one message per peer link)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    T = states_local.shape[1]
    counts = [shard_bounds(n_songs, r, world)[1] - shard_bounds(n_songs, r, world)[0] for r in range(world)]
    if states_local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {states_local.shape[0]} songs, partition says {counts[rank]}")
    bmax = max(counts) if counts else 0
    pad_s = states_local.new_full((bmax, T), -1)
    pad_l = loglik_local.new_zeros((bmax,))
    pad_s[: counts[rank]] = states_local
    pad_l[: counts[rank]] = loglik_local
    if rank == dst:
        buf_s = [torch.empty_like(pad_s) for _ in range(world)]
        buf_l = [torch.empty_like(pad_l) for _ in range(world)]
    else:
        buf_s = buf_l = None
    dist.gather(pad_s, buf_s, dst=dst, group=group)
    dist.gather(pad_l, buf_l, dst=dst, group=group)
    if rank != dst:
        return None, None
    states = torch.cat([buf_s[r][: counts[r]] for r in range(world)], dim=0)
    loglik = torch.cat([buf_l[r][: counts[r]] for r in range(world)], dim=0)
    return states, loglik


def gather_paths_async(states_local: torch.Tensor, loglik_local: torch.Tensor, out_states: Optional[torch.Tensor],
                       out_loglik: Optional[torch.Tensor], dst: int = 0, group=None):
    """Non-blocking gather for equal shards: rank ``dst`` passes ``out_states [world, b, T]`` / ``out_loglik [world, b]``
    (reused from call to call; the shards land in place, no concatenation), the others pass None.  The collective is
    ordered after the work already enqueued on the CURRENT stream and runs on the communicator's own stream, so the
    caller can keep launching kernels (the next batch's forward pass) while the paths travel.  Returns the work
    handles; ``wait()`` them (or synchronise the device) before reading the outputs or overwriting the inputs."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        if out_states is None or out_loglik is None or out_states.shape[0] != world or out_loglik.shape[0] != world:
            raise ValueError("destination rank needs out_states [world, b, T] and out_loglik [world, b]")
        if tuple(out_states.shape[1:]) != tuple(states_local.shape) or tuple(out_loglik.shape[1:]) != tuple(loglik_local.shape):
            raise ValueError("gather_paths_async needs equal shards")
        list_s = [out_states[r] for r in range(world)]
        list_l = [out_loglik[r] for r in range(world)]
    else:
        list_s = list_l = None
    w1 = dist.gather(states_local, list_s, dst=dst, group=group, async_op=True)
    w2 = dist.gather(loglik_local, list_l, dst=dst, group=group, async_op=True)
    return w1, w2


def decode_sharded(decode_fn: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
                   emissions_local: torch.Tensor, n_songs: int, dst: int = 0, group=None):
    """Decode this rank's block of songs with ``decode_fn`` (e.g. ``ViterbiDecoder.decode``) and gather
    the paths on ``dst``.  ``emissions_local`` is this rank's ``[b_r, T, S]`` block."""
    states, loglik = decode_fn(emissions_local)
    return gather_paths(states, loglik, n_songs, dst=dst, group=group)
