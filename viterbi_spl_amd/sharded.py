"""Multi-GPU decoding: independent songs shard across ranks, one gather of the decoded paths.

The reference has no distributed code at all (every script asserts a single GPU, e.g.
tonet/for_paper.py:144-149); songs are independent, so the only exchange is the final gather of
``states [B/G, T] int32`` + ``loglik [B/G]`` to rank 0.  One process per GPU,
``torch.distributed`` (backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU for the tests).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_songs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition: the first ``n_songs % world`` ranks get one extra song."""
    if world < 1 or not (0 <= rank < world) or n_songs < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_songs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_length(lengths: Sequence[int], world: int) -> List[np.ndarray]:
    """Length-aware partition for ragged batches (recordings are decoded whole: tonet/for_paper.py:2304-2305 in the
    reference, T = the song's length): longest-first greedy bin packing of the songs over ``world`` ranks by total
    frames -- a rank's decode time is proportional to the frames it holds, so the slowest rank of a block partition
    of sorted-by-nothing lengths sets the step time.  Deterministic (ties broken by song index, then by rank), so every
    rank computes the same assignment from the same ``lengths`` and no exchange is needed.  Returns, per rank, the
    ascending song indices it decodes; :func:`gather_paths_indexed` puts the paths back in the original order."""
    if world < 1:
        raise ValueError("bad shard request")
    n = np.asarray(lengths, dtype=np.int64)
    if n.ndim != 1 or (n < 0).any():
        raise ValueError("lengths must be a 1-D array of non-negative frame counts")
    order = np.lexsort((np.arange(len(n)), -n))            # longest first, index breaks ties
    load = np.zeros(world, dtype=np.int64)
    count = np.zeros(world, dtype=np.int64)
    bins: List[List[int]] = [[] for _ in range(world)]
    for j in order:
        r = int(np.lexsort((np.arange(world), count, load))[0])     # least frames, then fewest songs, then lowest rank
        bins[r].append(int(j))
        load[r] += n[j]
        count[r] += 1
    return [np.asarray(sorted(b), dtype=np.int64) for b in bins]


def gather_paths_indexed(states_local: torch.Tensor, loglik_local: torch.Tensor, shards: Sequence[np.ndarray], dst: int = 0,
                         group=None) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """Gather shards of DIFFERENT sizes whose songs are scattered over the batch (``shards[r]`` = the song indices rank r
    holds, in the order of its local tensors) and return, on ``dst``, ``states [n, T]`` / ``loglik [n]`` in the original
    song order.  One fixed-size gather per tensor: shards are padded to the largest one."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if len(shards) != world:
        raise ValueError("one index list per rank")
    counts = [len(x) for x in shards]
    if states_local.shape[0] != counts[rank] or loglik_local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {states_local.shape[0]} songs, the assignment says {counts[rank]}")
    n = sum(counts)
    flat = np.concatenate([np.asarray(x, dtype=np.int64) for x in shards]) if n else np.zeros(0, np.int64)
    if n and not np.array_equal(np.sort(flat), np.arange(n)):
        raise ValueError("shards must partition range(n_songs)")
    T = states_local.shape[1]
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
    states = states_local.new_empty((n, T))
    loglik = loglik_local.new_empty((n,))
    for r in range(world):
        idx = torch.as_tensor(np.asarray(shards[r], dtype=np.int64), device=states.device)
        states[idx] = buf_s[r][: counts[r]]
        loglik[idx] = buf_l[r][: counts[r]]
    return states, loglik


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


class GatherPipeline:
    """The step loop's gather side, as bench.py runs it: ``n_slots`` batches in flight, each with its own gather buffers on
    ``dst``; the gather of the batch in slot k is launched non-blocking behind the work already enqueued on the current
    stream and must have completed before slot k is written again.  Backend-agnostic (RCCL on GPUs, gloo in the CPU
    rehearsal tests/test_sharded_gloo.py runs with eight ranks)::

        pipe = GatherPipeline(b, T, device, n_slots=2)
        for i in range(steps):
            k = pipe.acquire(i)             # waits for the gather that last used slot k
            ... decode batch i into states[k], loglik[k] ...
            pipe.submit(k, states[k], loglik[k])
        pipe.drain()
    """

    def __init__(self, b: int, T: int, device, n_slots: int = 2, dst: int = 0, group=None):
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dst, self.group, self.n_slots = dst, group, n_slots
        self.pending: List[Optional[tuple]] = [None] * n_slots
        self.launched = 0
        if self.rank == dst:
            self.out = [(torch.empty((self.world, b, T), dtype=torch.int32, device=device),
                         torch.empty((self.world, b), dtype=torch.float32, device=device)) for _ in range(n_slots)]
        else:
            self.out = [(None, None)] * n_slots

    def acquire(self, i: int) -> int:
        k = i % self.n_slots
        self.wait(k)
        return k

    def wait(self, k: int) -> None:
        if self.pending[k] is not None:
            for w in self.pending[k]:
                w.wait()
            self.pending[k] = None

    def submit(self, k: int, states_local: torch.Tensor, loglik_local: torch.Tensor) -> None:
        if self.pending[k] is not None:
            raise RuntimeError(f"slot {k} still has a gather in flight: acquire() it first")
        self.pending[k] = gather_paths_async(states_local, loglik_local, self.out[k][0], self.out[k][1], dst=self.dst, group=self.group)
        self.launched += 1

    def drain(self) -> None:
        for k in range(self.n_slots):
            self.wait(k)

    def result(self, k: int):
        """(states [world, b, T], loglik [world, b]) of the batch last gathered through slot k (rank ``dst`` only)."""
        return self.out[k]

