"""World-size-2 gloo rehearsal of the multi-GPU path: block partition, local decode, gather.
The local decode is the CPU oracle here (test infrastructure); on the GPU box the same
``decode_sharded`` drives ``ViterbiDecoder.decode`` over RCCL (bench.py --gpus N)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import viterbi_oracle as vo
from viterbi_spl_amd import sharded, synth

N_SONGS, T, S = 5, 40, 97


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = synth.dense_random_log_transition(S, seed=5)
        pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
        lo, hi = sharded.shard_bounds(N_SONGS, rank, world)
        E = synth.emissions_dense(hi - lo, T, S, seed=3, first_song=lo)

        def decode_fn(e):
            st, ll = vo.decode_c(A, pi, e.numpy())
            return torch.from_numpy(st), torch.from_numpy(ll)

        states, loglik = sharded.decode_sharded(decode_fn, E, N_SONGS, dst=0)
        if rank == 0:
            q.put((states.numpy(), loglik.numpy()))
        else:
            assert states is None and loglik is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    states, loglik = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    A = synth.dense_random_log_transition(S, seed=5)
    pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
    E = synth.emissions_dense(N_SONGS, T, S, seed=3).numpy()
    ref, rl = vo.decode_c(A, pi, E)
    assert states.shape == (N_SONGS, T)
    assert np.array_equal(states, ref) and np.array_equal(loglik, rl)


def _worker_async(rank, world, port, q):
    """The bench's pipelined form: equal shards, two batches in flight, shards land in place on rank 0."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = synth.dense_random_log_transition(S, seed=5)
        pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
        b = 2
        outs = [(torch.empty((world, b, T), dtype=torch.int32), torch.empty((world, b), dtype=torch.float32)) if rank == 0 else (None, None)
                for _ in range(2)]
        pending, keep = [], []
        for step in range(2):
            E = synth.emissions_dense(b, T, S, seed=10 + step, first_song=rank * b)
            st, ll = vo.decode_c(A, pi, E.numpy())
            st_t, ll_t = torch.from_numpy(st.astype(np.int32)), torch.from_numpy(ll)
            keep.append((st_t, ll_t))
            pending.append(sharded.gather_paths_async(st_t, ll_t, outs[step][0], outs[step][1], dst=0))
        for w1, w2 in pending:
            w1.wait()
            w2.wait()
        if rank == 0:
            q.put([(o[0].numpy().copy(), o[1].numpy().copy()) for o in outs])
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_async_gather_in_place():
    world, b = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    A = synth.dense_random_log_transition(S, seed=5)
    pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
    for step, (st, ll) in enumerate(got):
        E = synth.emissions_dense(world * b, T, S, seed=10 + step).numpy()
        ref, rl = vo.decode_c(A, pi, E)
        assert np.array_equal(st.reshape(world * b, T), ref) and np.array_equal(ll.reshape(-1), rl)


# ---------------------------------------------------------------- length-aware sharding (ragged recordings), four ranks
RAGGED_LENGTHS = [40, 7, 33, 1, 40, 12, 25, 3, 18, 39, 2]          # 11 songs: uneven shards over four ranks


def test_shard_by_length_balances_frames():
    shards = sharded.shard_by_length(RAGGED_LENGTHS, 4)
    flat = np.sort(np.concatenate(shards))
    assert np.array_equal(flat, np.arange(len(RAGGED_LENGTHS)))              # a partition
    loads = [int(np.asarray(RAGGED_LENGTHS)[s].sum()) for s in shards]
    block = [sum(RAGGED_LENGTHS[slice(*sharded.shard_bounds(len(RAGGED_LENGTHS), r, 4))]) for r in range(4)]
    assert max(loads) <= max(block) and max(loads) - min(loads) <= max(RAGGED_LENGTHS)
    assert max(loads) <= -(-sum(RAGGED_LENGTHS) // 4) + max(RAGGED_LENGTHS)   # the LPT bound
    again = sharded.shard_by_length(np.asarray(RAGGED_LENGTHS), 4)           # deterministic: every rank computes the same
    assert all(np.array_equal(a, b) for a, b in zip(shards, again))
    assert [len(s) for s in sharded.shard_by_length([5, 5], 4)] == [1, 1, 0, 0]   # more ranks than songs: empty shards
    assert sharded.shard_by_length([], 3)[0].size == 0


def _worker_ragged(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = synth.dense_random_log_transition(S, seed=5)
        pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
        shards = sharded.shard_by_length(RAGGED_LENGTHS, world)
        mine = shards[rank]
        # every song is generated by its GLOBAL index, so the decoded batch does not depend on who holds what
        E = torch.cat([synth.emissions_dense(1, T, S, seed=3, first_song=int(j)) for j in mine], dim=0) if len(mine) else torch.empty((0, T, S))
        lens = np.asarray(RAGGED_LENGTHS, dtype=np.int64)[mine]
        if len(mine):
            st, ll = vo.decode_c(A, pi, E.numpy(), lengths=lens)
        else:
            st, ll = np.zeros((0, T), np.int32), np.zeros((0,), np.float32)
        states, loglik = sharded.gather_paths_indexed(torch.from_numpy(st.astype(np.int32)), torch.from_numpy(ll), shards, dst=0)
        if rank == 0:
            q.put((states.numpy(), loglik.numpy()))
        else:
            assert states is None and loglik is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_four_rank_length_aware_shards_and_indexed_gather():
    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    states, loglik = q.get(timeout=180)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    A = synth.dense_random_log_transition(S, seed=5)
    pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
    n = len(RAGGED_LENGTHS)
    E = synth.emissions_dense(n, T, S, seed=3).numpy()
    ref, rl = vo.decode_c(A, pi, E, lengths=np.asarray(RAGGED_LENGTHS, dtype=np.int64))
    assert states.shape == (n, T)
    assert np.array_equal(states, ref) and np.array_equal(loglik, rl)          # original song order, -1 past each song's end


# ---------------------------------------------------------------- bench.py's pipelined step loop, eight ranks
PIPE_STEPS = 5


def _worker_pipeline(rank, world, port, q):
    """sharded.GatherPipeline exactly as bench.py drives it: two slots, the gather of step i in flight while step i + 1
    decodes into the other slot, slot reuse guarded by acquire()."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        A = synth.dense_random_log_transition(S, seed=5)
        pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
        b = 2
        pipe = sharded.GatherPipeline(b, T, torch.device("cpu"), n_slots=2)
        st_k = [torch.empty((b, T), dtype=torch.int32) for _ in range(2)]
        ll_k = [torch.empty((b,), dtype=torch.float32) for _ in range(2)]
        seen = []
        for i in range(PIPE_STEPS):
            k = pipe.acquire(i)
            if rank == 0 and i >= 2:                       # the batch gathered through this slot two steps ago is complete here
                o = pipe.result(k)
                seen.append((i - 2, o[0].numpy().copy(), o[1].numpy().copy()))
            E = synth.emissions_dense(b, T, S, seed=20 + i, first_song=rank * b)
            st, ll = vo.decode_c(A, pi, E.numpy())
            st_k[k].copy_(torch.from_numpy(st.astype(np.int32)))
            ll_k[k].copy_(torch.from_numpy(ll))
            pipe.submit(k, st_k[k], ll_k[k])
        pipe.drain()
        assert pipe.launched == PIPE_STEPS and all(p is None for p in pipe.pending)
        if rank == 0:
            for i in range(max(0, PIPE_STEPS - 2), PIPE_STEPS):
                o = pipe.result(i % 2)
                seen.append((i, o[0].numpy().copy(), o[1].numpy().copy()))
            q.put(seen)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_eight_rank_pipelined_gather_rehearsal():
    world, b = 8, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen = q.get(timeout=300)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    A = synth.dense_random_log_transition(S, seed=5)
    pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
    assert sorted(i for i, _, _ in seen) == list(range(PIPE_STEPS))
    for i, st, ll in seen:
        E = synth.emissions_dense(world * b, T, S, seed=20 + i).numpy()
        ref, rl = vo.decode_c(A, pi, E)
        assert np.array_equal(st.reshape(world * b, T), ref) and np.array_equal(ll.reshape(-1), rl), f"step {i}"
