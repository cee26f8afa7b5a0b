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
