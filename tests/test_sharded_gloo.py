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
