"""Stub worker for tests/test_bench_launcher.py: what bench.py's workers do, without a GPU -- gloo process group from torchrun's
environment, the two-slot GatherPipeline step loop (the local decode is the CPU oracle: test infrastructure), ONE JSON line from
rank 0.  `--fail-rank R` makes rank R crash after the rendezvous (the other ranks then wait in a gather until the launcher ends them)."""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import viterbi_oracle as vo  # noqa: E402
from viterbi_spl_amd import sharded, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--fail-rank", type=int, default=-1)
args = ap.parse_args()
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert world == args.gpus, (world, args.gpus)
dist.init_process_group("gloo")
S, T, b = 97, 40, 2
A = synth.dense_random_log_transition(S, seed=5)
pi = synth.dense_random_log_transition(S, seed=6)[0].copy()
pipe = sharded.GatherPipeline(b, T, torch.device("cpu"), n_slots=2)
st_k = [torch.empty((b, T), dtype=torch.int32) for _ in range(2)]
ll_k = [torch.empty((b,), dtype=torch.float32) for _ in range(2)]
print(f"rank {rank} of {world} up", flush=True)
if rank == args.fail_rank:
    os._exit(3)          # a crashed worker (sys.exit would wait in the process group destructor for its peers)
for i in range(args.steps):
    k = pipe.acquire(i)
    E = synth.emissions_dense(b, T, S, seed=20 + i, first_song=rank * b)
    st, ll = vo.decode_c(A, pi, E.numpy())
    st_k[k].copy_(torch.from_numpy(st.astype(np.int32)))
    ll_k[k].copy_(torch.from_numpy(ll))
    pipe.submit(k, st_k[k], ll_k[k])
pipe.drain()
if rank == 0:
    o = pipe.result((args.steps - 1) % 2)
    E = synth.emissions_dense(world * b, T, S, seed=20 + args.steps - 1).numpy()
    ref, rl = vo.decode_c(A, pi, E)
    ok = bool(np.array_equal(o[0].numpy().reshape(world * b, T), ref) and np.array_equal(o[1].numpy().reshape(-1), rl))
    print(json.dumps({"world_size": dist.get_world_size(), "steps": args.steps, "gathers": pipe.launched, "last_step_equals_oracle": ok}), flush=True)
dist.barrier()
dist.destroy_process_group()
