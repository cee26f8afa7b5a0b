"""Minimal profiling target: N decodes of a [B, T, S=361] tonet batch with a chosen algo (run under rocprofv3).
usage: prof_target.py B algo [steps] [T] [option=value ...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

B = int(sys.argv[1])
algo = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
T = int(sys.argv[4]) if len(sys.argv) > 4 else 30000
dev = torch.device("cuda:0")
A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    dec.set_option(k, int(v))
base = synth.emissions_peaks(min(B, 32), T, 361, seed=1234, device=dev)      # songs repeat with period 32 (generation time)
E = base if B <= 32 else base.repeat((B + 31) // 32, 1, 1)[:B].contiguous()
st = torch.empty((B, T), dtype=torch.int32, device=dev)
ll = torch.empty((B,), dtype=torch.float32, device=dev)
for _ in range(steps):
    dec.decode_into(E, st, ll, algo=algo)
torch.cuda.synchronize()
print("done", B, algo, steps, T)
