"""Profiling target for the kernels outside the headline (run under rocprofv3 --kernel-trace --stats): configs[4] with the jdc
band and the Durrieu matrix ([256, 30000, 722] fp16), an unstructured 361-state matrix ([128, 3000, 361]), the wave form at
1024 songs, and the emission builders on the headline batch.  Two launches of each."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbi_spl_amd import ViterbiDecoder, emissions, synth  # noqa: E402

dev = torch.device("cuda:0")


def run(dec, E, algo="auto", n=2):
    B, T, _ = E.shape
    st = torch.empty((B, T), dtype=torch.int32, device=dev)
    ll = torch.empty((B,), dtype=torch.float32, device=dev)
    for _ in range(n):
        dec.decode_into(E, st, ll, algo=algo)
    torch.cuda.synchronize()
    dec._ws = None
    torch.cuda.empty_cache()


def tiled(gen, B, T, S, dtype):
    base = gen(32, T, S, seed=1234, device=dev, dtype=dtype)
    return base.repeat(B // 32, 1, 1).contiguous()


T = 30000
E = tiled(synth.emissions_peaks, 256, T, 722, torch.float16)
la, lp = synth.log_params(synth.tonet_transition(721, 40), synth.floored_prior(722))
run(ViterbiDecoder(la, lp, dev), E)
A = synth.durrieu_transition(721, 20)
run(ViterbiDecoder(np.require(np.log(A).astype(np.float32).T, np.float32, ["C"]), np.log(np.full(722, 1.0 / 722)).astype(np.float32), dev), E)
del E
torch.cuda.empty_cache()
la, lp = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
run(ViterbiDecoder(la, lp, dev), tiled(synth.emissions_peaks, 1024, T, 361, torch.float32))
torch.cuda.empty_cache()
D = synth.dense_random_log_transition(361, seed=3)
run(ViterbiDecoder(D, D[0].copy(), dev), tiled(synth.emissions_dense, 128, 3000, 361, torch.float32), algo="dense")
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.randn((128, T, 360), generator=g, device=dev) * 3.0
for _ in range(2):
    emissions.shaun_log_emissions(x)
y = torch.randn((128, T, 361), generator=g, device=dev) * 3.0
for _ in range(2):
    emissions.softmax_log_emissions(y)
torch.cuda.synchronize()
print("done")
