"""Profiling target of round 4 (run under rocprofv3): argv = which parts to run, any of
   headline  B = 128, 10 decodes (workgroup kernels)
   wave      B = 1024 and 2048, full history, 10 decodes each (the wave kernel + the sparse back-trace)
   lane      B = 2048, 5 back-traces with one stream per lane (backtrace_form 4) behind one forward pass
   packed    3250 ragged songs, 61.44 M frames, 5 packed decodes
   configs4  [256, 30000, 722] fp16: jdc band and Durrieu, 10 decodes each
   obs       the emission builders on [128, 30000, 360 / 361] logits, 10 launches each
   b1024     B = 1024 full history, 2 decodes (the PMC traffic passes)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, emissions, synth  # noqa: E402

dev = torch.device("cuda:0")
parts = set(sys.argv[1:]) or {"headline", "wave", "lane", "packed", "configs4", "obs"}
T = 30000


def run(dec, E, algo="auto", n=10, phase_bt_only=0):
    B, T_, _ = E.shape
    st = torch.empty((B, T_), dtype=torch.int32, device=dev)
    ll = torch.empty((B,), dtype=torch.float32, device=dev)
    for _ in range(n):
        dec.decode_into(E, st, ll, algo=algo)
    for _ in range(phase_bt_only):
        dec.decode_into(E, st, ll, algo=algo, phase="backtrace")
    torch.cuda.synchronize()
    dec._ws = None
    torch.cuda.empty_cache()


def tiled(gen, B, S, dtype=torch.float32):
    base = gen(min(B, 32), T, S, seed=1234, device=dev, dtype=dtype)
    return base if B <= 32 else base.repeat(B // 32, 1, 1).contiguous()


A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
if "headline" in parts:
    run(dec, tiled(synth.emissions_peaks, 128, 361), "banded")
if "wave" in parts:
    for B in (1024, 2048):
        run(dec, tiled(synth.emissions_peaks, B, 361), "banded")
if "b1024" in parts:
    run(dec, tiled(synth.emissions_peaks, 1024, 361), "banded", n=2)
if "lane" in parts:
    dec.set_option("backtrace_form", 4)
    run(dec, tiled(synth.emissions_peaks, 2048, 361), "banded", n=1, phase_bt_only=4)
    dec.set_option("reset", 0)
if "packed" in parts:
    r = bench.packed_row(dec, A, pi, T, 2048 * T, dev, 5)
    print("packed", round(r["ms_per_step"], 3), "ms per step", flush=True)
if "configs4" in parts:
    E = tiled(synth.emissions_peaks, 256, 722, torch.float16)
    la, lp = synth.log_params(synth.tonet_transition(721, 40), synth.floored_prior(722))
    run(ViterbiDecoder(la, lp, dev), E)
    D = synth.durrieu_transition(721, 20)
    run(ViterbiDecoder(np.require(np.log(D).astype(np.float32).T, np.float32, ["C"]), np.log(np.full(722, 1.0 / 722)).astype(np.float32), dev), E)
    del E
    torch.cuda.empty_cache()
if "obs" in parts:
    x = synth.pitch_logits(32, T, 360, seed=5, device=dev).repeat(4, 1, 1).contiguous()
    out = torch.empty((128, T, 361), dtype=torch.float32, device=dev)
    for _ in range(10):
        emissions.shaun_log_emissions(x, out=out)
    y = torch.cat([torch.zeros((128, T, 1), device=dev), x], dim=2).contiguous()
    for _ in range(10):
        emissions.softmax_log_emissions(y, out=out)
    torch.cuda.synchronize()
print("done", sorted(parts))
