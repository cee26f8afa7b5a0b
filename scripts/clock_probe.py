"""Timing experiment: shader cycles and wall ticks per frame inside banded_forward_kernel (debug flags 16/32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import ViterbiDecoder, synth
dev = torch.device("cuda:0")
logA_T, log_pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(logA_T, log_pi, dev)
for B in (1, 128, 512):
    E = synth.emissions_peaks(B, 30000, 361, seed=1, device=dev)
    st = torch.empty((B, 30000), dtype=torch.int32, device=dev); ll = torch.empty(B, device=dev)
    out = {}
    for flag, name in ((16, "cycles"), (32, "ticks100MHz")):
        os.environ["VIT_DEBUG_FLAGS"] = str(flag)
        for _ in range(2):
            dec.decode_into(E, st, ll, algo="banded", phase="forward")
        torch.cuda.synchronize()
        out[name] = ll.float().mean().item()
    os.environ["VIT_DEBUG_FLAGS"] = "0"
    ghz = out["cycles"] / out["ticks100MHz"] * 0.1
    print(f"B={B}: {out['cycles']:.0f} cycles/frame, {out['ticks100MHz']*10:.0f} ns/frame, clock {ghz:.2f} GHz")
