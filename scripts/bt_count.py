"""Timing experiment: how often does the back-trace take the full-evaluation step? (debug flag 32)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import ViterbiDecoder, synth
dev = torch.device("cuda:0")
logA_T, log_pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(logA_T, log_pi, dev)
for kind, gen in (("peaks", synth.emissions_peaks), ("dense", synth.emissions_dense)):
    B, T = 32, 30000
    E = gen(B, T, 361, seed=1, device=dev)
    st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty(B, device=dev)
    dec.decode_into(E, st, ll, algo="banded")
    torch.cuda.synchronize()
    nbytes = dec.workspace_bytes(B, T)
    base = (dec._ws.data_ptr() + 255) & ~255
    cnt = dec._ws[base - dec._ws.data_ptr() + nbytes - 256: base - dec._ws.data_ptr() + nbytes - 240].view(torch.int64)
    cnt.zero_()
    os.environ["VIT_DEBUG_FLAGS"] = "32"
    dec.decode_into(E, st, ll, algo="banded", phase="backtrace")
    torch.cuda.synchronize()
    os.environ["VIT_DEBUG_FLAGS"] = "0"
    c = cnt.cpu().tolist()
    print(kind, "frames", c[0], "full steps", c[1], "fraction", c[1] / max(c[0], 1))
