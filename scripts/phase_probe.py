"""Timing experiment: per-wave cycles per frame in the four segments of banded_forward_kernel
(work before barrier B | barrier B | merge | barrier A), debug flag 256."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import ViterbiDecoder, synth
dev = torch.device("cuda:0")
logA_T, log_pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(logA_T, log_pi, dev)
B, T, S = int(os.environ.get("PB", "128")), 30000, 361
E = synth.emissions_peaks(B, T, S, seed=1, device=dev)
st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty(B, device=dev)
extra = int(os.environ.get("PFLAGS", "0"))
os.environ["VIT_DEBUG_FLAGS"] = str(256 | extra)
for _ in range(2):
    dec.decode_into(E, st, ll, algo="banded", phase="forward")
torch.cuda.synchronize()
os.environ["VIT_DEBUG_FLAGS"] = "0"
SD = (S + 5) // 4 * 4
off = ((B * T * SD * 4 + 255) // 256) * 256
base = (dec._ws.data_ptr() + 255) & ~255
ws = dec._ws[base - dec._ws.data_ptr():]
fm = ws[off: off + B * 64 * 4].view(torch.float32).view(B, 64)
names = ["work1", "barB", "merge", "barA"]
for w, role in [(0, "target0"), (3, "target3"), (5, "target5"), (6, "prefix"), (7, "suffix"), (8, "dense")]:
    v = fm[:, 4 * w: 4 * w + 4].mean(dim=0).tolist()
    print(f"{role:8s} " + "  ".join(f"{n}={x:7.1f}" for n, x in zip(names, v)) + f"   total={sum(v):.0f}")
