#!/bin/bash
# high-resolution grid (BASELINE configs[4]): S = 722, fp16 emissions, B = 256, wide bands
cd "$(dirname "$0")/.."
for dm in 14 30 40 56; do python - <<PY
import sys, torch
sys.path.insert(0, ".")
from viterbi_spl_amd import ViterbiDecoder, synth
dev = torch.device("cuda:0")
logA_T, log_pi = synth.log_params(synth.tonet_transition(721, $dm), synth.floored_prior(722))
dec = ViterbiDecoder(logA_T, log_pi, dev)
B, T = 256, 30000
E = synth.emissions_peaks(B, T, 722, seed=1, device=dev, dtype=torch.float16)
st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty(B, device=dev)
for _ in range(2):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    evs[0].record(); dec.decode_into(E, st, ll, algo="banded", phase="forward"); evs[1].record(); dec.decode_into(E, st, ll, algo="banded", phase="backtrace"); evs[2].record()
    torch.cuda.synchronize()
f, b = evs[0].elapsed_time(evs[1]), evs[1].elapsed_time(evs[2])
print("S=722 f16 B=256 d_max", $dm, "W", dec.info["group_window"], "fwd_ms", round(f, 2), "bt_ms", round(b, 2), "Mframes/s", round(B * T / (f + b) / 1e3, 1))
PY
done
