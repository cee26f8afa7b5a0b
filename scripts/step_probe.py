"""Step-structured (Durrieu) forward kernel: time per frame by batch size (timing experiment: is one workgroup per CU latency-bound?)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
S, T = 722, 3000
A = synth.durrieu_transition(S - 1, 20)
dec = ViterbiDecoder(np.require(np.log(A).astype(np.float32).T, np.float32, ["C"]), np.log(np.full(S, 1.0 / S)).astype(np.float32), dev)
base = synth.emissions_peaks(32, T, S, seed=1, device=dev, dtype=torch.float16)
for B in (64, 128, 256, 512, 1024):
    E = base.repeat(B // 32, 1, 1).contiguous()
    st = torch.empty((B, T), dtype=torch.int32, device=dev)
    ll = torch.empty((B,), dtype=torch.float32, device=dev)
    for form in (0, 3, 1):
        dec.set_option("step_form", form)
        dec.decode_into(E, st, ll, phase="forward")
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        dec.decode_into(E, st, ll, phase="forward")
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])
        print(f"B {B} step_form {form}: {ms:.2f} ms -> {B*T/ms/1e3:.1f} Mframes/s; {ms*1e-3/T*2.4e9:.0f} cycles per frame", flush=True)

# phase stamps (hooks build only: make -C viterbi_spl_amd/csrc TIMING=1)
try:
    dec.set_option("step_form", 0)
    dec.set_option("timing", 256)
except Exception as e:  # release build
    print("no timing hooks in this build:", e)
    sys.exit(0)
B = 256
E = base.repeat(B // 32, 1, 1).contiguous()
st = torch.empty((B, T), dtype=torch.int32, device=dev)
ll = torch.empty((B,), dtype=torch.float32, device=dev)
dec.decode_into(E, st, ll, phase="forward")
torch.cuda.synchronize()
ws = dec._ws
a256 = lambda x: (x + 255) // 256 * 256
off = dec.workspace_bytes(B, T) - a256(B * 32 * 4) - a256(B * 4) - a256(B * 64 * 4)
sc = ws[off:off + B * 64 * 4].view(torch.float32).view(B, 64).cpu().numpy()
for w in range(7):
    ph = sc[:, 4 * w:4 * w + 4]
    print(f"wave {w}: publish {ph[:,0].mean():.0f}  barrier {ph[:,1].mean():.0f}  consume {ph[:,2].mean():.0f}  store+prefetch {ph[:,3].mean():.0f}  (cycles, mean over songs)")
