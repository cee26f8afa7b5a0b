"""Ablations of the wave forward kernel (results WRONG -- timing only): what does the kernel cost without its history stores,
without its emission loads, without both?  Needs the hooks build (`make -C viterbi_spl_amd/csrc TIMING=1`), which this script
loads instead of the release library.  argv: batch sizes (default 1024 2048)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbi_spl_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libviterbi_hip_timing.so")
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
T = 30000
base = synth.emissions_peaks(32, T, 361, seed=1234, device=dev)
for B in [int(x) for x in sys.argv[1:]] or [1024, 2048]:
    E = base.repeat((B + 31) // 32, 1, 1)[:B].contiguous()
    st = torch.empty((B, T), dtype=torch.int32, device=dev)
    ll = torch.empty((B,), dtype=torch.float32, device=dev)
    for label, opts in (("full history", {"wave_history": 1}), ("half history", {"wave_history": 2}),
                        ("no stores", {"wave_history": 1, "timing": 1}), ("no loads", {"wave_history": 1, "timing": 2}),
                        ("no loads, no stores", {"wave_history": 1, "timing": 3}),
                        ("full, two waves per SIMD", {"wave_history": 1, "wave_two": 1}), ("half, two waves per SIMD", {"wave_history": 2, "wave_two": 1}),
                        ("no stores, two waves", {"wave_history": 1, "timing": 1, "wave_two": 1}), ("neither, two waves", {"wave_history": 1, "timing": 3, "wave_two": 1})):
        dec.set_option("reset", 0)
        for k, v in opts.items():
            dec.set_option(k, v)
        dec.decode_into(E, st, ll, algo="wave", phase="forward")
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 5
        ev[0].record()
        for _ in range(n):
            dec.decode_into(E, st, ll, algo="wave", phase="forward")
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / n
        print(f"B {B} {label:28s} forward {ms:7.2f} ms  {B*T/ms/1e3:6.0f} Mframes/s  {B*T*2166/(ms*1e-3)/8e12:.3f} of the roofline", flush=True)
    del E, st, ll
    dec._ws = None
    torch.cuda.empty_cache()
