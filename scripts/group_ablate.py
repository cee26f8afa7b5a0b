"""Ablations of the one-song-per-workgroup forward kernel (results WRONG -- timing only): what does the LDS return path of the
window reads cost a frame?  Needs the hooks builds `make -C viterbi_spl_amd/csrc TIMING=1 [ABL=n]` (n = window quads actually read per
chunk of eight, the others reuse them; compile-time, so that the loop is otherwise the release loop).  argv: the build's suffix ("" / 4 / 1),
then batch sizes (default 128 256)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbi_spl_amd import _lib  # noqa: E402

ABL = sys.argv[1] if len(sys.argv) > 1 else ""
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"libviterbi_hip_timing{ABL}.so")
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
T = 30000
for S, dmax, dt in ((361, 14, torch.float32), (722, 40, torch.float16)):
    A, pi = synth.log_params(synth.tonet_transition(S - 1, dmax), synth.floored_prior(S))
    dec = ViterbiDecoder(A, pi, dev)
    base = synth.emissions_peaks(32, T, S, seed=1234, device=dev, dtype=dt)
    for B in [int(x) for x in sys.argv[2:]] or [128, 256]:
        E = base.repeat((B + 31) // 32, 1, 1)[:B].contiguous()
        st = torch.empty((B, T), dtype=torch.int32, device=dev)
        ll = torch.empty((B,), dtype=torch.float32, device=dev)
        for label, opts in ((f"window quads read per chunk: {ABL or 'all'}", {}),) * 2:
            dec.set_option("reset", 0)
            for k, v in opts.items():
                dec.set_option(k, v)
            dec.decode_into(E, st, ll, algo="group", phase="forward")
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            n = 5
            ev[0].record()
            for _ in range(n):
                dec.decode_into(E, st, ll, algo="group", phase="forward")
            ev[1].record()
            torch.cuda.synchronize()
            ms = ev[0].elapsed_time(ev[1]) / n
            print(f"S {S} W {dec.info['group_window']} B {B} {label:28s}: {ms:7.2f} ms  {ms * 1e-3 / (T - 1) * 2.4e9:6.0f} cycles per frame at 2.4 GHz", flush=True)
        del E, st, ll
        dec._ws = None
        torch.cuda.empty_cache()
    del dec, base
