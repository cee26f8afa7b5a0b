"""Dense forward kernel: time per frame by songs per workgroup (option dense_songs) and batch size (timing experiment)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
S, T = (int(sys.argv[1]) if len(sys.argv) > 1 else 361), 3000
A = synth.dense_random_log_transition(S, seed=3)
pi = synth.dense_random_log_transition(S, seed=4)[0].copy()
dec = ViterbiDecoder(A, pi, dev)
base = synth.emissions_dense(32, T, S, seed=1, device=dev)
for B in (128, 256, 1024):
    E = base.repeat(B // 32, 1, 1).contiguous()
    st = torch.empty((B, T), dtype=torch.int32, device=dev)
    ll = torch.empty((B,), dtype=torch.float32, device=dev)
    for ns, kt1 in ((0, 0), (1, 0), (2, 0)):
        dec.set_option("dense_form", 0 if ns == 0 else 1)     # first entry: the matrix-resident form
        dec.set_option("dense_songs", ns)
        dec.set_option("dense_one_thread", kt1)
        dec.decode_into(E, st, ll, algo="dense", phase="forward")
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        dec.decode_into(E, st, ll, algo="dense", phase="forward")
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1])
        wgs = (B + ns - 1) // ns if ns else B
        print(f"B {B} songs/wg {ns} one_thread {kt1}: {ms:.1f} ms -> {B*T/ms/1e3:.1f} Mframes/s; {ms*1e-3/T*2.4e9:.0f} cycles per frame per workgroup ({wgs} workgroups)", flush=True)
