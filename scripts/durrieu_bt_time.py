"""Back-trace time of the Durrieu matrix at configs[4] ([256 / 512, 30000, 722] fp16) for several chunk counts / warm-ups of the whole-row
kernel (test infrastructure; GPU box)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from viterbi_spl_amd import ViterbiDecoder, synth
import bench
dev = torch.device("cuda:0")
A, pi = bench.make_params("durrieu", 722, 14)
dec = ViterbiDecoder(A, pi, dev)
T = 30000
for B in (256, 512):
    E = bench.tiled_emissions(synth.emissions_peaks, B, T, 722, 1234, dev, torch.float16)
    st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty((B,), dtype=torch.float32, device=dev)
    dec.set_option("reset", 0)
    dec.decode_into(E, st, ll, algo="auto"); torch.cuda.synchronize()
    ref = st.clone()
    for chunks, warm in ((0, -1), (8, 128), (16, 128), (32, 128), (16, 256), (32, 256), (32, 64)):
        dec.set_option("reset", 0)
        dec.set_option("bt_chunks", chunks)
        dec.set_option("bt_warm", warm)
        dec.decode_into(E, st, ll, algo="auto", phase="backtrace"); torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(3): dec.decode_into(E, st, ll, algo="auto", phase="backtrace")
        ev[1].record(); torch.cuda.synchronize()
        ct = dec.backtrace_counters(B, T)
        print(f"B {B} chunks {chunks} warm {warm}: back-trace {ev[0].elapsed_time(ev[1]) / 3:.2f} ms  same: {bool(torch.equal(ref, st))}  {ct}", flush=True)
    del E, st, ll, ref
    dec._ws = None
    torch.cuda.empty_cache()
