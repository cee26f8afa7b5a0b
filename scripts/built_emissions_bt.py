"""Back-trace on the emissions the builder makes from pitch logits (bench.py's pipeline block) against the synthetic "peaks" rows
of the sweep: time and per-1000-frame event counters of the one-stream-per-wavefront and the one-stream-per-lane kernels at
B = 1024 (test infrastructure; run on the GPU box)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from viterbi_spl_amd import ViterbiDecoder, emissions, synth
dev = torch.device("cuda:0")
A, pi = bench.make_params("tonet", 361, 14)
dec = ViterbiDecoder(A, pi, dev)
T, B = 30000, int(sys.argv[1]) if len(sys.argv) > 1 else 1024


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty((B,), dtype=torch.float32, device=dev)
for kind in ("built", "built_segments", "peaks", "dense"):
    if kind.startswith("built"):
        X = synth.pitch_logits(32, T, 360, seed=5, device=dev, voicing="segments" if kind.endswith("segments") else "toggle").repeat(B // 32, 1, 1).contiguous()
        E = torch.empty((B, T, 361), dtype=torch.float32, device=dev)
        emissions.shaun_log_emissions(X, out=E)
        del X
    else:
        gen = synth.emissions_peaks if kind == "peaks" else synth.emissions_dense
        E = gen(32, T, 361, seed=1234, device=dev).repeat(B // 32, 1, 1).contiguous()
    ref = None
    for form in (0, 4):
        dec.set_option("reset", 0); dec.set_option("backtrace_form", form)
        dec.decode_into(E, st, ll, algo="banded"); torch.cuda.synchronize()
        if ref is None: ref = st.clone()
        t = timed(lambda: dec.decode_into(E, st, ll, algo="banded", phase="backtrace"))
        ct = dec.backtrace_counters(B, T)
        voiced = float((st != 360).float().mean())
        print(f"{kind} B {B} form {form}: back-trace {t:.2f} ms  same {bool(torch.equal(ref, st))}  voiced {voiced:.3f}  per 1000 frames "
              f"{ {k: round(v * 1000.0 / (B * T), 2) for k, v in ct.items()} }", flush=True)
    del E, ref
    torch.cuda.empty_cache()
