"""A/B of the scalars a full-history row of the wave form carries (test infrastructure; run on the GPU box): its own only ("wave_two" bit 2,
the layout up to round 3) against its own and those of the two frames before it (one scalar line per three frames in the back-trace).
Forward / back-trace (one stream per wavefront, one per lane) / two-stream step at B = 1024 and 2048, alternating on one box; paths compared."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from viterbi_spl_amd import ViterbiDecoder, synth
dev = torch.device("cuda:0")
A, pi = bench.make_params("tonet", 361, 14)
dec = ViterbiDecoder(A, pi, dev)
T = 30000


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for B in [int(x) for x in sys.argv[1:]] or [1024, 2048]:
    base = synth.emissions_peaks(32, T, 361, seed=1234, device=dev)
    E = base.repeat(B // 32, 1, 1).contiguous()
    del base
    st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty((B,), dtype=torch.float32, device=dev)
    ref = None
    for rep in range(2):
        for flag in (4, 0):
            dec.set_option("reset", 0)
            dec.set_option("wave_two", flag)
            st.fill_(-1)
            dec.decode_into(E, st, ll, algo="banded"); torch.cuda.synchronize()
            if ref is None: ref = st.clone()
            same = bool(torch.equal(ref, st))
            f = timed(lambda: dec.decode_into(E, st, ll, algo="banded", phase="forward"))
            b = timed(lambda: dec.decode_into(E, st, ll, algo="banded", phase="backtrace"))
            dec.set_option("backtrace_form", 4)
            bl = timed(lambda: dec.decode_into(E, st, ll, algo="banded", phase="backtrace"))
            same &= bool(torch.equal(ref, st))
            dec.set_option("backtrace_form", 0)
            dec._ws = None
            torch.cuda.empty_cache()
            two = bench.time_overlapped(dec, E, "banded", 6)
            two = two[0] if isinstance(two, tuple) else two
            print(f"B {B} {'own scalars only ' if flag else 'three frames a row'}: forward {f:.2f} ms  back-trace {b:.2f}  lane form {bl:.2f}  two streams {two:.2f} ms per step  same paths {same}", flush=True)
    del E, st, ll, ref
    dec._ws = None
    torch.cuda.empty_cache()
