"""Back-trace time of the wave form's full history at B = 1024 / 2048 (S = 361, T = 30000, peak-sparse emissions): five launches behind one
forward pass (test infrastructure; run on the GPU box)."""
import sys, torch
sys.path.insert(0, ".")
from viterbi_spl_amd import ViterbiDecoder, synth
dev = torch.device("cuda:0")
A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
T = 30000
base = synth.emissions_peaks(32, T, 361, seed=1234, device=dev)
for B in (1024, 2048):
    E = base.repeat(B // 32, 1, 1).contiguous()
    st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty((B,), dtype=torch.float32, device=dev)
    dec.decode_into(E, st, ll, algo="wave"); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5): dec.decode_into(E, st, ll, algo="wave", phase="backtrace")
    ev[1].record(); torch.cuda.synchronize()
    print(f"B {B} backtrace {ev[0].elapsed_time(ev[1]) / 5:.3f} ms", flush=True)
    del E, st, ll; dec._ws = None; torch.cuda.empty_cache()
