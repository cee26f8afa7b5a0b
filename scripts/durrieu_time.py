"""Forward time of the Durrieu matrix at configs[4] ([256, 30000, 722] fp16) with an oracle spot check (test infrastructure; GPU box)."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from viterbi_spl_amd import ViterbiDecoder, synth
from oracle import viterbi_oracle as vo
import bench
dev = torch.device("cuda:0")
A, pi = bench.make_params("durrieu", 722, 14)
dec = ViterbiDecoder(A, pi, dev)
E = bench.tiled_emissions(synth.emissions_peaks, 256, 30000, 722, 1234, dev, torch.float16)
st = torch.empty((256, 30000), dtype=torch.int32, device=dev); ll = torch.empty((256,), dtype=torch.float32, device=dev)
dec.decode_into(E, st, ll, algo="auto"); torch.cuda.synchronize()
rs, rl = vo.decode_c(A, pi, E[:2].float().cpu().numpy())
print("oracle:", np.array_equal(st[:2].cpu().numpy(), rs), np.array_equal(ll[:2].cpu().numpy(), rl))
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(3): dec.decode_into(E, st, ll, algo="auto", phase="forward")
    ev[1].record(); torch.cuda.synchronize()
    print(f"durrieu forward {ev[0].elapsed_time(ev[1]) / 3:.2f} ms", flush=True)
