"""Two-stream schedule with the HALF history at B = 1024 (one forward wave per SIMD leaves 256 registers of every SIMD free): does the half
back-trace of batch i run UNDER the forward pass of batch i + 1 when its workgroups are small enough to start beside the resident
forward waves?  (test infrastructure; run on the GPU box)   argv: batch sizes (default 1024)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
A, pi = bench.make_params("tonet", 361, 14)
dec = ViterbiDecoder(A, pi, dev)
T = 30000
for B in [int(x) for x in sys.argv[1:] if x.isdigit()] or [1024]:
    E = bench.tiled_emissions(synth.emissions_peaks, B, T, 361, 1234, dev, torch.float32)
    ref = None
    for hist, waves, chunks in ((1, 0, 0), (2, 0, 0), (2, 8, 0), (2, 8, 2), (2, 4, 0), (2, 8, 8)):
        dec.set_option("reset", 0)
        dec.set_option("wave_history", hist)
        dec.set_option("bt_block_waves", waves)
        r, st, ll = bench.time_serial(dec, E, "wave", steps=5)
        ser = r["forward_ms"] + r["backtrace_ms"]
        # time_overlapped sets bt_chunks itself (chunks_beside_forward is 0 at this batch size): set ours afterwards through a wrapper
        orig = dec.chunks_beside_forward
        dec.chunks_beside_forward = lambda b, c=chunks: c
        wall, st2, ll2 = bench.time_overlapped(dec, E, "wave", steps=5)
        dec.chunks_beside_forward = orig
        dec.set_option("wave_history", hist)
        same = bool(torch.equal(st, st2)) and (ref is None or bool(torch.equal(ref, st)))
        if ref is None:
            ref = st.clone()
        print(f"B {B} history {'full' if hist == 1 else 'half'} block waves {waves or 16} chunks {chunks or 'auto'}: one stream fwd {r['forward_ms']:.2f} + bt {r['backtrace_ms']:.2f} = {ser:.2f} ms "
              f"({B * T / ser / 1e3:.0f} Mframes/s); two streams {wall:.2f} ms per step ({B * T / wall / 1e3:.0f} Mframes/s, {B * T * 2172 / (wall * 1e-3) / 8e12:.3f} of the roofline)  same paths: {same}", flush=True)
        del st, ll, st2, ll2
        dec._ws = None
        dec._ws_slots = {}
        torch.cuda.empty_cache()
    del E, ref
    torch.cuda.empty_cache()
