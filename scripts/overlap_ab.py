"""A/B of the back-trace's row loops under the headline's two-stream schedule (B = 128, forward of step i+1 beside the back-trace of
step i): does the faster back-trace take issue slots from the forward workgroups it shares CUs with?  (test infrastructure; GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
E = bench.tiled_emissions(synth.emissions_peaks, 128, 30000, 361, 1234, dev, torch.float32)
for rep in range(2):
    for fast in (0, 1):       # (bench.time_overlapped sets bt_chunks = chunks_beside_forward(B) itself: 16 at B = 128)
        dec.set_option("reset", 0)
        dec.set_option("bt_fast_rows", fast)
        wall, st, ll = bench.time_overlapped(dec, E, "auto", steps=40)
        dec.set_option("bt_fast_rows", fast)
        r, st2, ll2 = bench.time_serial(dec, E, "auto", steps=10)
        print(f"bt_fast_rows {fast}: two streams {wall:.3f} ms per step; one stream forward {r['forward_ms']:.3f} + back-trace {r['backtrace_ms']:.3f} ms", flush=True)
