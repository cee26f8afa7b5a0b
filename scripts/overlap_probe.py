"""Serial vs overlapped schedule (back-trace of batch i under the forward pass of batch i + 1, two workspace slots) for the
wave form with a full and with a half history.  argv: batch sizes (default 1024 2048)."""
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
T = 30000
for B in [int(x) for x in sys.argv[1:]] or [1024, 2048]:
    E = bench.tiled_emissions(synth.emissions_peaks, B, T, 361, 1234, dev, torch.float32)
    for label, opts in (("full history", {"wave_history": 1}), ("half history", {"wave_history": 2})):
        dec.set_option("reset", 0)
        for k, v in opts.items():
            dec.set_option(k, v)
        dec._ws = None
        dec._ws_slots = {}
        torch.cuda.empty_cache()
        r, st, ll = bench.time_serial(dec, E, "wave", steps=5)
        line = f"B {B} {label}: serial fwd {r['forward_ms']:.2f} + bt {r['backtrace_ms']:.2f} = {r['forward_ms'] + r['backtrace_ms']:.2f} ms ({B*T*2172/((r['forward_ms']+r['backtrace_ms'])*1e-3)/8e12:.3f})"
        try:
            dec._ws = None
            torch.cuda.empty_cache()
            wall, st2, ll2 = bench.time_overlapped(dec, E, "wave", steps=6)
            line += f"; overlapped {wall:.2f} ms per step ({B*T*2172/(wall*1e-3)/8e12:.3f}), same paths {bool(torch.equal(st, st2))}"
        except torch.OutOfMemoryError:
            line += "; overlapped: two slots do not fit"
        print(line, flush=True)
    del E
    torch.cuda.empty_cache()
