"""Throughput of the emission builders: pitch logits -> log-emission rows in the decoder's layout.
Elementwise / HBM-bound: bytes = logits read + emission rows written."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import emissions
dev = torch.device("cuda:0")
B, T, U = int(os.environ.get("OB", "32")), 30000, 360
g = torch.Generator(device=dev); g.manual_seed(3)
for name, fn, cols in (("shaun", emissions.shaun_log_emissions, U), ("softmax", emissions.softmax_log_emissions, U + 1)):
    x = torch.randn((B, T, cols), generator=g, device=dev) * 3.0
    for _ in range(2):
        y = fn(x)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(5):
        y = fn(x)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 5
    nbytes = x.numel() * 4 + y.numel() * 4
    print(f"{name}: [{B},{T},{cols}] -> [{B},{T},{U + 1}]  {ms:.3f} ms  {B * T / ms / 1e3:.0f} Mframes/s  {nbytes / ms / 1e6:.0f} GB/s")
