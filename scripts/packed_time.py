"""Packed ragged decode (vit_decode_packed) at full size: per-kernel times from HIP events around whole calls (test infrastructure)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from viterbi_spl_amd import ViterbiDecoder
dev = torch.device("cuda:0")
A, pi = bench.make_params("tonet", 361, 14)
dec = ViterbiDecoder(A, pi, dev)
for total in (1024, 2048):
    r = bench.packed_row(dec, A, pi, 30000, total * 30000, dev, 5)
    print(total, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()}, flush=True)
