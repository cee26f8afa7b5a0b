"""Randomised check of the emission builders (test infrastructure; run on the GPU box): random bin counts, window half-widths,
frame counts, quantised logits (ties) and the three builder forms against the oracle restatement of the reference builders:
identical peak sets (structural log(tiny) entries) and probabilities within 1e-5.  argv: seconds (default 120), seed."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import observation_oracle as oo  # noqa: E402
from viterbi_spl_amd import emissions as em  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
tiny = np.float32(np.finfo(np.float32).tiny)
floor = np.log(np.float32(0) + tiny)
t0 = time.time()
n_cases = 0
fails = 0
while time.time() - t0 < budget:
    U = int(rng.choice([rng.integers(8, 768), 320, 360, 720]))
    spw = int(min(rng.choice([1, 2, 3, 5, 7, 15, 16, 31, 33, 64]), U - 1))
    n = int(rng.choice([1, 2, 7, 64, 300]))
    quant = int(rng.choice([0, 0, 1, 2, 8]))
    x = (rng.standard_normal((n, U)) * float(rng.choice([0.5, 3, 10]))).astype(np.float32)
    if quant:
        x = (np.round(x * quant) / quant).astype(np.float32)
    xt = torch.from_numpy(x).to(dev)
    mode = int(rng.integers(3))
    if mode == 0:
        vth = float(rng.choice([0.1, 0.32, 0.5, 0.9]))
        want = np.log(oo.shaun_observation_probs(x, voicing_threshold=vth, spw=spw).T + tiny)
        got = em.shaun_log_emissions(xt, voicing_threshold=vth, single_side_peak_width=spw).cpu().numpy()
        cmp_log = False
    elif mode == 1:
        y = np.ascontiguousarray(np.concatenate([(rng.standard_normal((n, 1)) * 3).astype(np.float32), x], axis=1))
        want = np.log(oo.softmax_observation_probs(y, spw=spw) + tiny)
        got = em.softmax_log_emissions(torch.from_numpy(y).to(dev), single_side_peak_width=spw).cpu().numpy()
        cmp_log = False
    else:
        prior = rng.random(U + 1).astype(np.float32) + np.float32(1e-3)
        prior /= prior.sum()
        vth = float(rng.choice([0.2, 0.5, 0.8]))
        scaled = bool(rng.integers(2))
        want = np.log(oo.softmax_scaled_observation_probs(x, vth, prior, scaled=scaled, spw=spw) + tiny)
        got = em.softmax_scaled_log_emissions(xt, vth, torch.from_numpy(prior).to(dev) if scaled else None, single_side_peak_width=spw).cpu().numpy()
        cmp_log = True
    n_cases += 1
    same_set = np.array_equal(got == floor, want == floor)
    live = want != floor
    close = np.allclose(got[live], want[live], rtol=0, atol=3e-5) if cmp_log else np.allclose(np.exp(got[live]), np.exp(want[live]), rtol=1e-5, atol=1e-30)
    if not (same_set and close):
        fails += 1
        print("FAIL", mode, U, spw, n, quant, "peak set", same_set, "values", close, flush=True)
print(f"done: {n_cases} cases, {fails} failures")
sys.exit(1 if fails else 0)
