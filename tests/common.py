"""Shared helpers for the tests (input regeneration from the golden manifest)."""
import hashlib

import numpy as np
import torch

from viterbi_spl_amd import synth

GEN = {"peaks": synth.emissions_peaks, "dense": synth.emissions_dense, "ties": synth.emissions_ties,
       "scaled": synth.emissions_scaled}


def case_params(golden, case):
    p = golden["params"]
    return p[f"{case['params']}_logA_T"], p[f"{case['params']}_log_pi"]


def case_emissions(case, device="cpu", as_stored=False):
    """[T,S] emissions of a golden case.  as_stored=True keeps fp16 storage for f16 cases."""
    dt = torch.float16 if case["f16"] else torch.float32
    e = GEN[case["kind"]](1, case["T"], case["S"], seed=case["seed"], dtype=dt, device=device)[0]
    return e if as_stored else e.to(torch.float32)


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def logits_case(seed, n_frames, n_bins):
    """Seeded pitch logits for the emission-builder tests: weak noise floor (unvoiced frames), melody-like
    bumps in two frames out of three, and some exactly tied values."""
    rng = np.random.default_rng(seed)
    x = rng.normal(-8.0, 1.5, (n_frames, n_bins)).astype(np.float32)
    for f in range(n_frames):
        if f % 3:
            c = int(rng.integers(8, n_bins - 8))
            x[f, c - 2:c + 3] += np.asarray([1.0, 3.0, 6.0, 3.0, 1.0], np.float32) * np.float32(rng.uniform(0.5, 2.5))
    x[:, ::37] = np.round(x[:, ::37])
    return x
