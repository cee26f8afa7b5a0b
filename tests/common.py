"""Shared helpers for the tests (input regeneration from the golden manifest)."""
import hashlib

import numpy as np
import torch

from viterbi_spl_amd import synth

GEN = {"peaks": synth.emissions_peaks, "dense": synth.emissions_dense, "ties": synth.emissions_ties}


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
