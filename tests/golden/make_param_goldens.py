#!/usr/bin/env python3
"""Pin viterbi_spl_amd/params.py, synth.durrieu_transition and datfile.py to the reference (build container only).

* tonet/viterbi_transition_post_processing.py and tonet/p_steady_post_processing.py are scripts (module-level code that
  reads a .dat file, plots, writes a .dat file).  They are RUN here as they stand, with their I/O stubbed: a fake
  `self_defined` module serves a synthetic count matrix / stationary distribution and captures what the script saves,
  and dummy `matplotlib` modules swallow the plots.  Committed: the synthetic inputs' seed and the scripts' outputs.
* imm/transition_matrix.py::gen_transition_matrix_fn is imported and compared with synth.durrieu_transition.
* the two shipped parameter files msnet/viterbi_*.dat are read with the reference's loader and with datfile.py; their
  SHA-256 and header lines are committed so that a test can rebuild the exact file bytes from tests/golden/params.npz.
Nothing from /root/reference is copied: outputs are data (tests/golden/param_goldens.npz, param_manifest.json)."""
import hashlib
import importlib.util
import json
import os
import runpy
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from viterbi_spl_amd import datfile, params, synth  # noqa: E402


def synthetic_counts(n_bins, seed):
    """Note -> note transition counts as viterbi_ini_probs_and_transition.py would tally them: mass near the diagonal
    falling off with distance, a sprinkle of far jumps, and voiced/unvoiced switches in the last row / column."""
    rng = np.random.default_rng(seed)
    C = np.zeros((n_bins + 1, n_bins + 1), np.int64)
    r = np.arange(n_bins)
    for d in range(-30, 31):
        lam = 4000.0 * np.exp(-abs(d) / 2.5) if abs(d) <= 20 else 0.2
        idx = r[(r + d >= 0) & (r + d < n_bins)]
        C[idx, idx + d] = rng.poisson(lam * (0.2 + rng.random(len(idx))), len(idx))
    C[:n_bins, n_bins] = rng.poisson(30, n_bins)
    C[n_bins, :n_bins] = rng.poisson(30, n_bins)
    C[n_bins, n_bins] = 50000
    return C


def run_script(path, served):
    """Run a reference post-processing script with `self_defined` I/O and matplotlib stubbed; returns what it saved."""
    saved = {}
    fake = types.ModuleType("self_defined")
    fake.load_np_array_from_file_fn = lambda file_name: served[file_name]
    fake.save_np_array_to_file_fn = lambda file_name, arr, name: saved.__setitem__(file_name, (name, np.array(arr)))
    mpl = types.ModuleType("matplotlib")
    mpl.use = lambda *a, **k: None
    plt = types.ModuleType("matplotlib.pyplot")
    for fn in ("scatter", "plot", "savefig", "close", "figure"):
        setattr(plt, fn, lambda *a, **k: None)
    mpl.pyplot = plt
    old = {k: sys.modules.get(k) for k in ("self_defined", "matplotlib", "matplotlib.pyplot")}
    sys.modules.update({"self_defined": fake, "matplotlib": mpl, "matplotlib.pyplot": plt})
    try:
        runpy.run_path(path, run_name="__ref_script__")
    finally:
        for k, v in old.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return saved


def sha_bytes(b):
    return hashlib.sha256(b).hexdigest()


def main():
    out, man = {}, {}
    # ---- tonet transition recipe on synthetic counts (n_bins = 360, d_max = 14)
    counts = synthetic_counts(360, seed=2024)
    saved = run_script(f"{REF}/tonet/viterbi_transition_post_processing.py", {"transition_int.dat": ("transition_int", counts)})
    name, A_ref = saved["viterbi_transition_matrix.dat"]
    assert name == "viterbi_transition_matrix" and A_ref.dtype == np.float32 and A_ref.shape == (361, 361)
    d_max = params.single_side_d_max_fn(h=0.01, B=60)
    assert d_max == 14
    A_mine = params.toeplitz_from_counts(counts, d_max)
    assert A_mine.dtype == np.float32 and A_mine.tobytes() == A_ref.tobytes(), "toeplitz_from_counts differs from the reference script"
    out["counts360"] = counts.astype(np.int32)
    out["transition360"] = A_ref
    # ---- floored prior
    rng = np.random.default_rng(7)
    p = rng.random(361) ** 6
    p[rng.integers(0, 360, 80)] = 0.0
    p[:-1] *= 0.45 / p[:-1].sum()
    p[-1] = 1.0 - p[:-1].sum()
    saved = run_script(f"{REF}/tonet/p_steady_post_processing.py", {"p_steady.dat": ("p_steady", p)})
    name, pi_ref = saved["viterbi_init_probs.dat"]
    assert name == "viterbi_init_probs" and pi_ref.dtype == np.float32
    pi_mine = params.floored_prior(p)
    assert pi_mine.tobytes() == pi_ref.tobytes(), "floored_prior differs from the reference script"
    out["p_steady361"] = p
    out["init_probs361"] = pi_ref
    # ---- Durrieu matrix
    spec = importlib.util.spec_from_file_location("ref_imm_transition_matrix", f"{REF}/imm/transition_matrix.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for bps, nb in ((20, 721), (20, 720), (5, 180)):
        D_ref = mod.gen_transition_matrix_fn(bps, nb)
        D_mine = synth.durrieu_transition(nb, bps)
        assert D_ref.dtype == np.float64 and D_mine.tobytes() == D_ref.tobytes(), "durrieu_transition differs from the reference"
        man[f"durrieu_{bps}_{nb}_sha256"] = sha_bytes(D_ref.tobytes())
    # ---- the shipped .dat files: reference loader == datfile.py; header + hash so that tests can rebuild the bytes
    spec = importlib.util.spec_from_file_location("ref_load_dat", f"{REF}/self_defined/load_np_array_from_file.py")
    lmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lmod)
    P = np.load(os.path.join(HERE, "params.npz"))
    for fname, key in (("viterbi_transition_matrix.dat", "msnet321_A"), ("viterbi_init_probs.dat", "msnet321_pi")):
        path = f"{REF}/msnet/{fname}"
        n_ref, a_ref = lmod.load_np_array_from_file_fn(path)
        n_me, a_me = datfile.load_np_array_from_file_fn(path)
        assert n_ref == n_me and a_ref.dtype == a_me.dtype and a_ref.shape == a_me.shape and a_ref.tobytes() == a_me.tobytes()
        assert a_ref.tobytes() == P[key].tobytes(), "params.npz does not hold the shipped parameters"
        raw = open(path, "rb").read()
        header = raw[:raw.index(b"\n") + 1]
        assert header + a_ref.tobytes() == raw
        man[fname] = {"header": header.decode(), "sha256": sha_bytes(raw), "params_key": key}
    np.savez_compressed(os.path.join(HERE, "param_goldens.npz"), **out)
    with open(os.path.join(HERE, "param_manifest.json"), "w") as fh:
        json.dump(man, fh, indent=1)
    print("wrote param_goldens.npz, param_manifest.json")


if __name__ == "__main__":
    main()
