"""Golden vectors for the S = 361 production entry points at full length, generated in the build container
from the reference's own methods:

  * ``Viterbi.viterbi_librosa_fn(self, probs_st)``        tonet/for_paper.py:1833-1870  (family B: F-order [S, T] probabilities)
  * ``SoftMaxViterbi.viterbi_librosa_fn(self, probs_ts)``  tonet/for_paper.py:1999-2037  (family C: C-order [T, S] probabilities)

Only the two ``ast.FunctionDef`` nodes are compiled (the module itself needs TensorFlow, torch-lightning, librosa, medleydb); ``self``
is a plain namespace carrying what ``__init__`` prepares (:1780-1815: log(x + tiny) in float32, the transition transposed, C order).
Commits data only: the decoded states (uint16), the seed and the SHA-256 of the regenerated inputs.

    python tests/golden/make_familyB_golden.py
"""
import ast
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from viterbi_spl_amd import synth                # noqa: E402

S, T, SEED = 361, 30000, 41


def inputs(seed=SEED, T=T):
    """[S, T] float32 probabilities, F-order: a k/4096 grid with exact zeros (what tests/golden/make_goldens.py's prob cases use)."""
    rows = torch.arange(S, dtype=torch.int64)
    cols = torch.arange(T, dtype=torch.int64)
    h = synth._cell_hash(synth._mix32(rows ^ seed), cols)
    return np.asfortranarray(((h % 4096).to(torch.float32) / 4096.0).numpy())


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def reference_methods():
    tree = ast.parse(open(f"{REF}/tonet/for_paper.py").read())
    fns = {}
    for cls in tree.body:
        if isinstance(cls, ast.ClassDef) and cls.name in ("Viterbi", "SoftMaxViterbi"):
            node = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "viterbi_librosa_fn")
            ns = {"np": np}
            exec(compile(ast.Module(body=[node], type_ignores=[]), f"ref_{cls.name}_viterbi_librosa_fn", "exec"), ns)
            fns[cls.name] = ns["viterbi_librosa_fn"]
    assert set(fns) == {"Viterbi", "SoftMaxViterbi"}
    return fns


def main():
    A = synth.tonet_transition(360, 14)
    pi = synth.floored_prior(S)
    tiny = np.finfo(np.float32).tiny
    me = types.SimpleNamespace(num_freq_bins=360,                                           # what __init__ prepares (:1780-1815)
                               log_transition_matrix_T=np.require(np.log(A + tiny).T, np.float32, ["C"]),
                               log_ini_probs=np.require(np.log(pi + tiny), np.float32, ["C"]))
    fns = reference_methods()
    P = inputs()
    sB = fns["Viterbi"](me, P.copy(order="F"))
    sC = fns["SoftMaxViterbi"](me, np.array(P.T, order="C"))
    assert sB.dtype == np.int64 and sB.shape == (T,) and np.array_equal(sB, sC)
    np.savez_compressed(os.path.join(HERE, "familyB_goldens.npz"), states_B=sB.astype(np.uint16), states_C=sC.astype(np.uint16))
    with open(os.path.join(HERE, "familyB_manifest.json"), "w") as fh:
        json.dump({"S": S, "T": T, "seed": SEED, "sha256_probs_st": sha(P), "params": "synth.tonet_transition(360, 14), synth.floored_prior(361)",
                   "sha256_params": sha(A, pi), "unvoiced_frames": int(np.sum(sB == S - 1)), "distinct_states": int(len(np.unique(sB))),
                   "reference": ["tonet/for_paper.py:1833-1870", "tonet/for_paper.py:1999-2037"]}, fh, indent=1)
    print("family B / C goldens at S = 361, T = 30000 written;", int(np.sum(sB == S - 1)), "unvoiced frames,", len(np.unique(sB)), "distinct states")


if __name__ == "__main__":
    main()
