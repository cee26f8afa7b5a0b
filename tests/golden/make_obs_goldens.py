#!/usr/bin/env python3
"""Pin oracle/observation_oracle.py to the reference's emission builders (build container only).

The classes live in tonet/for_paper.py, whose module top imports TF / torch / librosa / medleydb;
only the ast.FunctionDef nodes of the needed methods are compiled (with `np` in scope) and bound to
a bare namespace object carrying the attributes their __init__ would set (for_paper.py:1691-1701,
:1881-1887).  Outputs: tests/golden/obs_goldens.npz (seeds -> expected probabilities)."""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/tonet/for_paper.py"

from oracle import observation_oracle as oo  # noqa: E402


def extract(class_name, method_names, path=None):
    tree = ast.parse(open(path or REF).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == class_name)
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in method_names]
    assert len(fns) == len(method_names)
    # strip decorators (staticmethod) -- functions are called explicitly below
    for f in fns:
        f.decorator_list = []
    ns = {"np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), f"ref_{class_name}", "exec"), ns)
    return ns


from tests.common import logits_case  # noqa: E402  (shared with the tests)


def main():
    v = extract("Viterbi", ["expit", "find_peaks_all_at_once_np_fn", "observation_probs_fn"])
    s = extract("SoftMaxViterbi", ["find_peaks_all_at_once_np_fn", "observation_probs_fn"])
    out = {}
    for k, (seed, n) in enumerate([(1, 64), (2, 200), (3, 1)]):
        # --- Viterbi ("shaun") ---
        self_v = types.SimpleNamespace(num_freq_bins=360, single_side_peak_width=5,
                                       threshold=np.log(0.32 / (1. - 0.32)))
        self_v.find_peaks_all_at_once_np_fn = lambda fl, _s=self_v: v["find_peaks_all_at_once_np_fn"](_s, fl)
        v["Viterbi"] = types.SimpleNamespace(expit=v["expit"])      # observation_probs_fn calls Viterbi.expit(...)
        x = logits_case(seed, n, 360)
        ref = v["observation_probs_fn"](self_v, x.copy())
        mine = oo.shaun_observation_probs(x.copy(), 0.32)
        assert ref.shape == (361, n) and ref.dtype == np.float32 and ref.flags["F_CONTIGUOUS"]
        assert ref.tobytes(order="A") == mine.tobytes(order="A"), f"shaun restatement differs (case {k})"
        out[f"shaun{k}_seed"] = np.asarray([seed, n])
        out[f"shaun{k}_probs"] = np.ascontiguousarray(ref.T)          # [n, 361]
        # --- SoftMaxViterbi ---
        self_s = types.SimpleNamespace(num_freq_bins=360, single_side_peak_width=15)
        self_s.find_peaks_all_at_once_np_fn = lambda lg, _s=self_s: s["find_peaks_all_at_once_np_fn"](_s, lg)
        y = logits_case(seed + 10, n, 361)
        ref = s["observation_probs_fn"](self_s, y.copy())
        mine = oo.softmax_observation_probs(y.copy())
        assert ref.tobytes() == mine.tobytes(), f"softmax restatement differs (case {k})"
        out[f"softmax{k}_seed"] = np.asarray([seed + 10, n])
        out[f"softmax{k}_probs"] = ref
        print(f"case {k}: n={n} ok; voiced frames (shaun) = {int(np.sum(out[f'shaun{k}_probs'][:, -1] < 0.5))}")
    # --- dcnet's SoftMaxViterbi: scaled likelihoods p / prior (dcnet/softmax_viterbi.py:2530-2579), 320 bins, +/-5-bin
    #     peaks, the unvoiced logit padded from the voicing-threshold probability; prior = the shipped msnet
    #     viterbi_init_probs.dat (same 320-bin grid).  Values above 1 (positive log-emissions) are the point.
    d = extract("SoftMaxViterbi", ["find_peaks_all_at_once_np_fn", "observation_probs_fn"], "/root/reference/dcnet/softmax_viterbi.py")
    prior = np.load(os.path.join(HERE, "params.npz"))["msnet321_pi"]
    for k, (seed, n, vth, scaled) in enumerate([(21, 64, 0.5, True), (22, 200, 0.2, True), (23, 1, 0.5, True), (24, 120, 0.35, False)]):
        self_d = types.SimpleNamespace(num_freq_bins=320, single_side_peak_width=5, scaled=scaled, ini_probs=prior,
                                       voicing_threshold_prob_tf_var=types.SimpleNamespace(numpy=lambda _v=vth: np.float32(_v)))
        self_d.find_peaks_all_at_once_np_fn = lambda lg, _s=self_d: d["find_peaks_all_at_once_np_fn"](_s, lg)
        z = logits_case(seed, n, 320)
        ref = d["observation_probs_fn"](self_d, z.copy())
        mine = oo.softmax_scaled_observation_probs(z.copy(), np.float32(vth), prior, scaled=scaled)
        assert ref.dtype == np.float32 and ref.shape == (n, 321)
        assert ref.tobytes() == mine.tobytes(), f"scaled-softmax restatement differs (case {k})"
        out[f"scaled{k}_seed"] = np.asarray([seed, n])
        out[f"scaled{k}_vth"] = np.asarray([vth, 1.0 if scaled else 0.0], np.float32)
        out[f"scaled{k}_probs"] = ref
        print(f"scaled case {k}: n={n} scaled={scaled} ok; max likelihood {ref.max():.4g}")
    np.savez_compressed(os.path.join(HERE, "obs_goldens.npz"), **out)
    print("wrote obs_goldens.npz")


if __name__ == "__main__":
    main()
