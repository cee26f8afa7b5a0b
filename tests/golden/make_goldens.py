#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own NumPy decoders.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):   python tests/golden/make_goldens.py

What is imported from the reference (nothing is copied into this repo):
  * imm/tf_viterbi.py::viterbi_librosa_fn (:75-109), loaded by file path with an
    empty stub registered for its unused top-level ``import tensorflow``;
  * dcnet/tf_viterbi_decoding.py::viterbi_librosa_c_fn (:156-207) and its
    float64-accumulating sibling viterbi_librosa_fn (:209-263), obtained by
    compiling only those ast.FunctionDef nodes (the module itself needs TF, Numba
    and three unshipped .dat files at import time);
  * self_defined/load_np_array_from_file.py (the .dat reader) by file path, to
    read msnet/viterbi_{transition_matrix,init_probs}.dat (data fixtures).

Every case is also run through oracle/viterbi_oracle.py (NumPy restatement and
C restatement) and must agree bit-for-bit before anything is written.
The committed outputs are data only: inputs (or the seeds + SHA-256 of
regenerated inputs) and expected outputs.
"""
import ast
import hashlib
import importlib.util
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

from oracle import viterbi_oracle as vo          # noqa: E402
from viterbi_spl_amd import synth                # noqa: E402


def load_reference():
    sys.modules.setdefault("tensorflow", types.ModuleType("tensorflow"))
    spec = importlib.util.spec_from_file_location("ref_imm_tf_viterbi", f"{REF}/imm/tf_viterbi.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    log_fn = mod.viterbi_librosa_fn

    src = open(f"{REF}/dcnet/tf_viterbi_decoding.py").read()
    tree = ast.parse(src)
    nodes = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("viterbi_librosa_c_fn", "viterbi_librosa_fn")]
    assert len(nodes) == 2
    ns = {"np": np}
    exec(compile(ast.Module(body=nodes, type_ignores=[]), "ref_dcnet_fns", "exec"), ns)
    prob_fn = ns["viterbi_librosa_c_fn"]
    global ref_f64_fn
    ref_f64_fn = ns["viterbi_librosa_fn"]          # float64 T1: the tolerance reference

    spec = importlib.util.spec_from_file_location("ref_load_dat", f"{REF}/self_defined/load_np_array_from_file.py")
    lmod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lmod)
    return log_fn, prob_fn, lmod.load_np_array_from_file_fn


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    ref_log_fn, ref_prob_fn, ref_load_dat = load_reference()

    # ---------------------------------------------------------------- parameter fixtures
    name_a, A_ms = ref_load_dat(f"{REF}/msnet/viterbi_transition_matrix.dat")
    name_p, pi_ms = ref_load_dat(f"{REF}/msnet/viterbi_init_probs.dat")
    assert name_a == "viterbi_transition_matrix" and A_ms.shape == (321, 321) and A_ms.dtype == np.float32
    assert name_p == "viterbi_init_probs" and pi_ms.shape == (321,)
    A_to = synth.tonet_transition(360, 14)
    pi_to = synth.floored_prior(361)
    A_du = synth.durrieu_transition(721, 20)
    params = {}
    params["msnet321"] = synth.log_params(A_ms, pi_ms)
    params["tonet361"] = synth.log_params(A_to, pi_to)
    logA_du = np.require(np.log(A_du).astype(np.float32).T, np.float32, ["C"])
    params["durrieu722"] = (logA_du, np.log(np.full(722, 1.0 / 722)).astype(np.float32))
    # the 722-state grids with the band half-widths of jdc (40) and imm's post-processing matrix (56): W = 96 / 128 windows
    params["jdc722"] = synth.log_params(synth.tonet_transition(721, 40), synth.floored_prior(722))
    params["imm722w"] = synth.log_params(synth.tonet_transition(721, 56), synth.floored_prior(722))
    params["dense361"] = (synth.dense_random_log_transition(361, seed=3),
                          synth.dense_random_log_transition(361, seed=4)[0].copy())
    params["dense97"] = (synth.dense_random_log_transition(97, seed=5),
                         synth.dense_random_log_transition(97, seed=6)[0].copy())
    # BASELINE's literal "S=721": 720 bins + the unvoiced state, jdc band half-width and the Durrieu recipe
    params["jdc721"] = synth.log_params(synth.tonet_transition(720, 40), synth.floored_prior(721))
    A_du721 = synth.durrieu_transition(720, 20)
    params["durrieu721"] = (np.require(np.log(A_du721).astype(np.float32).T, np.float32, ["C"]),
                            np.log(np.full(721, 1.0 / 721)).astype(np.float32))
    np.savez_compressed(os.path.join(HERE, "params.npz"),
                        msnet321_A=A_ms, msnet321_pi=pi_ms,
                        **{f"{k}_logA_T": v[0] for k, v in params.items()},
                        **{f"{k}_log_pi": v[1] for k, v in params.items()})

    # ---------------------------------------------------------------- log-domain cases
    cases = []
    def add(pname, kind, T, seed, f16=False):
        cases.append(dict(params=pname, kind=kind, T=T, seed=seed, f16=f16))
    for T in (1, 2, 7, 1000):
        for kind in ("peaks", "dense", "ties"):
            add("tonet361", kind, T, 10 + T)
            add("msnet321", kind, T, 20 + T)
    add("tonet361", "dense", 1000, 77, f16=True)
    add("tonet361", "peaks", 1000, 78, f16=True)
    add("dense361", "dense", 1000, 31)
    add("dense361", "ties", 500, 32)
    add("dense97", "dense", 333, 33)
    add("dense97", "ties", 64, 34)
    add("durrieu722", "dense", 300, 41)
    add("durrieu722", "peaks", 300, 42)
    add("durrieu722", "dense", 200, 43, f16=True)
    add("jdc722", "peaks", 300, 51)
    add("jdc722", "ties", 200, 52, f16=True)
    add("imm722w", "peaks", 200, 53)
    add("imm722w", "dense", 150, 54, f16=True)
    add("tonet361", "peaks", 30000, 1)
    add("tonet361", "dense", 30000, 2)
    add("msnet321", "peaks", 30000, 3)
    add("dense361", "dense", 30000, 4)
    # (appended in round 2; the cases above keep their indices)
    # positive log-emissions ("scaled likelihood" p / prior, dcnet/softmax_viterbi.py:2571-2572): values up to +6
    add("tonet361", "scaled", 1000, 61)
    add("msnet321", "scaled", 1000, 62)
    add("tonet361", "scaled", 700, 63, f16=True)
    add("jdc722", "scaled", 300, 64, f16=True)
    add("durrieu722", "scaled", 200, 65)
    add("dense361", "scaled", 400, 66)
    add("tonet361", "scaled", 30000, 67)
    # literal S = 721
    add("jdc721", "peaks", 300, 71)
    add("jdc721", "ties", 200, 72, f16=True)
    add("jdc721", "scaled", 250, 73)
    add("durrieu721", "dense", 200, 74)
    add("durrieu721", "peaks", 200, 75, f16=True)

    gen = {"peaks": synth.emissions_peaks, "dense": synth.emissions_dense, "ties": synth.emissions_ties,
           "scaled": synth.emissions_scaled}
    out = {}
    manifest = []
    for k, c in enumerate(cases):
        logA_T, log_pi = params[c["params"]]
        S = logA_T.shape[0]
        dt = torch.float16 if c["f16"] else torch.float32
        logE = gen[c["kind"]](1, c["T"], S, seed=c["seed"], dtype=dt)[0].to(torch.float32).numpy()
        # the reference takes emissions as [S,T]
        ref_states = ref_log_fn(log_transition_matrix_T=logA_T, log_prob_init=log_pi,
                                log_probs_st=np.asfortranarray(logE.T))
        assert ref_states.dtype == np.int64 and ref_states.shape == (c["T"],)
        st_np, ll_np, d_np = vo.decode_numpy(logA_T, log_pi, logE, return_delta=True)
        st_c, ll_c, d_c = vo.decode_c(logA_T, log_pi, logE, return_delta=True)
        assert np.array_equal(ref_states, st_np), f"numpy restatement differs from reference in case {k}"
        assert np.array_equal(ref_states, st_c), f"C restatement differs from reference in case {k}"
        assert d_np.tobytes() == d_c.tobytes() and np.float32(ll_np).tobytes() == np.float32(ll_c).tobytes()
        out[f"c{k}_states"] = ref_states.astype(np.uint16)
        out[f"c{k}_delta"] = d_c
        c.update(index=k, S=int(S), loglik=float(ll_c), sha256=sha(logA_T, log_pi, logE),
                 jumps=int(np.sum(ref_states[1:] != ref_states[:-1])), max_logE=float(np.max(logE)))
        manifest.append(c)
        print(f"case {k:2d} {c['params']:>10s} {c['kind']:>5s} T={c['T']:<6d} f16={c['f16']!s:5s} ok "
              f"loglik={ll_c:.6g} jumps={c['jumps']}")

    # ---------------------------------------------------------------- prob-domain (family A) cases
    pcases = []
    for k, (T, seed) in enumerate([(1, 5), (3, 6), (500, 7)]):
        rows = torch.arange(321, dtype=torch.int64)
        cols = torch.arange(T, dtype=torch.int64)
        h = synth._cell_hash(synth._mix32(rows ^ seed), cols)
        probs_st = np.asfortranarray(((h % 4096).to(torch.float32) / 4096.0).numpy())  # [S,T], zeros allowed
        ref_states = ref_prob_fn(transition_matrix=A_ms, prob_init=pi_ms, probs_st=probs_st.copy(order="F"))
        mine = vo.decode_probs_numpy(A_ms, pi_ms, probs_st)
        assert np.array_equal(ref_states, mine)
        out[f"p{k}_states"] = ref_states.astype(np.uint16)
        out[f"p{k}_probs_st"] = probs_st
        pcases.append(dict(index=k, T=T, seed=seed, S=321))
        print(f"prob case {k} T={T} ok")

    # ---------------------------------------------------------------- the float64 sibling (a2): tolerance reference
    # dcnet/tf_viterbi_decoding.py:209-263 keeps T1 in float64, so its sums round differently from the float32 family
    # (families A-D, the TF graph, the Numba core: what this repo reproduces bit for bit).  The reference's own
    # cross-check (:284) asserts equal paths on its unshipped data; on long inputs the two disagree in a few frames.
    # Committed: its states, the float32 function's states on the same input, and both paths' exact (float64) scores.
    def prob_inputs(seed, T, dense):
        rows = torch.arange(321, dtype=torch.int64)
        cols = torch.arange(T, dtype=torch.int64)
        h = synth._cell_hash(synth._mix32(rows ^ seed), cols)
        if dense:      # Dirichlet-like dense columns: every state plausible (where the two variants drift apart)
            p = ((h % 4093) + 1).to(torch.float64)
            p = (p / p.sum(dim=0, keepdim=True)).to(torch.float32)
            return np.asfortranarray(p.numpy())
        return np.asfortranarray(((h % 4096).to(torch.float32) / 4096.0).numpy())

    def path_score64(states, probs_st):
        tiny = np.finfo(np.float32).tiny
        lA = np.log(A_ms.astype(np.float64) + tiny)
        lpi = np.log(pi_ms.astype(np.float64) + tiny)
        lE = np.log(probs_st.astype(np.float64) + tiny)
        s = lpi[states[0]] + lE[states[0], 0]
        s += np.sum(lA[states[:-1], states[1:]]) + np.sum(lE[states[1:], np.arange(1, len(states))])
        return float(s)

    f64cases = []
    for k, (T, seed, dense) in enumerate([(500, 7, False), (4000, 7, True), (30000, 11, True), (30000, 12, False)]):
        probs_st = prob_inputs(seed, T, dense)
        s64 = np.asarray(ref_f64_fn(transition_matrix=A_ms, prob_init=pi_ms, probs_st=probs_st.copy(order="F")), np.int64)
        s32 = ref_prob_fn(transition_matrix=A_ms, prob_init=pi_ms, probs_st=probs_st.copy(order="F"))
        assert np.array_equal(s32, vo.decode_probs_numpy(A_ms, pi_ms, probs_st))
        agree = float(np.mean(s64 == s32))
        sc64, sc32 = path_score64(s64, probs_st), path_score64(s32, probs_st)
        out[f"f64_{k}_states64"] = s64.astype(np.uint16)
        out[f"f64_{k}_states32"] = s32.astype(np.uint16)
        f64cases.append(dict(index=k, T=T, seed=seed, dense=dense, S=321, agreement=agree, differing_frames=int(np.sum(s64 != s32)),
                             score64=sc64, score32=sc32, rel_score_diff=abs(sc64 - sc32) / abs(sc64), sha256=sha(probs_st)))
        print(f"f64 case {k} T={T} dense={dense}: {int(np.sum(s64 != s32))} of {T} frames differ, "
              f"exact scores {sc64:.10g} vs {sc32:.10g} (rel {abs(sc64 - sc32) / abs(sc64):.3g})")

    np.savez_compressed(os.path.join(HERE, "goldens.npz"), **out)
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump({"log_cases": manifest, "prob_cases": pcases, "f64_cases": f64cases,
                   "numpy": np.__version__, "generated_from": "imm/tf_viterbi.py:75-109, "
                   "dcnet/tf_viterbi_decoding.py:156-207 (imported, not copied)"}, fh, indent=1)
    print("wrote goldens.npz, params.npz, manifest.json")


if __name__ == "__main__":
    main()
