"""Race hunt (test infrastructure): large batches decoded several times must give identical bytes, and every kernel family
applicable to a matrix (one song per workgroup, one song per wavefront, dense; sparse and whole-row back-trace) must agree
on every song."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
ok = True


def run(name, logA_T, log_pi, B, T, f16, kind="dense"):
    global ok
    S = logA_T.shape[0]
    dec = ViterbiDecoder(logA_T, log_pi, dev)
    gen = {"peaks": synth.emissions_peaks, "dense": synth.emissions_dense, "scaled": synth.emissions_scaled}[kind]
    E = gen(B, T, S, seed=99, device=dev, dtype=torch.float16 if f16 else torch.float32)
    lens = torch.randint(T // 2, T + 1, (B,), device=dev, dtype=torch.int64)
    ref = None
    forms = [("auto", 0)] * 3
    if dec.info["banded_ok"]:
        forms += [("group", 0), ("group", 2)]
        if dec.info["n_dense_rows"] == 0:
            forms += [("group", 4), ("group", 4)]          # one (song, chunk) stream per lane
        if dec.info["wave_ok"]:
            forms += [("wave", 0), ("wave", 2), ("wave", 0), ("wave", 4), ("packed", 0), ("packed", 0)]
    forms += [("dense", 0)]
    for algo, btf in forms:
        dec.set_option("backtrace_form", btf)
        if algo == "packed":                               # the songs' valid frames in one buffer (vit_decode_packed); compared after unpacking
            ln = lens.cpu().numpy()
            off = np.zeros(B + 1, np.int64)
            off[1:] = np.cumsum(ln)
            Ep = torch.cat([E[b, :int(ln[b])] for b in range(B)], dim=0).contiguous()
            sp, ll = dec.decode_packed(Ep, off, out_dtype=torch.int32)
            st = torch.full((B, T), -1, dtype=torch.int32, device=dev)
            idx = torch.arange(T, device=dev)[None, :] < lens[:, None]
            st[idx] = sp
            del Ep
        else:
            st, ll = dec.decode(E, lengths=lens, algo=algo, out_dtype=torch.int32)
        cur = (st.cpu().numpy().tobytes(), ll.cpu().numpy().tobytes())
        if ref is None:
            ref = cur
        elif cur != ref:
            ok = False
            print(name, algo, "backtrace_form", btf, "DIFFERS from the first run")
    print(name, "W", dec.info["group_window"], "wave" if dec.info["wave_ok"] else "-", f"{len(forms)} runs identical:", ok, flush=True)


A = synth.durrieu_transition(721, 20)
run("durrieu722", np.require(np.log(A).astype(np.float32).T, np.float32, ["C"]), np.log(np.full(722, 1.0 / 722)).astype(np.float32), 300, 4000, True)
for dm in (14, 40, 56):
    la, lp = synth.log_params(synth.tonet_transition(721, dm), synth.floored_prior(722))
    run(f"band722_dmax{dm}", la, lp, 300, 4000, True, "peaks")
la, lp = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
run("tonet361_B600_peaks", la, lp, 600, 6000, False, "peaks")
run("tonet361_B1500_dense", la, lp, 1500, 3000, False, "dense")
run("tonet361_B1100_scaled_f16", la, lp, 1100, 3000, True, "scaled")
la, lp = synth.log_params(synth.tonet_transition(320, 12), synth.floored_prior(321))
run("dcnet321_B900", la, lp, 900, 4000, False, "peaks")
sys.exit(0 if ok else 1)
