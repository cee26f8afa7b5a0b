"""Randomised parity run (test infrastructure; run on the GPU box): random state counts, band half-widths, matrix families,
batch shapes, ragged lengths, emission kinds and storage types; every kernel family the plan allows and both back-trace forms, the wave
form's half history, the checkpointed decode, the lane form of the back-trace (chunk counts up to 256) and the packed decode against the C
restatement in oracle/.  argv: seconds to run (default 240), seed (default 1)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import viterbi_oracle as vo  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dev = torch.device("cuda:0")
GEN = {"peaks": synth.emissions_peaks, "dense": synth.emissions_dense, "ties": synth.emissions_ties, "scaled": synth.emissions_scaled}


def random_matrix():
    fam = rng.choice(["band", "band", "band", "band_novoice", "dense", "durrieu", "band_edit"])
    if fam == "durrieu":
        n = int(rng.choice([705, 721, 740, 767]))
        A = synth.durrieu_transition(n, 20)
        la = np.require(np.log(A).astype(np.float32).T, np.float32, ["C"])
        lp = np.log(np.full(n + 1, 1.0 / (n + 1))).astype(np.float32)
        return f"durrieu{n + 1}", la, lp
    if fam == "dense":
        S = int(rng.choice([rng.integers(2, 200), rng.integers(129, 257), rng.integers(257, 369), rng.integers(369, 420)]))
        la = synth.dense_random_log_transition(S, seed=int(rng.integers(1 << 20)))
        lp = synth.dense_random_log_transition(S, seed=int(rng.integers(1 << 20)))[0].copy()
        return f"dense{S}", la, lp
    n = int(rng.choice([rng.integers(40, 383), rng.integers(300, 383), 320, 360, rng.integers(400, 760), 720, 721]))
    dmax = int(rng.integers(1, 8)) if n < 100 else int(rng.choice([rng.integers(2, 15), 12, 14, rng.integers(15, 60)]))
    dmax = min(dmax, n // 3)
    A = synth.tonet_transition(n, dmax)
    pi = synth.floored_prior(n + 1)
    if fam == "band_novoice":            # voiced block only: no extra column
        A = A[:n, :n] / A[:n, :n].sum(axis=1, keepdims=True)
        pi = pi[:n] / pi[:n].sum()
    la, lp = synth.log_params(A, pi)
    if fam == "band_edit" and la.shape[0] > 64:   # a few far entries in a few rows: dense-row outliers or wider windows
        for _ in range(int(rng.integers(1, 4))):
            j, i = int(rng.integers(la.shape[0])), int(rng.integers(la.shape[0]))
            la[j, i] = np.float32(-1.0 - rng.integers(0, 2000) / 256.0)
    return f"{fam}{la.shape[0]}_d{dmax}", la, lp


t0 = time.time()
n_cases = n_runs = 0
fails = []
while time.time() - t0 < budget:
    name, la, lp = random_matrix()
    S = la.shape[0]
    try:
        dec = ViterbiDecoder(la, lp, dev)
    except Exception as e:   # noqa: BLE001
        fails.append((name, "plan", repr(e)))
        print("FAIL plan", name, e, flush=True)
        continue
    for _ in range(3):
        B = int(rng.choice([1, 2, 3, 5, 9, 17]))
        T = int(rng.choice([1, 2, 3, 17, 64, 65, 129, 300, 601, 1100]))
        kind = str(rng.choice(list(GEN) if S >= 8 else ["dense", "ties"]))
        f16 = bool(rng.integers(2))
        E = GEN[kind](B, T, S, seed=int(rng.integers(1 << 20)), device=dev, dtype=torch.float16 if f16 else torch.float32)
        lens_np = np.where(rng.random(B) < 0.5, T, rng.integers(1, T + 1, B)).astype(np.int64)
        lens = torch.from_numpy(lens_np).to(dev)
        ref_s, ref_l = vo.decode_c(la, lp, E.float().cpu().numpy(), lengths=lens_np)
        forms = [("auto", 0)]
        if dec.info["banded_ok"]:
            forms += [("group", 0), ("group", 2)]
            if dec.info["n_dense_rows"] == 0:
                forms += [("group", 4)]           # one (song, chunk) stream per lane
            if dec.info["wave_ok"]:
                forms += [("wave", 0), ("wave", 2), ("wave", 4), ("wave-half", 0), ("checkpointed", 0), ("packed", 0)]
        if S <= 400 or rng.random() < 0.3:
            forms += [("dense", 0)]
        chunks = int(rng.choice([0, 0, 2, 7, 32]))
        lane_chunks = int(rng.choice([0, 1, 3, 64, 100, 256]))
        warm = int(rng.choice([0, 1, 16, 64]))
        n_cases += 1
        for algo, btf in forms:
            dec.set_option("backtrace_form", btf)
            dec.set_option("bt_fast_rows", n_cases % 3 == 0)      # every third case: all rows through the sparse kernels' general code
            if btf == 4:
                dec.set_option("bt_chunks", lane_chunks)
                dec.set_option("bt_warm", warm)
            elif chunks:
                dec.set_option("bt_chunks", chunks)
                dec.set_option("bt_warm", warm)
            try:
                if algo == "wave-half":          # even delta rows only; refused (loudly) where the plan does not allow it
                    dec.set_option("wave_history", 2)
                    st, ll = dec.decode(E, lengths=lens, algo="wave", out_dtype=torch.int32)
                elif algo == "packed":           # the songs' valid frames in one buffer, no padding
                    off = np.zeros(B + 1, np.int64)
                    off[1:] = np.cumsum(lens_np)
                    Ep = torch.cat([E[b, :int(lens_np[b])] for b in range(B)], dim=0).contiguous()
                    sp, ll = dec.decode_packed(Ep, off, out_dtype=torch.int32)
                    st = torch.full((B, T), -1, dtype=torch.int32, device=dev)
                    for b in range(B):
                        st[b, :int(lens_np[b])] = sp[off[b]:off[b + 1]]
                elif algo == "checkpointed":
                    st, ll = dec.decode_checkpointed(E, segment_frames=int(rng.choice([64, 65, 100, 256, 2048])), lengths=lens, out_dtype=torch.int32)
                else:
                    st, ll = dec.decode(E, lengths=lens, algo=algo, out_dtype=torch.int32)
            except Exception as e:   # noqa: BLE001  a forced family may be refused (loudly); "auto" must always decode
                dec.set_option("reset", 0)
                print("refused" if algo != "auto" else "FAIL", name, {k: dec.info[k] for k in ("banded_ok", "wave_ok", "group_window", "n_extras", "n_dense_rows")}, B, T, algo, btf, e, flush=True)
                if algo == "auto":
                    fails.append((name, "auto refused"))
                continue
            dec.set_option("reset", 0)
            n_runs += 1
            st = st.cpu().numpy()
            okp = all(np.array_equal(st[b, :lens_np[b]], ref_s[b, :lens_np[b]]) for b in range(B))
            okl = np.array_equal(ll.cpu().numpy(), ref_l)
            if not (okp and okl):
                fails.append((name, dec.info, B, T, kind, f16, algo, btf, chunks, warm, okp, okl))
                print("FAIL", name, B, T, kind, f16, algo, btf, chunks, warm, "paths", okp, "loglik", okl, flush=True)
    if n_cases % 30 == 0:
        print(f"{n_cases} cases, {n_runs} decodes, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {n_runs} decodes, {len(fails)} failures")
sys.exit(1 if fails else 0)
