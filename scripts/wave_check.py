"""Wave-form forward kernel: parity against the CPU oracle on small ragged batches, then forward / back-trace timing
at large batch sizes (test infrastructure; run on the GPU box).  argv: list of batch sizes to time (default 1024 2048)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import viterbi_oracle as vo  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
p = np.load(os.path.join(ROOT, "tests", "golden", "params.npz"))
ok = True
for name in ("tonet361", "msnet321"):
    A, pi = p[f"{name}_logA_T"], p[f"{name}_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    assert dec.info["wave_ok"], dec.info
    S = dec.S
    for kind, gen in (("peaks", synth.emissions_peaks), ("dense", synth.emissions_dense), ("ties", synth.emissions_ties)):
        for f16 in (False, True):
            for two in (0, 1):
                E = gen(9, 301, S, seed=5, device=dev, dtype=torch.float16 if f16 else torch.float32)
                lens = torch.tensor([301, 1, 2, 150, 300, 3, 64, 65, 66], dtype=torch.int64, device=dev)
                ref_s, ref_l = vo.decode_c(A, pi, E.float().cpu().numpy(), lengths=lens.cpu().numpy())
                dec.set_option("wave_two", two)
                st, ll = dec.decode(E, lengths=lens, algo="wave", out_dtype=torch.int32)
                good = np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l)
                ok &= good
                print(name, kind, "f16" if f16 else "f32", "two" if two else "one", "OK" if good else "MISMATCH", flush=True)
                if not good:
                    bad = np.argwhere(st.cpu().numpy() != ref_s)
                    print("  first mismatches (song, frame):", bad[:5].tolist(), "loglik", ll.cpu().numpy()[:4], ref_l[:4])
    dec.set_option("reset", 0)
print("PARITY", "PASS" if ok else "FAIL", flush=True)
if not ok:
    sys.exit(1)

A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
T = 30000
for B in [int(x) for x in (sys.argv[1:] or ["1024", "2048"])]:
    E = synth.emissions_peaks(B, T, 361, seed=1234, device=dev)
    st = torch.empty((B, T), dtype=torch.int32, device=dev)
    ll = torch.empty((B,), dtype=torch.float32, device=dev)
    for algo, two in (("wave", 0), ("wave", 1), ("group", 0)):
        dec.set_option("wave_two", two)
        dec.decode_into(E, st, ll, algo=algo)      # warm
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        n = 3
        tf = tb = 0.0
        for _ in range(n):
            ev[0].record()
            dec.decode_into(E, st, ll, algo=algo, phase="forward")
            ev[1].record()
            dec.decode_into(E, st, ll, algo=algo, phase="backtrace")
            ev[2].record()
            torch.cuda.synchronize()
            tf += ev[0].elapsed_time(ev[1])
            tb += ev[1].elapsed_time(ev[2])
        tf /= n
        tb /= n
        print(f"B {B} algo {algo} two {two}: fwd {tf:.2f} ms  bt {tb:.2f} ms  -> fwd {B*T/tf/1e3:.0f} Mframes/s, whole {B*T/(tf+tb)/1e3:.0f} Mframes/s", flush=True)
        if algo == "wave" and two == 0:
            ref = st.clone()
        else:
            print("   same paths as wave:", bool(torch.equal(ref, st)), flush=True)
    # spot check against the oracle
    sub = [0, B // 2, B - 1]
    rs, rl = vo.decode_c(A, pi, E[sub].cpu().numpy())
    print("   oracle spot check:", np.array_equal(ref[sub].cpu().numpy(), rs), flush=True)
    del E, st, ll
    torch.cuda.empty_cache()
