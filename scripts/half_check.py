"""Half-history wave form (wave.hip HM 1 + backtrace_half.hip): parity against the CPU oracle on small ragged batches (every
emission kind, storage type, chunking, forced bad guesses), then forward / back-trace timing against the full history at
large batch sizes (test infrastructure; run on the GPU box).  argv: batch sizes to time (default 1024 2048); "peaks" /
"dense" selects the emission kind; "notime" skips the timing."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import viterbi_oracle as vo  # noqa: E402
from viterbi_spl_amd import ViterbiDecoder, synth  # noqa: E402

dev = torch.device("cuda:0")
p = np.load(os.path.join(ROOT, "tests", "golden", "params.npz"))
args = [x for x in sys.argv[1:] if x.isdigit()]
kinds = [x for x in sys.argv[1:] if x in ("peaks", "dense")] or ["peaks"]
ok = True
for name in ("tonet361", "msnet321"):
    A, pi = p[f"{name}_logA_T"], p[f"{name}_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    assert dec.info["wave_ok"], dec.info
    S = dec.S
    for kind, gen in (("peaks", synth.emissions_peaks), ("dense", synth.emissions_dense), ("ties", synth.emissions_ties)):
        for f16 in (False, True):
            for Tm in (701, 702):
                E = gen(11, Tm, S, seed=5, device=dev, dtype=torch.float16 if f16 else torch.float32)
                lens = torch.tensor([Tm, 1, 2, 150, Tm - 1, 3, 64, 65, 66, 4, 5], dtype=torch.int64, device=dev)
                ref_s, ref_l = vo.decode_c(A, pi, E.float().cpu().numpy(), lengths=lens.cpu().numpy())
                for algo, hist, btf in (("wave", 2, 0), ("wave", 1, 0), ("group", 1, 0)):
                    for chunks, warm in ((0, -1), (7, 0), (32, 1), (1, -1), (5, 33), (2, 7)):
                        dec.set_option("reset", 0)
                        dec.set_option("wave_history", hist)
                        dec.set_option("backtrace_form", btf)
                        dec.set_option("bt_chunks", chunks)
                        dec.set_option("bt_warm", warm)
                        st, ll = dec.decode(E, lengths=lens, algo=algo, out_dtype=torch.int32)
                        good = np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l)
                        ok &= good
                        if not good:
                            bad = np.argwhere(st.cpu().numpy() != ref_s)
                            print(name, kind, "f16" if f16 else "f32", Tm, algo, "hist", hist, "bt", btf, chunks, warm, "MISMATCH", len(bad), bad[:6].tolist(), flush=True)
        print(name, kind, "checked", flush=True)
    dec.set_option("reset", 0)
print("PARITY", "PASS" if ok else "FAIL", flush=True)
if not ok or "notime" in sys.argv:
    sys.exit(0 if ok else 1)

A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
T = 30000
NU = 32
for kind in kinds:
    gen = synth.emissions_peaks if kind == "peaks" else synth.emissions_dense
    base = gen(NU, T, 361, seed=1234, device=dev)
    for B in [int(x) for x in (args or ["1024", "2048"])]:
        E = base.repeat((B + NU - 1) // NU, 1, 1)[:B].contiguous() if B > NU else base[:B].contiguous()
        st = torch.empty((B, T), dtype=torch.int32, device=dev)
        ll = torch.empty((B,), dtype=torch.float32, device=dev)
        ref = None
        for hist in (1, 2):
            dec.set_option("reset", 0)
            dec.set_option("wave_history", hist)
            dec._ws = None
            torch.cuda.empty_cache()
            dec.decode_into(E, st, ll, algo="wave")      # warm
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            n = 5
            tf = tb = 0.0
            for _ in range(n):
                ev[0].record()
                dec.decode_into(E, st, ll, algo="wave", phase="forward")
                ev[1].record()
                dec.decode_into(E, st, ll, algo="wave", phase="backtrace")
                ev[2].record()
                torch.cuda.synchronize()
                tf += ev[0].elapsed_time(ev[1])
                tb += ev[1].elapsed_time(ev[2])
            tf /= n
            tb /= n
            same = "" if ref is None else f"  same paths: {bool(torch.equal(ref, st))}"
            fr = B * T * 2166 / (tf * 1e-3) / 8e12
            wr = B * T * 2172 / ((tf + tb) * 1e-3) / 8e12
            print(f"{kind} B {B} history {'full' if hist == 1 else 'half'} : fwd {tf:.2f} ms  bt {tb:.2f} ms  -> fwd {B*T/tf/1e3:.0f} Mframes/s ({fr:.3f} of the roofline), "
                  f"whole {B*T/(tf+tb)/1e3:.0f} Mframes/s ({wr:.3f}){same}", flush=True)
            ct = dec.backtrace_counters(B, T)
            print("      per 1000 frames:", {k: round(v * 1000.0 / (B * T), 3) for k, v in ct.items()}, flush=True)
            if ref is None:
                ref = st.clone()
        sub = [0, min(B, NU) // 2, min(B, NU) - 1]
        rs, rl = vo.decode_c(A, pi, E[sub].cpu().numpy())
        print("   oracle spot check:", np.array_equal(ref[sub].cpu().numpy(), rs), flush=True)
        del E, st, ll, ref
        dec._ws = None
        torch.cuda.empty_cache()
    del base
