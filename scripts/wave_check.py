"""Wave-form forward kernel + sparse back-trace: parity against the CPU oracle on small ragged batches (every forward form
x back-trace form, forced bad guesses), then forward / back-trace timing at large batch sizes (test infrastructure; run
on the GPU box).  argv: batch sizes to time (default 1024 2048); "peaks" / "dense" selects the emission kind."""
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
            E = gen(9, 701, S, seed=5, device=dev, dtype=torch.float16 if f16 else torch.float32)
            lens = torch.tensor([701, 1, 2, 150, 700, 3, 64, 65, 66], dtype=torch.int64, device=dev)
            ref_s, ref_l = vo.decode_c(A, pi, E.float().cpu().numpy(), lengths=lens.cpu().numpy())
            for algo in ("wave", "group"):
                for btf in (0, 2):
                    for chunks, warm in ((0, -1), (7, 0), (32, 1)):
                        dec.set_option("reset", 0)
                        dec.set_option("backtrace_form", btf)
                        dec.set_option("bt_chunks", chunks)
                        dec.set_option("bt_warm", warm)
                        st, ll = dec.decode(E, lengths=lens, algo=algo, out_dtype=torch.int32)
                        good = np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l)
                        ok &= good
                        if not good:
                            bad = np.argwhere(st.cpu().numpy() != ref_s)
                            print(name, kind, "f16" if f16 else "f32", algo, "bt", btf, chunks, warm, "MISMATCH", bad[:4].tolist(), flush=True)
        print(name, kind, "checked", flush=True)
    dec.set_option("reset", 0)
print("PARITY", "PASS" if ok else "FAIL", flush=True)
if not ok:
    sys.exit(1)

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
        for algo, btf in (("wave", 0), ("wave", 2), ("group", 0), ("group", 2)):
            dec.set_option("reset", 0)
            dec.set_option("backtrace_form", btf)
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
            same = "" if ref is None else f"  same paths: {bool(torch.equal(ref, st))}"
            print(f"{kind} B {B} fwd {algo} bt {btf}: fwd {tf:.2f} ms  bt {tb:.2f} ms  -> fwd {B*T/tf/1e3:.0f} Mframes/s, whole {B*T/(tf+tb)/1e3:.0f} Mframes/s{same}", flush=True)
            if ref is None:
                ref = st.clone()
        sub = [0, min(B, NU) // 2, min(B, NU) - 1]
        rs, rl = vo.decode_c(A, pi, E[sub].cpu().numpy())
        print("   oracle spot check:", np.array_equal(ref[sub].cpu().numpy(), rs), flush=True)
        del E, st, ll, ref
        torch.cuda.empty_cache()
    del base
