"""Lane form of the back-trace (backtrace_lane.hip: one (song, chunk) stream per lane): parity against the CPU oracle on small ragged
batches (the reference's matrices, every emission kind and storage type, chunk counts up to 256, forced bad guesses), then timing
against the one-stream-per-wavefront kernel at large batch sizes (test infrastructure; run on the GPU box).
argv: batch sizes to time (default 128 1024 2048); "notime" skips the timing; "chunks" sweeps the chunk count."""
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
ok = True
for name in ("tonet361", "msnet321", "jdc722", "imm722w", "jdc721"):
    A, pi = p[f"{name}_logA_T"], p[f"{name}_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    assert dec.info["banded_ok"], dec.info
    S = dec.S
    small = S > 400
    for kind, gen in (("peaks", synth.emissions_peaks), ("dense", synth.emissions_dense), ("ties", synth.emissions_ties)):
        for f16 in (False, True):
            if small and f16 and kind != "peaks":
                continue
            for Tm in ((701,) if small else (701, 1702)):
                E = gen(11, Tm, S, seed=5, device=dev, dtype=torch.float16 if f16 else torch.float32)
                lens = torch.tensor([Tm, 1, 2, 150, Tm - 1, 3, 64, 65, 66, 4, 5], dtype=torch.int64, device=dev)
                ref_s, ref_l = vo.decode_c(A, pi, E.float().cpu().numpy(), lengths=lens.cpu().numpy())
                algos = ("group",) if not dec.info["wave_ok"] else ("wave", "group")
                for algo in algos:
                    for chunks, warm in ((0, -1), (7, 0), (32, 1), (1, -1), (5, 33), (2, 7), (64, 0), (200, 3), (256, -1), (255, 0)):
                        for use_len in (True, False):
                            dec.set_option("reset", 0)
                            dec.set_option("backtrace_form", 4)
                            dec.set_option("bt_chunks", chunks)
                            dec.set_option("bt_warm", warm)
                            st, ll = dec.decode(E, lengths=lens if use_len else None, algo=algo, out_dtype=torch.int32)
                            if use_len:
                                good = np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l)
                            else:     # the full-length song 0 only
                                good = np.array_equal(st[0].cpu().numpy(), ref_s[0]) and float(ll[0]) == float(ref_l[0])
                            ok &= good
                            if not good:
                                bad = np.argwhere(st.cpu().numpy()[: (11 if use_len else 1)] != ref_s[: (11 if use_len else 1)])
                                print(name, kind, "f16" if f16 else "f32", Tm, algo, chunks, warm, use_len, "MISMATCH", len(bad), bad[:6].tolist(), flush=True)
        print(name, kind, "checked", dec.backtrace_counters(11, Tm), flush=True)
    dec.set_option("reset", 0)
    del dec
print("PARITY", "PASS" if ok else "FAIL", flush=True)
if not ok or "notime" in sys.argv:
    sys.exit(0 if ok else 1)

A, pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(A, pi, dev)
T = 30000
NU = 32


def time_bt(E, st, ll, algo, n=5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    dec.decode_into(E, st, ll, algo=algo, phase="backtrace")
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(n):
        dec.decode_into(E, st, ll, algo=algo, phase="backtrace")
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


for kind in ("peaks", "dense"):
    gen = synth.emissions_peaks if kind == "peaks" else synth.emissions_dense
    base = gen(NU, T, 361, seed=1234, device=dev)
    for B in [int(x) for x in (args or ["128", "1024", "2048"])]:
        E = base.repeat((B + NU - 1) // NU, 1, 1)[:B].contiguous() if B > NU else base[:B].contiguous()
        st = torch.empty((B, T), dtype=torch.int32, device=dev)
        ll = torch.empty((B,), dtype=torch.float32, device=dev)
        algo = "banded"
        dec.set_option("reset", 0)
        dec.decode_into(E, st, ll, algo=algo)
        torch.cuda.synchronize()
        ref = st.clone()
        t_sparse = time_bt(E, st, ll, algo)
        ct = dec.backtrace_counters(B, T)
        print(f"{kind} B {B}: one stream per wavefront {t_sparse:.3f} ms   per 1000 frames: { {k: round(v * 1000.0 / (B * T), 3) for k, v in ct.items()} }", flush=True)
        sweeps = ((0, -1),) if "chunks" not in sys.argv else ((0, -1), (32, -1), (64, -1), (128, -1), (234, -1), (256, 32), (128, 32), (128, 128), (64, 128))
        for chunks, warm in sweeps:
            dec.set_option("reset", 0)
            dec.set_option("backtrace_form", 4)
            dec.set_option("bt_chunks", chunks)
            dec.set_option("bt_warm", warm)
            st.fill_(-7)
            t_lane = time_bt(E, st, ll, algo)
            ct = dec.backtrace_counters(B, T)
            print(f"{kind} B {B}: one stream per lane, chunks {chunks} warm {warm}: {t_lane:.3f} ms  same paths: {bool(torch.equal(ref, st))}   per 1000 frames: "
                  f"{ {k: round(v * 1000.0 / (B * T), 3) for k, v in ct.items()} }", flush=True)
        sub = [0, min(B, NU) // 2, min(B, NU) - 1]
        rs, rl = vo.decode_c(A, pi, E[sub].cpu().numpy())
        print("   oracle spot check:", np.array_equal(st[sub].cpu().numpy(), rs), flush=True)
        del E, st, ll, ref
        dec._ws = None
        torch.cuda.empty_cache()
    del base
