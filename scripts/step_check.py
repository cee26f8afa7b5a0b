"""Parity + timing of the step-structured kernel (Durrieu matrix) against the CPU oracle and the plain dense kernel."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import ViterbiDecoder, synth
from oracle import viterbi_oracle as vo
dev = torch.device("cuda:0")
A = synth.durrieu_transition(721, 20)
logA_T = np.require(np.log(A).astype(np.float32).T, np.float32, ["C"])
log_pi = np.log(np.full(722, 1.0 / 722)).astype(np.float32)
dec = ViterbiDecoder(logA_T, log_pi, dev)
print(dec.info)
ok = True
for (B, T, kind, f16) in ((3, 300, "dense", False), (5, 500, "peaks", False), (4, 400, "ties", True), (16, 2000, "dense", True)):
    gen = {"peaks": synth.emissions_peaks, "ties": synth.emissions_ties, "dense": synth.emissions_dense}[kind]
    E = gen(B, T, 722, seed=7, device=dev, dtype=torch.float16 if f16 else torch.float32)
    lens = torch.randint(1, T + 1, (B,), device=dev, dtype=torch.int64)
    st, ll = dec.decode(E, lengths=lens, algo="auto", out_dtype=torch.int32)
    torch.cuda.synchronize()
    ref_s, ref_l = vo.decode_c(logA_T, log_pi, E.float().cpu().numpy(), lengths=lens.cpu().numpy())
    e = np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l)
    ok = ok and e
    print(B, T, kind, f16, "bit-exact", e)
if ok and len(sys.argv) > 1:
    B, T = 256, 3000
    E = synth.emissions_dense(B, T, 722, seed=1, device=dev, dtype=torch.float16)
    st = torch.empty((B, T), dtype=torch.int32, device=dev); ll = torch.empty(B, device=dev)
    for algo in ("auto", "dense"):
        for _ in range(2):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record(); dec.decode_into(E, st, ll, algo=algo, phase="forward"); ev[1].record(); dec.decode_into(E, st, ll, algo=algo, phase="backtrace"); ev[2].record()
            torch.cuda.synchronize()
        print(algo, "fwd_ms", round(ev[0].elapsed_time(ev[1]), 2), "bt_ms", round(ev[1].elapsed_time(ev[2]), 2))
sys.exit(0 if ok else 1)
