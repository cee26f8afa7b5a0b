"""Race hunt (test infrastructure): large batches decoded several times must give identical bytes, and the structured
kernels must agree with the independent plain dense kernel on every song."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import ViterbiDecoder, synth
dev = torch.device("cuda:0")
ok = True
def run(name, logA_T, log_pi, B, T, f16, kind="dense"):
    global ok
    S = logA_T.shape[0]
    dec = ViterbiDecoder(logA_T, log_pi, dev)
    gen = {"peaks": synth.emissions_peaks, "dense": synth.emissions_dense}[kind]
    E = gen(B, T, S, seed=99, device=dev, dtype=torch.float16 if f16 else torch.float32)
    lens = torch.randint(T // 2, T + 1, (B,), device=dev, dtype=torch.int64)
    ref = None
    for rep in range(4):
        st, ll = dec.decode(E, lengths=lens, algo="auto", out_dtype=torch.int32)
        cur = (st.cpu().numpy().tobytes(), ll.cpu().numpy().tobytes())
        if ref is None: ref = cur
        elif cur != ref: ok = False; print(name, "run", rep, "DIFFERS from run 0")
    sd, ld = dec.decode(E, lengths=lens, algo="dense", out_dtype=torch.int32)
    same = (sd.cpu().numpy().tobytes(), ld.cpu().numpy().tobytes()) == ref
    ok = ok and same
    print(name, dec.info["group_window"], "repeatable, equals dense kernel:", same)
A = synth.durrieu_transition(721, 20)
run("durrieu722", np.require(np.log(A).astype(np.float32).T, np.float32, ["C"]), np.log(np.full(722, 1.0 / 722)).astype(np.float32), 300, 4000, True)
for dm in (14, 40, 56):
    la, lp = synth.log_params(synth.tonet_transition(721, dm), synth.floored_prior(722))
    run(f"band722_dmax{dm}", la, lp, 300, 4000, True, "peaks")
la, lp = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
run("tonet361_B600", la, lp, 600, 6000, False, "peaks")
sys.exit(0 if ok else 1)
