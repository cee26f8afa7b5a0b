"""Parity of a banded forward form (argv[1]: option forward_form, 1 .. 5) against the CPU oracle (test infrastructure)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viterbi_spl_amd import ViterbiDecoder, synth
from oracle import viterbi_oracle as vo
dev = torch.device("cuda:0")
logA_T, log_pi = synth.log_params(synth.tonet_transition(360, 14), synth.floored_prior(361))
dec = ViterbiDecoder(logA_T, log_pi, dev)
dec.set_option("forward_form", int(sys.argv[1]))
ok = True
for (B, T, kind) in ((3, 400, "peaks"), (8, 2000, "ties"), (16, 5000, "dense"), (128, 3000, "peaks")):
    gen = {"peaks": synth.emissions_peaks, "ties": synth.emissions_ties, "dense": synth.emissions_dense}[kind]
    E = gen(B, T, 361, seed=5, device=dev)
    lens = torch.randint(1, T + 1, (B,), device=dev, dtype=torch.int64)
    st, ll = dec.decode(E, lengths=lens, algo="banded", out_dtype=torch.int32)
    torch.cuda.synchronize()
    ref_s, ref_l = vo.decode_c(logA_T, log_pi, E.cpu().numpy(), lengths=lens.cpu().numpy())
    e = np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l)
    ok = ok and e
    print(B, T, kind, "bit-exact", e)
sys.exit(0 if ok else 1)
