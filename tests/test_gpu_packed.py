"""GPU parity tests of round 4, all through the C ABI: the packed (ragged) decode, the lane form of the back-trace, the checked
back-trace phase.  Bar: states bit-exact, log-likelihood bit-equal against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import viterbi_oracle as vo
from viterbi_spl_amd import ViterbiDecoder, _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    _lib.load()
    return torch.device("cuda:0")


def _pack(E, lens):
    """[B, T, S] + lengths -> packed [sum T_b, S], offsets."""
    off = np.zeros(len(lens) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    return torch.cat([E[b, :int(n)] for b, n in enumerate(lens)], dim=0).contiguous(), off


def _unpack(states, off, T):
    out = np.full((len(off) - 1, T), -1, np.int32)
    for b in range(len(off) - 1):
        out[b, :off[b + 1] - off[b]] = states[off[b]:off[b + 1]]
    return out


@pytest.mark.parametrize("name", ["tonet361", "msnet321"])
def test_packed_decode_small(golden, dev, name):
    """vit_decode_packed on a handful of songs (lengths 1, 2, odd, even, equal), fp32 and fp16 storage, every emission kind: the
    states and log-likelihoods of the padded decode, i.e. of the oracle run on every song alone."""
    A, pi = golden["params"][f"{name}_logA_T"], golden["params"][f"{name}_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    assert dec.info["wave_ok"]
    S, T = dec.S, 333
    lens = np.array([T, 1, 2, 150, T - 1, 3, 64, 65, 66, 4, 5, T, 129, 1, 77], np.int64)
    for kind, gen in (("peaks", synth.emissions_peaks), ("dense", synth.emissions_dense), ("ties", synth.emissions_ties)):
        for dt in (torch.float32, torch.float16):
            E = gen(len(lens), T, S, seed=9, device=dev, dtype=dt)
            ref_s, ref_l = vo.decode_c(A, pi, E.float().cpu().numpy(), lengths=lens)
            Ep, off = _pack(E, lens)
            st, ll = dec.decode_packed(Ep, off, out_dtype=torch.int32)
            assert np.array_equal(_unpack(st.cpu().numpy(), off, T), ref_s), (name, kind, dt)
            assert np.array_equal(ll.cpu().numpy(), ref_l), (name, kind, dt)
    with pytest.raises(ValueError):
        dec.decode_packed(Ep, off[:-1])                      # offsets must end at the number of rows
    with pytest.raises(ValueError):
        dec.decode_packed(Ep, np.concatenate([off[:3], off[2:]]))   # an empty song


def test_packed_decode_needs_the_wave_form(golden, dev):
    dec = ViterbiDecoder(golden["params"]["dense97_logA_T"], golden["params"]["dense97_log_pi"], dev)
    E = synth.emissions_dense(1, 10, 97, seed=1, device=dev)[0]
    with pytest.raises(_lib.ViterbiHipError):
        dec.decode_packed(E, [0, 10])


def test_packed_decode_3000_ragged_songs(golden, dev):
    """3072 recordings with lengths uniform in [T/4, T] (and a few of one or two frames) in one packed buffer: more songs than
    forward slots (8 per compute unit), so every wavefront walks several songs back to back; every state and log-likelihood
    equals the oracle's decode of that song alone."""
    A, pi = golden["params"]["tonet361_logA_T"], golden["params"]["tonet361_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    S, T, B, NU = 361, 1200, 3072, 96
    rng = np.random.default_rng(31)
    lens = rng.integers(T // 4, T + 1, B).astype(np.int64)
    lens[[5, 777, 2048, B - 1]] = (1, 2, 1, T)
    base = synth.emissions_peaks(NU, T, S, seed=77, device=dev)                 # songs repeat with period 96, lengths do not
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum(lens)
    Ep = torch.empty((int(off[-1]), S), dtype=torch.float32, device=dev)
    for b in range(B):
        Ep[off[b]:off[b + 1]] = base[b % NU, :lens[b]]
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    assert B > 8 * n_cus or n_cus > 384, "the test wants more songs than slots"
    st, ll = dec.decode_packed(Ep, off, out_dtype=torch.int32)
    st, ll = st.cpu().numpy(), ll.cpu().numpy()
    base_h = base.cpu().numpy()
    for u in range(NU):                                                          # the oracle, once per distinct (song, length)
        idx = np.arange(u, B, NU)
        Eu = np.broadcast_to(base_h[u], (len(idx), T, S))
        rs, rl = vo.decode_c(A, pi, np.ascontiguousarray(Eu), lengths=lens[idx])
        for k, b in enumerate(idx):
            assert np.array_equal(st[off[b]:off[b + 1]], rs[k, :lens[b]]), (b, int(lens[b]))
            assert ll[b] == rl[k], b
    st2, ll2 = dec.decode_packed(Ep, off, out_dtype=torch.int32)                 # a second run returns identical bytes
    assert np.array_equal(st2.cpu().numpy(), st) and np.array_equal(ll2.cpu().numpy(), ll)


@pytest.mark.parametrize("name", ["tonet361", "msnet321", "jdc722", "imm722w"])
def test_lane_form_of_the_backtrace(golden, dev, name):
    """backtrace_form 4: one (song, chunk) stream per lane (backtrace_lane.hip), up to 256 chunks per song, forced bad guesses
    (zero warm-up: the verify / repair passes decide), ragged lengths down to one frame, behind both forward forms."""
    A, pi = golden["params"][f"{name}_logA_T"], golden["params"][f"{name}_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    S, T = dec.S, 701
    lens = torch.tensor([T, 1, 2, 150, T - 1, 3, 64, 65, 66, 4, 5], dtype=torch.int64, device=dev)
    for kind, gen in (("peaks", synth.emissions_peaks), ("dense", synth.emissions_dense), ("ties", synth.emissions_ties)):
        for dt in ((torch.float32, torch.float16) if S < 400 else (torch.float32,)):
            E = gen(11, T, S, seed=5, device=dev, dtype=dt)
            ref_s, ref_l = vo.decode_c(A, pi, E.float().cpu().numpy(), lengths=lens.cpu().numpy())
            for algo in (("wave", "group") if dec.info["wave_ok"] else ("group",)):
                for chunks, warm in ((0, -1), (7, 0), (1, -1), (64, 0), (200, 3), (256, -1)):
                    dec.set_option("reset", 0)
                    dec.set_option("backtrace_form", 4)
                    dec.set_option("bt_chunks", chunks)
                    dec.set_option("bt_warm", warm)
                    st, ll = dec.decode(E, lengths=lens, algo=algo, out_dtype=torch.int32)
                    assert np.array_equal(st.cpu().numpy(), ref_s), (name, kind, dt, algo, chunks, warm)
                    assert np.array_equal(ll.cpu().numpy(), ref_l), (name, kind, dt, algo, chunks, warm)
    dec.set_option("reset", 0)
    dense = ViterbiDecoder(golden["params"]["dense97_logA_T"], golden["params"]["dense97_log_pi"], dev)
    dense.set_option("backtrace_form", 4)
    with pytest.raises(_lib.ViterbiHipError):                                    # unstructured matrix: refused loudly, no silent other kernel
        dense.decode(synth.emissions_dense(1, 20, 97, seed=1, device=dev))


def test_backtrace_phase_refuses_another_emission_tensor(golden, dev):
    """The split form's lifetime rule: the back-trace phase hands the emission pointer to the library
    (vit_backtrace_checked), which refuses a tensor other than the one the forward pass decoded -- with the half history it
    would read the odd frames' emissions from it."""
    A, pi = golden["params"]["tonet361_logA_T"], golden["params"]["tonet361_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    E = synth.emissions_peaks(3, 120, 361, seed=2, device=dev)
    E2 = E.clone()
    st = torch.empty((3, 120), dtype=torch.int32, device=dev)
    ref_s, _ = vo.decode_c(A, pi, E.cpu().numpy())
    for hist in (1, 2):
        dec.set_option("wave_history", hist)
        dec.decode_into(E, st, None, algo="wave", phase="forward")
        with pytest.raises(_lib.ViterbiHipError, match="invalid argument"):
            dec.decode_into(E2, st, None, algo="wave", phase="backtrace")        # same values, another buffer
        with pytest.raises(_lib.ViterbiHipError, match="invalid argument"):
            dec.decode_into(E.half(), st, None, algo="wave", phase="backtrace")  # another storage type
        dec.decode_into(E, st, None, algo="wave", phase="backtrace")
        assert np.array_equal(st.cpu().numpy(), ref_s), hist
    dec.set_option("reset", 0)


def test_forward_family_query(golden, dev):
    """vit_forward_family: what bench.py asks instead of guessing the library's batch-size thresholds."""
    dec = ViterbiDecoder(golden["params"]["tonet361_logA_T"], golden["params"]["tonet361_log_pi"], dev)
    n_cus = torch.cuda.get_device_properties(dev).multi_processor_count
    assert dec.forward_family(1, "banded") == "group" and dec.forward_family(2 * n_cus, "banded") == "group"
    assert dec.forward_family(2 * n_cus + 1, "banded") == "wave" and dec.forward_family(1, "wave") == "wave"
    assert dec.forward_family(4096, "dense") == "dense"
    dense = ViterbiDecoder(golden["params"]["dense97_logA_T"], golden["params"]["dense97_log_pi"], dev)
    assert dense.forward_family(64, "auto") == "dense"
    with pytest.raises(_lib.ViterbiHipError):
        dense.forward_family(64, "wave")


def test_workspace_budget_policy(golden, dev):
    """ViterbiDecoder.decode(max_workspace_bytes=...): full history -> half history -> checkpointed decode as the budget shrinks,
    the same states and log-likelihoods in every mode, a loud error when nothing fits."""
    A, pi = golden["params"]["tonet361_logA_T"], golden["params"]["tonet361_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    B, T = 9, 4100
    E = synth.emissions_peaks(B, T, 361, seed=3, device=dev)
    lens = torch.tensor([T, 1, 2, 1500, T - 1, 3, 2049, 65, 4096], dtype=torch.int64, device=dev)
    ref_s, ref_l = vo.decode_c(A, pi, E.cpu().numpy(), lengths=lens.cpu().numpy())
    full = dec.workspace_bytes(B, T, "auto")
    seen = []
    for budget in (None, full, full - 1, full // 2 + 4096, full // 3, full // 8, full // 20):
        mode = dec.plan_workspace(B, T, "auto", budget)
        seen.append(mode["mode"])
        assert budget is None or mode["workspace_bytes"] <= budget
        st, ll = dec.decode(E, lengths=lens, out_dtype=torch.int32, max_workspace_bytes=budget)
        assert np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l), (budget, mode)
    assert seen[:2] == ["full", "full"] and "half" in seen and seen[-1] == "checkpointed", seen
    assert dec._options.get("wave_history", 0) == 0                           # the policy leaves the plan's options as it found them
    with pytest.raises(_lib.ViterbiHipError, match="fits a workspace"):
        dec.decode(E, max_workspace_bytes=100000)
    dense = ViterbiDecoder(golden["params"]["dense97_logA_T"], golden["params"]["dense97_log_pi"], dev)
    with pytest.raises(_lib.ViterbiHipError, match="fits a workspace"):      # no wave form: nothing to fall back to
        dense.decode(synth.emissions_dense(4, 300, 97, seed=1, device=dev), max_workspace_bytes=1000)


def test_family_b_and_c_at_full_length(dev):
    """The S = 361 production entry points themselves at T = 30000: Viterbi.viterbi_librosa_fn on F-order
    [S, T] probabilities and SoftMaxViterbi.viterbi_librosa_fn on C-order [T, S] probabilities, against the states the reference's
    own methods (tonet/for_paper.py:1833-1870, :1999-2037; AST-extracted by tests/golden/make_familyB_golden.py) returned."""
    import hashlib
    import json
    import os
    from tests.golden import make_familyB_golden as mk
    from viterbi_spl_amd import reference_api as ra
    gdir = os.path.dirname(mk.__file__)
    man = json.load(open(os.path.join(gdir, "familyB_manifest.json")))
    gold = np.load(os.path.join(gdir, "familyB_goldens.npz"))
    A, pi = synth.tonet_transition(360, 14), synth.floored_prior(361)
    assert mk.sha(A, pi) == man["sha256_params"]
    P = mk.inputs(man["seed"], man["T"])
    assert P.flags["F_CONTIGUOUS"] and hashlib.sha256(np.ascontiguousarray(P).tobytes()).hexdigest() == man["sha256_probs_st"]
    want = gold["states_B"].astype(np.int64)
    got = ra.Viterbi(A, pi, device=dev).viterbi_librosa_fn(P.copy(order="F"))
    assert got.dtype == np.int64 and got.shape == (man["T"],) and np.array_equal(got, want)
    got = ra.SoftMaxViterbi(A, pi, device=dev).viterbi_librosa_fn(np.array(P.T, order="C"))
    assert np.array_equal(got, gold["states_C"].astype(np.int64))


def test_packed_decode_error_paths(golden, dev):
    """vit_decode_packed at the C ABI: status codes, not exceptions -- bad offsets, a workspace that is too small, an empty batch."""
    import ctypes
    lib = _lib.load()
    A, pi = golden["params"]["tonet361_logA_T"], golden["params"]["tonet361_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    E = synth.emissions_peaks(1, 50, 361, seed=1, device=dev)[0].contiguous()
    st = torch.empty((50,), dtype=torch.int32, device=dev)
    ll = torch.empty((2,), dtype=torch.float32, device=dev)
    need = dec.workspace_bytes_packed(2, 50)
    assert need > 50 * 384 * 4
    ws = torch.empty(need + 256, dtype=torch.uint8, device=dev)
    wp = (ws.data_ptr() + 255) & ~255

    def call(off, nbytes=need, B=None):
        off = np.asarray(off, np.int64)
        return lib.vit_decode_packed(dec._plan, E.data_ptr(), _lib.VIT_F32, len(off) - 1 if B is None else B, off.ctypes.data, wp, nbytes,
                                     st.data_ptr(), ll.data_ptr(), None)
    assert call([0, 20, 50]) == 0
    torch.cuda.synchronize()
    ref_s, ref_l = vo.decode_c(A, pi, np.stack([np.pad(E[:20].cpu().numpy(), ((0, 30), (0, 0))), np.pad(E[20:].cpu().numpy(), ((0, 20), (0, 0)))]),
                               lengths=np.array([20, 30], np.int64))
    assert np.array_equal(st[:20].cpu().numpy(), ref_s[0, :20]) and np.array_equal(st[20:].cpu().numpy(), ref_s[1, :30])
    assert np.array_equal(ll.cpu().numpy(), ref_l)
    assert call([1, 20, 50]) == -1                         # offsets must start at 0           (VIT_EINVAL)
    assert call([0, 20, 20]) == -1                         # an empty song
    assert call([0, 30, 20]) == -1                         # decreasing
    assert call([0, 20, 50], nbytes=need - 1) == -4        # VIT_EWORKSPACE
    assert call([0], B=0) == 0                             # nothing to do
    assert lib.vit_decode_packed(dec._plan, E.data_ptr(), 7, 2, np.asarray([0, 20, 50], np.int64).ctypes.data, wp, need, st.data_ptr(), ll.data_ptr(), None) == -1
    assert int(lib.vit_workspace_bytes_packed(None, 2, 50)) == 0


def test_postprocessor_over_many_recordings(golden, dev):
    """reference_api.Viterbi / SoftMaxViterbi.decode_recordings: recordings of different lengths through one builder launch, one packed
    decode and one voicing map -- every recording's (voiced, bins) equal to the post-processor run on that recording alone."""
    from tests.common import logits_case
    from viterbi_spl_amd import reference_api as ra
    A, pi = synth.tonet_transition(360, 14), synth.floored_prior(361)
    lens = [257, 1, 64, 1000, 2, 333, 129]
    for cls, cols in ((ra.Viterbi, 360), (ra.SoftMaxViterbi, 361)):
        vit = cls(A, pi, device=dev)
        recs = [logits_case(100 + k, n, cols) for k, n in enumerate(lens)]
        got = vit.decode_recordings(recs)
        assert len(got) == len(lens)
        for k, x in enumerate(recs):
            v1, b1 = vit.decode_logits(x)
            assert got[k][0].shape == (lens[k],) and torch.equal(got[k][0], v1) and torch.equal(got[k][1], b1), (cls.__name__, k)
        assert vit.decode_recordings([]) == []


def test_half_backtrace_with_small_workgroups(golden, dev):
    """bt_block_waves: the half history's back-trace in workgroups of eight / four waves (the form that can start beside resident
    forward waves) decodes the same bits as the sixteen-wave default, chunked and with forced bad guesses."""
    A, pi = golden["params"]["tonet361_logA_T"], golden["params"]["tonet361_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    E = synth.emissions_dense(19, 901, 361, seed=8, device=dev)
    lens = torch.tensor([901, 1, 2, 900, 450, 3, 64, 65, 66, 4, 5, 901, 333, 17, 129, 800, 2, 640, 77], dtype=torch.int64, device=dev)
    ref_s, ref_l = vo.decode_c(A, pi, E.cpu().numpy(), lengths=lens.cpu().numpy())
    for waves in (0, 8, 4):
        for chunks, warm in ((0, -1), (7, 0), (2, 5)):
            dec.set_option("reset", 0)
            dec.set_option("wave_history", 2)
            dec.set_option("bt_block_waves", waves)
            dec.set_option("bt_chunks", chunks)
            dec.set_option("bt_warm", warm)
            st, ll = dec.decode(E, lengths=lens, algo="wave", out_dtype=torch.int32)
            assert np.array_equal(st.cpu().numpy(), ref_s) and np.array_equal(ll.cpu().numpy(), ref_l), (waves, chunks, warm)
    dec.set_option("reset", 0)


@pytest.mark.parametrize("name", ["tonet361", "msnet321"])
def test_scalars_of_three_frames_per_history_row(golden, dev, name):
    """Full history of the wave form with one extra column: row t carries the scalars (frame maximum, delta of the extra column) of
    frames t, t-1, t-2, and the back-trace kernels read frame t's from the carrier row t - t % 3 + 2, or from the song's last row
    (csrc/kernels.hpp wave_aux_frames / wave_aux_row).  Every song length 1 .. 13 (all residues, the clamp at the last row), chunk
    boundaries at every position, both back-trace kernels and the packed decode -- against the oracle, and against the layout in
    which a row carries its own scalars only ("wave_two" bit 2)."""
    A, pi = golden["params"][f"{name}_logA_T"], golden["params"][f"{name}_log_pi"]
    dec = ViterbiDecoder(A, pi, dev)
    S = dec.S
    lens = np.array(list(range(1, 14)) + [97, 98, 99, 100], np.int64)
    B, T = len(lens), 100
    for kind, gen in (("peaks", synth.emissions_peaks), ("ties", synth.emissions_ties)):
        E = gen(B, T, S, seed=31, device=dev)
        ref_s, ref_l = vo.decode_c(A, pi, E.cpu().numpy(), lengths=lens)
        for own_only in (0, 4):
            for form, chunks, warm in ((0, 0, -1), (0, 7, 0), (0, 32, 1), (4, 0, -1), (4, 5, 0), (4, 64, 2), (4, 13, 1)):
                dec.set_option("reset", 0)
                dec.set_option("wave_two", own_only)
                dec.set_option("backtrace_form", form)
                dec.set_option("bt_chunks", chunks)
                dec.set_option("bt_warm", warm)
                st, ll = dec.decode(E, lengths=torch.from_numpy(lens).to(dev), algo="wave", out_dtype=torch.int32)
                assert np.array_equal(st.cpu().numpy(), ref_s), (name, kind, own_only, form, chunks, warm)
                assert np.array_equal(ll.cpu().numpy(), ref_l)
            dec.set_option("reset", 0)
            dec.set_option("wave_two", own_only)
            Ep, off = _pack(E, lens)
            st, ll = dec.decode_packed(Ep, off, out_dtype=torch.int32)
            assert np.array_equal(_unpack(st.cpu().numpy(), off, T), ref_s), (name, kind, own_only, "packed")
            assert np.array_equal(ll.cpu().numpy(), ref_l)
    dec.set_option("reset", 0)
