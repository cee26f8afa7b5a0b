"""Host-side logic that needs no GPU: synthetic generators, .dat I/O, partitioning, adapter asserts."""
import numpy as np
import pytest
import torch

from viterbi_spl_amd import datfile, sharded, synth


def test_generators_are_deterministic_and_on_grid():
    a = synth.emissions_dense(2, 50, 361, seed=5)
    b = synth.emissions_dense(1, 50, 361, seed=5, first_song=1)
    assert torch.equal(a[1], b[0])
    assert torch.equal(a * 256, torch.round(a * 256)) and a.min() >= -30 and a.max() <= 0
    p = synth.emissions_peaks(1, 200, 361, seed=1)[0]
    floor = np.float32(synth.LOG_TINY32)
    nonfloor = (p != floor).sum(dim=1)
    assert nonfloor.min() >= 1 and nonfloor.max() <= 6
    t = synth.emissions_ties(1, 20, 50, seed=1)
    assert set(np.unique(t.numpy())) <= {0.0, -1.0, -2.0}


def test_tonet_recipe_matrix():
    A = synth.tonet_transition(360, 14)
    assert A.shape == (361, 361) and A.dtype == np.float32
    assert np.allclose(A.sum(axis=1), 1)
    assert np.count_nonzero(A[100]) == 30 and np.count_nonzero(A[360]) == 361
    logA_T, log_pi = synth.log_params(A, synth.floored_prior(361))
    assert logA_T.flags["C_CONTIGUOUS"] and logA_T.dtype == np.float32
    assert logA_T[5, 100] == np.float32(synth.LOG_TINY32)  # target 5 <- source 100 is structurally zero
    assert not np.isneginf(logA_T).any()


def test_dat_roundtrip(tmp_path, golden):
    A = golden["params"]["msnet321_A"]
    f = tmp_path / "viterbi_transition_matrix.dat"
    datfile.save_np_array_to_file_fn(str(f), A, "viterbi_transition_matrix")
    head = open(f, "rb").readline()
    assert head == b"viterbi_transition_matrix C float32 321 321\n"
    name, B = datfile.load_np_array_from_file_fn(str(f))
    assert name == "viterbi_transition_matrix" and B.tobytes() == A.tobytes()
    # header form of the shipped msnet files (no C/F flag)
    g = tmp_path / "old.dat"
    g.write_bytes(b"viterbi_init_probs float32 321\n" + golden["params"]["msnet321_pi"].tobytes())
    name, pi = datfile.load_np_array_from_file_fn(str(g))
    assert name == "viterbi_init_probs" and pi.shape == (321,)
    # Fortran-ordered 2-D arrays come back F-contiguous
    F = np.asfortranarray(np.arange(12, dtype=np.float32).reshape(3, 4))
    h = tmp_path / "f.dat"
    datfile.save_np_array_to_file_fn(str(h), F, "x")
    _, F2 = datfile.load_np_array_from_file_fn(str(h))
    assert F2.flags["F_CONTIGUOUS"] and np.array_equal(F, F2)


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 128, 1024, 1025):
        for w in (1, 2, 3, 8):
            spans = [sharded.shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharded.shard_bounds(4, 2, 2)


def test_adapters_keep_reference_assertions(golden):
    from viterbi_spl_amd import reference_api as ra
    A = golden["params"]["msnet321_A"].copy()
    pi = golden["params"]["msnet321_pi"]
    P = np.full((321, 4), 0.5, np.float32, order="F")
    bad = A.copy()
    bad[0, 0] += 0.5
    with pytest.raises(AssertionError):
        ra.viterbi_librosa_c_fn(transition_matrix=bad, prob_init=pi, probs_st=P)
    with pytest.raises(AssertionError):
        ra.viterbi_librosa_fn(log_transition_matrix_T=np.asfortranarray(A), log_prob_init=pi, log_probs_st=P)
    with pytest.raises(AssertionError):
        ra.viterbi_librosa_fn(log_transition_matrix_T=A.astype(np.float64), log_prob_init=pi, log_probs_st=P)


def test_decode_refuses_cpu_tensors(golden):
    import viterbi_spl_amd as v
    with pytest.raises(ValueError):
        v.decode(torch.zeros(3, 321), golden["params"]["msnet321_logA_T"], golden["params"]["msnet321_log_pi"])


def test_host_emission_builders_match_reference_goldens(golden, monkeypatch):
    """The adapters' host-exact builders (used with exact_emissions=True) reproduce the reference bit for bit.
    Constructed without a GPU: only the NumPy methods are exercised."""
    import os
    from tests.common import logits_case
    from viterbi_spl_amd import reference_api as ra
    og = np.load(os.path.join(os.path.dirname(__file__), "golden", "obs_goldens.npz"))
    v = ra.Viterbi.__new__(ra.Viterbi)
    v.num_freq_bins, v.single_side_peak_width, v.threshold = 360, 5, np.log(0.32 / (1. - 0.32))
    s = ra.SoftMaxViterbi.__new__(ra.SoftMaxViterbi)
    s.num_freq_bins, s.single_side_peak_width = 360, 15
    for k in range(3):
        seed, n = og[f"shaun{k}_seed"]
        got = v.observation_probs_fn(logits_case(int(seed), int(n), 360))
        assert got.flags["F_CONTIGUOUS"] and np.array_equal(np.ascontiguousarray(got.T), og[f"shaun{k}_probs"])
        seed, n = og[f"softmax{k}_seed"]
        assert np.array_equal(s.observation_probs_fn(logits_case(int(seed), int(n), 361)), og[f"softmax{k}_probs"])
