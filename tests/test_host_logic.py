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
    with pytest.raises(AssertionError):      # float64 parameters: not the arithmetic this library reproduces
        ra.viterbi_librosa_c_fn(transition_matrix=A.astype(np.float64), prob_init=pi, probs_st=P)
    with pytest.raises(AssertionError):
        ra.viterbi_librosa_fn(log_transition_matrix_T=np.asfortranarray(A), log_prob_init=pi, log_probs_st=P)
    with pytest.raises(AssertionError):
        ra.viterbi_librosa_fn(log_transition_matrix_T=A.astype(np.float64), log_prob_init=pi, log_probs_st=P)


def test_decode_refuses_cpu_tensors(golden):
    import viterbi_spl_amd as v
    with pytest.raises(ValueError):
        v.decode(torch.zeros(3, 321), golden["params"]["msnet321_logA_T"], golden["params"]["msnet321_log_pi"])


def test_scaled_observation_oracle_matches_reference_goldens(golden):
    """dcnet's scaled-likelihood builder (dcnet/softmax_viterbi.py:2530-2579): the oracle restatement equals the
    reference's own outputs (values far above 1: positive log-emissions)."""
    import os
    from oracle import observation_oracle as oo
    from tests.common import logits_case
    og = np.load(os.path.join(os.path.dirname(__file__), "golden", "obs_goldens.npz"))
    prior = golden["params"]["msnet321_pi"]
    for k in range(4):
        seed, n = og[f"scaled{k}_seed"]
        vth, scaled = og[f"scaled{k}_vth"]
        got = oo.softmax_scaled_observation_probs(logits_case(int(seed), int(n), 320), np.float32(vth), prior, scaled=bool(scaled))
        assert np.array_equal(got, og[f"scaled{k}_probs"])
    assert og["scaled0_probs"].max() > 1000


def test_parameter_recipes_match_the_reference_scripts(golden):
    """viterbi_spl_amd/params.py against the outputs of the reference's own post-processing scripts (run with stubbed
    I/O by tests/golden/make_param_goldens.py): banded Toeplitz transition from counts, floored prior."""
    import json
    import os
    from viterbi_spl_amd import params
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    pg = np.load(os.path.join(gdir, "param_goldens.npz"))
    assert params.single_side_d_max_fn(h=0.01, B=60) == 14                 # tonet
    assert params.single_side_d_max_fn(h=0.01, B=240) == 56                # imm (imm/viterbi_transition_post_processing.py:44)
    A = params.toeplitz_from_counts(pg["counts360"].astype(np.int64), 14)
    assert A.dtype == np.float32 and A.tobytes() == pg["transition360"].tobytes()
    assert np.count_nonzero(A[100]) == 30 and np.count_nonzero(A[360]) == 361 and np.allclose(A.sum(axis=1), 1)
    pi = params.floored_prior(pg["p_steady361"])
    assert pi.dtype == np.float32 and pi.tobytes() == pg["init_probs361"].tobytes()
    assert pi[:-1].min() >= 0.9 * (1.0 / 361 / 10.0) * (1 - pi[-1])
    # ... and the structure analyser takes the recipe's output: band + unvoiced row / column, wave form available
    from tests.plan_replay import HostPlan
    plan = HostPlan(*synth.log_params(A, pi))
    assert plan.ok and plan.max_window == 29 and plan.extras == [360] and plan.floor_ok and plan.wave_ok
    # Durrieu's matrix and the shipped .dat files: hashes recorded from the reference in the build container
    man = json.load(open(os.path.join(gdir, "param_manifest.json")))
    import hashlib
    for bps, nb in ((20, 721), (20, 720), (5, 180)):
        assert hashlib.sha256(synth.durrieu_transition(nb, bps).tobytes()).hexdigest() == man[f"durrieu_{bps}_{nb}_sha256"]


def test_datfile_reads_the_shipped_parameter_files(golden, tmp_path):
    """The two parameter files the reference ships (msnet/viterbi_*.dat) are rebuilt byte for byte from the committed
    arrays + recorded header (SHA-256 recorded from the real files) and read back with datfile.py."""
    import hashlib
    import json
    import os
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "param_manifest.json")))
    for fname in ("viterbi_transition_matrix.dat", "viterbi_init_probs.dat"):
        rec = man[fname]
        arr = golden["params"][rec["params_key"]]
        raw = rec["header"].encode() + arr.tobytes()
        assert hashlib.sha256(raw).hexdigest() == rec["sha256"]
        f = tmp_path / fname
        f.write_bytes(raw)
        name, got = datfile.load_np_array_from_file_fn(str(f))
        assert name == fname[:-4] and got.dtype == arr.dtype and got.shape == arr.shape and got.tobytes() == arr.tobytes()
