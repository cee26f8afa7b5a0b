"""Plan analysis + exactness of the banded decomposition, on the CPU (no GPU needed)."""
import numpy as np
import pytest

from oracle import viterbi_oracle as vo
from tests.plan_replay import HostPlan, replay_banded, replay_dense_image, replay_wave
from viterbi_spl_amd import synth


def test_tonet_structure_is_proven(golden):
    p = golden["params"]
    plan = HostPlan(p["tonet361_logA_T"], p["tonet361_log_pi"])
    assert plan.ok and plan.S == 361 and plan.SP == 384
    assert plan.max_window == 29 and plan.W == 32
    assert plan.extras == [360] and plan.dense_rows == []      # the unvoiced row is banded with its own constant
    assert plan.c0 == np.float32(synth.LOG_TINY32)
    assert np.all(plan.rowc[:360] == np.float32(synth.LOG_TINY32)) and plan.rowc[360] == p["tonet361_logA_T"][360, 0]
    assert np.all(plan.lo[:361] + plan.W <= 361) and np.all(plan.lo[:361] >= 0)


def test_msnet_real_parameters_are_banded(golden):
    p = golden["params"]
    plan = HostPlan(p["msnet321_logA_T"], p["msnet321_log_pi"])
    assert plan.ok and plan.max_window == 25 and plan.W == 32
    assert plan.extras == [320] and plan.dense_rows == []


def test_unstructured_matrices_fall_back(golden):
    p = golden["params"]
    assert not HostPlan(p["dense361_logA_T"], p["dense361_log_pi"]).ok
    assert not HostPlan(p["durrieu722_logA_T"], p["durrieu722_log_pi"]).ok
    assert not HostPlan(p["dense97_logA_T"], p["dense97_log_pi"]).ok


def test_image_packing_roundtrip(golden):
    p = golden["params"]
    A = p["dense97_logA_T"]
    plan = HostPlan(A, p["dense97_log_pi"])
    E = synth.emissions_dense(1, 50, 97, seed=2)[0].numpy()
    a, la = replay_dense_image(plan, E)
    b, lb = vo.decode_numpy(A, p["dense97_log_pi"], E)
    assert np.array_equal(a, b) and la == lb
    assert np.all(np.isneginf(plan.A4[:, 97:, :])) and np.all(np.isneginf(plan.log_pi[97:]))


@pytest.mark.parametrize("pname,kind,T,seed", [
    ("tonet361", "peaks", 120, 1), ("tonet361", "dense", 120, 2), ("tonet361", "ties", 120, 3),
    ("msnet321", "peaks", 100, 4), ("msnet321", "ties", 100, 5),
])
def test_banded_replay_is_bit_exact(golden, pname, kind, T, seed):
    from tests.common import GEN
    p = golden["params"]
    A, pi = p[f"{pname}_logA_T"], p[f"{pname}_log_pi"]
    plan = HostPlan(A, pi)
    E = GEN[kind](1, T, A.shape[0], seed=seed)[0].numpy()
    st, ll, delta = replay_banded(plan, E)
    ref, rl, rdelta = vo.decode_c(A, pi, E, return_delta=True)
    assert np.array_equal(st, ref)
    assert delta.tobytes() == rdelta.tobytes() and np.float32(ll) == np.float32(rl)
    # the reference's matrices also qualify for the one-maximum ("floor-max") form of the forward kernel
    assert plan.floor_ok
    st2, ll2, delta2 = replay_banded(plan, E, floor=True)
    assert np.array_equal(st2, ref) and delta2.tobytes() == rdelta.tobytes()
    # ... and for the two-targets-per-lane form (one W-wide window per pair of adjacent targets)
    assert plan.pair_ok
    st3, ll3, delta3 = replay_banded(plan, E, floor=True, pair=True)
    assert np.array_equal(st3, ref) and delta3.tobytes() == rdelta.tobytes()


def _banded_matrix(S, half, rng, extras=(), dense_rows=(), floor=-50.0, quant=4):
    A = np.full((S, S), floor, np.float32)
    for j in range(S):
        lo, hi = max(0, j - half), min(S, j + half + 1)
        A[j, lo:hi] = -(rng.integers(0, 40, hi - lo) / quant)
    for x in extras:
        A[:, x] = -(rng.integers(0, 40, S) / quant)
    for r in dense_rows:
        A[r, :] = -(rng.integers(0, 40, S) / quant)
    return A.astype(np.float32)


@pytest.mark.parametrize("seed", range(6))
def test_random_banded_structures_with_ties(seed):
    """Adversarial: coarse value grid (many exact ties), extras in the middle, dense rows anywhere,
    floor values that regularly win."""
    rng = np.random.default_rng(seed)
    S = int(rng.integers(70, 200))
    half = int(rng.integers(1, 7))
    extras = sorted(set(int(x) for x in rng.integers(0, S, int(rng.integers(0, 3)))))
    dense_rows = sorted(set(int(x) for x in rng.integers(0, S, int(rng.integers(0, 3)))))
    A = _banded_matrix(S, half, rng, extras, dense_rows, floor=-3.0 if seed % 2 else -50.0, quant=2)
    pi = -(rng.integers(0, 8, S) / 2).astype(np.float32)
    plan = HostPlan(A, pi)
    assert plan.ok, (S, half, extras, dense_rows)
    E = -(rng.integers(0, 6, (60, S)) / 2).astype(np.float32)
    st, ll, delta = replay_banded(plan, E)
    ref, rl, rdelta = vo.decode_c(A, pi, E, return_delta=True)
    assert np.array_equal(st, ref)
    assert delta.tobytes() == rdelta.tobytes()
    # floor-max form: only proven when no in-window entry is below the row constant and there are no dense rows
    assert plan.floor_ok == (plan.n_dense == 0 and _window_ge_floor(A, plan))
    if plan.floor_ok:
        st2, _, delta2 = replay_banded(plan, E, floor=True)
        assert np.array_equal(st2, ref) and delta2.tobytes() == rdelta.tobytes()
        if plan.pair_ok:
            st3, _, delta3 = replay_banded(plan, E, floor=True, pair=True)
            assert np.array_equal(st3, ref) and delta3.tobytes() == rdelta.tobytes()


def _window_ge_floor(A, plan):
    S, W = plan.S, plan.W
    for j in range(S):
        for i in range(plan.lo[j], plan.lo[j] + W):
            if i not in plan.extras and not A[j, i] >= plan.rowc[j]:
                return False
    return True


def test_floor_form_is_refused_when_a_window_entry_is_below_the_row_constant():
    """The one-maximum form would be WRONG here (an in-window source could win through the row constant):
    the plan must not offer it, and the scan form must stay exact."""
    rng = np.random.default_rng(5)
    S = 96
    A = _banded_matrix(S, 4, rng, floor=-2.0, quant=1)     # window entries down to -39, floor -2
    pi = np.zeros(S, np.float32)
    plan = HostPlan(A, pi)
    assert plan.ok and not plan.floor_ok
    E = -(rng.integers(0, 4, (50, S))).astype(np.float32)
    st, _, delta = replay_banded(plan, E)
    ref, _, rdelta = vo.decode_c(A, pi, E, return_delta=True)
    assert np.array_equal(st, ref) and delta.tobytes() == rdelta.tobytes()


def test_banded_with_minus_inf_floor():
    rng = np.random.default_rng(11)
    S = 128
    A = _banded_matrix(S, 3, rng, extras=(S - 1,), dense_rows=(S - 1,), floor=-np.inf)
    pi = np.full(S, -np.inf, np.float32)
    pi[5] = 0
    plan = HostPlan(A, pi)
    assert plan.ok
    E = -(rng.integers(0, 6, (40, S)) / 2).astype(np.float32)
    st, ll, delta = replay_banded(plan, E)
    ref, rl, rdelta = vo.decode_c(A, pi, E, return_delta=True)
    assert np.array_equal(st, ref) and delta.tobytes() == rdelta.tobytes()


@pytest.mark.parametrize("n_bins,d_max,W", [(721, 40, 84), (721, 46, 96), (499, 56, 128), (721, 30, 64)])
def test_wide_bands_of_the_high_resolution_grids(n_bins, d_max, W):
    """jdc: d_max = 40 on the 721-bin grid (82 + unvoiced exceptions per row), imm: d_max = 56; the plan proves them
    banded with W = 84 / 128 (84: the narrowest whole-float4 window over the 81 sources of a row; 96 for a wider band) and the floor-max form replays bit for bit."""
    logA_T, log_pi = synth.log_params(synth.tonet_transition(n_bins, d_max), synth.floored_prior(n_bins + 1))
    plan = HostPlan(logA_T, log_pi)
    assert plan.ok and plan.W == W and plan.max_window == 2 * d_max + 1 and plan.extras == [n_bins] and plan.n_dense == 0
    assert plan.floor_ok
    E = synth.emissions_peaks(1, 60, n_bins + 1, seed=9)[0].numpy()
    st, ll, delta = replay_banded(plan, E, floor=True)
    ref, rl, rdelta = vo.decode_c(logA_T, log_pi, E, return_delta=True)
    assert np.array_equal(st, ref) and delta.tobytes() == rdelta.tobytes()


def test_step_structure_of_the_durrieu_matrix(golden):
    """imm's own decoder uses a dense matrix whose columns are piecewise constant in 20-bin distance bands; the plan must
    prove that from the bits (and refuse as soon as one entry breaks it), and the max of band-window maxima plus ONE
    far maximum must reproduce the dense recursion bit for bit."""
    p = golden["params"]
    A, pi = p["durrieu722_logA_T"], p["durrieu722_log_pi"]
    plan = HostPlan(A, pi)
    assert not plan.ok and plan.step_ok and plan.step_bw == 20 and plan.step_kb == 9
    B = A.copy()
    B[300, 310] = np.nextafter(B[300, 310], np.float32(0))          # one entry one ulp off: not a step matrix any more
    assert not HostPlan(B, pi).step_ok
    assert not HostPlan(p["dense361_logA_T"], p["dense361_log_pi"]).step_ok
    # host replay of step_forward_kernel's arithmetic
    S, n, BW, KB = 722, 721, 20, 9
    E = synth.emissions_dense(1, 40, S, seed=12)[0].numpy()
    ref, rl, rdelta = vo.decode_c(A, pi, E, return_delta=True)
    C = np.stack([A[np.minimum(np.arange(n) + k * BW, n - 1), np.arange(n)] if k * BW < n else None for k in range(KB + 1)])
    for i in range(n):                                               # band value of source i at distance k*BW (either side)
        for k in range(KB + 1):
            j = i + k * BW if i + k * BW < n else i - k * BW
            C[k, i] = A[j, i]
    delta = (pi + E[0]).astype(np.float32)
    ninf = np.float32(-np.inf)
    for t in range(1, E.shape[0]):
        V = (delta[None, :n] + C).astype(np.float32)                 # [KB+1, n]
        M = max(np.max(V[KB]), np.float32(delta[n] + A[0, n]))
        m = np.full(S, ninf, np.float32)
        for j in range(n):
            best = M
            for i in range(max(0, j - KB * BW + 1), min(n, j + KB * BW)):
                best = max(best, V[abs(i - j) // BW, i])
            m[j] = best
        m[n] = np.max((delta + A[n]).astype(np.float32))
        delta = (m + E[t]).astype(np.float32)
    assert delta.tobytes() == rdelta.tobytes()


# ------------------------------------------------------------------ wave form (wave_forward_kernel: one song per wavefront)
@pytest.mark.parametrize("pname,kind,T,seed", [
    ("tonet361", "peaks", 90, 1), ("tonet361", "dense", 90, 2), ("tonet361", "ties", 90, 3),
    ("msnet321", "peaks", 80, 4), ("msnet321", "ties", 80, 5),
])
def test_wave_form_replay_is_bit_exact(golden, pname, kind, T, seed):
    """The reference's matrices qualify for the wave form (every exception span within 14 sources of its target, extra-column
    entries above the row constant); the packed table tabV + the frame maximum over ALL sources replay bit for bit."""
    from tests.common import GEN
    p = golden["params"]
    A, pi = p[f"{pname}_logA_T"], p[f"{pname}_log_pi"]
    plan = HostPlan(A, pi)
    assert plan.wave_ok and plan.floor_all_ok and plan.wave_npl == 6 and plan.wave_dk == 14
    assert plan.wave_d == {"tonet361": 14, "msnet321": 12}[pname]
    E = GEN[kind](1, T, A.shape[0], seed=seed)[0].numpy()
    hist, delta = replay_wave(plan, E)
    _, _, rdelta = vo.decode_c(A, pi, E, return_delta=True)
    assert delta.tobytes() == rdelta.tobytes()
    # the history layout the back-trace reads: state i in column 64*npl - S + i, the frame maximum in column 0
    o = 64 * plan.wave_npl - plan.S
    assert hist.shape[1] == 64 * plan.wave_npl and hist[-1, o:].tobytes() == rdelta.tobytes()
    assert hist[-1, 0] == np.max(rdelta)


@pytest.mark.parametrize("seed", range(8))
def test_wave_form_on_random_banded_structures(seed):
    """Six states per lane (S in 321..383), half-widths up to 14, 0-2 extra columns anywhere, coarse value grid (ties),
    floors that regularly win.  Where the plan offers the wave form its replay must equal the dense oracle; it must
    not offer it when an extra-column entry lies below a row constant (M over all sources would then be wrong)."""
    rng = np.random.default_rng(500 + seed)
    S = int(rng.integers(321, 384))
    half = int(rng.integers(1, 15))
    extras = sorted(set(int(x) for x in rng.integers(0, S, int(rng.integers(0, 3)))))
    floor = -50.0 if seed % 2 else -3.0
    A = _banded_matrix(S, half, rng, extras, (), floor=floor, quant=2)
    pi = -(rng.integers(0, 8, S) / 2).astype(np.float32)
    plan = HostPlan(A, pi)
    assert plan.ok
    extras_dominate = all(np.all(A[:, x] >= plan.rowc[:S]) for x in plan.extras)
    assert plan.floor_all_ok == (plan.floor_ok and extras_dominate)
    assert plan.wave_ok == (plan.floor_all_ok and plan.n_dense == 0 and plan.wave_d <= 14 and plan.n_extras <= 2)
    if floor == -50.0:
        assert plan.wave_ok, (S, half, extras)
    if plan.wave_ok:
        E = -(rng.integers(0, 6, (50, S)) / 2).astype(np.float32)
        _, delta = replay_wave(plan, E)
        _, _, rdelta = vo.decode_c(A, pi, E, return_delta=True)
        assert delta.tobytes() == rdelta.tobytes()


def test_wave_form_is_refused_when_an_extra_column_is_below_the_row_constant():
    rng = np.random.default_rng(77)
    S = 350
    A = _banded_matrix(S, 6, rng, extras=(S - 1,), floor=-10.0, quant=4)   # band and extra-column entries in [-9.75, 0]
    assert HostPlan(A, np.zeros(S, np.float32)).wave_ok
    A[:, S - 1] -= 15.0                                                     # extra-column entries now below the floor
    plan = HostPlan(A, np.zeros(S, np.float32))
    assert plan.ok and plan.floor_ok and not plan.floor_all_ok and not plan.wave_ok


def test_wave_form_needs_six_states_per_lane_and_narrow_bands():
    logA_T, log_pi = synth.log_params(synth.tonet_transition(720, 40), synth.floored_prior(721))
    assert not HostPlan(logA_T, log_pi).wave_ok                      # 12 states per lane, half-width 40: workgroup kernels
    logA_T, log_pi = synth.log_params(synth.tonet_transition(360, 20), synth.floored_prior(361))
    p = HostPlan(logA_T, log_pi)
    assert p.ok and p.wave_d == 20 and not p.wave_ok                 # half-width beyond the instantiated 14
