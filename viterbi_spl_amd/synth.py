"""Synthetic HMM parameters and emissions for tests and bench.py.

Everything here is deterministic integer hashing on a k/256 grid, so no libm
call decides a value: the same (seed, song, frame, state) gives the same bits
with NumPy on the host and with torch on the GPU.

Parameter recipes follow the reference's offline scripts (paths relative to
/root/reference):

* :func:`tonet_transition` -- tonet/viterbi_transition_post_processing.py:44-88
  (banded Toeplitz voiced block, d_max = 14, voiced/unvoiced switch matrix).
* :func:`floored_prior`    -- tonet/p_steady_post_processing.py:7-24.
* :func:`durrieu_transition` -- imm/transition_matrix.py:3-27 (fully dense).
* :func:`log_params`       -- tonet/for_paper.py:1780-1815 (log(x + tiny), transpose).
"""
from __future__ import annotations

import numpy as np
import torch

TINY32 = float(np.finfo(np.float32).tiny)
LOG_TINY32 = float(np.log(np.float32(0.0) + np.finfo(np.float32).tiny))  # -87.33655

_M32 = 0xFFFFFFFF


# --------------------------------------------------------------------------- hashing
def _mul32(x: torch.Tensor, c: int) -> torch.Tensor:
    """(x * c) mod 2^32 on int64 tensors holding values < 2^32, without int64 overflow."""
    lo = x * (c & 0xFFFF)
    hi = ((x * (c >> 16)) & 0xFFFF) << 16
    return (lo + hi) & _M32


def _mix32(x: torch.Tensor) -> torch.Tensor:
    x = x & _M32
    x = x ^ (x >> 16)
    x = _mul32(x, 0x7FEB352D)
    x = x ^ (x >> 15)
    x = _mul32(x, 0x846CA68B)
    x = x ^ (x >> 16)
    return x


def _song_base(seed: int, song: int) -> int:
    t = torch.tensor([(seed * 0x9E3779B1 + song * 0x85EBCA6B + 0x165667B1) & _M32], dtype=torch.int64)
    return int(_mix32(t)[0])


def _row_hash(base: int, T: int, device) -> torch.Tensor:
    t = torch.arange(T, dtype=torch.int64, device=device)
    return _mix32(_mul32(t, 0x9E3779B9) ^ base)


def _cell_hash(row: torch.Tensor, cols: torch.Tensor) -> torch.Tensor:
    return _mix32(row[:, None] + _mul32(cols, 0xC2B2AE35)[None, :])


# --------------------------------------------------------------------------- emissions
def emissions_dense(B: int, T: int, S: int, seed: int = 0, device="cpu", dtype=torch.float32,
                    first_song: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
    """Dense i.i.d. log-emissions on the k/256 grid in [-30, 0]; shape [B,T,S]."""
    if out is None:
        out = torch.empty((B, T, S), dtype=dtype, device=device)
    cols = torch.arange(S, dtype=torch.int64, device=device)
    for b in range(B):
        row = _row_hash(_song_base(seed, first_song + b), T, device)
        h = _cell_hash(row, cols)
        out[b] = ((h % 7681).to(torch.float32) * (-1.0 / 256.0)).to(dtype)
    return out


def emissions_peaks(B: int, T: int, S: int, seed: int = 0, device="cpu", dtype=torch.float32,
                    first_song: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
    """Peak-sparse log-emissions mimicking Viterbi.observation_probs_fn
    (tonet/for_paper.py:1733-1778): per frame 0-5 pitch peaks plus the unvoiced
    state (last), every other state exactly log(tiny)."""
    if S < 8:
        raise ValueError("emissions_peaks needs at least 8 states (peak positions are drawn from S - 5 centres)")
    if out is None:
        out = torch.empty((B, T, S), dtype=dtype, device=device)
    frames = torch.arange(T, dtype=torch.int64, device=device)
    for b in range(B):
        row = _row_hash(_song_base(seed ^ 0x5BD1E995, first_song + b), T, device)
        e = torch.full((T, S), LOG_TINY32, dtype=torch.float32, device=device)
        # distractor peaks: 0-4 random bins per frame
        n_peaks = _mix32(row ^ 0x1000) % 5
        for k in range(4):
            pos = _mix32(row ^ (0x2000 + k)) % (S - 1)
            val = -2.0 - (_mix32(row ^ (0x3000 + k)) % 2048).to(torch.float32) * (1.0 / 256.0)
            on = n_peaks > k
            e[frames[on], pos[on]] = val[on]
        # melody peak: 48-frame notes, 3 of 4 voiced, jittering +/-2 bins around the note's bin
        note = _mix32((frames // 48) ^ _song_base(seed ^ 0x1B873593, first_song + b))
        voiced = (note % 4) != 0
        centre = 2 + (note >> 8) % (S - 5)
        pos = centre + (_mix32(row ^ 0x5000) % 5) - 2
        val = (_mix32(row ^ 0x6000) % 512).to(torch.float32) * (-1.0 / 256.0)
        e[frames[voiced], pos[voiced]] = val[voiced]
        # unvoiced state (last): likely between notes, unlikely inside them
        u = (_mix32(row ^ 0x4000) % 1024).to(torch.float32) * (-1.0 / 256.0)
        e[:, S - 1] = torch.where(voiced, u - 6.0, u)
        out[b] = e.to(dtype)
    return out


def emissions_scaled(B: int, T: int, S: int, seed: int = 0, device="cpu", dtype=torch.float32,
                     first_song: int = 0) -> torch.Tensor:
    """Peak-sparse log-emissions of a "scaled likelihood" builder (p / prior, dcnet/softmax_viterbi.py:2571-2572 in the
    reference): like :func:`emissions_peaks`, but peak values reach well ABOVE zero (up to +6: a posterior near 1 over a
    prior of a few 1e-3) and the unvoiced state swings between -3 and +3; every other state exactly log(tiny)."""
    if S < 8:
        raise ValueError("emissions_scaled needs at least 8 states (peak positions are drawn from S - 5 centres)")
    out = torch.empty((B, T, S), dtype=dtype, device=device)
    frames = torch.arange(T, dtype=torch.int64, device=device)
    for b in range(B):
        row = _row_hash(_song_base(seed ^ 0x3C6EF372, first_song + b), T, device)
        e = torch.full((T, S), LOG_TINY32, dtype=torch.float32, device=device)
        n_peaks = _mix32(row ^ 0x1100) % 5
        for k in range(4):
            pos = _mix32(row ^ (0x2100 + k)) % (S - 1)
            val = 4.0 - (_mix32(row ^ (0x3100 + k)) % 2048).to(torch.float32) * (1.0 / 256.0)      # [-4, +4]
            on = n_peaks > k
            e[frames[on], pos[on]] = val[on]
        note = _mix32((frames // 48) ^ _song_base(seed ^ 0x7F4A7C15, first_song + b))
        voiced = (note % 4) != 0
        centre = 2 + (note >> 8) % (S - 5)
        pos = centre + (_mix32(row ^ 0x5100) % 5) - 2
        val = 6.0 - (_mix32(row ^ 0x6100) % 512).to(torch.float32) * (1.0 / 256.0)                 # (+4, +6]
        e[frames[voiced], pos[voiced]] = val[voiced]
        u = 3.0 - (_mix32(row ^ 0x4100) % 1024).to(torch.float32) * (1.0 / 256.0)                  # (-1, +3]
        e[:, S - 1] = torch.where(voiced, u - 2.0, u)
        out[b] = e.to(dtype)
    return out


def emissions_ties(B: int, T: int, S: int, seed: int = 0, device="cpu", dtype=torch.float32) -> torch.Tensor:
    """Adversarial-tie emissions: values from {0, -1, -2} only, so many
    candidates coincide exactly and the lowest-index tie-break decides."""
    out = torch.empty((B, T, S), dtype=dtype, device=device)
    cols = torch.arange(S, dtype=torch.int64, device=device)
    for b in range(B):
        row = _row_hash(_song_base(seed ^ 0x27D4EB2F, b), T, device)
        out[b] = (-(_cell_hash(row, cols) % 3)).to(torch.float32).to(dtype)
    return out


# --------------------------------------------------------------------------- parameters
def tonet_transition(n_bins: int = 360, d_max: int = 14) -> np.ndarray:
    """Row-stochastic float32 [n_bins+1, n_bins+1] (source -> target), band +/- d_max."""
    d = np.arange(-d_max, d_max + 1)
    profile = 1.0 / (1.0 + np.abs(d)).astype(np.float64) ** 3      # positive, decreasing in |d|
    profile = profile / np.sum(profile)
    A = np.zeros((n_bins, n_bins), np.float32)
    for off, w in zip(d, profile):
        idx = np.arange(max(0, -off), min(n_bins, n_bins - off))
        A[idx, idx + off] = w
    A = A / np.sum(A, axis=1)[:, None]
    switch = np.asarray([[0.97790518, 0.02209482], [0.01720512, 0.98279488]], np.float32)
    A = np.pad(A, [(0, 1), (0, 1)])
    A[:n_bins, :n_bins] *= switch[0, 0]
    A[:n_bins, n_bins] = switch[0, 1]
    A[n_bins, :n_bins] = switch[1, 0] / n_bins
    A[n_bins, n_bins] = switch[1, 1]
    assert np.allclose(np.sum(A, axis=1), 1.0)
    return A.astype(np.float32)


def durrieu_transition(n_bins: int = 721, bins_per_semitone: int = 20) -> np.ndarray:
    """Fully dense float64 [n_bins+1, n_bins+1]: piecewise-constant exp(-semitones)."""
    per_dist = np.exp(-(np.arange(n_bins) // bins_per_semitone).astype(np.float64))
    cutoff = 10 * bins_per_semitone
    per_dist[cutoff:] = per_dist[cutoff - 1]
    r = np.arange(n_bins)
    A = np.empty((n_bins + 1, n_bins + 1), np.float64)
    A[:n_bins, :n_bins] = per_dist[np.abs(r[:, None] - r[None, :])]
    cp = per_dist[cutoff - 1]
    A[:n_bins, n_bins] = cp * 10.0 ** (-90)
    A[n_bins, :n_bins] = cp * 10.0 ** (-80)
    A[n_bins, n_bins] = cp * 10.0 ** (-100)
    return A / np.sum(A, axis=1)[:, None]


def floored_prior(S: int) -> np.ndarray:
    """A floored-stationary style prior: mass concentrated mid-range + unvoiced."""
    n = S - 1
    x = np.arange(n, dtype=np.float64)
    ps = 1.0 / (1.0 + ((x - n / 2.0) / (n / 8.0)) ** 2)
    ps = ps / np.sum(ps)
    ps = np.maximum(ps, 1.0 / S / 10.0)
    ps = ps / np.sum(ps) * 0.45
    return np.append(ps, 0.55).astype(np.float32)


def uniform_prior(S: int) -> np.ndarray:
    return np.full((S,), 1.0 / S, np.float32)


def log_params(transition_matrix: np.ndarray, prob_init: np.ndarray):
    """log(x + tiny) in float32; transition transposed to [target, source], C order."""
    A = np.asarray(transition_matrix).astype(np.float32)
    logA_T = np.require(np.log(A + np.float32(TINY32)).T, np.float32, ["C"])
    log_pi = np.log(np.asarray(prob_init, np.float32) + np.float32(TINY32)).astype(np.float32)
    return logA_T, log_pi


def dense_random_log_transition(S: int, seed: int = 0) -> np.ndarray:
    """Unstructured log-domain [target, source] matrix on the k/256 grid in [-20, 0]."""
    rows = torch.arange(S, dtype=torch.int64)
    h = _cell_hash(_mix32(rows ^ _song_base(seed ^ 0x68E31DA4, 0)), rows)
    return ((h % 5121).to(torch.float32) * (-1.0 / 256.0)).numpy()


def pitch_logits(B: int, T: int, n_bins: int, seed: int = 0, device="cpu", voicing: str = "toggle") -> torch.Tensor:
    """Synthetic pitch logits ``[B, T, n_bins]`` float32 for the emission builders (bench.py's pipeline block): a weak noise floor
    (mean -8, sd 1.5: an unvoiced frame) and, in the voiced frames, a melody-like five-bin bump whose centre drifts by a bin or two
    per frame -- the shape of tests/common.logits_case, generated on the device.  ``voicing``: "toggle" = two frames out of three are
    voiced (a voicing switch every other frame: the hardest case for the back-trace's row bound); "segments" = voiced and unvoiced
    runs of ~120 frames on average (a two-state chain that flips with probability 1/120 per frame: what a recording looks like)."""
    g = torch.Generator(device=device)
    g.manual_seed(1000003 * seed + 17)
    x = torch.randn((B, T, n_bins), generator=g, device=device) * 1.5 - 8.0
    steps = torch.randint(-2, 3, (B, T), generator=g, device=device)
    centre = (n_bins // 2 + torch.cumsum(steps, dim=1)) % (n_bins - 16) + 8
    amp = torch.rand((B, T), generator=g, device=device) * 2.0 + 0.5
    if voicing == "segments":
        flips = (torch.rand((B, T), generator=g, device=device) < 1.0 / 120.0).to(torch.int32)
        on = ((torch.cumsum(flips, dim=1) + torch.arange(B, device=device)[:, None]) % 2 == 0).to(x.dtype)
    elif voicing == "toggle":
        on = (torch.arange(T, device=device) % 3 != 0).to(x.dtype)[None, :]
    else:
        raise ValueError("voicing must be 'toggle' or 'segments'")
    voiced = on * amp
    shape = torch.tensor([1.0, 3.0, 6.0, 3.0, 1.0], device=device)
    for k in range(5):
        x.scatter_add_(2, (centre + (k - 2)).unsqueeze(-1), (voiced * shape[k]).unsqueeze(-1))
    return x
