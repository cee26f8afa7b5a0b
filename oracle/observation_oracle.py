"""oracle/observation_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatements of the reference's emission builders (logits -> observation probabilities), the
Python per-frame loops that sit right upstream of the Viterbi decoder:

* :func:`shaun_observation_probs`   -- Viterbi.observation_probs_fn + find_peaks_all_at_once_np_fn +
  expit, tonet/for_paper.py:1703-1778 (peak picking in a +/-5-bin reflect-padded window, soft voicing
  decision on the strongest peak, per-frame renormalised exp over the peaks);
* :func:`softmax_observation_probs` -- SoftMaxViterbi.observation_probs_fn, tonet/for_paper.py:1890-1944
  (softmax over the peak set with a +/-15-bin window, unvoiced logit first, rolled to the last state).

* :func:`softmax_scaled_observation_probs` -- dcnet's SoftMaxViterbi.observation_probs_fn,
  dcnet/softmax_viterbi.py:2530-2579 ("scaled likelihood": softmax over the peak set divided by the state prior,
  unvoiced logit = the padded voicing-threshold logit, +/-5-bin window; values may exceed 1).

Pinned: tests/golden/make_obs_goldens.py runs the reference's own methods (AST-extracted) on seeded
logits and commits inputs' seeds + outputs; tests compare these restatements bit for bit.
"""
from __future__ import annotations

import numpy as np


def _peaks(frames_logits: np.ndarray, spw: int) -> np.ndarray:
    """bool [n_frames, n_bins]: bin is the FIRST maximum of its reflect-padded (2*spw+1)-window."""
    n_frames, n_bins = frames_logits.shape
    padded = np.pad(frames_logits, [(0, 0), (spw, spw)], mode="reflect")
    out = np.zeros((n_frames, n_bins), np.bool_)
    w = 2 * spw + 1
    for b in range(n_bins):
        out[:, b] = np.argmax(padded[:, b:b + w], axis=1) == spw
    return out


def _expit(s):
    if s > 0:
        return 1.0 / (1.0 + np.exp(-s))
    p = np.exp(s)
    return p / (1.0 + p)


def shaun_observation_probs(logits: np.ndarray, voicing_threshold: float = 0.32, spw: int = 5,
                            p: float = 0.8, scale: float = 2.0) -> np.ndarray:
    """logits float32 [n_frames, n_bins] -> probabilities float32 [n_bins+1, n_frames], F-order."""
    assert logits.dtype == np.float32 and logits.ndim == 2
    n_frames, n_bins = logits.shape
    threshold = np.log(voicing_threshold / (1.0 - voicing_threshold))
    offset = np.log(p / (1.0 - p))
    out = np.zeros((n_bins + 1, n_frames), np.float32, order="F")
    is_peak = _peaks(logits, spw)
    for f in range(n_frames):
        idx = np.where(is_peak[f])[0]
        if len(idx) == 0:
            out[-1, f] = 1
            continue
        pk = logits[f][idx]                       # a copy (fancy indexing), float32
        g = pk[np.argmax(pk)]
        if g >= threshold:
            s = scale * (g - threshold) + offset
        else:
            s = scale * (g - threshold) - offset
        p_voiced = _expit(s)
        pk -= g
        np.exp(pk, out=pk)
        t = p_voiced / np.sum(pk)
        np.multiply(pk, t, out=pk)
        out[idx, f] = pk
        out[-1, f] = 1.0 - p_voiced
    return out


def softmax_observation_probs(logits: np.ndarray, spw: int = 15) -> np.ndarray:
    """logits float32 [n_frames, n_bins+1] (column 0 = unvoiced) -> probabilities float32
    [n_frames, n_bins+1] with the unvoiced state LAST."""
    assert logits.dtype == np.float32 and logits.ndim == 2 and logits.flags["C_CONTIGUOUS"]
    n_frames = len(logits)
    n_bins = logits.shape[1] - 1
    is_peak = np.zeros((n_frames, n_bins + 1), np.bool_)
    is_peak[:, 0] = True
    is_peak[:, 1:] = _peaks(logits[:, 1:], spw)
    out = np.zeros((n_frames, n_bins + 1), np.float32)
    for f in range(n_frames):
        idx = np.where(is_peak[f])[0]
        if len(idx) == 1:
            out[f, 0] = 1
            continue
        pk = logits[f, idx]
        pk = pk - np.max(pk)
        pk = np.exp(pk)
        pk = pk / np.sum(pk)
        out[f, idx] = pk
    return np.roll(out, shift=-1, axis=1)


def softmax_scaled_observation_probs(logits: np.ndarray, voicing_threshold_prob: float, ini_probs: np.ndarray,
                                     scaled: bool = True, spw: int = 5) -> np.ndarray:
    """logits float32 [n_frames, n_bins] -> scaled likelihoods float32 [n_frames, n_bins+1], unvoiced state LAST.
    ini_probs: the state prior [n_bins+1] (unvoiced last), as loaded from viterbi_init_probs.dat."""
    assert logits.dtype == np.float32 and logits.ndim == 2
    n_frames, n_bins = logits.shape
    if scaled:
        prior = np.roll(np.asarray(ini_probs), 1).astype(np.float32)      # unvoiced first, like the padded logits
    else:
        prior = np.ones([n_bins + 1], np.float32)
    vth = np.log(voicing_threshold_prob / (1. - voicing_threshold_prob))
    lg = np.pad(logits, [[0, 0], [1, 0]], mode="constant", constant_values=vth)
    is_peak = np.zeros((n_frames, n_bins + 1), np.bool_)
    is_peak[:, 0] = True
    is_peak[:, 1:] = _peaks(lg[:, 1:], spw)
    out = np.zeros((n_frames, n_bins + 1), np.float32)
    for f in range(n_frames):
        idx = np.where(is_peak[f])[0]
        if len(idx) == 1:
            out[f, 0] = 1. / prior[0]
            continue
        pk = lg[f, idx]
        np.subtract(pk, np.max(pk), out=pk)
        np.exp(pk, out=pk)
        np.divide(pk, np.sum(pk), out=pk)
        np.divide(pk, prior[idx], out=pk)
        out[f, idx] = pk
    return np.roll(out, shift=-1, axis=1)
