"""Emission builders on the GPU: pitch logits -> log observation probabilities.

The reference builds the observation probabilities of a song with Python loops over its frames on the
host (``Viterbi.observation_probs_fn`` tonet/for_paper.py:1733-1778, ``SoftMaxViterbi.observation_probs_fn``
:1911-1944) and takes ``log(p + tiny)`` inside ``viterbi_librosa_fn``.  These functions do both steps in
one kernel launch (``vit_obs_shaun`` / ``vit_obs_softmax``) and return the ``[..., T, n_bins+1]`` float32
log-emission tensor that ``decode()`` consumes, so logits produced by an acoustic model on the GPU never
visit the host (the ``.cpu().numpy()`` round trip at tonet/for_paper.py:2282).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib


def _check(logits: torch.Tensor, last: int) -> torch.Tensor:
    if not isinstance(logits, torch.Tensor) or logits.device.type != "cuda":
        raise ValueError("logits must be a torch tensor on a ROCm GPU (there is no CPU path)")
    if logits.dtype != torch.float32 or logits.dim() < 2 or logits.shape[-1] != last:
        raise ValueError(f"logits must be float32 [..., frames, {last}]")
    if not logits.is_contiguous():
        raise ValueError("logits must be C-contiguous")
    return logits


def _out(out, shape, device) -> torch.Tensor:
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if out.dtype != torch.float32 or tuple(out.shape) != tuple(shape) or out.device != device or not out.is_contiguous():
        raise ValueError(f"out must be a contiguous float32 {tuple(shape)} tensor on the logits' device")
    return out


def shaun_log_emissions(logits: torch.Tensor, voicing_threshold: float = 0.32, single_side_peak_width: int = 5,
                        p: float = 0.8, scale: float = 2.0, out: torch.Tensor | None = None) -> torch.Tensor:
    """``Viterbi.observation_probs_fn`` + log: logits ``[..., T, n_bins]`` -> log-emissions ``[..., T, n_bins+1]`` (into ``out``
    when given: a caller's emission buffer)."""
    n_bins = logits.shape[-1]
    logits = _check(logits, n_bins)
    assert 0 < voicing_threshold < 1
    out = _out(out, logits.shape[:-1] + (n_bins + 1,), logits.device)
    n = logits.numel() // n_bins
    with torch.cuda.device(logits.device):
        rc = _lib.load().vit_obs_shaun(logits.data_ptr(), n, n_bins, single_side_peak_width,
                                       math.log(voicing_threshold / (1.0 - voicing_threshold)),
                                       math.log(p / (1.0 - p)), scale, out.data_ptr(),
                                       torch.cuda.current_stream(logits.device).cuda_stream)
    _lib.check(rc, "vit_obs_shaun")
    return out


def softmax_log_emissions(logits: torch.Tensor, single_side_peak_width: int = 15, out: torch.Tensor | None = None) -> torch.Tensor:
    """``SoftMaxViterbi.observation_probs_fn`` + log: logits ``[..., T, n_bins+1]`` (column 0 = unvoiced) ->
    log-emissions ``[..., T, n_bins+1]`` with the unvoiced state last."""
    S = logits.shape[-1]
    logits = _check(logits, S)
    out = _out(out, logits.shape, logits.device)
    n = logits.numel() // S
    with torch.cuda.device(logits.device):
        rc = _lib.load().vit_obs_softmax(logits.data_ptr(), n, S - 1, single_side_peak_width, out.data_ptr(),
                                         torch.cuda.current_stream(logits.device).cuda_stream)
    _lib.check(rc, "vit_obs_softmax")
    return out


def softmax_scaled_log_emissions(logits: torch.Tensor, voicing_threshold_prob: float, prior: torch.Tensor | None,
                                 single_side_peak_width: int = 5) -> torch.Tensor:
    """dcnet's ``SoftMaxViterbi.observation_probs_fn`` (dcnet/softmax_viterbi.py:2530-2579) + log: logits ``[..., T, n_bins]``
    -> log of the scaled likelihoods ``[..., T, n_bins+1]`` (unvoiced last; values above 0 are normal).  ``prior``: the
    state prior ``[n_bins+1]`` (unvoiced last) on the GPU, or None for the unscaled variant."""
    n_bins = logits.shape[-1]
    logits = _check(logits, n_bins)
    assert 0 < voicing_threshold_prob < 1
    if prior is not None:
        if prior.device != logits.device or prior.dtype != torch.float32 or tuple(prior.shape) != (n_bins + 1,) or not prior.is_contiguous():
            raise ValueError(f"prior must be a contiguous float32 [{n_bins + 1}] tensor on the logits' device")
    out = torch.empty(logits.shape[:-1] + (n_bins + 1,), dtype=torch.float32, device=logits.device)
    n = logits.numel() // n_bins
    # the reference pads with log(vth / (1. - vth)) computed from an np.float32 scalar (dcnet/softmax_viterbi.py:2546-2548): float32
    # arithmetic under NumPy >= 2 -- the NumPy that generated tests/golden/obs_goldens.npz -- (NumPy 1.x promoted the scalar
    # expression to float64 before the float32 pad: at most one ulp apart); hand the kernel exactly the float32 value
    vth = np.float32(voicing_threshold_prob)
    unvoiced_logit = float(np.log(vth / (np.float32(1) - vth)))
    with torch.cuda.device(logits.device):
        rc = _lib.load().vit_obs_softmax_scaled(logits.data_ptr(), n, n_bins, single_side_peak_width,
                                                unvoiced_logit,
                                                prior.data_ptr() if prior is not None else None, out.data_ptr(),
                                                torch.cuda.current_stream(logits.device).cuda_stream)
    _lib.check(rc, "vit_obs_softmax_scaled")
    return out
