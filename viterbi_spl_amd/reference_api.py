"""Signature-compatible adapters for the reference's Viterbi entry points.

Same names, keyword arguments, argument meaning and assertion behaviour as the reference
functions they replace (paths relative to the reference repo); the forward recursion and the
back-trace run on the GPU through ``ViterbiDecoder``.  The prob -> log step stays on the host in
NumPy float32 exactly where the reference does it, because ``np.log`` is the one operation whose
last bit a GPU cannot be trusted to reproduce.

  family A  viterbi_librosa_c_fn / viterbi_numba_fn(*, transition_matrix, prob_init, probs_st)
            dcnet/tf_viterbi_decoding.py:119-207, dcnet/main.py:2417-2468 (+10 copies)
  family D  viterbi_librosa_fn(*, log_transition_matrix_T, log_prob_init, log_probs_st)
            imm/tf_viterbi.py:75-109
  AOT core  viterbi_numba_core(B, prob_init, probs)   dcnet/aot_viterbi_core.py:8-54
  family B  Viterbi.viterbi_librosa_fn(self, probs_st)          tonet/for_paper.py:1833-1870
  family C  SoftMaxViterbi.viterbi_librosa_fn(self, probs_ts)   tonet/for_paper.py:1999-2037
"""
from __future__ import annotations

import numpy as np
import torch

from .decoder import ViterbiDecoder, get_decoder

_TINY = np.finfo(np.float32).tiny


def _run(logA_T: np.ndarray, log_pi: np.ndarray, logE_ts: np.ndarray, decoder: ViterbiDecoder | None = None) -> np.ndarray:
    dec = decoder if decoder is not None else get_decoder(logA_T, log_pi)
    e = torch.from_numpy(np.ascontiguousarray(logE_ts, dtype=np.float32)).to(dec.device)
    states, _ = dec.decode(e, out_dtype=torch.int64)
    return states.cpu().numpy()


# ----------------------------------------------------------------------------- family D
def viterbi_librosa_fn(*, log_transition_matrix_T, log_prob_init, log_probs_st):
    """Log-domain core (imm/tf_viterbi.py:75-109): emissions are [S, T]; returns int64[T]."""
    B = log_transition_matrix_T
    assert B.flags['C_CONTIGUOUS']
    assert B.dtype == np.float32
    S = len(B)
    assert len(log_prob_init) == S
    assert log_probs_st.dtype == np.float32
    assert log_probs_st.shape[0] == S
    logE = np.require(log_probs_st.T, requirements=['C'])
    return _run(B, np.asarray(log_prob_init, np.float32), logE)


# ----------------------------------------------------------------------------- family A
def _check_probs(transition_matrix, prob_init, probs_st):
    S = len(transition_matrix)
    assert transition_matrix.shape == (S, S)
    assert probs_st.shape[0] == S and probs_st.ndim == 2
    assert np.allclose(np.sum(transition_matrix, axis=1), 1.)
    assert len(prob_init) == S
    assert np.isclose(np.sum(prob_init), 1.)
    return S


def viterbi_librosa_c_fn(*, transition_matrix, prob_init, probs_st):
    """Probabilities in, [S, T] emissions (dcnet/tf_viterbi_decoding.py:156-207); int64[T] out."""
    _check_probs(transition_matrix, prob_init, probs_st)
    tinyp = np.finfo(probs_st.dtype).tiny
    logA_T = np.require(np.log(transition_matrix.T + tinyp), np.float32, ['C'])
    log_pi = np.log(prob_init + tinyp).astype(np.float32)
    logE = np.require(np.log(probs_st.T + tinyp), np.float32, ['C'])
    return _run(logA_T, log_pi, logE)


def viterbi_numba_fn(*, transition_matrix, prob_init, probs_st):
    """The marshalling wrapper around the AOT core (dcnet/tf_viterbi_decoding.py:119-153)."""
    _check_probs(transition_matrix, prob_init, probs_st)
    B = np.require(transition_matrix.T, requirements=['C']).copy()
    probs = np.require(probs_st.T, requirements=['C']).copy()
    return viterbi_numba_core(B, prob_init.copy(), probs)


def viterbi_numba_core(B, prob_init, probs):
    """Stand-in for ``viterbi_numba.core`` ('i8[:](f4[:, ::1], f4[:], f4[:, ::1])',
    dcnet/aot_viterbi_core.py:8-54).  B is [S,S] "target <- source", probs is [T,S].
    Like the Numba core it log-transforms its arguments IN PLACE (:23-25)."""
    assert B.dtype == np.float32 and prob_init.dtype == np.float32 and probs.dtype == np.float32
    assert B.flags['C_CONTIGUOUS'] and probs.flags['C_CONTIGUOUS']
    S = B.shape[0]
    assert prob_init.shape[0] == S and probs.shape[1] == S
    tinyp = np.float32(1.1754944e-38)
    B[:] = np.log(B + tinyp)
    prob_init[:] = np.log(prob_init + tinyp)
    probs[:] = np.log(probs + tinyp)
    return _run(B, prob_init, probs)


# ----------------------------------------------------------------------------- families B / C
class _PreparedViterbi:
    """Parameters logged and transposed once (tonet/for_paper.py:1780-1815)."""

    def __init__(self, transition_matrix, init_probs, num_freq_bins=None, device=None):
        transition_matrix = np.asarray(transition_matrix)
        init_probs = np.asarray(init_probs)
        U = transition_matrix.shape[0] - 1 if num_freq_bins is None else int(num_freq_bins)
        self.num_freq_bins = U
        assert transition_matrix.shape == (U + 1, U + 1)
        assert np.all(np.isclose(np.sum(transition_matrix, axis=1), 1))
        assert init_probs.shape == (U + 1,)
        assert np.isclose(np.sum(init_probs), 1)
        tiny = np.finfo(np.float32).tiny
        t = np.log(transition_matrix + tiny)
        assert not np.any(np.isneginf(t))
        t = np.require(t.T, np.float32, ['C'])
        t.flags['WRITEABLE'] = False
        self.log_transition_matrix_T = t
        p = np.log(init_probs + tiny)
        assert not np.any(np.isneginf(p))
        p = np.require(p, np.float32)
        p.flags['WRITEABLE'] = False
        self.log_ini_probs = p
        self._decoder = ViterbiDecoder(t, p, device)

    @classmethod
    def from_dat_files(cls, transition_file, init_file, **kw):
        from .datfile import load_np_array_from_file_fn
        name, A = load_np_array_from_file_fn(transition_file)
        assert name == 'viterbi_transition_matrix'
        name, pi = load_np_array_from_file_fn(init_file)
        assert name == 'viterbi_init_probs'
        return cls(A, pi, **kw)

    def _post(self, bins):
        n_bins = self.num_freq_bins
        voiced = bins < n_bins
        bins = np.minimum(bins, n_bins - 1)
        return voiced, bins


def _find_peaks(frames_logits, spw):
    """find_peaks_all_at_once_np_fn (tonet/for_paper.py:1714-1731): first maximum of the reflect-padded window."""
    n_frames, n_bins = frames_logits.shape
    padded = np.pad(frames_logits, [(0, 0), (spw, spw)], mode='reflect')
    w = 2 * spw + 1
    are_peaks = np.zeros([n_frames, n_bins], np.bool_)
    for bin_idx in range(n_bins):
        are_peaks[:, bin_idx] = np.argmax(padded[:, bin_idx:bin_idx + w], axis=1) == spw
    return are_peaks


class Viterbi(_PreparedViterbi):
    """Family B (tonet/for_paper.py:1683-1870).  ``viterbi_librosa_fn`` takes F-contiguous [S, T]
    probabilities and logs them IN PLACE, as in the reference.  ``__call__(logits)`` is the whole
    post-processor (:1817-1831): emission builder -> decode -> (voiced, bins)."""

    def __init__(self, transition_matrix, init_probs, voicing_threshold=0.32, num_freq_bins=None, device=None):
        super().__init__(transition_matrix, init_probs, num_freq_bins, device)
        assert 0 < voicing_threshold < 1
        self.voicing_threshold = voicing_threshold
        self.threshold = np.log(voicing_threshold / (1. - voicing_threshold))
        self.single_side_peak_width = 5

    @staticmethod
    def expit(s):
        if s > 0:
            return 1. / (1. + np.exp(-s))
        p = np.exp(s)
        return p / (1. + p)

    def observation_probs_fn(self, logits):
        """Host NumPy builder, operation for operation the reference's (:1733-1778): bit-exact, slow."""
        assert isinstance(logits, np.ndarray) and logits.dtype == np.float32
        n_frames, n_freq_bins = logits.shape
        assert n_freq_bins == self.num_freq_bins
        offset = np.log(0.8 / (1. - 0.8))
        scale = 2.
        melodies_frames = np.zeros([n_freq_bins + 1, n_frames], np.float32, order='F')
        are_peaks = _find_peaks(logits, self.single_side_peak_width)
        for frame_idx in range(n_frames):
            peak_indices = np.where(are_peaks[frame_idx])[0]
            if len(peak_indices) == 0:
                melodies_frames[-1, frame_idx] = 1
                continue
            peak_logits = logits[frame_idx][peak_indices]
            g = peak_logits[np.argmax(peak_logits)]
            if g >= self.threshold:
                s = scale * (g - self.threshold) + offset
            else:
                s = scale * (g - self.threshold) - offset
            p_voiced = Viterbi.expit(s)
            peak_logits -= g
            np.exp(peak_logits, out=peak_logits)
            t = p_voiced / np.sum(peak_logits)
            np.multiply(peak_logits, t, out=peak_logits)
            melodies_frames[peak_indices, frame_idx] = peak_logits
            melodies_frames[-1, frame_idx] = 1. - p_voiced
        assert np.all(np.isclose(np.sum(melodies_frames, axis=0), 1))
        return melodies_frames

    def viterbi_librosa_fn(self, probs_st):
        S = self.num_freq_bins + 1
        assert probs_st.shape[0] == S
        assert probs_st.dtype == np.float32
        assert probs_st.flags['F_CONTIGUOUS']
        np.add(probs_st, _TINY, out=probs_st)
        np.log(probs_st, out=probs_st)
        probs = np.require(probs_st.T, np.float32, ['C'])
        return _run(self.log_transition_matrix_T, self.log_ini_probs, probs, self._decoder)

    def __call__(self, logits, exact_emissions=False):
        """logits: [n_frames, n_bins] float32 (NumPy, or a torch tensor already on the GPU).
        exact_emissions=True builds the observation probabilities on the host exactly like the reference
        (bit-exact end to end); the default builds them on the GPU (vit_obs_shaun: same decisions, exp/log
        within a few ulp) so that GPU-resident logits never visit the host."""
        if exact_emissions:
            lg = logits.detach().cpu().numpy() if isinstance(logits, torch.Tensor) else logits
            return self._post(self.viterbi_librosa_fn(self.observation_probs_fn(lg)))
        from .emissions import shaun_log_emissions
        lg = logits if isinstance(logits, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(logits, np.float32))
        lg = lg.to(self._decoder.device).contiguous()
        logE = shaun_log_emissions(lg, self.voicing_threshold, self.single_side_peak_width)
        states, _ = self._decoder.decode(logE, out_dtype=torch.int64)
        return self._post(states.cpu().numpy())


class SoftMaxViterbi(_PreparedViterbi):
    """Family C (tonet/for_paper.py:1873-2037): C-contiguous [T, S] probabilities (values may
    exceed 1 for scaled likelihoods), logged IN PLACE.  ``__call__(logits)`` as above."""

    def __init__(self, transition_matrix, init_probs, num_freq_bins=None, device=None):
        super().__init__(transition_matrix, init_probs, num_freq_bins, device)
        self.single_side_peak_width = 15

    def observation_probs_fn(self, logits):
        """Host NumPy builder following :1911-1944 (unvoiced logit first, rolled to the last state)."""
        assert isinstance(logits, np.ndarray) and logits.dtype == np.float32 and logits.ndim == 2
        assert logits.shape[1] == self.num_freq_bins + 1 and logits.flags['C_CONTIGUOUS']
        n_frames, n_bins = len(logits), self.num_freq_bins
        prob_ts = np.zeros([n_frames, 1 + n_bins], np.float32)
        are_peaks_ts = np.zeros([n_frames, n_bins + 1], np.bool_)
        are_peaks_ts[:, 0] = True
        are_peaks_ts[:, 1:] = _find_peaks(logits[:, 1:], self.single_side_peak_width)
        for frame_idx, are_peaks in enumerate(are_peaks_ts):
            peak_indices = np.where(are_peaks)[0]
            if len(peak_indices) == 1:
                prob_ts[frame_idx, 0] = 1
                continue
            peak_logits = logits[frame_idx, peak_indices]
            peak_logits = np.exp(peak_logits - np.max(peak_logits))
            prob_ts[frame_idx, peak_indices] = peak_logits / np.sum(peak_logits)
        assert np.allclose(np.sum(prob_ts, axis=1), 1)
        return np.roll(prob_ts, shift=-1, axis=1)

    def viterbi_librosa_fn(self, probs_ts):
        S = self.num_freq_bins + 1
        assert probs_ts.ndim == 2
        assert probs_ts.shape[1] == S
        assert probs_ts.dtype == np.float32
        assert probs_ts.flags['C_CONTIGUOUS']
        np.add(probs_ts, _TINY, out=probs_ts)
        np.log(probs_ts, out=probs_ts)
        return _run(self.log_transition_matrix_T, self.log_ini_probs, probs_ts, self._decoder)

    def __call__(self, logits, exact_emissions=False):
        if exact_emissions:
            lg = logits.detach().cpu().numpy() if isinstance(logits, torch.Tensor) else logits
            return self._post(self.viterbi_librosa_fn(self.observation_probs_fn(np.ascontiguousarray(lg))))
        from .emissions import softmax_log_emissions
        lg = logits if isinstance(logits, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(logits, np.float32))
        lg = lg.to(self._decoder.device).contiguous()
        logE = softmax_log_emissions(lg, self.single_side_peak_width)
        states, _ = self._decoder.decode(logE, out_dtype=torch.int64)
        return self._post(states.cpu().numpy())
