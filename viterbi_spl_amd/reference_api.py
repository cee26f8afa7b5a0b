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
  TF graph  viterbi_tf_fn(transition_matrix, prob_init, probs_st) -> int32        dcnet/tf_viterbi_decoding.py:23-72
  TF eager  tf_viterbi_librosa_fn(*, tf_log_transition_matrix_T, ...) -> int32    imm/tf_viterbi.py:8-72
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
    # float32 inputs only: with float64 parameters the reference's NumPy code (dcnet/tf_viterbi_decoding.py:183-197) adds a
    # float32 delta to float64 log-probabilities and rounds once, which float32 kernels do not reproduce; the compiled core it
    # replaces takes f4 arrays only ('i8[:](f4[:, ::1], f4[:], f4[:, ::1])', dcnet/aot_viterbi_core.py:8)
    assert transition_matrix.dtype == np.float32 and np.asarray(prob_init).dtype == np.float32 and probs_st.dtype == np.float32
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


# ----------------------------------------------------------------------------- the TensorFlow variants (rows a3 / a7)
def _to_numpy(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return x.numpy() if hasattr(x, 'numpy') else np.asarray(x)       # tf.Tensor / tf.Variable expose .numpy() as well


def viterbi_tf_fn(transition_matrix, prob_init, probs_st):
    """Call surface of the TF-graph variant (dcnet/tf_viterbi_decoding.py:23-72): positional probabilities in (tensors or
    arrays, converted to float32 as ``tf.convert_to_tensor(..., tf.float32)`` does), S hard-wired to 321 there, the
    ``tf.debugging.assert_near`` checks on the row sums, ``int32[T]`` states out.  Same float32 recursion as family A; the
    log is NumPy's here (TensorFlow's own log may differ in the last bit: that variant's outputs are parity unpinned)."""
    A = np.asarray(_to_numpy(transition_matrix), np.float32)
    pi = np.asarray(_to_numpy(prob_init), np.float32)
    P = np.asarray(_to_numpy(probs_st), np.float32)
    assert np.allclose(np.sum(A, axis=1), 1., atol=1e-6 * A.shape[1]) and np.isclose(np.sum(pi), 1., atol=1e-5)
    return viterbi_librosa_c_fn(transition_matrix=A, prob_init=pi, probs_st=np.asfortranarray(P)).astype(np.int32)


def tf_viterbi_librosa_fn(*, tf_log_transition_matrix_T, tf_log_prob_init, tf_or_np_log_probs_st):
    """Call surface of the TF-eager log-domain variant (imm/tf_viterbi.py:8-72): ``[S,S]`` target <- source log-matrix,
    ``[S]`` log-prior, ``[S,T]`` log-emissions (tensor or array, taken as float32), ``int32[T]`` states out."""
    B = np.require(_to_numpy(tf_log_transition_matrix_T), np.float32, ['C'])
    prob_init = np.asarray(_to_numpy(tf_log_prob_init), np.float32)
    probs = np.asarray(_to_numpy(tf_or_np_log_probs_st), np.float32)
    S = len(B)
    assert B.shape == (S, S)
    assert len(prob_init) == S
    assert probs.shape == (S, probs.shape[1])
    return viterbi_librosa_fn(log_transition_matrix_T=B, log_prob_init=prob_init, log_probs_st=probs).astype(np.int32)


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
        self.ini_probs = init_probs
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

    # ---- the post-processor on the GPU: logits -> log-emissions -> decode -> (voiced, bins)
    def _log_emissions(self, logits: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def decode_logits(self, logits, note_range=None):
        """Whole post-processor with everything on the GPU.  logits: NumPy or torch, ``[frames, columns]``.
        Returns torch tensors on the decoder's device: ``(voiced bool[T], bins int32[T])``, plus ``notes float32[T]``
        (``note_range[bins]``, 0 where unvoiced: tonet/for_paper.py:2106-2115, :2207) when ``note_range`` is given."""
        lg = logits if isinstance(logits, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(logits, np.float32))
        lg = lg.to(self._decoder.device).contiguous()
        states, _ = self._decoder.decode(self._log_emissions(lg), out_dtype=torch.int32)
        if note_range is None:
            return self._decoder.voicing(states, self.num_freq_bins)
        return self._decoder.voicing_notes(states, note_range, self.num_freq_bins)

    def decode_recordings(self, logits_list):
        """Many recordings in ONE pass: what the reference does recording by recording (one ``viterbi(logits)`` call each,
        tonet/for_paper.py:2304-2319) as a single emission-builder launch over the concatenated frames, a single packed decode
        (``vit_decode_packed``: no padding, forward slots packed by length) and a single voicing map.  ``logits_list``: NumPy or torch
        ``[T_b, columns]`` per recording.  Returns a list of ``(voiced bool[T_b], bins int32[T_b])`` torch tensors on the GPU;
        every pair equals ``decode_logits`` of that recording alone (the builders work frame by frame, the decode is bit-identical)."""
        dev = self._decoder.device
        parts = [(lg if isinstance(lg, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(lg, np.float32))).to(dev) for lg in logits_list]
        if not parts:
            return []
        offsets = np.concatenate([[0], np.cumsum([int(x.shape[0]) for x in parts])]).astype(np.int64)
        assert all(x.dim() == 2 and x.shape[1] == parts[0].shape[1] and x.shape[0] >= 1 for x in parts)
        E = self._log_emissions(torch.cat(parts, dim=0).contiguous())
        states, _ = self._decoder.decode_packed(E, offsets, out_dtype=torch.int32)
        voiced, bins = self._decoder.voicing(states, self.num_freq_bins)
        return [(voiced[offsets[b]:offsets[b + 1]], bins[offsets[b]:offsets[b + 1]]) for b in range(len(parts))]

    def __call__(self, logits):
        """``viterbi(logits) -> (voiced bool[T], bins int64[T])`` as NumPy arrays, the reference's call surface
        (tonet/for_paper.py:1817-1831, :2309).  The observation probabilities are built on the GPU (``vit_obs_*``: peak
        picking, voicing decision and structural zeros exact; exp / log / sums within a few ulp of NumPy's).  Measured
        against the host-exact pipeline (tests/test_gpu_parity.py::test_postprocessor_agreement_at_full_length): T = 30000,
        five seeds per builder ("shaun", softmax, scaled likelihoods) -- 100 % of the frames of all fifteen songs decode to
        the reference's states; the test's bar is 99.9 % per song with every differing frame on a path whose exact score
        ties the reference's to 1e-6.  Callers that need the reference's bits by construction build the probabilities
        themselves and call :meth:`viterbi_librosa_fn`, which is bit-exact."""
        voiced, bins = self.decode_logits(logits)
        return voiced.cpu().numpy(), bins.cpu().numpy().astype(np.int64)


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

    def viterbi_librosa_fn(self, probs_st):
        S = self.num_freq_bins + 1
        assert probs_st.shape[0] == S
        assert probs_st.dtype == np.float32
        assert probs_st.flags['F_CONTIGUOUS']
        np.add(probs_st, _TINY, out=probs_st)
        np.log(probs_st, out=probs_st)
        probs = np.require(probs_st.T, np.float32, ['C'])
        return _run(self.log_transition_matrix_T, self.log_ini_probs, probs, self._decoder)

    def _log_emissions(self, logits):
        from .emissions import shaun_log_emissions
        return shaun_log_emissions(logits, self.voicing_threshold, self.single_side_peak_width)


class SoftMaxViterbi(_PreparedViterbi):
    """Family C (tonet/for_paper.py:1873-2037): C-contiguous [T, S] probabilities (values may
    exceed 1 for scaled likelihoods), logged IN PLACE.  ``__call__(logits)`` as above; logits carry the unvoiced
    column first ([T, n_bins+1])."""

    def __init__(self, transition_matrix, init_probs, num_freq_bins=None, device=None):
        super().__init__(transition_matrix, init_probs, num_freq_bins, device)
        self.single_side_peak_width = 15

    def viterbi_librosa_fn(self, probs_ts):
        S = self.num_freq_bins + 1
        assert probs_ts.ndim == 2
        assert probs_ts.shape[1] == S
        assert probs_ts.dtype == np.float32
        assert probs_ts.flags['C_CONTIGUOUS']
        np.add(probs_ts, _TINY, out=probs_ts)
        np.log(probs_ts, out=probs_ts)
        return _run(self.log_transition_matrix_T, self.log_ini_probs, probs_ts, self._decoder)

    def _log_emissions(self, logits):
        from .emissions import softmax_log_emissions
        return softmax_log_emissions(logits, self.single_side_peak_width)


class ScaledSoftMaxViterbi(SoftMaxViterbi):
    """dcnet's SoftMaxViterbi (dcnet/softmax_viterbi.py:2486-2674; same class in ftanet / jdc / msnet): logits carry
    the n_bins pitch columns only, the unvoiced logit is the voicing-threshold logit, peaks use a +/-5-bin window and --
    with ``scaled=True`` -- every softmax probability is divided by its state prior ("scaled likelihood": values far
    above 1, i.e. positive log-emissions)."""

    def __init__(self, transition_matrix, init_probs, voicing_threshold_prob, scaled, num_freq_bins=None, device=None):
        super().__init__(transition_matrix, init_probs, num_freq_bins, device)
        assert 0 < voicing_threshold_prob < 1
        self.voicing_threshold_prob = float(voicing_threshold_prob)
        self.scaled = bool(scaled)
        self.single_side_peak_width = 5
        if self.scaled:
            assert self.ini_probs.min() > 0.3 / (self.num_freq_bins * 10)       # dcnet/softmax_viterbi.py:2536
        self._prior_dev = torch.from_numpy(np.ascontiguousarray(self.ini_probs, np.float32)).to(self._decoder.device) if self.scaled else None

    def _log_emissions(self, logits):
        from .emissions import softmax_scaled_log_emissions
        return softmax_scaled_log_emissions(logits, self.voicing_threshold_prob, self._prior_dev, self.single_side_peak_width)


class RecordingAccumulator:
    """Per-recording logits kept on the GPU (replaces tonet/for_paper.py:2282-2309, where every batch of snippets is
    copied to the host, transposed, and the recording concatenated in NumPy before ``viterbi(logits)``).

        acc = RecordingAccumulator(viterbi, max_frames)          # viterbi: Viterbi ("shaun"), SoftMaxViterbi or ScaledSoftMaxViterbi
        for batch in recording: acc.append(pitch_logits, padded_frames)   # [n_snippets, channels, frames] on the GPU
        voiced, bins, notes = acc.finish(note_range)             # torch tensors on the GPU

    ``append`` transposes the snippets into time-major rows directly behind the rows already held (one kernel:
    ``vit_snippets_append``; "shaun" rows are relative to the unvoiced channel, :2296-2297), ``finish`` runs emission
    builder, decoder, voicing map and the bin -> note gather without a host visit.  Channels per snippet: n_bins + 1 with the
    unvoiced channel first for Viterbi (subtracted and dropped) and SoftMaxViterbi (kept); the n_bins pitch channels only for
    ScaledSoftMaxViterbi, whose unvoiced logit is a constant (dcnet/softmax_viterbi.py:2548-2550)."""

    def __init__(self, viterbi: _PreparedViterbi, max_frames: int):
        self.viterbi = viterbi
        if isinstance(viterbi, Viterbi):
            self.mode, self.channels, self.cols = 0, viterbi.num_freq_bins + 1, viterbi.num_freq_bins
        elif isinstance(viterbi, ScaledSoftMaxViterbi):       # (a SoftMaxViterbi subclass: test it first)
            self.mode, self.channels, self.cols = 1, viterbi.num_freq_bins, viterbi.num_freq_bins
        elif isinstance(viterbi, SoftMaxViterbi):
            self.mode, self.channels, self.cols = 1, viterbi.num_freq_bins + 1, viterbi.num_freq_bins + 1
        else:
            raise TypeError("RecordingAccumulator needs a Viterbi, SoftMaxViterbi or ScaledSoftMaxViterbi")
        dev = viterbi._decoder.device
        self._rows = torch.empty((int(max_frames), self.cols), dtype=torch.float32, device=dev)
        self.n_frames = 0

    def reset(self):
        self.n_frames = 0

    def append(self, pitch_logits: torch.Tensor, padded_frames: int = 0):
        from . import _lib
        if not isinstance(pitch_logits, torch.Tensor) or pitch_logits.device != self._rows.device:
            raise ValueError("pitch_logits must be a torch tensor on the decoder's device")
        if pitch_logits.dtype != torch.float32 or pitch_logits.dim() != 3 or pitch_logits.shape[1] != self.channels:
            raise ValueError(f"pitch_logits must be float32 [snippets, {self.channels}, frames]")
        x = pitch_logits.contiguous()
        n, C, F = x.shape
        rows = n * F - int(padded_frames)
        if rows < 0 or self.n_frames + rows > self._rows.shape[0]:
            raise ValueError("recording longer than max_frames (or padded_frames out of range)")
        with torch.cuda.device(x.device):
            rc = _lib.load().vit_snippets_append(x.data_ptr(), n, C, F, self.mode, self._rows[self.n_frames:].data_ptr(), rows,
                                                 torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(rc, "vit_snippets_append")
        self.n_frames += rows

    def logits(self) -> torch.Tensor:
        """The recording so far: ``[n_frames, columns]`` on the GPU (a view)."""
        return self._rows[: self.n_frames]

    def finish(self, note_range=None):
        out = self.viterbi.decode_logits(self.logits(), note_range)
        self.reset()
        return out
