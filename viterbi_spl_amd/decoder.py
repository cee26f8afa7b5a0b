"""Host side of the MI355X Viterbi decoder: torch tensors in, ctypes C ABI underneath.

``decode(emission_logits, transition_matrix, init_probs)`` is the call surface named by
BASELINE.json; it follows the reference's log-domain core
``viterbi_librosa_fn(*, log_transition_matrix_T, log_prob_init, log_probs_st)``
(imm/tf_viterbi.py:75-109 in the reference repo) with these conventions:

* ``transition_matrix`` -- LOG-domain, transposed: ``[S, S]`` float32 with row ``j`` = log-probabilities
  INTO target ``j`` (what the reference calls ``log_transition_matrix_T`` / ``B``,
  tonet/for_paper.py:1798-1815);
* ``init_probs``        -- LOG-domain ``[S]`` float32 (``log_prob_init``);
* ``emission_logits``   -- LOG-domain ``[T, S]`` or ``[B, T, S]`` float32/float16, C-contiguous, on the GPU
  (time-major rows, the layout the reference hands its core: dcnet/tf_viterbi_decoding.py:145).

PyTorch is plumbing only here (device memory, streams); all arithmetic happens in
libviterbi_hip.so.  Nothing in this module falls back to the CPU.
"""
from __future__ import annotations

import ctypes
import hashlib
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib


def _as_host_f32(x, shape=None) -> np.ndarray:
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    a = np.ascontiguousarray(x, dtype=np.float32)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


class ViterbiDecoder:
    """A transition matrix + prior analysed once and resident on one GPU.

    Mirrors the reference's ``Viterbi.__init__`` (tonet/for_paper.py:1685-1701): parameters are
    prepared once, then many songs are decoded.
    """

    def __init__(self, log_transition_matrix_T, log_prob_init, device: Optional[torch.device] = None):
        lib = _lib.load()
        A = _as_host_f32(log_transition_matrix_T)
        if A.ndim != 2 or A.shape[0] != A.shape[1]:
            raise ValueError("log_transition_matrix_T must be [S, S]")
        S = A.shape[0]
        pi = _as_host_f32(log_prob_init, (S,))
        if np.isnan(A).any() or np.isnan(pi).any():
            raise ValueError("NaN in HMM parameters")
        self.S = S
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise ValueError("ViterbiDecoder needs a GPU device (there is no CPU path)")
        self._plan = ctypes.c_void_p()
        _lib.check(lib.vit_plan_create(A.ctypes.data, pi.ctypes.data, S, ctypes.byref(self._plan)), "vit_plan_create")
        info = _lib.PlanInfo()
        _lib.check(lib.vit_plan_query(self._plan, ctypes.byref(info)), "vit_plan_query")
        self.info = {
            "S": int(info.S), "banded_ok": bool(info.banded_ok), "n_extras": int(info.n_extras),
            "n_dense_rows": int(info.reserved[0]), "floor_ok": bool(info.reserved[1]), "lo_affine": bool(info.reserved[2] & 1), "pair_ok": bool(info.reserved[2] & 2), "step_ok": bool(info.reserved[2] & 4), "wave_ok": bool(info.reserved[2] & 8), "max_window": int(info.max_window),
            "group_window": int(info.group_window), "row_constant": float(info.consts[0]),
            "extras": [int(info.extras[k]) for k in range(int(info.n_extras))],
        }
        nbytes = int(lib.vit_plan_image_bytes(self._plan))
        with torch.cuda.device(self.device):
            self._image = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._image_ptr = (self._image.data_ptr() + 255) & ~255
            stream = torch.cuda.current_stream(self.device)
            _lib.check(lib.vit_plan_upload(self._plan, self._image_ptr, nbytes, stream.cuda_stream), "vit_plan_upload")
            stream.synchronize()  # host image is pageable: make the copy visible before anything else
        self._ws: Optional[torch.Tensor] = None
        self._ws_slots: dict = {}
        self._options: dict = {}

    def __del__(self):
        try:
            if getattr(self, "_plan", None) is not None and self._plan.value:
                _lib.load().vit_plan_destroy(self._plan)
                self._plan = ctypes.c_void_p()
        except Exception:
            pass

    def set_option(self, key: str, value: int) -> None:
        """Kernel-selection override (``vit_plan_set_option``; keys in include/viterbi_hip.h).  Every setting decodes the
        same bits; ``set_option("reset", 0)`` restores the defaults."""
        _lib.check(_lib.load().vit_plan_set_option(self._plan, key.encode(), int(value)), f"vit_plan_set_option({key})")
        if key == "reset":
            self._options = {}
        else:
            self._options[key] = int(value)

    def chunks_beside_forward(self, B: int) -> int:
        """``bt_chunks`` for the two-stream schedule (the back-trace of batch i beside the forward pass of batch i + 1): while one
        song per workgroup leaves compute units idle (B below the CU count), the back-trace gets the chunks that fill THOSE units
        -- sixteen waves each -- instead of every unit of the chip, where its workgroups sit next to the latency-bound forward
        workgroups and slow them down (B = 128: 10.04 instead of 10.22 ms per step, scripts/overlap_ab.py).  0 = the library's own
        count (what a single stream should use: the back-trace alone is fastest with every unit)."""
        n_cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        if B <= 0 or 2 * (n_cus - B) < B:                  # fewer idle units than half the songs: the idle units alone would serialise the
            return 0                                        # back-trace (B = 250 on 256 units: ONE chunk per song) -- the library's own count
        return max(4, min(32, 16 * (n_cus - B) // B))      # (32: the library's cap, kBtMaxChunks)

    FAMILIES = {1: "dense", 2: "group", 3: "wave"}

    def forward_family(self, B: int, algo: str = "auto") -> str:
        """The forward kernel family the library launches for this batch size (``vit_forward_family``): "dense" (any matrix, or
        the step-structured kernel), "group" (banded, one song per workgroup) or "wave" (banded, one song per wavefront).  The
        thresholds scale with the device's compute units -- ask, do not guess."""
        fam = int(_lib.load().vit_forward_family(self._plan, int(B), _lib.ALGO[algo]))
        if fam < 0:
            _lib.check(fam, "vit_forward_family")
        return self.FAMILIES[fam]

    # ------------------------------------------------------------------ workspace
    def workspace_bytes(self, B: int, T: int, algo: Optional[str] = None) -> int:
        """Bytes of workspace a [B,T,S] decode needs: for one `algo` (``vit_workspace_bytes_for`` -- the wave form keeps
        half a history), or enough for any of them (``vit_workspace_bytes``) when `algo` is None."""
        lib = _lib.load()
        if algo is None:
            return int(lib.vit_workspace_bytes(self._plan, B, T))
        need = int(lib.vit_workspace_bytes_for(self._plan, B, T, _lib.ALGO[algo]))
        if need == 0 and B > 0:
            raise _lib.ViterbiHipError(f"algo {algo!r} is not available for this plan")
        return need

    def history_mode(self, B: int, T: int, algo: str = "auto") -> str:
        """"half" if the forward kernel `algo` resolves to for this batch stores only the delta rows of even frames (the wave form;
        the back-trace rebuilds the odd ones), else "full" -- read off the workspace the library asks for."""
        need = self.workspace_bytes(B, T, algo)
        return "half" if need * 1.5 < self.workspace_bytes(B, T) else "full"

    def _workspace(self, B: int, T: int, slot: int = 0, algo: Optional[str] = None) -> Tuple[int, int]:
        """Workspace `slot` (callers that overlap the back-trace of one batch with the forward pass of the next on
        two streams give each batch in flight its own slot; slot 0 is the default).  A buffer that is already large
        enough is kept, so alternating algos on one decoder settle on the largest need."""
        need = self.workspace_bytes(B, T, algo)
        ws = self._ws if slot == 0 else self._ws_slots.get(slot)
        if ws is None or ws.numel() < need + 256:
            ws = None
            if slot == 0:
                self._ws = None
            else:
                self._ws_slots.pop(slot, None)
            ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            if slot == 0:
                self._ws = ws
            else:
                self._ws_slots[slot] = ws
        return (ws.data_ptr() + 255) & ~255, ws.numel() - 256

    # ------------------------------------------------------------------ checks
    def _check_emissions(self, logE: torch.Tensor) -> Tuple[torch.Tensor, bool, int]:
        if not isinstance(logE, torch.Tensor):
            raise TypeError("emission_logits must be a torch tensor on the GPU")
        if logE.device != self.device:
            raise ValueError(f"emission_logits is on {logE.device}, decoder is on {self.device}")
        if logE.dtype == torch.float32:
            dt = _lib.VIT_F32
        elif logE.dtype == torch.float16:
            dt = _lib.VIT_F16
        else:
            raise TypeError("emission_logits must be float32 or float16")
        single = logE.dim() == 2
        if single:
            logE = logE.unsqueeze(0)
        if logE.dim() != 3 or logE.shape[2] != self.S or logE.shape[1] < 1:
            raise ValueError(f"emission_logits must be [B,T,{self.S}] or [T,{self.S}] with T >= 1")
        if not logE.is_contiguous():
            raise ValueError("emission_logits must be C-contiguous (the reference requires it too)")
        return logE, single, dt

    # ------------------------------------------------------------------ decode
    def decode_into(self, logE: torch.Tensor, states: torch.Tensor, loglik: Optional[torch.Tensor] = None,
                    lengths: Optional[torch.Tensor] = None, algo: str = "auto", phase: str = "both", slot: int = 0) -> None:
        """Enqueue a decode on the current stream.  states: int32 [B,T]; loglik: float32 [B].
        `phase` "forward" / "backtrace" run the two halves separately (same `slot` = same workspace).

        Lifetime rule of the split form: `logE` must be the SAME tensor, unchanged, in both phases -- with the wave form's half
        history (``set_option("wave_history", 2)``) the back-trace reads 32 emission values of every odd frame again.  The
        back-trace phase hands the pointer to the library (``vit_backtrace_checked``), which refuses a tensor other than the one
        its forward pass decoded ("invalid argument"); a caller that double-buffers its emissions must keep a batch's buffer
        untouched until that batch's back-trace has run."""
        lib = _lib.load()
        logE, _, dt = self._check_emissions(logE)
        B, T, _ = logE.shape
        if states.dtype != torch.int32 or tuple(states.shape) != (B, T) or not states.is_contiguous():
            raise ValueError("states must be a contiguous int32 [B,T] tensor")
        if loglik is not None and (loglik.dtype != torch.float32 or tuple(loglik.shape) != (B,)):
            raise ValueError("loglik must be float32 [B]")
        if lengths is not None:
            if lengths.dtype != torch.int64 or tuple(lengths.shape) != (B,) or lengths.device != self.device:
                raise ValueError("lengths must be an int64 [B] tensor on the decoder's device")
        ws_ptr, ws_bytes = self._workspace(B, T, slot, algo)
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            len_ptr = lengths.data_ptr() if lengths is not None else None
            ll_ptr = loglik.data_ptr() if loglik is not None else None
            a = _lib.ALGO[algo]
            if phase == "both":
                rc = lib.vit_decode(self._plan, logE.data_ptr(), dt, B, T, len_ptr, ws_ptr, ws_bytes,
                                    states.data_ptr(), ll_ptr, a, stream)
            elif phase == "forward":
                rc = lib.vit_forward(self._plan, logE.data_ptr(), dt, B, T, len_ptr, ws_ptr, ws_bytes, ll_ptr, a, stream)
            elif phase == "backtrace":
                rc = lib.vit_backtrace_checked(self._plan, logE.data_ptr(), dt, B, T, len_ptr, ws_ptr, ws_bytes, states.data_ptr(), a, stream)
            else:
                raise ValueError(phase)
        _lib.check(rc, f"vit_{phase if phase != 'both' else 'decode'}")

    COUNTERS = ("tiles_fetched", "span_misses", "whole_row_evaluations", "odd_rows_rebuilt", "chunks_repaired", "frames_repaired")

    def backtrace_counters(self, B: int, T: int, slot: int = 0) -> dict:
        """Event counts of the last back-trace that ran on workspace `slot` (``vit_backtrace_counters``), summed over the songs.
        Synchronises the device."""
        ws = self._ws if slot == 0 else self._ws_slots.get(slot)
        if ws is None:
            raise _lib.ViterbiHipError("no workspace in this slot")
        base = (ws.data_ptr() + 255) & ~255
        off, n = ctypes.c_size_t(), ctypes.c_int32()
        _lib.check(_lib.load().vit_backtrace_counters(self._plan, B, T, base, ctypes.byref(off), ctypes.byref(n)), "vit_backtrace_counters")
        torch.cuda.synchronize(self.device)
        start = base - ws.data_ptr() + off.value
        ct = ws[start:start + B * n.value * 4].view(torch.int32).view(B, n.value).sum(dim=0).cpu().tolist()
        return {k: int(ct[i]) for i, k in enumerate(self.COUNTERS)}

    def plan_workspace(self, B: int, T: int, algo: str = "auto", max_workspace_bytes: Optional[int] = None) -> dict:
        """How a [B, T, S] batch is decoded under a workspace budget (bytes; None = whatever `algo` asks for):

        * ``{"mode": "full"}``  -- the normal decode fits (one delta row per frame, or what the options already select);
        * ``{"mode": "half"}``  -- the wave form with the rows of even frames only (``wave_history`` 2: half the workspace, a
          slower back-trace), where the plan has it;
        * ``{"mode": "checkpointed", "segment_frames": K}`` -- ``vit_decode_checkpointed`` with the largest K that fits (about
          twice the forward work; wave-form plans).

        Raises ViterbiHipError when nothing fits.  Every mode decodes the same bits."""
        need = self.workspace_bytes(B, T, algo)
        if max_workspace_bytes is None or need <= max_workspace_bytes:
            return {"mode": "full", "workspace_bytes": need}
        if self.info["wave_ok"] and algo in ("auto", "banded", "wave") and T >= 2:
            lib = _lib.load()
            prev = self._options.get("wave_history", 0)
            _lib.check(lib.vit_plan_set_option(self._plan, b"wave_history", 2), "vit_plan_set_option(wave_history)")
            half = int(lib.vit_workspace_bytes_for(self._plan, B, T, _lib.ALGO["wave"]))
            _lib.check(lib.vit_plan_set_option(self._plan, b"wave_history", prev), "vit_plan_set_option(wave_history)")
            if 0 < half <= max_workspace_bytes:
                return {"mode": "half", "workspace_bytes": half}
            K = 8192
            while K >= 64:
                ck = int(lib.vit_workspace_bytes_checkpointed(self._plan, B, T, K))
                if 0 < ck <= max_workspace_bytes:       # the largest segment that fits: fewest launches
                    return {"mode": "checkpointed", "segment_frames": K, "workspace_bytes": ck}
                K //= 2
        raise _lib.ViterbiHipError(f"no decode of a [{B}, {T}, {self.S}] batch fits a workspace of {max_workspace_bytes} bytes "
                                   f"(the normal decode needs {need})")

    def decode(self, emission_logits: torch.Tensor, lengths: Optional[torch.Tensor] = None, algo: str = "auto",
               out_dtype: torch.dtype = torch.int64, max_workspace_bytes: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns (states [B,T] or [T] of ``out_dtype`` (reference: int64), loglik float32 [B] or scalar).

        ``max_workspace_bytes``: a budget for the delta history (the reference keeps its work buffers for ONE song,
        tonet/for_paper.py:1852-1853; a batch of 2048 full-length songs asks for 94 GB).  The decode falls from the full history to
        the wave form's half history to the checkpointed decode (``plan_workspace``) instead of running out of memory; the result
        is the same in every mode."""
        logE, single, _ = self._check_emissions(emission_logits)
        B, T, _ = logE.shape
        if B > 0 and max_workspace_bytes is not None:
            mode = self.plan_workspace(B, T, algo, int(max_workspace_bytes))
            if mode["mode"] == "checkpointed":
                st, ll = self.decode_checkpointed(logE, segment_frames=mode["segment_frames"], lengths=lengths, out_dtype=out_dtype)
                return (st[0], ll[0]) if single else (st, ll)
            if mode["mode"] == "half":
                prev = self._options.get("wave_history", 0)
                self.set_option("wave_history", 2)
                self._ws = None                              # (a larger buffer kept from an earlier decode would defeat the budget)
                try:
                    st, ll = self.decode(logE, lengths=lengths, algo="wave", out_dtype=out_dtype)
                finally:
                    self.set_option("wave_history", prev)
                return (st[0], ll[0]) if single else (st, ll)
        states = torch.empty((B, T), dtype=torch.int32, device=self.device)
        loglik = torch.empty((B,), dtype=torch.float32, device=self.device)
        if B > 0:
            self.decode_into(logE, states, loglik, lengths, algo)
        if out_dtype != torch.int32:
            states = states.to(out_dtype)
        return (states[0], loglik[0]) if single else (states, loglik)

    # ------------------------------------------------------------------ bounded-workspace decode
    def workspace_bytes_checkpointed(self, B: int, T: int, segment_frames: int) -> int:
        need = int(_lib.load().vit_workspace_bytes_checkpointed(self._plan, B, T, int(segment_frames)))
        if need == 0 and B > 0:
            raise _lib.ViterbiHipError("checkpointed decode needs a plan with the wave form and segment_frames >= 64")
        return need

    def decode_checkpointed(self, emission_logits: torch.Tensor, segment_frames: int = 1024, lengths: Optional[torch.Tensor] = None,
                            out_dtype: torch.dtype = torch.int64, workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """``decode`` with a bounded workspace (``vit_decode_checkpointed``): about ``T / segment_frames + segment_frames`` delta
        rows per song instead of ``T`` -- the reference keeps its work buffers for one song at a time
        (tonet/for_paper.py:1852-1853).  Same states, same log-likelihood; about twice the forward work.  ``workspace``: an
        optional uint8 tensor of at least ``workspace_bytes_checkpointed(...) + 256`` bytes to decode in."""
        lib = _lib.load()
        logE, single, dt = self._check_emissions(emission_logits)
        B, T, _ = logE.shape
        states = torch.empty((B, T), dtype=torch.int32, device=self.device)
        loglik = torch.empty((B,), dtype=torch.float32, device=self.device)
        if lengths is not None and (lengths.dtype != torch.int64 or tuple(lengths.shape) != (B,) or lengths.device != self.device):
            raise ValueError("lengths must be an int64 [B] tensor on the decoder's device")
        if B > 0:
            need = self.workspace_bytes_checkpointed(B, T, segment_frames)
            ws = workspace if workspace is not None else torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            if ws.dtype != torch.uint8 or ws.device != self.device or ws.numel() < need + 256:
                raise ValueError(f"workspace must be a uint8 tensor of at least {need + 256} bytes on the decoder's device")
            with torch.cuda.device(self.device):
                rc = lib.vit_decode_checkpointed(self._plan, logE.data_ptr(), dt, B, T, lengths.data_ptr() if lengths is not None else None,
                                                 (ws.data_ptr() + 255) & ~255, ws.numel() - 256, states.data_ptr(), loglik.data_ptr(),
                                                 int(segment_frames), torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(rc, "vit_decode_checkpointed")
            torch.cuda.current_stream(self.device).synchronize()      # a workspace allocated here must outlive the kernels
        if out_dtype != torch.int32:
            states = states.to(out_dtype)
        return (states[0], loglik[0]) if single else (states, loglik)

    # ------------------------------------------------------------------ packed (ragged) decode
    def decode_packed(self, emission_logits: torch.Tensor, offsets, out_dtype: torch.dtype = torch.int64,
                      workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Decode B recordings of different lengths without padding (``vit_decode_packed``): ``emission_logits`` is
        ``[sum T_b, S]`` (the rows of recording b are ``offsets[b] : offsets[b+1]``), ``offsets`` a host sequence of B + 1
        frame offsets starting at 0.  Returns ``(states [sum T_b], loglik [B])``, packed like the input -- what the reference
        computes recording by recording (tonet/for_paper.py:2304-2309).  Plans with the wave form only."""
        lib = _lib.load()
        if not isinstance(emission_logits, torch.Tensor) or emission_logits.device != self.device:
            raise ValueError("emission_logits must be a torch tensor on the decoder's device")
        if emission_logits.dim() != 2 or emission_logits.shape[1] != self.S or not emission_logits.is_contiguous():
            raise ValueError(f"emission_logits must be a C-contiguous [sum T_b, {self.S}] tensor")
        if emission_logits.dtype == torch.float32:
            dt = _lib.VIT_F32
        elif emission_logits.dtype == torch.float16:
            dt = _lib.VIT_F16
        else:
            raise TypeError("emission_logits must be float32 or float16")
        off = np.ascontiguousarray(offsets.cpu().numpy() if isinstance(offsets, torch.Tensor) else offsets, dtype=np.int64)
        if off.ndim != 1 or off.size < 1 or off[0] != 0 or (np.diff(off) < 1).any() or off[-1] != emission_logits.shape[0]:
            raise ValueError("offsets must be B + 1 strictly increasing frame offsets from 0 to the number of emission rows")
        B, N = off.size - 1, int(off[-1])
        states = torch.empty((N,), dtype=torch.int32, device=self.device)
        loglik = torch.empty((B,), dtype=torch.float32, device=self.device)
        if B > 0:
            need = int(lib.vit_workspace_bytes_packed(self._plan, B, N))
            if need == 0:
                raise _lib.ViterbiHipError("the packed decode needs a plan with the wave form")
            ws = workspace if workspace is not None else torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            if ws.dtype != torch.uint8 or ws.device != self.device or ws.numel() < need + 256:
                raise ValueError(f"workspace must be a uint8 tensor of at least {need + 256} bytes on the decoder's device")
            with torch.cuda.device(self.device):
                rc = lib.vit_decode_packed(self._plan, emission_logits.data_ptr(), dt, B, off.ctypes.data, (ws.data_ptr() + 255) & ~255,
                                           ws.numel() - 256, states.data_ptr(), loglik.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(rc, "vit_decode_packed")
            if workspace is None:
                torch.cuda.current_stream(self.device).synchronize()      # a workspace allocated here must outlive the kernels
        if out_dtype != torch.int32:
            states = states.to(out_dtype)
        return states, loglik

    def workspace_bytes_packed(self, B: int, total_frames: int) -> int:
        return int(_lib.load().vit_workspace_bytes_packed(self._plan, int(B), int(total_frames)))

    def voicing(self, states: torch.Tensor, n_bins: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """voiced = state < n_bins, bins = min(state, n_bins-1) (tonet/for_paper.py:1828-1829)."""
        n_bins = self.S - 1 if n_bins is None else int(n_bins)
        st = states.to(torch.int32).contiguous()
        voiced = torch.empty(st.shape, dtype=torch.uint8, device=st.device)
        bins = torch.empty(st.shape, dtype=torch.int32, device=st.device)
        with torch.cuda.device(st.device):
            rc = _lib.load().vit_voicing_map(st.data_ptr(), st.numel(), n_bins, voiced.data_ptr(), bins.data_ptr(),
                                             torch.cuda.current_stream(st.device).cuda_stream)
        _lib.check(rc, "vit_voicing_map")
        return voiced.bool(), bins


    def voicing_notes(self, states: torch.Tensor, note_range, n_bins: Optional[int] = None):
        """(voiced, bins, notes): the voicing map plus ``notes = note_range[bins]`` with 0 where unvoiced
        (tonet/for_paper.py:2106-2115, :2207), all on the GPU."""
        n_bins = self.S - 1 if n_bins is None else int(n_bins)
        st = states.to(torch.int32).contiguous()
        nr = note_range if isinstance(note_range, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(note_range, np.float32))
        nr = nr.to(device=st.device, dtype=torch.float32).contiguous()
        if tuple(nr.shape) != (n_bins,):
            raise ValueError(f"note_range must have {n_bins} entries")
        voiced = torch.empty(st.shape, dtype=torch.uint8, device=st.device)
        bins = torch.empty(st.shape, dtype=torch.int32, device=st.device)
        notes = torch.empty(st.shape, dtype=torch.float32, device=st.device)
        with torch.cuda.device(st.device):
            rc = _lib.load().vit_voicing_notes(st.data_ptr(), st.numel(), n_bins, nr.data_ptr(), voiced.data_ptr(), bins.data_ptr(),
                                               None, notes.data_ptr(), torch.cuda.current_stream(st.device).cuda_stream)
        _lib.check(rc, "vit_voicing_notes")
        return voiced.bool(), bins, notes


_DECODERS: dict = {}


def get_decoder(transition_matrix, init_probs, device=None) -> ViterbiDecoder:
    """Decoder cache keyed by the parameter bits and the device (parameters are prepared once per
    process in the reference too: tonet/for_paper.py:281-286)."""
    A = _as_host_f32(transition_matrix)
    pi = _as_host_f32(init_probs)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (hashlib.sha256(A.tobytes()).digest(), hashlib.sha256(pi.tobytes()).digest(), A.shape, str(dev))
    dec = _DECODERS.get(key)
    if dec is None:
        if len(_DECODERS) >= 8:
            _DECODERS.pop(next(iter(_DECODERS)))
        dec = ViterbiDecoder(A, pi, dev)
        _DECODERS[key] = dec
    return dec


def decode(emission_logits: torch.Tensor, transition_matrix, init_probs, lengths: Optional[torch.Tensor] = None,
           algo: str = "auto", out_dtype: torch.dtype = torch.int64) -> Tuple[torch.Tensor, torch.Tensor]:
    """decode(emission_logits, transition_matrix, init_probs) -> (states, loglik).  See module docstring."""
    if not isinstance(emission_logits, torch.Tensor) or emission_logits.device.type != "cuda":
        raise ValueError("emission_logits must be a torch tensor on a ROCm GPU (there is no CPU path)")
    dec = get_decoder(transition_matrix, init_probs, emission_logits.device)
    return dec.decode(emission_logits, lengths=lengths, algo=algo, out_dtype=out_dtype)
