"""ctypes binding of libviterbi_hip.so (the C ABI declared in include/viterbi_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C viterbi_spl_amd/csrc``.
There is no CPU fallback: if the shared object is missing or fails to load, every
decode entry point raises.
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libviterbi_hip.so")

VIT_OK = 0
VIT_F32, VIT_F16 = 0, 1
ALGO = {"auto": 0, "dense": 1, "banded": 2, "wave": 3, "group": 4}

EXPORTS = (
    "vit_abi_version", "vit_status_string", "vit_last_hip_error", "vit_plan_create", "vit_plan_destroy",
    "vit_plan_query", "vit_plan_image_bytes", "vit_plan_upload", "vit_workspace_bytes", "vit_decode",
    "vit_forward", "vit_backtrace", "vit_voicing_map", "vit_obs_shaun", "vit_obs_softmax", "vit_obs_softmax_scaled", "vit_plan_set_option", "vit_snippets_append", "vit_voicing_notes",
    "vit_workspace_bytes_for", "vit_debug_scan", "vit_backtrace_counters", "vit_workspace_bytes_checkpointed", "vit_decode_checkpointed",
    "vit_backtrace_checked", "vit_forward_family", "vit_workspace_bytes_packed", "vit_decode_packed",
)
ABI_VERSION = 4


class PlanInfo(ctypes.Structure):
    _fields_ = [
        ("S", ctypes.c_int64), ("banded_ok", ctypes.c_int32), ("n_consts", ctypes.c_int32),
        ("n_extras", ctypes.c_int32), ("max_window", ctypes.c_int32), ("group_window", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 3), ("consts", ctypes.c_float * 4), ("extras", ctypes.c_int32 * 4),
    ]


class ViterbiHipError(RuntimeError):
    pass


_lib = None


def _preload_hip_runtime():
    """Make sure the HIP runtime PyTorch uses is the one our library binds to
    (same SONAME libamdhip64.so.7; the first one loaded wins)."""
    import torch  # noqa: F401  (loads torch/lib/libamdhip64.so)
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tlib):
        ctypes.CDLL(tlib, mode=ctypes.RTLD_GLOBAL)


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ViterbiHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C viterbi_spl_amd/csrc`). There is no CPU fallback.")
    _preload_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    vp, i64, i32, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_size_t
    lib.vit_abi_version.restype = i32
    lib.vit_status_string.restype = ctypes.c_char_p
    lib.vit_status_string.argtypes = [i32]
    lib.vit_last_hip_error.restype = i32
    lib.vit_plan_create.restype = i32
    lib.vit_plan_create.argtypes = [vp, vp, i64, ctypes.POINTER(vp)]
    lib.vit_plan_destroy.restype = None
    lib.vit_plan_destroy.argtypes = [vp]
    lib.vit_plan_query.restype = i32
    lib.vit_plan_query.argtypes = [vp, ctypes.POINTER(PlanInfo)]
    lib.vit_plan_set_option.restype = i32
    lib.vit_plan_set_option.argtypes = [vp, ctypes.c_char_p, i64]
    lib.vit_plan_image_bytes.restype = sz
    lib.vit_plan_image_bytes.argtypes = [vp]
    lib.vit_plan_upload.restype = i32
    lib.vit_plan_upload.argtypes = [vp, vp, sz, vp]
    lib.vit_workspace_bytes.restype = sz
    lib.vit_workspace_bytes.argtypes = [vp, i64, i64]
    lib.vit_workspace_bytes_for.restype = sz
    lib.vit_workspace_bytes_for.argtypes = [vp, i64, i64, i32]
    lib.vit_workspace_bytes_checkpointed.restype = sz
    lib.vit_workspace_bytes_checkpointed.argtypes = [vp, i64, i64, i64]
    lib.vit_decode_checkpointed.restype = i32
    lib.vit_decode_checkpointed.argtypes = [vp, vp, i32, i64, i64, vp, vp, sz, vp, vp, i64, vp]
    lib.vit_backtrace_counters.restype = i32
    lib.vit_backtrace_counters.argtypes = [vp, i64, i64, vp, ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_int32)]
    lib.vit_decode.restype = i32
    lib.vit_decode.argtypes = [vp, vp, i32, i64, i64, vp, vp, sz, vp, vp, i32, vp]
    lib.vit_forward.restype = i32
    lib.vit_forward.argtypes = [vp, vp, i32, i64, i64, vp, vp, sz, vp, i32, vp]
    lib.vit_backtrace.restype = i32
    lib.vit_backtrace.argtypes = [vp, i64, i64, vp, vp, sz, vp, i32, vp]
    lib.vit_backtrace_checked.restype = i32
    lib.vit_backtrace_checked.argtypes = [vp, vp, i32, i64, i64, vp, vp, sz, vp, i32, vp]
    lib.vit_workspace_bytes_packed.restype = sz
    lib.vit_workspace_bytes_packed.argtypes = [vp, i64, i64]
    lib.vit_decode_packed.restype = i32
    lib.vit_decode_packed.argtypes = [vp, vp, i32, i64, vp, vp, sz, vp, vp, vp]
    lib.vit_forward_family.restype = i32
    lib.vit_forward_family.argtypes = [vp, i64, i32]
    lib.vit_voicing_map.restype = i32
    lib.vit_voicing_map.argtypes = [vp, i64, i32, vp, vp, vp]
    f64 = ctypes.c_double
    lib.vit_obs_shaun.restype = i32
    lib.vit_obs_shaun.argtypes = [vp, i64, i32, i32, f64, f64, f64, vp, vp]
    lib.vit_obs_softmax.restype = i32
    lib.vit_obs_softmax.argtypes = [vp, i64, i32, i32, vp, vp]
    lib.vit_obs_softmax_scaled.restype = i32
    lib.vit_obs_softmax_scaled.argtypes = [vp, i64, i32, i32, f64, vp, vp, vp]
    lib.vit_snippets_append.restype = i32
    lib.vit_snippets_append.argtypes = [vp, i32, i32, i32, i32, vp, i64, vp]
    lib.vit_voicing_notes.restype = i32
    lib.vit_voicing_notes.argtypes = [vp, i64, i32, vp, vp, vp, vp, vp, vp]
    lib.vit_debug_scan.restype = i32
    lib.vit_debug_scan.argtypes = [vp, i32, i32, vp, vp, vp]
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != VIT_OK:
        lib = load()
        msg = lib.vit_status_string(rc).decode()
        extra = f" (hipError_t {lib.vit_last_hip_error()})" if rc == -3 else ""
        raise ViterbiHipError(f"{what}: {msg}{extra}")
