"""The C-ABI shared library loads and exports every symbol include/viterbi_hip.h declares.
No compute calls here (no GPU in this tier); only host-side entry points are exercised."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "viterbi_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vit_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from viterbi_spl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _lib.load()


def test_header_and_loader_agree(lib):
    from viterbi_spl_amd import _lib
    names = declared_functions()
    assert len(names) >= 13
    assert set(names) == set(_lib.EXPORTS)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in viterbi_hip.h but not exported"


def test_version_and_status_strings(lib):
    from viterbi_spl_amd import _lib
    assert lib.vit_abi_version() == _lib.ABI_VERSION == 4
    assert lib.vit_status_string(0) == b"ok"
    assert b"workspace" in lib.vit_status_string(-4)


def test_plan_create_query_is_host_only(lib, golden):
    from viterbi_spl_amd import _lib
    A = np.ascontiguousarray(golden["params"]["tonet361_logA_T"])
    pi = np.ascontiguousarray(golden["params"]["tonet361_log_pi"])
    plan = ctypes.c_void_p()
    assert lib.vit_plan_create(A.ctypes.data, pi.ctypes.data, 361, ctypes.byref(plan)) == 0
    info = _lib.PlanInfo()
    assert lib.vit_plan_query(plan, ctypes.byref(info)) == 0
    assert info.S == 361 and info.banded_ok == 1 and info.group_window == 32 and info.max_window == 29
    assert info.n_extras == 1 and info.extras[0] == 360
    assert info.reserved[2] & 8          # the wave form (one song per wavefront) is available for the tonet matrix
    assert lib.vit_plan_image_bytes(plan) > 361 * 361 * 4
    ws = lib.vit_workspace_bytes(plan, 128, 30000)
    assert ws >= 128 * 30000 * 384 * 4   # float32 delta history; the wave form's slot-order rows (64 * 6 floats) are the widest
    # selection overrides: known keys only, and the result-breaking timing mask is refused by the release build
    assert lib.vit_plan_set_option(plan, b"forward_form", 4) == 0
    assert lib.vit_plan_set_option(plan, b"bt_chunks", 7) == 0
    assert lib.vit_plan_set_option(plan, b"no_such_key", 1) == -1
    assert lib.vit_plan_set_option(plan, b"timing", 1) == -5
    assert lib.vit_plan_set_option(plan, b"timing", 0) == 0
    assert lib.vit_plan_set_option(plan, b"reset", 0) == 0
    # decode before upload is refused, not executed
    dummy = ctypes.c_void_p(256 * 1024)
    rc = lib.vit_decode(plan, dummy, 0, 1, 10, None, dummy, ws, dummy, None, 0, None)
    assert rc == -6
    lib.vit_plan_destroy(plan)


def test_bad_arguments_are_rejected(lib):
    plan = ctypes.c_void_p()
    A = np.zeros((4, 4), np.float32)
    assert lib.vit_plan_create(None, A.ctypes.data, 4, ctypes.byref(plan)) == -1
    assert lib.vit_plan_create(A.ctypes.data, A.ctypes.data, 0, ctypes.byref(plan)) == -1
    assert lib.vit_plan_create(A.ctypes.data, A.ctypes.data, 5000, ctypes.byref(plan)) == -1
    assert lib.vit_workspace_bytes(None, 1, 1) == 0
