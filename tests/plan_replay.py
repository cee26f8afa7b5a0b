"""Host replay of the banded decomposition, driven by the SAME packed plan image the GPU kernel
reads (built by viterbi_spl_amd/csrc/plan.cpp compiled with g++).  Test infrastructure: it lets
the CPU suite prove that plan + merge order reproduce the dense recursion bit for bit."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "viterbi_spl_amd", "libviterbi_plan_host.so")
BIG = 0x7FFFFFFF


def _lib():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "viterbi_spl_amd", "csrc"), "../libviterbi_plan_host.so"])
    lib = ctypes.CDLL(LIB)
    lib.vph_create.restype = ctypes.c_void_p
    lib.vph_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.vph_destroy.argtypes = [ctypes.c_void_p]
    lib.vph_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.vph_offsets.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.vph_image.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    return lib


class HostPlan:
    def __init__(self, logA_T, log_pi):
        lib = _lib()
        A = np.ascontiguousarray(logA_T, np.float32)
        pi = np.ascontiguousarray(log_pi, np.float32)
        S = A.shape[0]
        h = lib.vph_create(A.ctypes.data, pi.ctypes.data, S)
        info = np.zeros(16, np.int32)
        c0 = np.zeros(1, np.float32)
        off = np.zeros(9, np.int64)
        lib.vph_info(h, info.ctypes.data, c0.ctypes.data)
        lib.vph_offsets(h, off.ctypes.data)
        img = np.zeros(int(off[7]), np.uint8)
        lib.vph_image(h, img.ctypes.data)
        lib.vph_destroy(h)
        self.ok, self.S, self.SP, self.W = bool(info[0]), int(info[1]), int(info[2]), int(info[3])
        self.n_extras, self.n_dense, self.max_window = int(info[4]), int(info[5]), int(info[6])
        self.extras = [int(x) for x in info[7:7 + self.n_extras]]
        self.dense_rows = [int(x) for x in info[11:11 + self.n_dense]]
        self.S4 = int(info[15])
        self.c0 = np.float32(c0[0])
        SP, S4 = self.SP, self.S4

        def sec(k, dtype, count):
            return img[int(off[k]): int(off[k]) + count * np.dtype(dtype).itemsize].view(dtype)
        self.log_pi = sec(0, np.float32, SP)
        self.A4 = sec(1, np.float32, S4 * SP * 4).reshape(S4, SP, 4)
        self.lo = sec(2, np.int32, SP)
        self.kind = sec(3, np.int32, SP)
        self.tabA = sec(4, np.float32, max(self.W, 1) * SP).reshape(max(self.W, 1), SP)
        self.extraA = sec(5, np.float32, 4 * SP).reshape(4, SP)
        self.denseA = sec(6, np.float32, 4 * SP).reshape(4, SP)
        self.Arow = sec(8, np.float32, self.S * SP).reshape(self.S, SP)


def _first_max_prefix(g, gi):
    """inclusive scan, later element replaces only if strictly greater"""
    S = len(g)
    pv = np.empty(S, np.float32)
    pi = np.empty(S, np.int64)
    bv, bi = np.float32(-np.inf), BIG
    for i in range(S):
        if g[i] > bv:
            bv, bi = g[i], gi[i]
        pv[i], pi[i] = bv, bi
    return pv, pi


def _first_max_suffix(g, gi):
    """scan from the top; the next lower index replaces on >= (so the lowest index wins ties)"""
    S = len(g)
    sv = np.empty(S, np.float32)
    si = np.empty(S, np.int64)
    bv, bi = np.float32(-np.inf), BIG
    for i in range(S - 1, -1, -1):
        if g[i] >= bv:
            bv, bi = g[i], gi[i]
        sv[i], si[i] = bv, bi
    return sv, si


def replay_banded(plan: HostPlan, logE):
    """Returns (states int64[T], loglik, final delta[S]) following banded_forward_kernel step by step."""
    assert plan.ok
    S, W = plan.S, plan.W
    logE = np.ascontiguousarray(logE, np.float32)
    T = logE.shape[0]
    delta = (plan.log_pi[:S] + logE[0]).astype(np.float32)
    psi = np.zeros((T, S), np.int64)
    lo = plan.lo[:S].astype(np.int64)
    kind = plan.kind[:S]
    win_idx = lo[:, None] + np.arange(W)[None, :]            # [S,W]
    tab = plan.tabA[:, :S].T                                  # [S,W]
    masked = np.zeros(S, bool)
    masked[plan.extras] = True
    ar = np.arange(S)
    for t in range(1, T):
        g = np.where(masked, np.float32(-np.inf), (delta + plan.c0).astype(np.float32))
        gi = np.where(masked, BIG, ar)
        pv, pi = _first_max_prefix(g, gi)
        sv, si = _first_max_suffix(g, gi)
        cand = (delta[win_idx] + tab).astype(np.float32)
        warg = np.argmax(cand, axis=1)
        wbest = cand[ar, warg]
        new = np.empty(S, np.float32)
        for j in range(S):
            av, ai = np.float32(-np.inf), BIG
            if kind[j] == -1:
                if lo[j] > 0 and pv[lo[j] - 1] > av:
                    av, ai = pv[lo[j] - 1], pi[lo[j] - 1]
                wv = wbest[j]
                wi = BIG if not (wv > -np.inf) else lo[j] + warg[j]
                if wv > av:
                    av, ai = wv, wi
                qs = lo[j] + W
                if qs < S and sv[qs] > av:
                    av, ai = sv[qs], si[qs]
                for k, x in enumerate(plan.extras):
                    v = np.float32(delta[x] + plan.extraA[k, j])
                    if v > av or (v == av and x < ai and ai != BIG):
                        av, ai = v, x
            else:
                d = kind[j]
                c = (delta + plan.denseA[d, :S]).astype(np.float32)
                ai = int(np.argmax(c))
                av = c[ai]
                if not (av > -np.inf):
                    ai = BIG
            if ai == BIG:
                ai = 0
            psi[t, j] = ai
            new[j] = av + logE[t, j]
        delta = new
    s = int(np.argmax(delta))
    path = np.empty(T, np.int64)
    path[-1] = s
    for t in range(T - 2, -1, -1):
        s = psi[t + 1, s]
        path[t] = s
    return path, delta[path[-1]], delta


def replay_dense_image(plan: HostPlan, logE):
    """Dense recursion driven by the packed A4 image (checks the packing the dense kernel reads)."""
    S = plan.S
    A = plan.A4[:, :S, :].transpose(1, 0, 2).reshape(S, -1)[:, :S]   # [target, source]
    from oracle import viterbi_oracle as vo
    return vo.decode_numpy(A, plan.log_pi[:S], logE)
