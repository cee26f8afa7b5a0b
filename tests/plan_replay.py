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
    lib.vph_wave.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
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
        off = np.zeros(12, np.int64)
        lib.vph_info(h, info.ctypes.data, c0.ctypes.data)
        lib.vph_offsets(h, off.ctypes.data)
        img = np.zeros(int(off[7]), np.uint8)
        lib.vph_image(h, img.ctypes.data)
        wv = np.zeros(6, np.int64)
        lib.vph_wave(h, wv.ctypes.data)
        lib.vph_destroy(h)
        self.wave_ok, self.wave_npl, self.wave_d, self.wave_dk = bool(wv[0]), int(wv[1]), int(wv[2]), int(wv[3])
        self.floor_all_ok = bool(wv[4])
        self.ok, self.S, self.SP, self.W = bool(info[0]), int(info[1]), int(info[2]), int(info[3])
        self.n_extras, self.n_dense, self.max_window = int(info[4]), int(info[5]), int(info[6])
        self.extras = [int(x) for x in info[7:7 + self.n_extras]]
        self.dense_rows = [int(x) for x in info[11:11 + self.n_dense]]
        self.S4 = int(info[15]) & 0xFFFF
        self.pair_ok = bool(int(info[15]) & 0x10000)
        self.floor_ok = bool(int(info[15]) & 0x20000)
        self.step_ok = bool(int(info[15]) & 0x40000)
        self.step_kb = (int(info[15]) >> 20) & 0xF
        self.step_bw = (int(info[15]) >> 24) & 0x7F
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
        self.rowc = sec(9, np.float32, SP)
        self.lo2 = sec(10, np.int32, SP // 2)
        self.tabP = sec(11, np.float32, max(self.W, 1) * SP).reshape(max(self.W, 1), SP)
        if self.wave_ok:   # [own state k][pair m][half h][lane]
            n = self.wave_npl * (self.wave_dk + 1) * 2 * 64
            self.tabV = img[int(wv[5]): int(wv[5]) + 4 * n].view(np.float32).reshape(self.wave_npl, self.wave_dk + 1, 2, 64)


def replay_banded(plan: HostPlan, logE, floor=False, pair=False):
    """Follows the GPU kernels step by step on the host.

    Forward (banded_forward_kernel): value-only; per target the max of the W window sums, of
    fl(max(prefix-max, suffix-max of the RAW delta) + c_j) and of the extra-column sums; dense rows
    take the max over every source.  The delta row of every frame is kept.
    Back-trace (lazy_backtrace_kernel): for the path state j at t+1 rebuild every candidate
    fl(delta_t[i] + logA_T[j][i]) from the plan tables and take the LOWEST index attaining the max;
    the fast path (window + extras only) is taken when fl(max_i delta_t[i] + c_j) < window max.
    floor=True follows banded_floor_forward_kernel instead: the out-of-window term is fl(M + c_j) with M the
    max of the raw delta over ALL non-extra sources (needs plan.floor_ok).
    Returns (states int64[T], loglik, final delta[S])."""
    assert plan.ok
    assert plan.floor_ok or not floor
    assert (plan.pair_ok and floor) or not pair
    S, W = plan.S, plan.W
    logE = np.ascontiguousarray(logE, np.float32)
    T = logE.shape[0]
    lo = plan.lo[:S].astype(np.int64)
    kind = plan.kind[:S]
    rowc = plan.rowc[:S]
    win_idx = lo[:, None] + np.arange(W)[None, :]            # [S,W]
    tab = plan.tabA[:, :S].T                                  # [S,W]
    if pair:   # banded_floor_pair_forward_kernel: targets 2p, 2p+1 evaluate the common window lo2[p] .. lo2[p]+W
        lo2 = plan.lo2[np.arange(S) // 2].astype(np.int64)
        win_idx = lo2[:, None] + np.arange(W)[None, :]
        tab = plan.tabP[:, :S].T
    masked = np.zeros(S, bool)
    masked[plan.extras] = True
    ninf = np.float32(-np.inf)

    hist = np.empty((T, S), np.float32)
    dmax = np.empty(T, np.float32)
    delta = (plan.log_pi[:S] + logE[0]).astype(np.float32)
    hist[0] = delta
    for t in range(1, T):
        raw = np.where(masked, ninf, delta)
        pv = np.concatenate([[ninf], np.maximum.accumulate(raw)])          # pv[q] = max raw[:q]
        sv = np.concatenate([np.maximum.accumulate(raw[::-1])[::-1], [ninf]])  # sv[q] = max raw[q:]
        dmax[t - 1] = pv[S]
        m = np.max((delta[win_idx] + tab).astype(np.float32), axis=1)
        if floor:
            outside = (pv[S] + rowc).astype(np.float32)
        else:
            outside = (np.maximum(pv[lo], sv[lo + W]) + rowc).astype(np.float32)
        m = np.maximum(m, outside)
        for k, x in enumerate(plan.extras):
            m = np.maximum(m, (delta[x] + plan.extraA[k, :S]).astype(np.float32))
        for j in np.nonzero(kind >= 0)[0]:
            m[j] = np.max((delta + plan.denseA[kind[j], :S]).astype(np.float32))
        delta = (m + logE[t]).astype(np.float32)
        hist[t] = delta

    s = int(np.argmax(delta))
    path = np.empty(T, np.int64)
    path[-1] = s
    ar = np.arange(S)
    n_fast = 0
    for t in range(T - 2, -1, -1):
        d = hist[t]
        j = s
        if kind[j] == -1:
            cand_i = list(range(lo[j], lo[j] + W)) + list(plan.extras)
            cand_v = [np.float32(d[lo[j] + w] + plan.tabA[w, j]) for w in range(W)] + \
                     [np.float32(d[x] + plan.extraA[k, j]) for k, x in enumerate(plan.extras)]
            m = max(cand_v) if cand_v else ninf
            if np.float32(dmax[t] + rowc[j]) < m:              # fast path: no row-constant candidate can tie
                s = min(i for i, v in zip(cand_i, cand_v) if v == m)
                n_fast += 1
                path[t] = s
                continue
            excl = masked | ((ar >= lo[j]) & (ar < lo[j] + W))
            vf = np.where(excl, ninf, (d + rowc[j]).astype(np.float32))
            m = max(m, np.max(vf))
            idx = [i for i, v in zip(cand_i, cand_v) if v == m] + list(np.nonzero(vf == m)[0])
            s = int(min(idx)) if idx else 0
        else:
            v = (d + plan.denseA[kind[j], :S]).astype(np.float32)
            s = int(np.argmax(v))
        path[t] = s
    replay_banded.last_fast_fraction = n_fast / max(T - 1, 1)
    return path, delta[path[-1]], delta


def replay_wave(plan: HostPlan, logE):
    """Follows wave_forward_kernel (viterbi_spl_amd/csrc/wave.hip) on the host, driven by the packed tabV table.

    Slot q = npl*lane + k holds state q - o (o = 64*npl - S).  Own state k of lane l evaluates the dk+1 source pairs at
    neighbourhood positions p0e(k) + 2m + h, i.e. slots npl*(l - H) + p; a slot outside the wave delivers 0 (the DPP shift's
    bound_ctrl fill) -- its weight must be -inf.  m_j = max(window candidates, fl(M + c_j), extra-column candidates) with
    M over ALL sources.  Returns (history [T, 64*npl] in the kernel's slot order incl. M in column 0, final delta [S])."""
    assert plan.ok and plan.wave_ok and plan.floor_all_ok
    S, npl, dk = plan.S, plan.wave_npl, plan.wave_dk
    H = (dk + npl - 1) // npl
    NQ = 64 * npl
    o = NQ - S
    logE = np.ascontiguousarray(logE, np.float32)
    T = logE.shape[0]
    ninf = np.float32(-np.inf)
    lane = np.arange(NQ) // npl
    k = np.arange(NQ) % npl
    p0e = (k + npl * H - dk) & ~1
    m = np.arange(dk + 1)
    src_slot = (npl * (lane - H) + p0e)[:, None, None] + 2 * m[None, :, None] + np.arange(2)[None, None, :]   # [NQ, dk+1, 2]
    w = plan.tabV[k[:, None, None], m[None, :, None], np.arange(2)[None, None, :], lane[:, None, None]]        # [NQ, dk+1, 2]
    inside = (src_slot >= 0) & (src_slot < NQ)
    assert np.all(np.isneginf(w[~inside])), "a source slot outside the wave must carry a -inf weight"
    state = np.arange(NQ) - o
    valid = state >= 0
    assert np.all(np.isneginf(w[~valid])), "idle targets carry -inf weights"
    # the table holds the true matrix entries: window positions that exist
    cj = np.where(valid, plan.rowc[np.clip(state, 0, S - 1)], ninf).astype(np.float32)
    xa = [np.where(valid, plan.extraA[e, np.clip(state, 0, S - 1)], ninf).astype(np.float32) for e in range(plan.n_extras)]
    lpi = np.where(valid, plan.log_pi[np.clip(state, 0, S - 1)], ninf).astype(np.float32)
    sc = np.clip(src_slot, 0, NQ - 1)

    hist = np.empty((T, NQ), np.float32)
    d = np.full(NQ, ninf, np.float32)
    d[valid] = (lpi[valid] + logE[0]).astype(np.float32)
    with np.errstate(invalid="ignore"):
        for t in range(T):
            if t > 0:
                dz = np.where(inside, d[sc], np.float32(0))                    # 0 where a shift has no source lane
                cand = (dz + w).astype(np.float32)
                acc = np.max(cand.reshape(NQ, -1), axis=1)
                mm = np.maximum(acc, (M + cj).astype(np.float32))
                for e, x in enumerate(plan.extras):
                    mm = np.maximum(mm, (d[o + x] + xa[e]).astype(np.float32))
                e_t = np.full(NQ, np.float32(0))                               # idle slots: whatever the clamped load returns
                e_t[valid] = logE[t]
                d = (mm + e_t).astype(np.float32)
                assert np.all(np.isneginf(d[~valid]))
            M = np.max(d)
            hist[t] = d
            hist[t, 0] = M
            for e, x in enumerate(plan.extras):                                    # copies of the extra columns' delta
                hist[t, 1 + e] = d[o + x]
    return hist, d[valid].copy()


def replay_dense_image(plan: HostPlan, logE):
    """Dense recursion driven by the packed A4 image (checks the packing the dense kernel reads)."""
    S = plan.S
    A = plan.A4[:, :S, :].transpose(1, 0, 2).reshape(S, -1)[:, :S]   # [target, source]
    from oracle import viterbi_oracle as vo
    return vo.decode_numpy(A, plan.log_pi[:S], logE)
