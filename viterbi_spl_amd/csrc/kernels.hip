// kernels.hip -- gfx950 (MI355X) kernels for the float32 log-domain Viterbi decoder.
//
// Semantics (reference: imm/tf_viterbi.py:91-107,
// tonet/for_paper.py:1855-1868):
//   delta_0[j]  = fl32(log_pi[j] + logE[0][j])
//   m_j         = max_i fl32(delta_{t-1}[i] + logA_T[j][i]);  psi_t[j] = LOWEST i attaining it
//   delta_t[j]  = fl32(m_j + logE[t][j])
//   s_{T-1}     = lowest argmax_j delta_{T-1}[j];  s_t = psi_{t+1}[s_{t+1}]
// Only add / compare / select: built with -ffp-contract=off, results are bit-identical to
// the reference's NumPy float32 loop.
//
// Layout: one workgroup per song (dense kernel: NS songs per workgroup sharing every
// transition tile it streams from L2), one thread per target state, delta resident in
// LDS for the whole song, emissions read with coalesced loads one frame ahead,
// back-pointers written as uint16 rows padded to 16 bytes.  No MFMA: the recurrence is
// max-plus, not an add-contract.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace vit {

constexpr int kBig = 0x7fffffff;

struct VI {
    float v;
    int i;
};

__device__ __forceinline__ VI vi_identity() { return VI{-INFINITY, kBig}; }

// first-max: `later` (higher source index) replaces `earlier` only if strictly greater
__device__ __forceinline__ VI op_fwd(VI earlier, VI later) { return later.v > earlier.v ? later : earlier; }
// pieces visited in DESCENDING source order: the next (lower-index) piece wins ties
__device__ __forceinline__ VI op_rev(VI acc, VI next) { return next.v >= acc.v ? next : acc; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ VI dpp_fetch(VI x) {
    VI r;
    r.v = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(x.v), CTRL,
                                                     ROW_MASK, 0xf, false));
    r.i = __builtin_amdgcn_update_dpp(kBig, x.i, CTRL, ROW_MASK, 0xf, false);
    return r;
}

// Inclusive wave64 scan over lanes 0..63 with the ordered first-max operator.
// REV = false: lanes ascend in source index; REV = true: lanes DESCEND in source index.
// DPP: row_shr:1/2/4/8 inside each row of 16, then row_bcast:15 (rows 1,3), row_bcast:31 (rows 2,3).
template <bool REV>
__device__ __forceinline__ VI wave_scan(VI x) {
#define VIT_SCAN_STEP(CTRL, MASK)                        \
    {                                                    \
        VI s = dpp_fetch<CTRL, MASK>(x);                 \
        x = REV ? op_rev(s, x) : op_fwd(s, x);           \
    }
    VIT_SCAN_STEP(0x111, 0xf)
    VIT_SCAN_STEP(0x112, 0xf)
    VIT_SCAN_STEP(0x114, 0xf)
    VIT_SCAN_STEP(0x118, 0xf)
    VIT_SCAN_STEP(0x142, 0xa)
    VIT_SCAN_STEP(0x143, 0xc)
#undef VIT_SCAN_STEP
    return x;
}

template <typename ET>
__device__ __forceinline__ float load_e(const ET* p);
template <>
__device__ __forceinline__ float load_e<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load_e<__half>(const __half* p) { return __half2float(*p); }

__device__ __forceinline__ int song_length(const int64_t* lengths, int song, int T) {
    if (!lengths) return T;
    long long v = lengths[song];
    v = v < 1 ? 1 : v;
    return v > T ? T : (int)v;
}

// Workgroup-wide lowest-index argmax of delta (terminal state), result to thread 0.
__device__ __forceinline__ void terminal_argmax(float dj, int j, int S, VI* tot, int nw, int32_t* last_state,
                                                float* loglik, int song) {
    VI x{j < S ? dj : -INFINITY, j < S ? j : kBig};
    x = wave_scan<false>(x);
    if ((j & 63) == 63) tot[j >> 6] = x;
    __syncthreads();
    if (j == 0) {
        VI acc = vi_identity();
        for (int b = 0; b < nw; ++b) acc = op_fwd(acc, tot[b]);
        if (acc.i == kBig) acc.i = 0;
        last_state[song] = acc.i;
        if (loglik) loglik[song] = acc.v;
    }
}

// ---------------------------------------------------------------------------------------
// Dense forward kernel: NS songs per workgroup; every thread owns one target state and
// walks all S sources four at a time.  A4[q][j][0..3] = logA_T[j][4q..4q+3] is a coalesced
// 16-byte load per lane (L2 resident, 4*S*S bytes), reused for the NS songs; the delta
// vectors are read from LDS as wave-uniform (broadcast) 16-byte reads.
// ---------------------------------------------------------------------------------------
template <int NS, typename ET>
__global__ void __launch_bounds__(dense_max_threads(NS)) dense_forward_kernel(FwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int S = a.S, SP = a.SP, S4 = a.S4, T = a.T;
    const int SD = S4 * 4;                       // delta row length in LDS (multiple of 4)
    float* dl = reinterpret_cast<float*>(smem);  // [2][NS][SD]
    VI* tot = reinterpret_cast<VI*>(dl + 2 * NS * SD);

    const int j = threadIdx.x;
    const int nw = blockDim.x >> 6;
    const int song0 = blockIdx.x * NS;
    const float4* __restrict__ A4 = reinterpret_cast<const float4*>(a.image + a.off_A4);
    const float* __restrict__ log_pi = reinterpret_cast<const float*>(a.image + a.off_logpi);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE);

    int Tb[NS];
    bool live[NS];
    int Tmax = 1;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        live[s] = song0 + s < a.B;
        Tb[s] = live[s] ? song_length(a.lengths, song0 + s, T) : 1;
        Tmax = Tb[s] > Tmax ? Tb[s] : Tmax;
    }

    float enext[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const size_t base = (size_t)(song0 + s) * T * S;
        float d = -INFINITY;
        if (live[s] && j < S) d = log_pi[j] + load_e<ET>(E + base + j);
        if (j < SD) { dl[(0 * NS + s) * SD + j] = d; dl[(1 * NS + s) * SD + j] = -INFINITY; }
        enext[s] = (live[s] && j < S && Tb[s] > 1) ? load_e<ET>(E + base + S + j) : 0.f;
    }
    __syncthreads();

    int cur = 0;
    for (int t = 1; t < Tmax; ++t) {
        float ecur[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ecur[s] = enext[s];
            if (live[s] && j < S && t + 1 < Tb[s])
                enext[s] = load_e<ET>(E + ((size_t)(song0 + s) * T + t + 1) * S + j);
        }
        float best[NS];
        int arg[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { best[s] = -INFINITY; arg[s] = kBig; }
        const float* dcur = dl + cur * NS * SD;
#pragma unroll 4
        for (int q = 0; q < S4; ++q) {
            const float4 av = A4[(size_t)q * SP + j];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float4 dv = *reinterpret_cast<const float4*>(dcur + s * SD + 4 * q);
                float v;
                v = dv.x + av.x; if (v > best[s]) { best[s] = v; arg[s] = 4 * q; }
                v = dv.y + av.y; if (v > best[s]) { best[s] = v; arg[s] = 4 * q + 1; }
                v = dv.z + av.z; if (v > best[s]) { best[s] = v; arg[s] = 4 * q + 2; }
                v = dv.w + av.w; if (v > best[s]) { best[s] = v; arg[s] = 4 * q + 3; }
            }
        }
        float* dnxt = dl + (cur ^ 1) * NS * SD;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (j < S) {
                if (live[s] && t < Tb[s]) {
                    const int ai = arg[s] == kBig ? 0 : arg[s];
                    dnxt[s * SD + j] = best[s] + ecur[s];
                    a.psi[((size_t)(song0 + s) * T + t) * a.SPSI + j] = (uint16_t)ai;
                } else {
                    dnxt[s * SD + j] = dcur[s * SD + j];
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }

#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (live[s]) {
            const float dj = j < S ? dl[(cur * NS + s) * SD + j] : -INFINITY;
            terminal_argmax(dj, j, S, tot, nw, a.last_state, a.loglik, song0 + s);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// Banded forward kernel: one song per workgroup.
//
// Per frame, for a banded target j (window [lo_j, lo_j+W), shared constant c0, extra
// columns X) the candidates, merged in increasing source order with strict '>':
//   prefix  first-max_{i < lo_j, i not in X}  fl(delta_i + c0)      Pp[lo_j]
//   window  fl(delta_i + logA_T[j][i]),  i in [lo_j, lo_j+W)        W register-resident entries
//   suffix  first-max_{i >= lo_j+W, i not in X} fl(delta_i + c0)    Sf[lo_j+W]
//   extras  fl(delta_x + logA_T[j][x]), x in X                      lexicographic merge
// Dense rows (e.g. the "unvoiced" target) are reduced over all sources.
//
// Wave roles (NWT = target waves, 64*NWT >= S):
//   waves 0..NWT-1  one thread per target: window candidates, merge, delta_t, back-pointer
//   wave  NWT       prefix scan over all sources (NWT elements per lane) + dense rows 0,2
//   wave  NWT+1     suffix scan (lanes hold the sources in descending blocks) + dense rows 1,3
// The scan waves run beside the window phase; two workgroup barriers per frame.
// ---------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ VI dpp_shift_in(VI x) {  // whole-wave shift by one lane, lane 0 gets identity
    VI r;
    r.v = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(x.v), CTRL, 0xf, 0xf, false));
    r.i = __builtin_amdgcn_update_dpp(kBig, x.i, CTRL, 0xf, 0xf, false);
    return r;
}

template <int W, int NWT, typename ET>
__global__ void __launch_bounds__((NWT + 2) * 64) banded_forward_kernel(FwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NP = NWT * 64;   // padded state count handled by the target waves
    constexpr int EPL = NWT;       // sources per lane in the scan waves
    const int S = a.S, SP = a.SP, T = a.T;
    float* dl = reinterpret_cast<float*>(smem);   // [NP]       delta_{t-1}; entries >= S stay -inf
    VI* Pp = reinterpret_cast<VI*>(dl + NP);      // [NP+1]     Pp[q] = first-max over sources < q
    VI* Sf = Pp + NP + 1;                         // [NP+1]     Sf[q] = first-max over sources >= q
    VI* Dr = Sf + NP + 1;                         // [4]        dense-row results
    VI* tot = Dr + kMaxDenseRows;                 // [16]       terminal argmax scratch

    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    uint16_t* __restrict__ psi = a.psi + (size_t)song * T * a.SPSI;
    const float c0 = a.c0;
    const int nx = a.n_extras, nd = a.n_dense;
    const int dbg = a.debug;
    const float* __restrict__ daT = reinterpret_cast<const float*>(a.image + a.off_denseA);

    // ---------------- per-role setup
    const bool is_target = wv < NWT;
    const int j = tid;                       // target index (target waves)
    const bool tvalid = is_target && j < S;
    const int jc = j < SP ? j : 0;           // clamp for image reads (image rows are SP wide)
    int lo = 0, kind = -2;
    float aw[W];
    float xa[kMaxExtras];
    int xcol[kMaxExtras];
#pragma unroll
    for (int k = 0; k < kMaxExtras; ++k) { xa[k] = -INFINITY; xcol[k] = a.extras[k]; }
    if (is_target) {
        lo = reinterpret_cast<const int32_t*>(a.image + a.off_lo)[jc];
        kind = j < SP ? reinterpret_cast<const int32_t*>(a.image + a.off_kind)[jc] : -2;
        const float* __restrict__ tab = reinterpret_cast<const float*>(a.image + a.off_tabA);
        const float* __restrict__ xaT = reinterpret_cast<const float*>(a.image + a.off_extraA);
#pragma unroll
        for (int w = 0; w < W; ++w) aw[w] = tab[(size_t)w * SP + jc];
#pragma unroll
        for (int k = 0; k < kMaxExtras; ++k) xa[k] = xaT[(size_t)k * SP + jc];
    } else {
#pragma unroll
        for (int w = 0; w < W; ++w) aw[w] = 0.f;
    }
    // scan waves: lane owns sources [blk*EPL, blk*EPL+EPL); the suffix wave walks blocks downwards
    const bool is_pre = wv == NWT;
    const int blk = is_pre ? lane : 63 - lane;
    const int i0 = blk * EPL;
    bool smask[EPL];           // source is padding or an extra column: excluded from the c0 scans
    float dA0[EPL], dA1[EPL];  // this wave's two dense rows
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = i0 + e;
        bool m = i >= S;
#pragma unroll
        for (int k = 0; k < kMaxExtras; ++k) m |= (k < nx && i == xcol[k]);
        smask[e] = m;
        const int d0 = is_pre ? 0 : 1;
        dA0[e] = (!is_target && i < S && d0 < nd) ? daT[(size_t)d0 * SP + i] : -INFINITY;
        dA1[e] = (!is_target && i < S && d0 + 2 < nd) ? daT[(size_t)(d0 + 2) * SP + i] : -INFINITY;
    }

    // ---------------- frame 0
    if (is_target) {
        float d = -INFINITY;
        if (tvalid) d = reinterpret_cast<const float*>(a.image + a.off_logpi)[j] + load_e<ET>(E + j);
        dl[j] = d;
    } else if (lane == 0) {
        if (is_pre) Pp[0] = vi_identity(); else Sf[NP] = vi_identity();
    }
    // Emission rows are fetched two frames ahead and consumed only at the end of a frame: vmcnt
    // retires in order, so a wait on a younger load would also wait for the previous frame's
    // back-pointer store.
    float e_a = (tvalid && Tb > 1) ? load_e<ET>(E + S + j) : 0.f;
    float e_b = (tvalid && Tb > 2) ? load_e<ET>(E + 2 * (size_t)S + j) : 0.f;
    // Retire every set-up load here, so that inside the frame loop the only vector-memory
    // operations the wait-count pass has to reason about are the two it issues per frame.
#pragma unroll
    for (int w = 0; w < W; ++w) asm volatile("" ::"v"(aw[w]));
#pragma unroll
    for (int k = 0; k < kMaxExtras; ++k) asm volatile("" ::"v"(xa[k]));
#pragma unroll
    for (int e = 0; e < EPL; ++e) asm volatile("" ::"v"(dA0[e]), "v"(dA1[e]));
    asm volatile("" ::"v"(lo), "v"(kind), "v"(e_a), "v"(e_b));
    __syncthreads();

    auto frame = [&](const int t, float& e_slot) {
        float best = -INFINITY;
        int arg = kBig;
        float xv[kMaxExtras];
#pragma unroll
        for (int k = 0; k < kMaxExtras; ++k) xv[k] = -INFINITY;

        if (is_target) {
            // ---- window candidates (everything here reads delta_{t-1})
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k)
                if (k < nx) xv[k] = dl[xcol[k]] + xa[k];
            if (!(dbg & 1)) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const float v = dl[lo + w] + aw[w];
                    if (v > best) { best = v; arg = w; }
                }
            }
        } else if (!(dbg & 2)) {
            float d[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) d[e] = dl[i0 + e];
            VI p[EPL];
            if (is_pre) {
                // inclusive first-max prefix over this lane's sources, ascending
                VI run = vi_identity();
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const VI x{smask[e] ? -INFINITY : d[e] + c0, smask[e] ? kBig : i0 + e};
                    run = op_fwd(run, x);
                    p[e] = run;
                }
                const VI ex = dpp_shift_in<0x138>(wave_scan<false>(run));  // sources of all lower lanes
#pragma unroll
                for (int e = 0; e < EPL; ++e) Pp[i0 + e + 1] = op_fwd(ex, p[e]);
            } else {
                // inclusive first-max suffix over this lane's sources, descending
                VI run = vi_identity();
#pragma unroll
                for (int e = EPL - 1; e >= 0; --e) {
                    const VI x{smask[e] ? -INFINITY : d[e] + c0, smask[e] ? kBig : i0 + e};
                    run = op_rev(run, x);
                    p[e] = run;
                }
                const VI ex = dpp_shift_in<0x138>(wave_scan<true>(run));   // sources of all higher blocks
#pragma unroll
                for (int e = 0; e < EPL; ++e) Sf[i0 + e] = op_rev(ex, p[e]);
            }
            // dense rows: full first-max over every source
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int dr = (is_pre ? 0 : 1) + 2 * r;
                if (dr < nd) {
                    VI run = vi_identity();
                    if (is_pre) {
#pragma unroll
                        for (int e = 0; e < EPL; ++e)
                            run = op_fwd(run, VI{d[e] + (r ? dA1[e] : dA0[e]), i0 + e < S ? i0 + e : kBig});
                        run = wave_scan<false>(run);
                    } else {
#pragma unroll
                        for (int e = EPL - 1; e >= 0; --e)
                            run = op_rev(run, VI{d[e] + (r ? dA1[e] : dA0[e]), i0 + e < S ? i0 + e : kBig});
                        run = wave_scan<true>(run);
                    }
                    if (lane == 63) Dr[dr] = run;
                }
            }
        }
        __syncthreads();

        if (is_target && !(dbg & 4)) {
            // ---- merge in increasing source order, write delta_t and the back-pointer
            VI acc = Pp[lo];
            acc = op_fwd(acc, VI{best, arg == kBig ? kBig : lo + arg});
            acc = op_fwd(acc, Sf[lo + W]);
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k)
                if (k < nx && (xv[k] > acc.v || (xv[k] == acc.v && xcol[k] < acc.i && acc.i != kBig)))
                    acc = VI{xv[k], xcol[k]};  // acc.i == kBig <=> everything so far is -inf: stays "none" -> 0
            const VI dres = Dr[kind >= 0 ? kind : 0];
            if (kind >= 0) acc = dres;
            if (tvalid) {
                if (acc.i == kBig) acc.i = 0;
                dl[j] = acc.v + e_slot;
                if (!(dbg & 8)) {
                    psi[(size_t)t * a.SPSI + j] = (uint16_t)acc.i;
                    if (t + 2 < Tb) e_slot = load_e<ET>(E + (size_t)(t + 2) * S + j);
                }
            }
        }
        __syncthreads();
    };
    int t = 1;
    for (; t + 1 < Tb; t += 2) {
        frame(t, e_a);
        frame(t + 1, e_b);
    }
    if (t < Tb) frame(t, e_a);

    terminal_argmax(is_target ? dl[j] : -INFINITY, tid, S, tot, NWT + 2, a.last_state, a.loglik, song);
}

// ---------------------------------------------------------------------------------------
// Back-trace: one workgroup per song.  Tiles of K consecutive back-pointer rows are staged
// through LDS with coalesced 16-byte loads (the next tile is fetched into registers while
// lane 0 chases the current one), the chase itself runs on LDS latency, and the decoded
// states of a tile are written back coalesced.
// ---------------------------------------------------------------------------------------
constexpr int kBtThreads = 256;
constexpr int kBtMaxVec = 12;  // 16-byte vectors per thread per tile -> tile <= 48 KiB
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bt_fetch(u32x4 (&stage)[kBtMaxVec], const u32x4* __restrict__ psi, int top, int K,
                                         int rowv, int tid) {
    const int first = top - K + 1 > 1 ? top - K + 1 : 1;  // first row of the tile
    const int nvec = (top - first + 1) * rowv;
#pragma unroll
    for (int v = 0; v < kBtMaxVec; ++v) {
        const int idx = tid + v * kBtThreads;
        stage[v] = psi[(size_t)first * rowv + (idx < nvec ? idx : nvec - 1)];  // clamped: always in the tile
    }
}

__global__ void __launch_bounds__(kBtThreads) backtrace_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int K = a.K;
    const int rowv = a.SPSI / 8;                       // u32x4 per row
    u32x4* tile = reinterpret_cast<u32x4*>(smem);      // [kBtMaxVec*kBtThreads] (K*rowv of it used)
    int32_t* out = reinterpret_cast<int32_t*>(tile + kBtMaxVec * kBtThreads);  // [K]
    int32_t& s_cur = out[K];                                             // chase cursor (lane 0 only)

    const int song = blockIdx.x;
    const int tid = threadIdx.x;
    const int T = a.T;
    const int Tb = song_length(a.lengths, song, T);
    const u32x4* __restrict__ psi = reinterpret_cast<const u32x4*>(a.psi + (size_t)song * T * a.SPSI);
    int32_t* __restrict__ states = a.states + (size_t)song * T;

    for (int t = Tb + tid; t < T; t += kBtThreads) states[t] = -1;
    if (tid == 0) {
        s_cur = a.last_state[song];
        states[Tb - 1] = s_cur;
    }

    // rows t in [1, Tb-1] are consumed from the top; tile covers rows (hi-K, hi]
    u32x4 stage[kBtMaxVec];
    int hi = Tb - 1;
    if (hi >= 1) bt_fetch(stage, psi, hi, K, rowv, tid);
    while (hi >= 1) {
        const int first = hi - K + 1 > 1 ? hi - K + 1 : 1;
        const int rows = hi - first + 1;
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int v = 0; v < kBtMaxVec; ++v) tile[tid + v * kBtThreads] = stage[v];  // slots >= nvec hold clamped copies
        __syncthreads();
        const int next_hi = first - 1;
        if (next_hi >= 1) bt_fetch(stage, psi, next_hi, K, rowv, tid);
        if (tid == 0) {
            const uint16_t* rowsp = reinterpret_cast<const uint16_t*>(tile);
            int cur = s_cur;
            for (int r = rows - 1; r >= 0; --r) {
                cur = rowsp[(size_t)r * a.SPSI + cur];
                out[r] = cur;                 // state at frame first + r - 1
            }
            s_cur = cur;
        }
        __syncthreads();
        for (int r = tid; r < rows; r += kBtThreads) states[first + r - 1] = out[r];
        hi = next_hi;
    }
}

// voiced = state < n_bins; bins = min(state, n_bins-1)  (tonet/for_paper.py:1828-1829)
__global__ void voicing_map_kernel(const int32_t* __restrict__ states, int64_t n, int32_t n_bins,
                                   uint8_t* __restrict__ voiced, int32_t* __restrict__ bins) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = states[i];
        voiced[i] = (s >= 0 && s < n_bins) ? 1 : 0;
        bins[i] = s < 0 ? -1 : (s < n_bins - 1 ? s : n_bins - 1);
    }
}

// DPP scan self-test: out[lane] = inclusive first-max scan of (vals[lane], lane) per 64 lanes.
__global__ void scan_selftest_kernel(const float* __restrict__ vals, int rev, float* __restrict__ out_v,
                                     int32_t* __restrict__ out_i) {
    const int j = threadIdx.x + blockIdx.x * blockDim.x;
    VI x{vals[j], (int)threadIdx.x};
    x = rev ? wave_scan<true>(x) : wave_scan<false>(x);
    out_v[j] = x.v;
    out_i[j] = x.i;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <int NS, typename ET>
static hipError_t launch_dense_t(const FwdArgs& a, hipStream_t st) {
    const int SD = a.S4 * 4;
    const size_t lds = sizeof(float) * 2 * NS * SD + sizeof(VI) * 16;
    const int grid = (int)((a.B + NS - 1) / NS);
    hipLaunchKernelGGL((dense_forward_kernel<NS, ET>), dim3(grid), dim3(a.SP), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_dense(const FwdArgs& a, int ns, bool f16, hipStream_t st) {
    while (ns > 1 && a.SP > dense_max_threads(ns)) ns >>= 1;
    if (f16) {
        if (ns >= 4) return launch_dense_t<4, __half>(a, st);
        if (ns >= 2) return launch_dense_t<2, __half>(a, st);
        return launch_dense_t<1, __half>(a, st);
    }
    if (ns >= 4) return launch_dense_t<4, float>(a, st);
    if (ns >= 2) return launch_dense_t<2, float>(a, st);
    return launch_dense_t<1, float>(a, st);
}

template <int W, int NWT, typename ET>
static hipError_t launch_banded_t(const FwdArgs& a, hipStream_t st) {
    constexpr int NP = NWT * 64;
    const size_t lds = sizeof(float) * NP + sizeof(VI) * (2 * (NP + 1) + kMaxDenseRows + 16);
    hipLaunchKernelGGL((banded_forward_kernel<W, NWT, ET>), dim3((int)a.B), dim3((NWT + 2) * 64), lds, st, a);
    return hipGetLastError();
}

template <int W, typename ET>
static hipError_t launch_banded_w(const FwdArgs& a, hipStream_t st) {
    const int nwt = banded_target_waves(a.S, W);
    if constexpr (W <= 32) {
        switch (nwt) {
            case 2: return launch_banded_t<W, 2, ET>(a, st);
            case 4: return launch_banded_t<W, 4, ET>(a, st);
            case 6: return launch_banded_t<W, 6, ET>(a, st);
            case 8: return launch_banded_t<W, 8, ET>(a, st);
            case 12: return launch_banded_t<W, 12, ET>(a, st);
            default: return hipErrorInvalidConfiguration;
        }
    } else {
        switch (nwt) {
            case 2: return launch_banded_t<W, 2, ET>(a, st);
            case 4: return launch_banded_t<W, 4, ET>(a, st);
            case 6: return launch_banded_t<W, 6, ET>(a, st);
            default: return hipErrorInvalidConfiguration;
        }
    }
}

template <typename ET>
static hipError_t launch_banded_e(const FwdArgs& a, hipStream_t st) {
    switch (a.W) {
        case 16: return launch_banded_w<16, ET>(a, st);
        case 32: return launch_banded_w<32, ET>(a, st);
        case 64: return launch_banded_w<64, ET>(a, st);
        default: return hipErrorInvalidConfiguration;
    }
}

hipError_t launch_banded(const FwdArgs& a, bool f16, hipStream_t st) {
    return f16 ? launch_banded_e<__half>(a, st) : launch_banded_e<float>(a, st);
}

int backtrace_tile_rows(int SPSI) {
    int k = (kBtMaxVec * kBtThreads * 16) / (SPSI * 2);
    return k > 128 ? 128 : (k < 1 ? 1 : k);
}

hipError_t launch_backtrace(BtArgs a, hipStream_t st) {
    a.K = backtrace_tile_rows(a.SPSI);
    const size_t lds = (size_t)kBtMaxVec * kBtThreads * 16 + sizeof(int32_t) * (a.K + 1);
    hipLaunchKernelGGL(backtrace_kernel, dim3((int)a.B), dim3(kBtThreads), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_voicing_map(const int32_t* states, int64_t n, int32_t n_bins, uint8_t* voiced, int32_t* bins,
                              hipStream_t st) {
    if (n == 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(voicing_map_kernel, dim3((int)blocks), dim3(256), 0, st, states, n, n_bins, voiced, bins);
    return hipGetLastError();
}

hipError_t launch_scan_selftest(const float* vals, int n_waves, int rev, float* out_v, int32_t* out_i,
                                hipStream_t st) {
    hipLaunchKernelGGL(scan_selftest_kernel, dim3(n_waves), dim3(64), 0, st, vals, rev, out_v, out_i);
    return hipGetLastError();
}

}  // namespace vit
