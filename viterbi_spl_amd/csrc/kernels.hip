// kernels.hip -- gfx950 (MI355X) kernels for the float32 log-domain Viterbi decoder.
//
// Semantics (reference: imm/tf_viterbi.py:91-107, tonet/for_paper.py:1855-1868):
//   delta_0[j]  = fl32(log_pi[j] + logE[0][j])
//   m_j         = max_i fl32(delta_{t-1}[i] + logA_T[j][i]);  psi_t[j] = LOWEST i attaining it
//   delta_t[j]  = fl32(m_j + logE[t][j])
//   s_{T-1}     = lowest argmax_j delta_{T-1}[j];  s_t = psi_{t+1}[s_{t+1}]
// Only add / compare / select (built with -ffp-contract=off): bit-identical to the reference's
// NumPy float32 loop.
//
// "Lazy back-pointers": gfx950 retires one wave64 VALU instruction per 4 cycles per SIMD, and
// tracking the argmax index of every (frame, state) costs more instructions than the max itself,
// while the back-trace consumes ONE back-pointer per frame.  So
//   * the forward kernels are value-only (packed adds + max3) and store the delta row of every
//     frame (the reference's T1, tonet/for_paper.py:1852) instead of the back-pointer rows (T2);
//   * the back-trace recomputes psi_{t+1}[s_{t+1}] exactly -- the same fl32 sums, first index
//     attaining the max -- only for the state on the path.
// No MFMA: the recurrence is max-plus, not an add-contract.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "kernels.hpp"

namespace vit {

constexpr int kBig = 0x7fffffff;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct VI {
    float v;
    int i;
};

__device__ __forceinline__ VI vi_identity() { return VI{-INFINITY, kBig}; }
// first-max: `later` (higher index) replaces `earlier` only if strictly greater
__device__ __forceinline__ VI op_fwd(VI earlier, VI later) { return later.v > earlier.v ? later : earlier; }
// pieces visited in DESCENDING index order: the next (lower-index) piece wins ties
__device__ __forceinline__ VI op_rev(VI acc, VI next) { return next.v >= acc.v ? next : acc; }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ VI dpp_fetch(VI x) {
    VI r;
    r.v = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(x.v), CTRL,
                                                     ROW_MASK, 0xf, false));
    r.i = __builtin_amdgcn_update_dpp(kBig, x.i, CTRL, ROW_MASK, 0xf, false);
    return r;
}

// Inclusive wave64 scan with the ordered first-max operator (value, index).
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

// ---- value-only wave primitives
// Inclusive prefix max over lanes 0..lane: six v_max_f32 with a DPP source operand
// (row_shr:1/2/4/8, row_bcast:15 on rows 1,3, row_bcast:31 on rows 2,3).  A lane whose DPP source
// is invalid or whose row is masked is not written and keeps its own value.  The s_nop 1 pairs are
// the two wait states a DPP read of a just-written VGPR needs.
__device__ __forceinline__ float wave_scan_max(float x) {
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
    return x;
}
// lane l <- x[l-1], lane 0 <- fill   (wave_shr:1)
__device__ __forceinline__ float wave_shift_up(float x, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
// max over all 64 lanes, returned in every lane (wave-uniform)
__device__ __forceinline__ float wave_max_all(float x) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_scan_max(x)), 63));
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

// Workgroup-wide lowest-index argmax of delta (terminal state); every thread of the workgroup calls.
__device__ __forceinline__ void terminal_argmax(float dj, int j, bool valid, VI* tot, int nw, int32_t* last_state,
                                                float* loglik, int song) {
    VI x{valid ? dj : -INFINITY, valid ? j : kBig};
    x = wave_scan<false>(x);
    if ((threadIdx.x & 63) == 63) tot[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        VI acc = vi_identity();
        for (int b = 0; b < nw; ++b) acc = op_fwd(acc, tot[b]);
        if (acc.i == kBig) acc.i = 0;
        last_state[song] = acc.i;
        if (loglik) loglik[song] = acc.v;
    }
}

// ---------------------------------------------------------------------------------------
// Dense forward kernel (any matrix): NS songs per workgroup; every thread owns one target state
// and walks all S sources four at a time.  A4[q][j][0..3] = logA_T[j][4q..4q+3] is a coalesced
// 16-byte load per lane (L2 resident, 4*S*S bytes per frame), reused for the NS songs; the delta
// vectors are read from LDS as wave-uniform (broadcast) 16-byte reads.  Value-only: two packed
// adds and two max3 per four sources.
// ---------------------------------------------------------------------------------------
template <int NS, typename ET, int KT = 1>
__global__ void __launch_bounds__(KT == 2 ? 1024 : dense_max_threads(NS)) dense_forward_kernel(FwdArgs a) {
    // KT = 2: two threads per target, each walks half of the sources (twice the waves = twice the transition loads in
    // flight: the kernel is bound by the latency of streaming the matrix through L2, not by arithmetic); the halves meet
    // through LDS once per frame.
    extern __shared__ __align__(16) unsigned char smem[];
    const int S = a.S, SP = a.SP, S4 = a.S4, T = a.T, SD = a.SD;
    float* dl = reinterpret_cast<float*>(smem);  // [2][NS][SD]
    float* part = dl + 2 * NS * SD;              // [NS][SP] partial maxima of the upper half (KT == 2)
    VI* tot = reinterpret_cast<VI*>(part + (KT == 2 ? NS * SP : 0) + ((2 * NS * SD + (KT == 2 ? NS * SP : 0)) & 1));

    const int half = KT == 2 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >= (unsigned)SP)) : 0;   // SP is a multiple of 64
    const int j = threadIdx.x - half * SP;
    const bool lead = half == 0;                 // the thread that owns target j
    const int nw = blockDim.x >> 6;
    const int song0 = blockIdx.x * NS;
    const float4* __restrict__ A4 = reinterpret_cast<const float4*>(a.image + a.off_A4);
    const float* __restrict__ log_pi = reinterpret_cast<const float*>(a.image + a.off_logpi);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE);
    const int qmid = KT == 2 ? (S4 + 1) / 2 : S4;
    const int q_lo = half ? qmid : 0, q_hi = half ? S4 : qmid;

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
        if (lead && live[s] && j < S) {
            d = log_pi[j] + load_e<ET>(E + base + j);
            a.hist[(size_t)(song0 + s) * T * SD + j] = d;
        }
        if (lead && j < SD) { dl[(0 * NS + s) * SD + j] = d; dl[(1 * NS + s) * SD + j] = -INFINITY; }
        enext[s] = (lead && live[s] && j < S && Tb[s] > 1) ? load_e<ET>(E + base + S + j) : 0.f;
    }
    __syncthreads();

    int cur = 0;
    for (int t = 1; t < Tmax; ++t) {
        float ecur[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            ecur[s] = enext[s];
            if (lead && live[s] && j < S && t + 1 < Tb[s])
                enext[s] = load_e<ET>(E + ((size_t)(song0 + s) * T + t + 1) * S + j);
        }
        float b0[NS], b1[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) { b0[s] = -INFINITY; b1[s] = -INFINITY; }
        const float* dcur = dl + cur * NS * SD;
#pragma unroll 8
        for (int q = q_lo; q < q_hi; ++q) {
            const float4 av = A4[(size_t)q * SP + j];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const float4 dv = *reinterpret_cast<const float4*>(dcur + s * SD + 4 * q);
                b0[s] = fmaxf(fmaxf(b0[s], dv.x + av.x), dv.y + av.y);
                b1[s] = fmaxf(fmaxf(b1[s], dv.z + av.z), dv.w + av.w);
            }
        }
        if (KT == 2) {
            if (!lead) {
#pragma unroll
                for (int s = 0; s < NS; ++s) part[s * SP + j] = fmaxf(b0[s], b1[s]);
            }
            __syncthreads();
            if (lead) {
#pragma unroll
                for (int s = 0; s < NS; ++s) b0[s] = fmaxf(b0[s], part[s * SP + j]);
            }
        }
        float* dnxt = dl + (cur ^ 1) * NS * SD;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (lead && j < S) {
                if (live[s] && t < Tb[s]) {
                    const float dn = fmaxf(b0[s], b1[s]) + ecur[s];
                    dnxt[s * SD + j] = dn;
                    a.hist[((size_t)(song0 + s) * T + t) * SD + j] = dn;
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
            const bool valid = lead && j < S;
            const float dj = valid ? dl[(cur * NS + s) * SD + j] : -INFINITY;
            terminal_argmax(dj, j, valid, tot, nw, a.last_state, a.loglik, song0 + s);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// Dense forward kernel with the matrix RESIDENT on the CU (any matrix, 64 < S <= 368; one song per workgroup).
//
// dense_forward_kernel streams 4*S*S bytes of transition entries per frame and workgroup through a 64 B/clk vector L1:
// ~8100 cycles per frame at S = 361, four times the ~2000 its S*S packed adds and maxima take.  But a CU owns 512 KB of
// vector registers and 160 KB of LDS and the matrix is 521 KB: two threads per target, each holding its half row (HS = 184
// sources) as WRG = 132 registers + 13 float4 in LDS (conflict-free [q][thread] layout), keep every entry on the CU for the
// whole song.  A frame then reads only delta_{t-1}: the half row's 46 float4 as LDS broadcast reads (the even and the odd
// lanes of a wave read two addresses), 92 packed adds + 92 max3 per thread, one DPP exchange between the two lanes of a
// target, one barrier.  Same sums, same maxima as the streaming kernel: bit-identical history rows.
// ---------------------------------------------------------------------------------------
template <int HS, int WRG, int PF, typename ET>
__global__ void __launch_bounds__(768) dense_resident_forward_kernel(FwdArgs a) {
    static_assert(HS % 4 == 0 && WRG % 4 == 0 && WRG <= HS && PF % 2 == 0, "half rows in whole quads");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NT = 4 * HS;                   // threads that own a half row: two for each of up to 2*HS targets
    constexpr int QR = WRG / 4, QL = (HS - WRG) / 4;
    f32x4* awl = reinterpret_cast<f32x4*>(smem);             // [QL][NT]  the half rows' last QL quads
    float* dl = reinterpret_cast<float*>(awl + QL * NT);     // [2][2*HS] delta, double-buffered (entries >= S: -inf)
    VI* tot = reinterpret_cast<VI*>(dl + 4 * HS);
    const int S = a.S, SP = a.SP, S4 = a.S4, T = a.T, SD = a.SD;

    const int tid = threadIdx.x;
    const int j = tid >> 1, h = tid & 1;                     // target, half: lanes 2j and 2j+1 share target j
    const bool tvalid = j < S;
    const bool writer = tvalid && h == 0;
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* __restrict__ hist = a.hist + (size_t)song * T * SD;
    const int jc = tvalid ? j : S - 1;

    // ---------------- the half row: quads h*HS/4 .. of row j (quads beyond S4 and idle threads: -inf)
    const f32x4 ninf4 = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 aw[QR];
    {
        const f32x4* __restrict__ A4 = reinterpret_cast<const f32x4*>(a.image + a.off_A4);
        const int q0 = h * (HS / 4);
#pragma unroll
        for (int q = 0; q < QR; ++q) aw[q] = (tvalid && q0 + q < S4) ? A4[(size_t)(q0 + q) * SP + j] : ninf4;
        if (tid < NT) {
#pragma unroll
            for (int q = 0; q < QL; ++q) awl[q * NT + tid] = (tvalid && q0 + QR + q < S4) ? A4[(size_t)(q0 + QR + q) * SP + j] : ninf4;
        }
    }
    const int tl = tid < NT ? tid : 0;                       // (threads beyond NT own no target: j >= 2*HS >= S)
    for (int k = tid; k < 4 * HS; k += blockDim.x) dl[k] = -INFINITY;
    __syncthreads();

    // ---------------- frame 0
    {
        const float d0 = reinterpret_cast<const float*>(a.image + a.off_logpi)[jc] + load_e<ET>(E + jc);
        if (writer) { hist[j] = d0; dl[j] = d0; }
    }
    float er[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) er[k] = load_e<ET>(E + (size_t)(1 + k < Tb ? 1 + k : Tb - 1) * S + jc);
#pragma unroll
    for (int q = 0; q < QR; ++q) asm volatile("" ::"v"(aw[q].x), "v"(aw[q].y), "v"(aw[q].z), "v"(aw[q].w));
    __syncthreads();

    auto frame = [&](const int t, float& e_slot, const int RB) {
        const f32x4* __restrict__ dcur = reinterpret_cast<const f32x4*>(dl + RB * 2 * HS + h * HS);
        float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
        auto fold = [&](const f32x4 d, const f32x4 w, float& ma, float& mb) {
            const f32x2 c0_ = f32x2{d.x, d.y} + f32x2{w.x, w.y};
            const f32x2 c1_ = f32x2{d.z, d.w} + f32x2{w.z, w.w};
            ma = fmaxf(fmaxf(ma, c0_.x), c0_.y);
            mb = fmaxf(fmaxf(mb, c1_.x), c1_.y);
        };
        // Reads run one stage ahead of their use and never more: 132 weight registers leave ~16 for data in flight.
        // Register-resident weights: stages of two delta quads; weights in LDS: stages of one delta quad + its weight quad.
        {
            constexpr int NST = (QR + 1) / 2;
            f32x4 dv[2][2];
            auto issue = [&](const int c) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (2 * c + u < QR) dv[c & 1][u] = dcur[2 * c + u];
            };
            issue(0);
#pragma unroll
            for (int c = 0; c < NST; ++c) {
                if (c + 1 < NST) issue(c + 1);
                asm volatile("" ::: "memory");
                fold(dv[c & 1][0], aw[2 * c], m0, m1);
                if (2 * c + 1 < QR) fold(dv[c & 1][1], aw[2 * c + 1 < QR ? 2 * c + 1 : 0], m2, m3);
            }
        }
        {
            f32x4 dv[2], wv[2];
            auto issue = [&](const int c) {
                dv[c & 1] = dcur[QR + c];
                wv[c & 1] = awl[c * NT + tl];
            };
            if (QL > 0) issue(0);
#pragma unroll
            for (int c = 0; c < QL; ++c) {
                if (c + 1 < QL) issue(c + 1);
                asm volatile("" ::: "memory");
                if (c & 1) fold(dv[c & 1], wv[c & 1], m2, m3); else fold(dv[c & 1], wv[c & 1], m0, m1);
            }
        }
        float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        // the other half of the row: the neighbouring lane (quad_perm [1,0,3,2])
        m = fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), 0xB1, 0xf, 0xf, false)));
        const float dn = m + e_slot;
        const int tn = t + PF < Tb ? t + PF : Tb - 1;
        if (writer) {
            dl[(RB ^ 1) * 2 * HS + j] = dn;
            hist[(size_t)t * SD + j] = dn;
        }
        e_slot = load_e<ET>(E + (size_t)tn * S + jc);
        __syncthreads();
    };
    int t = 1;
    for (; t + PF - 1 < Tb; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) frame(t + k, er[k], k & 1);
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t + k < Tb) frame(t + k, er[k], k & 1);

    const int fb = (Tb - 1) & 1;                             // buffer holding delta_{Tb-1}
    terminal_argmax(writer ? dl[fb * 2 * HS + j] : -INFINITY, j, writer, tot, (int)(blockDim.x >> 6), a.last_state, a.loglik, song);
}

// ---------------------------------------------------------------------------------------
// Step-structured forward kernel (plan.step_ok: the Durrieu matrix of imm's own decoder, S = 722).
//
// For voiced source i and voiced target j, logA_T[j][i] = C[min(|i-j| / BW, KB)][i]: column i is piecewise constant in
// distance bands of BW bins and constant from distance KB*BW on.  So fl(delta_i + logA_T[j][i]) takes only KB+1 values
// per source: every thread publishes V_k[i] = fl(delta_i + C[k][i]) for its own state, and target j takes the max of
// the band windows V_k[j + k*BW .. j + k*BW + BW) and V_k(j - k*BW - BW .. j - k*BW] -- exactly the sums the dense
// recursion forms.  The far sources (distance >= KB*BW) reduce to ONE number, M = max_i V_KB[i]: if its arg-max is far
// from j it IS the far term, if it is near, its near-band value is >= M (no near band of a column is below the far
// value: checked by the plan; rounding is monotone) and every far term is <= M.  The unvoiced source contributes one
// value to every voiced target and joins M; the unvoiced target's row is arbitrary and gets a wave of its own.
// 2*KB*BW window reads per target instead of S global transition entries: the matrix is never streamed.
// One workgroup per song, NWV waves of voiced states + one wave for the unvoiced state, one barrier per frame
// (V is double-buffered).  Value-only like every forward kernel here; the generic back-trace follows.
// ---------------------------------------------------------------------------------------
template <int BW, int KB, int NWV, int PF, typename ET>
__global__ void __launch_bounds__((NWV + 1) * 64) step_forward_kernel(FwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NPV = NWV * 64;                 // padded voiced states
    constexpr int PAD = KB * BW + BW;             // -inf margin on both sides of every V_k
    constexpr int VLEN = NPV + 2 * PAD;
    constexpr int DLEN = NPV + 64;
    constexpr int NWM = (NWV + 1 + 3) / 4 * 4;    // wave maxima of V_KB + the unvoiced source's candidate
    float* V = reinterpret_cast<float*>(smem);    // [2][KB][VLEN]
    float* dl = V + 2 * KB * VLEN;                // [2][DLEN]  delta itself (for the unvoiced target's row)
    float* wm = dl + 2 * DLEN;                    // [2][NWM]
    VI* tot = reinterpret_cast<VI*>(wm + 2 * NWM);
    const int S = a.S, SP = a.SP, T = a.T, SD = a.SD;
    const int n = S - 1;                          // voiced states 0 .. n-1, unvoiced state n

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* __restrict__ hist = a.hist + (size_t)song * T * SD;
    const float* __restrict__ lpi = reinterpret_cast<const float*>(a.image + a.off_logpi);

    for (int k = tid; k < 2 * KB * VLEN + 2 * DLEN + 2 * NWM; k += (NWV + 1) * 64) V[k] = -INFINITY;
    __syncthreads();

    const bool voiced_wave = wv < NWV;
    const int j = tid;                                                    // voiced waves: own state
    const bool valid = voiced_wave && j < n;
    const int jld = valid ? j : n;                                        // emission column loaded (the unvoiced wave: column n)
    float c[KB + 1];                                                      // band values of source j
    {
        const float* __restrict__ sc = reinterpret_cast<const float*>(a.image + a.off_stepC);
#pragma unroll
        for (int k = 0; k <= KB; ++k) c[k] = valid ? sc[(size_t)k * SP + j] : -INFINITY;
    }
    constexpr int NQ = NWV + 1;                                           // sources per lane of the unvoiced target's wave
    float rown[NQ];
    {
        const float* __restrict__ ar = reinterpret_cast<const float*>(a.image + a.off_Arow) + (size_t)n * SP;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rown[q] = (!voiced_wave && lane + 64 * q < SP) ? ar[lane + 64 * q] : -INFINITY;   // -inf beyond S
    }
    const float cn = a.step_cn;

    float dn;                                                             // delta of the own state (unvoiced wave: of state n, every lane)
    {
        const float e0 = load_e<ET>(E + jld);
        dn = (valid || !voiced_wave) ? lpi[jld] + e0 : -INFINITY;
        if (valid || (!voiced_wave && lane == 0)) hist[jld] = dn;
    }
    float er[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) er[k] = load_e<ET>(E + (size_t)(1 + k < Tb ? 1 + k : Tb - 1) * S + jld);

    auto frame = [&](const int t, float& e_slot, const int b) {
        // ---- publish delta_{t-1} into buffer b
        if (voiced_wave) {
            float* vb = V + b * KB * VLEN + PAD + j;
#pragma unroll
            for (int k = 0; k < KB; ++k) vb[k * VLEN] = dn + c[k];
            if (valid) dl[b * DLEN + j] = dn;                              // (entry n belongs to the unvoiced wave)
            const float inc = wave_scan_max(dn + c[KB]);
            if (lane == 63) wm[b * NWM + wv] = inc;
        } else if (lane == 0) {
            dl[b * DLEN + n] = dn;
            wm[b * NWM + NWV] = dn + cn;
        }
        __syncthreads();
        // ---- consume
        float m;
        if (voiced_wave) {
            const float* rb = V + b * KB * VLEN + PAD + j;
            float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
            // band 0: distances 0 .. BW-1 on both sides (2*BW - 1 sources; one padded read keeps the pairs even)
#pragma unroll
            for (int r = -(BW - 1); r + 3 < BW + 1; r += 4) {
                m0 = fmaxf(fmaxf(m0, rb[r]), rb[r + 1]);
                m1 = fmaxf(fmaxf(m1, rb[r + 2]), r + 3 <= BW - 1 ? rb[r + 3] : -INFINITY);
            }
#pragma unroll
            for (int k = 1; k < KB; ++k) {
                const float* rk = rb + k * VLEN;
#pragma unroll
                for (int r = 0; r + 3 < BW; r += 4) {
                    m2 = fmaxf(fmaxf(m2, rk[k * BW + r]), rk[k * BW + r + 1]);
                    m3 = fmaxf(fmaxf(m3, rk[k * BW + r + 2]), rk[k * BW + r + 3]);
                    m0 = fmaxf(fmaxf(m0, rk[-k * BW - r]), rk[-k * BW - r - 1]);
                    m1 = fmaxf(fmaxf(m1, rk[-k * BW - r - 2]), rk[-k * BW - r - 3]);
                }
            }
            // the far sources and the unvoiced source: one maximum
            float M = -INFINITY;
#pragma unroll
            for (int q = 0; q < NWM / 4; ++q) {
                const f32x4 w = reinterpret_cast<const f32x4*>(wm + b * NWM)[q];
                M = fmaxf(fmaxf(fmaxf(M, w.x), w.y), fmaxf(w.z, w.w));
            }
            m = fmaxf(fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)), M);
        } else {
            // the unvoiced target: every source through its own (arbitrary) row
            float mm = -INFINITY;
#pragma unroll
            for (int q = 0; q < NQ; ++q) mm = fmaxf(mm, dl[b * DLEN + (lane + 64 * q < DLEN ? lane + 64 * q : DLEN - 1)] + rown[q]);
            m = wave_max_all(mm);
        }
        dn = (valid || !voiced_wave) ? m + e_slot : -INFINITY;
        const int tn = t + PF < Tb ? t + PF : Tb - 1;
        if (valid || (!voiced_wave && lane == 0)) hist[(size_t)t * SD + jld] = dn;
        e_slot = load_e<ET>(E + (size_t)tn * S + jld);
    };
    static_assert(BW % 4 == 0 && PF % 2 == 0, "band width in whole quads, even prefetch depth");
    int t = 1;
    for (; t + PF - 1 < Tb; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) frame(t + k, er[k], (k + 1) & 1);
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t + k < Tb) frame(t + k, er[k], (k + 1) & 1);

    __syncthreads();
    terminal_argmax(dn, voiced_wave ? j : n, valid || (!voiced_wave && lane == 0), tot, NWV + 1, a.last_state, a.loglik, song);
}

// ---------------------------------------------------------------------------------------
// Step-structured forward kernel, four targets per lane (what runs for the Durrieu matrix).
//
// Same arithmetic as step_forward_kernel.  What bounds that kernel is the LDS return path (359 four-byte window reads
// per target and frame).  With lane p owning states 4p .. 4p+3, the band windows of its four targets overlap in 17 of
// 20 sources, start on 16-byte boundaries (BW is a multiple of 4) and are read as contiguous float4s -- the ideal LDS
// pattern, a single copy of every V_k, 3.4x fewer bytes -- and the shared 17 sources are reduced once: ~15 instead of
// 40 max3 per band side for the four targets.  Three voiced waves + the unvoiced wave: one wave per SIMD.
// ---------------------------------------------------------------------------------------
template <int BW, int KB, int PF, typename ET>
__global__ void __launch_bounds__(256) step4_forward_kernel(FwdArgs a) {
    static_assert(BW == 20 && PF % 2 == 0, "written for 20-bin bands");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NW4 = 3;                        // voiced waves: 192 lanes x 4 states
    constexpr int NPV = NW4 * 256;                // padded voiced states
    constexpr int PAD = KB * BW + BW;             // -inf margin on both sides of every V_k (multiple of 4)
    constexpr int VLEN = NPV + 2 * PAD;
    constexpr int DLEN = NPV + 64;
    constexpr int FLEN = VLEN / 4;                // quads per V_k (PAD and NPV are multiples of 4)
    float* V = reinterpret_cast<float*>(smem);    // [2][KB][VLEN]
    float* F = V + 2 * KB * VLEN;                 // [2][KB][FLEN]  F_k[u] = max of the quad V_k[4u .. 4u+3]
    float* dl = F + 2 * KB * FLEN;                // [2][DLEN]  delta of the voiced states (for the unvoiced target's row)
    float* wm = dl + 2 * DLEN;                    // [2][4]     wave maxima of V_KB, slot 3 = the unvoiced source's candidate
    float* dun = wm + 2 * 4;                      // [2]        delta of the unvoiced state
    VI* tot = reinterpret_cast<VI*>(dun + 2 + 2);
    const int S = a.S, SP = a.SP, T = a.T, SD = a.SD;
    const int n = S - 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* __restrict__ hist = a.hist + (size_t)song * T * SD;
    const float* __restrict__ lpi = reinterpret_cast<const float*>(a.image + a.off_logpi);

    for (int k = tid; k < 2 * KB * VLEN + 2 * KB * FLEN + 2 * DLEN + 2 * 4 + 4; k += 256) V[k] = -INFINITY;
    __syncthreads();

    const bool voiced_wave = wv < NW4;
    const int j0 = 4 * tid;                                               // voiced lanes: first of the four own states
    bool val[4];
    int col[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        val[q] = voiced_wave && j0 + q < n;
        col[q] = voiced_wave ? (j0 + q < n ? j0 + q : n - 1) : n;          // emission column loaded (unvoiced wave: column n)
    }
    const bool all4 = voiced_wave && j0 + 3 < n;
    f32x4 c[KB + 1];
    {
        const float* __restrict__ sc = reinterpret_cast<const float*>(a.image + a.off_stepC);
#pragma unroll
        for (int k = 0; k <= KB; ++k) {
            c[k].x = val[0] ? sc[(size_t)k * SP + j0 + 0] : -INFINITY;
            c[k].y = val[1] ? sc[(size_t)k * SP + j0 + 1] : -INFINITY;
            c[k].z = val[2] ? sc[(size_t)k * SP + j0 + 2] : -INFINITY;
            c[k].w = val[3] ? sc[(size_t)k * SP + j0 + 3] : -INFINITY;
        }
    }
    constexpr int NQ = 13;                                                // sources per lane of the unvoiced target's wave
    float rown[NQ];
    {
        const float* __restrict__ ar = reinterpret_cast<const float*>(a.image + a.off_Arow) + (size_t)n * SP;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rown[q] = (!voiced_wave && lane + 64 * q < SP) ? ar[lane + 64 * q] : -INFINITY;
    }
    const float cn = a.step_cn;

    auto load4 = [&](const int row) -> f32x4 {
        const ET* __restrict__ r = E + (size_t)row * S;
        return f32x4{load_e<ET>(r + col[0]), load_e<ET>(r + col[1]), load_e<ET>(r + col[2]), load_e<ET>(r + col[3])};
    };
    auto store4 = [&](const int row, const f32x4 d) {
        float* __restrict__ h = hist + (size_t)row * SD;
        if (all4) {
            *reinterpret_cast<f32x4*>(h + j0) = d;
        } else if (voiced_wave) {
            if (val[0]) h[j0] = d.x;
            if (val[1]) h[j0 + 1] = d.y;
            if (val[2]) h[j0 + 2] = d.z;
        } else if (lane == 0) {
            h[n] = d.x;
        }
    };
    const f32x4 ninf4 = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    auto mask4 = [&](const f32x4 d) -> f32x4 {
        if (!voiced_wave) return d;
        return f32x4{val[0] ? d.x : -INFINITY, val[1] ? d.y : -INFINITY, val[2] ? d.z : -INFINITY, val[3] ? d.w : -INFINITY};
    };

    f32x4 dn;                                                             // delta of the own states (unvoiced wave: .x = state n, every lane)
    {
        const f32x4 e0 = load4(0);
        dn = mask4(f32x4{lpi[col[0]], lpi[col[1]], lpi[col[2]], lpi[col[3]]} + e0);
        store4(0, dn);
    }
    f32x4 er[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) er[k] = load4(1 + k < Tb ? 1 + k : Tb - 1);

    auto mx3 = [](float acc, float x, float y) { return fmaxf(fmaxf(acc, x), y); };

#ifdef VIT_TIMING_HOOKS
    const bool prof = (a.debug & 256) != 0;      // phase stamps: publish | barrier | consume | store + prefetch -> scratch[song][4*wave ..]
#else
    constexpr bool prof = false;
#endif
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;
    auto stamp = [&]() -> unsigned long long {
        unsigned long long v;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
        return v;
    };
    auto frame = [&](const int t, f32x4& e_slot, const int b) {
        const unsigned long long s0 = prof ? stamp() : 0ull;
        // ---- publish delta_{t-1} into buffer b
        if (voiced_wave) {
            float* vb = V + b * KB * VLEN + PAD + j0;
            float* fb = F + b * KB * FLEN + PAD / 4 + tid;
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                const f32x4 vq = dn + c[k];
                *reinterpret_cast<f32x4*>(vb + k * VLEN) = vq;
                fb[k * FLEN] = fmaxf(fmaxf(fmaxf(vq.x, vq.y), vq.z), vq.w);
            }
            *reinterpret_cast<f32x4*>(dl + b * DLEN + j0) = dn;
            const f32x4 vf = dn + c[KB];
            const float inc = wave_scan_max(fmaxf(fmaxf(vf.x, vf.y), fmaxf(vf.z, vf.w)));
            if (lane == 63) wm[b * 4 + wv] = inc;
        } else if (lane == 0) {
            dun[b] = dn.x;
            wm[b * 4 + 3] = dn.x + cn;
        }
        const unsigned long long s1 = prof ? stamp() : 0ull;
        __syncthreads();
        const unsigned long long s2 = prof ? stamp() : 0ull;
        // ---- consume
        f32x4 m = ninf4;
        if (voiced_wave) {
            const float* rb = V + b * KB * VLEN + PAD + j0;                 // 16-byte aligned; offsets below are relative to state j0
            const float* fr = F + b * KB * FLEN + PAD / 4 + tid;          // own quad's slot in F_0
            // A stage is one 24-source read window (six quads Q0 .. Q5 from a 16-byte aligned base): the four targets share the
            // four inner quads whole, so those arrive as their published maxima F (four floats) and only Q0 and Q5 are read in
            // full -- 2 ds_read_b128 + 4 ds_read_b32 and ~9 max operations per stage instead of 6 ds_read_b128 and ~19.
            // Stage 0, 1: band 0 (left-type window ending at the target, right-type window starting at it; the target's own
            // source is in both, which a maximum does not mind); stage 2k, 2k+1: band k right, left.
            // One wave per SIMD: nothing else covers the LDS latency, so the reads run LA stages ahead of the maxima.
            constexpr int LA = 3;                                         // stages in flight ahead of the one being reduced
            f32x4 q0[LA + 1], q5[LA + 1], fq[LA + 1];
            auto issue = [&](const int st) {
                const int k = st >> 1;
                const int qrel = st == 0 ? -5 : (st == 1 ? 0 : ((st & 1) ? -5 * k - 5 : 5 * k));     // first quad, relative to the own one
                const float* base = rb + k * VLEN + 4 * qrel;
                const float* fbase = fr + k * FLEN + qrel;
                q0[st % (LA + 1)] = *reinterpret_cast<const f32x4*>(base);
                q5[st % (LA + 1)] = *reinterpret_cast<const f32x4*>(base + 20);
                fq[st % (LA + 1)] = f32x4{fbase[1], fbase[2], fbase[3], fbase[4]};
            };
            float core = -INFINITY;                                       // sources every one of the four targets takes
            // right-type window (sources base .. base+23): target q takes read offsets q .. q+19
            auto reduce_right = [&](const int st) {
                const f32x4 a0 = q0[st % (LA + 1)], a5 = q5[st % (LA + 1)], f = fq[st % (LA + 1)];
                core = mx3(mx3(core, a0.w, f.x), f.y, fmaxf(f.z, f.w));
                const float lo2 = fmaxf(a0.y, a0.z), hi2 = fmaxf(a5.x, a5.y);
                m.x = mx3(m.x, a0.x, lo2);
                m.y = mx3(m.y, lo2, a5.x);
                m.z = mx3(m.z, a0.z, hi2);
                m.w = mx3(m.w, hi2, a5.z);
                // the element no target takes stays live up to here: a register that is dead on arrival is handed to the next
                // stage's address arithmetic, which then has to wait for this read to land (a queue drain per stage)
                asm volatile("" ::"v"(a5.w));
            };
            // left-type window (sources c-20 .. c+3, c the target-0 end of the window): target q takes read offsets q+1 .. q+20
            auto reduce_left = [&](const int st) {
                const f32x4 a0 = q0[st % (LA + 1)], a5 = q5[st % (LA + 1)], f = fq[st % (LA + 1)];
                core = mx3(mx3(core, a5.x, f.x), f.y, fmaxf(f.z, f.w));
                const float lo2 = fmaxf(a0.z, a0.w), hi2 = fmaxf(a5.y, a5.z);
                m.x = mx3(m.x, a0.y, lo2);
                m.y = mx3(m.y, lo2, a5.y);
                m.z = mx3(m.z, a0.w, hi2);
                m.w = mx3(m.w, hi2, a5.w);
                asm volatile("" ::"v"(a0.x));
            };
#pragma unroll
            for (int st = 0; st < LA; ++st) issue(st);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int st = 0; st < 2 * KB; ++st) {
                if (st + LA < 2 * KB) issue(st + LA);
                asm volatile("" ::: "memory");
                if (st & 1) {
                    if (st == 1) reduce_right(st); else reduce_left(st);
                } else {
                    if (st == 0) reduce_left(st); else reduce_right(st);
                }
            }
            m = f32x4{fmaxf(m.x, core), fmaxf(m.y, core), fmaxf(m.z, core), fmaxf(m.w, core)};
            // the far sources and the unvoiced source: one maximum
            const f32x4 w = *reinterpret_cast<const f32x4*>(wm + b * 4);
            const float M = fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w));
            m = f32x4{fmaxf(m.x, M), fmaxf(m.y, M), fmaxf(m.z, M), fmaxf(m.w, M)};
        } else {
            // the unvoiced target: every source through its own (arbitrary) row
            float mm = -INFINITY;
            const float du = dun[b];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int i = lane + 64 * q;                               // < DLEN; entries n .. DLEN-1 of dl hold -inf
                const float d = i == n ? du : dl[b * DLEN + i];
                mm = fmaxf(mm, d + rown[q]);
            }
            m.x = wave_max_all(mm);
        }
        dn = mask4(m + e_slot);
        const unsigned long long s3 = prof ? stamp() : 0ull;
        const int tn = t + PF < Tb ? t + PF : Tb - 1;
        store4(t, dn);
        e_slot = load4(tn);
        if (prof) {
            const unsigned long long s4 = stamp();
            ph0 += s1 - s0; ph1 += s2 - s1; ph2 += s3 - s2; ph3 += s4 - s3;
        }
    };
    int t = 1;
    for (; t + PF - 1 < Tb; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) frame(t + k, er[k], (k + 1) & 1);
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t + k < Tb) frame(t + k, er[k], (k + 1) & 1);
    if (prof && lane == 0 && Tb > 1) {
        float* o = a.fmax + (size_t)song * 64 + 4 * wv;
        const float nf = (float)(Tb - 1);
        o[0] = (float)ph0 / nf; o[1] = (float)ph1 / nf; o[2] = (float)ph2 / nf; o[3] = (float)ph3 / nf;
    }

    // terminal state: lowest-index argmax; a voiced lane holds four adjacent states
    __syncthreads();
    {
        VI x = vi_identity();
        if (voiced_wave) {
            if (val[0]) x = VI{dn.x, j0};
            if (val[1]) x = op_fwd(x, VI{dn.y, j0 + 1});
            if (val[2]) x = op_fwd(x, VI{dn.z, j0 + 2});
            if (val[3]) x = op_fwd(x, VI{dn.w, j0 + 3});
        } else if (lane == 0) {
            x = VI{dn.x, n};
        }
        x = wave_scan<false>(x);
        if (lane == 63) tot[wv] = x;
        __syncthreads();
        if (tid == 0) {
            VI acc = vi_identity();
            for (int bq = 0; bq < 4; ++bq) acc = op_fwd(acc, tot[bq]);
            if (acc.i == kBig) acc.i = 0;
            a.last_state[song] = acc.i;
            if (a.loglik) a.loglik[song] = acc.v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Step-structured forward kernel, four targets per lane, the bands split over two waves per lane group (what runs for
// the Durrieu matrix).
//
// Same arithmetic as step4_forward_kernel.  With one workgroup per CU (B = 256 on 256 CUs) that kernel has one wave per
// SIMD, and a lone wave issues one vector instruction per ~4 cycles: its ~50 publish and ~270 consume instructions ARE the
// frame time.  Here every group of 192 lanes x 4 states exists twice: waves 0-2 (half A) publish V_k and F_k for the bands
// k = 0..4 and delta itself and reduce the read stages 0..9, waves 3-5 (half B) take k = 5..8, the far-band maximum and the
// stages 10..17; the halves swap their partial maxima through LDS (one float4 each way) behind a second barrier and both form
// delta_t.  Two waves per SIMD issue alternately (2 cycles per instruction), each half moves half the bytes through the LDS
// store path, and with a barrier on either side of the reads V needs no second buffer: 60 KB of LDS instead of 112.
// ---------------------------------------------------------------------------------------
template <int BW, int KB, int PF, typename ET>
__global__ void __launch_bounds__(448) step4s_forward_kernel(FwdArgs a) {
    static_assert(BW == 20 && KB == 9 && PF % 2 == 0, "written for nine 20-bin bands");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NQL = 192;                      // lanes per half: 192 x 4 states
    constexpr int NPV = 4 * NQL;                  // padded voiced states
    constexpr int PAD = KB * BW + BW;             // -inf margin on both sides of every V_k (multiple of 4)
    constexpr int VLEN = NPV + 2 * PAD;
    constexpr int FLEN = VLEN / 4;
    constexpr int DLEN = NPV + 64;
    constexpr int KA = 5;                         // half A: bands 0 .. KA-1; half B: KA .. KB-1 and the far band KB
    constexpr int NST = 2 * KB, STA = 10;         // read stages; half A takes 0 .. STA-1 (half B also reduces the far band and scans)
    float* V = reinterpret_cast<float*>(smem);    // [KB][VLEN]
    float* F = V + KB * VLEN;                     // [KB][FLEN]  F_k[u] = max of the quad V_k[4u .. 4u+3]
    float* dl = F + KB * FLEN;                    // [DLEN]      delta of the voiced states (for the unvoiced target's row)
    f32x4* X = reinterpret_cast<f32x4*>(dl + DLEN);   // [2][NQL]    partial maxima of the two halves
    float* wm = reinterpret_cast<float*>(X + 2 * NQL);   // [4]  wave maxima of V_KB (half B), slot 3 = the unvoiced source's candidate
    float* dun = wm + 4;                          // [1 (+3)]    delta of the unvoiced state
    VI* tot = reinterpret_cast<VI*>(dun + 4);
    const int S = a.S, SP = a.SP, T = a.T, SD = a.SD;
    const int n = S - 1;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wv < 3 ? 0 : (wv < 6 ? 1 : 2);                        // 2: the unvoiced state's wave
    const int ql = tid - NQL * (half == 1 ? 1 : 0);                       // lane within the half (halves 0, 1)
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* __restrict__ hist = a.hist + (size_t)song * T * SD;
    const float* __restrict__ lpi = reinterpret_cast<const float*>(a.image + a.off_logpi);

    for (int k = tid; k < KB * VLEN + KB * FLEN + DLEN + 8 * NQL + 8; k += 448) V[k] = -INFINITY;
    __syncthreads();

    const bool voiced_wave = half < 2;
    const int j0 = 4 * ql;                                                // voiced lanes: first of the four own states
    bool val[4];
    int col[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        val[q] = voiced_wave && j0 + q < n;
        col[q] = voiced_wave ? (j0 + q < n ? j0 + q : n - 1) : n;          // emission column loaded (unvoiced wave: column n)
    }
    const bool all4 = voiced_wave && j0 + 3 < n;
    const int k0 = half == 1 ? KA : 0;                                    // first band of this half
    f32x4 c[KA];                                                          // half A: C_0 .. C_4; half B: C_5 .. C_8 and the far band C_9
    {
        const float* __restrict__ sc = reinterpret_cast<const float*>(a.image + a.off_stepC);
#pragma unroll
        for (int i = 0; i < KA; ++i) {
            const int k = k0 + i;
            c[i].x = val[0] ? sc[(size_t)k * SP + j0 + 0] : -INFINITY;
            c[i].y = val[1] ? sc[(size_t)k * SP + j0 + 1] : -INFINITY;
            c[i].z = val[2] ? sc[(size_t)k * SP + j0 + 2] : -INFINITY;
            c[i].w = val[3] ? sc[(size_t)k * SP + j0 + 3] : -INFINITY;
        }
    }
    constexpr int NQ = 13;                                                // sources per lane of the unvoiced target's wave
    float rown[NQ];
    {
        const float* __restrict__ ar = reinterpret_cast<const float*>(a.image + a.off_Arow) + (size_t)n * SP;
#pragma unroll
        for (int q = 0; q < NQ; ++q) rown[q] = (!voiced_wave && lane + 64 * q < SP) ? ar[lane + 64 * q] : -INFINITY;
    }
    const float cn = a.step_cn;

    auto load4 = [&](const int row) -> f32x4 {
        const ET* __restrict__ r = E + (size_t)row * S;
        return f32x4{load_e<ET>(r + col[0]), load_e<ET>(r + col[1]), load_e<ET>(r + col[2]), load_e<ET>(r + col[3])};
    };
    auto store4 = [&](const int row, const f32x4 d) {                     // half A and the unvoiced wave only
        float* __restrict__ h = hist + (size_t)row * SD;
        if (half == 0) {
            if (all4) {
                *reinterpret_cast<f32x4*>(h + j0) = d;
            } else {
                if (val[0]) h[j0] = d.x;
                if (val[1]) h[j0 + 1] = d.y;
                if (val[2]) h[j0 + 2] = d.z;
            }
        } else if (half == 2 && lane == 0) {
            h[n] = d.x;
        }
    };
    auto mask4 = [&](const f32x4 d) -> f32x4 {
        if (!voiced_wave) return d;
        return f32x4{val[0] ? d.x : -INFINITY, val[1] ? d.y : -INFINITY, val[2] ? d.z : -INFINITY, val[3] ? d.w : -INFINITY};
    };
    auto mx3 = [](float acc, float x, float y) { return fmaxf(fmaxf(acc, x), y); };

    f32x4 dn;                                                             // delta of the own states (unvoiced wave: .x = state n, every lane)
    {
        const f32x4 e0 = load4(0);
        dn = mask4(f32x4{lpi[col[0]], lpi[col[1]], lpi[col[2]], lpi[col[3]]} + e0);
        store4(0, dn);
    }
    f32x4 er[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) er[k] = load4(1 + k < Tb ? 1 + k : Tb - 1);

#ifdef VIT_TIMING_HOOKS
    const bool prof = (a.debug & 256) != 0;      // phase stamps: publish | barrier | reads | exchange + store -> scratch[song][4*wave ..]
#else
    constexpr bool prof = false;
#endif
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;
    auto stamp = [&]() -> unsigned long long {
        unsigned long long v;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
        return v;
    };

    const float* rb = V + PAD + j0;                                       // own quad in V_0 (16-byte aligned)
    const float* fr = F + PAD / 4 + ql;                                   // own quad's slot in F_0

    // reads of one stage (see step4_forward_kernel): Q0, Q5 in full, the four inner quads as their maxima
    struct Stage { f32x4 q0, q5, f; };
    auto issue = [&](const int st) -> Stage {
        const int k = st >> 1;
        const int qrel = st == 0 ? -5 : (st == 1 ? 0 : ((st & 1) ? -5 * k - 5 : 5 * k));
        const float* base = rb + k * VLEN + 4 * qrel;
        const float* fbase = fr + k * FLEN + qrel;
        Stage r;
        r.q0 = *reinterpret_cast<const f32x4*>(base);
        r.q5 = *reinterpret_cast<const f32x4*>(base + 20);
        r.f = f32x4{fbase[1], fbase[2], fbase[3], fbase[4]};
        return r;
    };

    auto frame = [&](const int t, f32x4& e_slot) {
        const unsigned long long s0 = prof ? stamp() : 0ull;
        // ---- publish delta_{t-1}
        if (voiced_wave) {
            float* vb = V + PAD + j0 + k0 * VLEN;
            float* fb = F + PAD / 4 + ql + k0 * FLEN;
#pragma unroll
            for (int i = 0; i < KA; ++i) {
                if (i < KA - 1 || half == 0) {                            // half B's fifth constant is the far band
                    const f32x4 vq = dn + c[i];
                    *reinterpret_cast<f32x4*>(vb + i * VLEN) = vq;
                    fb[i * FLEN] = fmaxf(fmaxf(fmaxf(vq.x, vq.y), vq.z), vq.w);
                }
            }
            if (half == 0) *reinterpret_cast<f32x4*>(dl + j0) = dn;
            if (half == 1) {
                const f32x4 vf = dn + c[KA - 1];
                const float inc = wave_scan_max(fmaxf(fmaxf(vf.x, vf.y), fmaxf(vf.z, vf.w)));
                if (lane == 63) wm[wv - 3] = inc;
            }
        } else if (lane == 0) {
            dun[0] = dn.x;
            wm[3] = dn.x + cn;
        }
        const unsigned long long s1 = prof ? stamp() : 0ull;
        __syncthreads();
        const unsigned long long s2 = prof ? stamp() : 0ull;
        // ---- reads
        f32x4 m = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        if (voiced_wave) {
            float core = -INFINITY;                                       // sources every one of the four targets takes
            auto reduce_right = [&](const Stage& r) {                     // target q takes read offsets q .. q+19
                const f32x4 a0 = r.q0, a5 = r.q5, f = r.f;
                core = mx3(mx3(core, a0.w, f.x), f.y, fmaxf(f.z, f.w));
                const float lo2 = fmaxf(a0.y, a0.z), hi2 = fmaxf(a5.x, a5.y);
                m.x = mx3(m.x, a0.x, lo2);
                m.y = mx3(m.y, lo2, a5.x);
                m.z = mx3(m.z, a0.z, hi2);
                m.w = mx3(m.w, hi2, a5.z);
                asm volatile("" ::"v"(a5.w));                             // (keeps a dead-on-arrival register out of the address arithmetic)
            };
            auto reduce_left = [&](const Stage& r) {                      // target q takes read offsets q+1 .. q+20
                const f32x4 a0 = r.q0, a5 = r.q5, f = r.f;
                core = mx3(mx3(core, a5.x, f.x), f.y, fmaxf(f.z, f.w));
                const float lo2 = fmaxf(a0.z, a0.w), hi2 = fmaxf(a5.y, a5.z);
                m.x = mx3(m.x, a0.y, lo2);
                m.y = mx3(m.y, lo2, a5.y);
                m.z = mx3(m.z, a0.w, hi2);
                m.w = mx3(m.w, hi2, a5.w);
                asm volatile("" ::"v"(a0.x));
            };
            constexpr int LA = 2;                                         // stages in flight ahead of the one being reduced
            auto run = [&](auto first, auto last) {
                constexpr int ST0 = decltype(first)::value, ST1 = decltype(last)::value;
                Stage buf[LA + 1];
#pragma unroll
                for (int st = ST0; st < ST0 + LA; ++st) buf[(st - ST0) % (LA + 1)] = issue(st);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int st = ST0; st < ST1; ++st) {
                    if (st + LA < ST1) buf[(st + LA - ST0) % (LA + 1)] = issue(st + LA);
                    asm volatile("" ::: "memory");
                    const bool left = st == 0 || (st >= 2 && (st & 1));
                    if (left) reduce_left(buf[(st - ST0) % (LA + 1)]); else reduce_right(buf[(st - ST0) % (LA + 1)]);
                }
            };
            if (half == 0) {
                run(std::integral_constant<int, 0>{}, std::integral_constant<int, STA>{});
            } else {
                run(std::integral_constant<int, STA>{}, std::integral_constant<int, NST>{});
                const f32x4 w = *reinterpret_cast<const f32x4*>(wm);        // the far sources and the unvoiced source: one maximum
                core = fmaxf(core, fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w)));
            }
            m = f32x4{fmaxf(m.x, core), fmaxf(m.y, core), fmaxf(m.z, core), fmaxf(m.w, core)};
            X[half * NQL + ql] = m;
        } else {
            // the unvoiced target: every source through its own (arbitrary) row
            float mm = -INFINITY;
            const float du = dun[0];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int i = lane + 64 * q;                               // < DLEN; entries n .. DLEN-1 of dl hold -inf
                const float d = i == n ? du : dl[i];
                mm = fmaxf(mm, d + rown[q]);
            }
            m.x = wave_max_all(mm);
        }
        const unsigned long long s3 = prof ? stamp() : 0ull;
        __syncthreads();
        // ---- both halves form delta_t
        if (voiced_wave) {
            const f32x4 o = X[(1 - half) * NQL + ql];
            m = f32x4{fmaxf(m.x, o.x), fmaxf(m.y, o.y), fmaxf(m.z, o.z), fmaxf(m.w, o.w)};
        }
        dn = mask4(m + e_slot);
        const int tn = t + PF < Tb ? t + PF : Tb - 1;
        store4(t, dn);
        e_slot = load4(tn);
        if (prof) {
            const unsigned long long s4 = stamp();
            ph0 += s1 - s0; ph1 += s2 - s1; ph2 += s3 - s2; ph3 += s4 - s3;
        }
    };
    int t = 1;
    for (; t + PF - 1 < Tb; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) frame(t + k, er[k]);
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t + k < Tb) frame(t + k, er[k]);
    if (prof && lane == 0 && Tb > 1) {
        float* o = a.fmax + (size_t)song * 64 + 4 * wv;
        const float nf = (float)(Tb - 1);
        o[0] = (float)ph0 / nf; o[1] = (float)ph1 / nf; o[2] = (float)ph2 / nf; o[3] = (float)ph3 / nf;
    }

    // terminal state: lowest-index argmax over half A's lanes (four adjacent states each) and the unvoiced state
    __syncthreads();
    {
        VI x = vi_identity();
        if (half == 0) {
            if (val[0]) x = VI{dn.x, j0};
            if (val[1]) x = op_fwd(x, VI{dn.y, j0 + 1});
            if (val[2]) x = op_fwd(x, VI{dn.z, j0 + 2});
            if (val[3]) x = op_fwd(x, VI{dn.w, j0 + 3});
        } else if (half == 2 && lane == 0) {
            x = VI{dn.x, n};
        }
        x = wave_scan<false>(x);
        if (lane == 63) tot[wv] = x;
        __syncthreads();
        if (tid == 0) {
            VI acc = vi_identity();
            for (int bq = 0; bq < 7; ++bq) acc = op_fwd(acc, tot[bq]);
            if (acc.i == kBig) acc.i = 0;
            a.last_state[song] = acc.i;
            if (a.loglik) a.loglik[song] = acc.v;
        }
    }
}

// ---------------------------------------------------------------------------------------
// Banded forward kernel: one song per workgroup, value-only.
//
// For a banded target j (window [lo_j, lo_j+W), row constant c_j, extra columns X):
//   m_j = max( max_w fl(delta_{lo_j+w} + logA_T[j][lo_j+w]),   W register-resident window entries
//              fl( max(Pv[lo_j], Sv[lo_j+W]) + c_j ),          Pv/Sv: prefix / suffix max of the RAW delta
//              fl(delta_x + logA_T[j][x]), x in X )             (extra columns are excluded from the scans)
// Rounding is monotone, so max_i fl(delta_i + c_j) = fl(max_i delta_i + c_j): the value equals the max
// of the fl32 sums the dense recursion forms, and delta is bit-identical.  Dense rows (none for the
// reference's matrices) are a full max over all sources.
//
// Wave roles (NWT target waves, 64*NWT >= S):
//   waves 0..NWT-1  one thread per target: window max, merge, delta_t, history row
//   wave  NWT       prefix-max scan over all sources (NWT per lane)
//   wave  NWT+1     suffix-max scan (lanes hold the sources in descending blocks)
//   wave  NWT+2     dense rows
// A SIMD retires one wave64 VALU instruction per 4 cycles, shared by the waves resident on it, so
// the frame time is set by the VALU instruction count of the busiest SIMD plus the two barriers.
// Two workgroup barriers per frame; emission rows are fetched two frames ahead.
// ---------------------------------------------------------------------------------------
// DBG = true adds the timing-experiment hooks (ablation mask, cycle stamps); the production instantiation has none.
// NXT >= 0 specialises for exactly NXT extra columns and no dense rows (the reference's matrices: NXT = 1);
// NXT < 0 is the generic form (run-time counts).  Every untaken branch costs an issue slot per frame.
template <int W, int NWT, bool DW, bool DBG, int NXT, typename ET>
__global__ void __launch_bounds__((NWT + (DW ? 3 : 2)) * 64) banded_forward_kernel(FwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NP = NWT * 64;
    constexpr int EPL = NWT;
    const int S = a.S, SP = a.SP, T = a.T, SD = a.SD;
    // delta_{t-1} is kept as four copies shifted by 0..3 entries (entry m of copy c = delta[m + c], copy c
    // at dls + c*DC + 4): every lane reads its window with aligned 16-byte LDS reads (256 B/clk instead of
    // the 128 B/clk of 4-/8-byte reads, which bound the window phase) from the copy that makes its window
    // start 16-byte aligned.  Each copy has 4 pad entries in front, so the four writes of a new delta value
    // (entry j - c of copy c) need no bounds test.  Entries >= S stay -inf.
    constexpr int DC = NP + 16;                   // copy stride = 16 banks mod 64: a 16-lane read group covers all 64 banks
    float* dls = reinterpret_cast<float*>(smem);  // [4][DC]
    float* dl = dls + 4;                          // copy 0 (unshifted)
    float* Pv = dls + 4 * DC;                     // [NP+1]  Pv[q] = max_{i<q}  raw delta (extras excluded)
    float* Sv = Pv + NP + 1;                      // [NP+1]  Sv[q] = max_{i>=q} raw delta
    float* Dv = Sv + NP + 1;                      // [4]     dense-row maxima
    VI* tot = reinterpret_cast<VI*>(Dv + kMaxDenseRows);  // [16] terminal argmax scratch (4*DC + 2*NP + 6 floats before: 8-byte aligned)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform for the compiler
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* __restrict__ hist = a.hist + (size_t)song * T * SD;
    constexpr bool GEN = NXT < 0;
    constexpr int NXL = GEN ? kMaxExtras : NXT;            // extra columns the loops are unrolled for
    const int nx = GEN ? a.n_extras : NXT, nd = GEN ? a.n_dense : 0;
#ifdef VIT_TIMING_HOOKS
    const int dbg = DBG ? a.debug : 0;
#else
    constexpr int dbg = 0;          // the ablation / probe hooks exist only in VIT_TIMING_HOOKS builds (scripts/)
#endif

    // ---------------- per-role setup
    const bool is_target = wv < NWT;
    const int j = tid;
    const bool tvalid = is_target && j < S;
    const int jst = j < S ? j : S + 1;                     // idle lanes store into pad column S+1 of the history row (SD >= S+2 always)
    const int jld = j < S ? j : S - 1;
    const int jc = j < SP ? j : 0;
    int lo = 0, kind = -2;
    float cj = 0.f;
    float aw[W];
    float xa[kMaxExtras];
    int xcol[kMaxExtras];
#pragma unroll
    for (int k = 0; k < kMaxExtras; ++k) { xa[k] = -INFINITY; xcol[k] = a.extras[k]; }
#pragma unroll
    for (int w = 0; w < W; ++w) aw[w] = 0.f;
    if (is_target) {
        cj = reinterpret_cast<const float*>(a.image + a.off_rowc)[jc];
        lo = reinterpret_cast<const int32_t*>(a.image + a.off_lo)[jc];
        kind = j < SP ? reinterpret_cast<const int32_t*>(a.image + a.off_kind)[jc] : -2;
        const float* __restrict__ tab = reinterpret_cast<const float*>(a.image + a.off_tabA);
        const float* __restrict__ xaT = reinterpret_cast<const float*>(a.image + a.off_extraA);
#pragma unroll
        for (int w = 0; w < W; ++w) aw[w] = tab[(size_t)w * SP + jc];
#pragma unroll
        for (int k = 0; k < kMaxExtras; ++k) xa[k] = xaT[(size_t)k * SP + jc];
    }
    const int role = wv - NWT;                 // 0 prefix, 1 suffix, 2 dense rows (DW) -- else the suffix wave reduces them
    constexpr int kDenseRole = DW ? 2 : 1;
    const int blk = role == 1 ? 63 - lane : lane;
    const int i0 = blk * EPL;
    bool smask[EPL];
    float dA[kMaxDenseRows][EPL];
    {
        const float* __restrict__ daT = reinterpret_cast<const float*>(a.image + a.off_denseA);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = i0 + e;
            bool m = i >= S;
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k) m |= (k < nx && i == xcol[k]);
            smask[e] = m;
#pragma unroll
            for (int d = 0; d < kMaxDenseRows; ++d)
                dA[d][e] = (role == kDenseRole && i < S && d < nd) ? daT[(size_t)d * SP + i] : -INFINITY;
        }
    }

    // ---------------- frame 0
    for (int k = tid; k < 4 * DC; k += blockDim.x) dls[k] = -INFINITY;
    __syncthreads();
    if (is_target) {
        if (tvalid) {
            const float d = reinterpret_cast<const float*>(a.image + a.off_logpi)[j] + load_e<ET>(E + j);
            hist[j] = d;
#pragma unroll
            for (int c = 0; c < 4; ++c) dl[c * DC + j - c] = d;
        }
    } else if (lane == 0) {
        if (role == 0) Pv[0] = -INFINITY;
        if (role == 1) Sv[NP] = -INFINITY;
    }
    // Emission rows are fetched two frames ahead and consumed only at the end of a frame: vmcnt
    // retires in order, so a wait on a younger load would also wait for the previous frame's store.
    float e_a = (tvalid && Tb > 1) ? load_e<ET>(E + S + j) : 0.f;
    float e_b = (tvalid && Tb > 2) ? load_e<ET>(E + 2 * (size_t)S + j) : 0.f;
    // Retire every set-up load here so the frame loop only sees the two memory ops it issues.
#pragma unroll
    for (int w = 0; w < W; ++w) asm volatile("" ::"v"(aw[w]));
#pragma unroll
    for (int k = 0; k < kMaxExtras; ++k) asm volatile("" ::"v"(xa[k]));
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
#pragma unroll
        for (int d = 0; d < kMaxDenseRows; ++d) asm volatile("" ::"v"(dA[d][e]));
    }
    asm volatile("" ::"v"(lo), "v"(kind), "v"(cj), "v"(e_a), "v"(e_b));
    __syncthreads();

    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0, ph3 = 0;   // timing experiments only (dbg & 256)
    auto stamp = [&]() -> unsigned long long {
        unsigned long long v;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");
        return v;
    };
    auto frame = [&](const int t, float& e_slot) {
        float m = -INFINITY;
        const bool prof = (dbg & 256) != 0;
        const unsigned long long s0 = prof ? stamp() : 0ull;
        if (is_target) {
            // ---- window max (reads delta_{t-1}); four independent max3 chains
            float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
            float xd[NXL > 0 ? NXL : 1];
#pragma unroll
            for (int k = 0; k < NXL; ++k) xd[k] = (!GEN || k < nx) ? dl[xcol[k]] : -INFINITY;
            if (!(dbg & 1)) {
                // copy (lo & 3), entry (lo & ~3): delta[lo .. lo+W) as W/4 aligned 16-byte reads
                const f32x4* __restrict__ win = reinterpret_cast<const f32x4*>(dl + (lo & 3) * DC + (lo & ~3));
#pragma unroll
                for (int w = 0; w + 7 < W; w += 8) {
                    const f32x4 da = win[w / 4], db = win[w / 4 + 1];
                    // v_pk_add_f32: two fl32 adds per instruction (each lane still rounds separately)
                    const f32x2 c0_ = f32x2{da.x, da.y} + f32x2{aw[w + 0], aw[w + 1]};
                    const f32x2 c1_ = f32x2{da.z, da.w} + f32x2{aw[w + 2], aw[w + 3]};
                    const f32x2 c2_ = f32x2{db.x, db.y} + f32x2{aw[w + 4], aw[w + 5]};
                    const f32x2 c3_ = f32x2{db.z, db.w} + f32x2{aw[w + 6], aw[w + 7]};
                    m0 = fmaxf(fmaxf(m0, c0_.x), c0_.y);
                    m1 = fmaxf(fmaxf(m1, c1_.x), c1_.y);
                    m2 = fmaxf(fmaxf(m2, c2_.x), c2_.y);
                    m3 = fmaxf(fmaxf(m3, c3_.x), c3_.y);
                }
            }
            // extra columns are consumed after the window so that their LDS read shares the window's wait
#pragma unroll
            for (int k = 0; k < NXL; ++k) m1 = fmaxf(m1, xd[k] + xa[k]);
            m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
        } else if (!(dbg & 2)) {
            float d[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) d[e] = dl[i0 + e];
            if (role == 0) {
                float p[EPL];
                float run = -INFINITY;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    run = fmaxf(run, smask[e] ? -INFINITY : d[e]);
                    p[e] = run;
                }
                const float inc = wave_scan_max(run);
                const float ex = wave_shift_up(inc, -INFINITY);                 // sources of all lower lanes
#pragma unroll
                for (int e = 0; e < EPL; ++e) Pv[i0 + e + 1] = fmaxf(ex, p[e]);
                // max of delta_{t-1} over the non-extra sources: bounds every row-constant candidate in the back-trace
                if (lane == 63 && !(dbg & 8)) hist[(size_t)(t - 1) * SD + S] = inc;   // pad column S of row t-1
            } else {
                if (role == 1) {
                    float p[EPL];
                    float run = -INFINITY;
#pragma unroll
                    for (int e = EPL - 1; e >= 0; --e) {
                        run = fmaxf(run, smask[e] ? -INFINITY : d[e]);
                        p[e] = run;
                    }
                    const float ex = wave_shift_up(wave_scan_max(run), -INFINITY);  // sources of all higher blocks
#pragma unroll
                    for (int e = 0; e < EPL; ++e) Sv[i0 + e] = fmaxf(ex, p[e]);
                }
                if (GEN && role == kDenseRole) {
#pragma unroll
                    for (int dr = 0; dr < kMaxDenseRows; ++dr) {
                        if (dr < nd) {
                            float dm = -INFINITY;
#pragma unroll
                            for (int e = 0; e < EPL; ++e) dm = fmaxf(dm, d[e] + dA[dr][e]);
                            dm = wave_max_all(dm);
                            if (lane == 0) Dv[dr] = dm;
                        }
                    }
                }
            }
        }
        const unsigned long long s1 = prof ? stamp() : 0ull;
        __syncthreads();
        const unsigned long long s2 = prof ? stamp() : 0ull;

        if (is_target && !(dbg & 4)) {
            m = fmaxf(m, fmaxf(Pv[lo], Sv[lo + W]) + cj);
            if (GEN) {
                const float dres = Dv[kind >= 0 ? kind : 0];
                if (kind >= 0) m = dres;
            }
            {
                // Every target lane stores and prefetches unconditionally (idle lanes: a pad column of the
                // history row / the last valid emission) so that the in-order vmcnt of the next use is exact
                // -- a conditional would make the compiler wait for the previous frame's store as well.
                const float dn = tvalid ? m + e_slot : -INFINITY;
#pragma unroll
                for (int c = 0; c < 4; ++c) dl[c * DC + j - c] = dn;
                if (!(dbg & 8)) {
                    float* __restrict__ hrow = hist + (size_t)t * SD;          // wave-uniform row bases:
                    const int tn = t + 2 < Tb ? t + 2 : Tb - 1;                 // scalar base + lane offset
                    const ET* __restrict__ erow = E + (size_t)tn * S;
                    hrow[jst] = dn;
                    e_slot = load_e<ET>(erow + jld);
                }
            }
        }
        const unsigned long long s3 = prof ? stamp() : 0ull;
        __syncthreads();
        if (prof) {
            const unsigned long long s4 = stamp();
            ph0 += s1 - s0; ph1 += s2 - s1; ph2 += s3 - s2; ph3 += s4 - s3;
        }
    };
    const unsigned long long clk0 = (dbg & 48) ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long rt0 = (dbg & 48) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int t = 1;
    for (; t + 1 < Tb; t += 2) {
        frame(t, e_a);
        frame(t + 1, e_b);
    }
    if (t < Tb) frame(t, e_a);

    terminal_argmax(is_target ? dl[j] : -INFINITY, j, tvalid, tot, NWT + (DW ? 3 : 2), a.last_state, a.loglik, song);
    if ((dbg & 256) && lane == 0 && Tb > 1) {   // per-wave phase averages -> fmax[song][4*wave .. 4*wave+3]
        float* o = a.fmax + (size_t)song * 64 + 4 * wv;
        const float n = (float)(Tb - 1);
        o[0] = (float)ph0 / n; o[1] = (float)ph1 / n; o[2] = (float)ph2 / n; o[3] = (float)ph3 / n;
    }
    if ((dbg & 48) && tid == 0) {  // timing experiments only: cycles (16) or 100 MHz ticks (32) per frame -> scratch slot 63
        const unsigned long long d = (dbg & 16) ? __builtin_amdgcn_s_memtime() - clk0 : __builtin_amdgcn_s_memrealtime() - rt0;
        a.fmax[(size_t)song * 64 + 63] = (float)d / (float)(Tb > 1 ? Tb - 1 : 1);
    }
}

// ---------------------------------------------------------------------------------------
// Banded forward kernel, "floor-max" form (plan.floor_ok): one song per workgroup, value-only, ONE barrier
// and no scan waves per frame.
//
// The plan proved that no in-window entry of a banded row is below the row constant c_j.  Let M be the max of
// the RAW delta_{t-1} over all non-extra sources, attained at i*.  If i* is outside the window of target j,
// fl(M + c_j) IS the out-of-window term.  If i* is inside, fl(M + c_j) <= fl(delta_i* + logA_T[j][i*]) (rounding
// is monotone and logA_T[j][i*] >= c_j), which the window max already contains, and every out-of-window term is
// <= fl(M + c_j) -- so in both cases
//   m_j = max( window max, fl(M + c_j), extra-column terms )
// is the value the dense recursion computes, bit for bit.  M is one number per frame: every wave reduces the
// delta values it has just produced (six DPP max steps) and publishes one float; after the frame's only barrier
// every lane combines the NWT wave maxima.  delta is double-buffered in LDS (four shifted copies each, see
// banded_forward_kernel), so nothing a wave reads in frame t is written before the barrier that ends frame t.
// M (= the back-trace's bound on every row-constant candidate) is stored in pad column S of the history row.
// ---------------------------------------------------------------------------------------
template <int W, int NWT, int NXT, int PF, typename ET>
__global__ void __launch_bounds__(NWT * 64) banded_floor_forward_kernel(FwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NP = NWT * 64;
    constexpr int DC = NP + 16;                   // copy stride (see banded_forward_kernel)
    constexpr int BUF = 4 * DC;                   // floats per delta buffer
    constexpr int NWM = (NWT + 3) / 4 * 4;        // wave maxima per buffer, padded to whole float4s with -inf
    float* dls = reinterpret_cast<float*>(smem);  // [2][4][DC]
    float* wm = dls + 2 * BUF;                    // [2][NWM]
    float* dump = wm + 2 * NWM;                   // [64 + NWM] per-lane dump slots (lanes that do not own a wave max)
    VI* tot = reinterpret_cast<VI*>(dump + 64 + NWM);
    // W = 128 with twelve waves (S > 512) leaves 168 registers per thread: the last 32 window weights then live in LDS
    // ([8][NP] float4-interleaved, read with conflict-free 16-byte reads next to the delta window)
    constexpr int WR = (W == 128 && NWT > 8) ? 96 : W;     // register-resident window weights
    f32x4* awl = reinterpret_cast<f32x4*>(tot + 16);       // [(W - WR) / 4][NP]
    const int S = a.S, SP = a.SP, T = a.T, SD = a.SD;
    constexpr bool GEN = NXT < 0;
    constexpr int NXL = GEN ? kMaxExtras : NXT;
    const int nx = GEN ? a.n_extras : NXT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* __restrict__ hist = a.hist + (size_t)song * T * SD;

    // ---------------- per-lane constants.  Idle lanes (j >= S) carry -inf tables: their delta stays -inf.
    const int j = tid;
    const bool tvalid = j < S;
    const int jc = tvalid ? j : 0;
    const int jld = tvalid ? j : S - 1;                                   // emission column an idle lane (harmlessly) loads
    // history store of frame t, relative to row t-1: own column of row t | lane S: M into pad column S of row t-1
    // | other idle lanes: pad column S+1 of row t (never read)
    const unsigned hoff = tvalid ? (unsigned)(SD + j) : (j == S ? (unsigned)S : (unsigned)(SD + S + 1));
    const bool is_fm = j == S;
    const int lo = reinterpret_cast<const int32_t*>(a.image + a.off_lo)[jc];
    const float cj = tvalid ? reinterpret_cast<const float*>(a.image + a.off_rowc)[jc] : -INFINITY;
    float aw[WR];
    float xa[NXL > 0 ? NXL : 1];
    int xcol[NXL > 0 ? NXL : 1];
    bool is_x = false;                                                    // this lane's state is an extra column: not part of M
    {
        const float* __restrict__ tab = reinterpret_cast<const float*>(a.image + a.off_tabA);
        const float* __restrict__ xaT = reinterpret_cast<const float*>(a.image + a.off_extraA);
#pragma unroll
        for (int w = 0; w < WR; ++w) aw[w] = tvalid ? tab[(size_t)w * SP + jc] : -INFINITY;
#pragma unroll
        for (int q = 0; q < (W - WR) / 4; ++q) {
            f32x4 wv4;
            wv4.x = tvalid ? tab[(size_t)(WR + 4 * q + 0) * SP + jc] : -INFINITY;
            wv4.y = tvalid ? tab[(size_t)(WR + 4 * q + 1) * SP + jc] : -INFINITY;
            wv4.z = tvalid ? tab[(size_t)(WR + 4 * q + 2) * SP + jc] : -INFINITY;
            wv4.w = tvalid ? tab[(size_t)(WR + 4 * q + 3) * SP + jc] : -INFINITY;
            awl[q * NP + j] = wv4;
        }
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            xcol[k] = k < nx ? a.extras[k] : 0;
            xa[k] = (tvalid && k < nx) ? xaT[(size_t)k * SP + jc] : -INFINITY;
            is_x |= (k < nx && j == xcol[k]);
        }
    }
    // delta[i] lives at float position 4 + sh + i - c of copy c (sh = a.win_shift): lane j reads its window from the copy
    // that makes delta[lo_j] 16-byte aligned.  With sh = lo_off mod 4 the sixteen lanes of one LDS read group start on
    // sixteen different 4-bank groups; without it the first and the last lane of a group collide (2-way conflict on
    // every ds_read_b128: 128 instead of 256 B/clk).
    const int sh = a.win_shift;
    const int lov = (tvalid ? lo : 0) + sh;
    const float* rp = dls + 4 + (lov & 3) * DC + (lov & ~3);              // window start in the copy that aligns it
    float* wp = dls + 4 + sh + j;                                         // own entry of copy 0 (copy c: + c*DC - c)
    float* wmp = lane == 63 ? wm + wv : dump + lane;                      // lane 63 ends up with the wave maximum

    for (int k = tid; k < 2 * BUF + 2 * NWM; k += NWT * 64) dls[k] = -INFINITY;
    __syncthreads();

    // produce(): publish a new delta value -- four shifted copies and the wave's share of M -- into buffer WB
    auto produce = [&](const float dn, const int WB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) wp[WB * BUF + c * DC - c] = dn;
        const float inc = wave_scan_max((NXL > 0 && is_x) ? -INFINITY : dn);
        wmp[WB * NWM] = inc;
    };

    // ---------------- frame 0
    {
        const float d0 = tvalid ? reinterpret_cast<const float*>(a.image + a.off_logpi)[j] + load_e<ET>(E + j) : -INFINITY;
        if (tvalid) hist[j] = d0;
        produce(d0, 0);
    }
    // Emission rows are fetched PF frames ahead (PF even): a global load takes ~2 us under load, several frame times,
    // and the s_waitcnt before a frame's "+ e" must not be what paces the recursion.
    float er[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) er[k] = load_e<ET>(E + (size_t)(1 + k < Tb ? 1 + k : Tb - 1) * S + jld);
#pragma unroll
    for (int w = 0; w < WR; ++w) asm volatile("" ::"v"(aw[w]));
#pragma unroll
    for (int k = 0; k < NXL; ++k) asm volatile("" ::"v"(xa[k]));
    asm volatile("" ::"v"(cj));
    __syncthreads();

    auto frame = [&](const int t, float& e_slot, const int RB) {
        const int WB = RB ^ 1;
        // ---- everything this frame reads from LDS: the window, the extra columns, the wave maxima
        const f32x4* __restrict__ win = reinterpret_cast<const f32x4*>(rp + RB * BUF);
        float xd[NXL > 0 ? NXL : 1];
        // the NWT wave maxima: whole float4s, plus one float2 when NWT % 4 is 2 or 3 (slots >= NWT hold -inf)
        f32x4 wq[NWT / 4 > 0 ? NWT / 4 : 1];
        f32x2 wr = f32x2{-INFINITY, -INFINITY};
        float wl = -INFINITY;
        // The small reads go out first and the first chunk of the window right behind them, and only then is M reduced: left to
        // itself the compiler reduces M before it issues the window reads -- a full LDS round trip with nothing else in flight.
        // (M reduced last instead lengthens the dependent tail after the last window read lands: measured slower.)
        auto small_reads = [&]() {
#pragma unroll
            for (int k = 0; k < NXL; ++k) xd[k] = dls[4 + sh + RB * BUF + xcol[k]];
#pragma unroll
            for (int q = 0; q < NWT / 4; ++q) wq[q] = reinterpret_cast<const f32x4*>(wm + RB * NWM)[q];
            if (NWT % 4 >= 2) wr = *reinterpret_cast<const f32x2*>(wm + RB * NWM + (NWT / 4) * 4);
            if (NWT % 4 == 1 || NWT % 4 == 3) wl = wm[RB * NWM + NWT - 1];
        };
        // the window in chunks of 32 sources (8 reads): wide windows (W = 96, 128) must not hold all their data at once
        float m0 = -INFINITY, m1 = -INFINITY, m2 = -INFINITY, m3 = -INFINITY;
        float M = -INFINITY;
        small_reads();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int w0 = 0; w0 < W; w0 += 32) {
        // W > 64: one chunk of reads in flight at a time (W register-resident weights leave no room for more; with
        // twelve waves per workgroup the other waves of the SIMD cover the read latency)
        if ((W > 64 || (W == 64 && NWT > 8)) && w0 > 0) asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3)::"memory");
        f32x4 dw[8];
#ifdef VIT_ABL_READS
        // result-breaking ablation (make TIMING=1 ABL=n builds only): read n of every chunk's window quads, the others reuse them --
        // what does the LDS return path cost a frame?
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (w0 + 4 * q < W) { if (q < VIT_ABL_READS) dw[q] = win[w0 / 4 + q]; else dw[q] = dw[q % VIT_ABL_READS]; }
#else
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (w0 + 4 * q < W) dw[q] = win[w0 / 4 + q];
#endif
        if (w0 == 0) {
            __builtin_amdgcn_sched_barrier(0);
            // M = max of delta_{t-1} over the non-extra sources
            M = wl;
            if (NWT % 4 >= 2) M = fmaxf(fmaxf(M, wr.x), wr.y);
#pragma unroll
            for (int q = 0; q < NWT / 4; ++q) M = fmaxf(fmaxf(fmaxf(M, wq[q].x), wq[q].y), fmaxf(wq[q].z, wq[q].w));
            m0 = M + cj;
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int w = w0; w + 7 < W && w < w0 + 32; w += 8) {
            const f32x4 da = dw[(w - w0) / 4], db = dw[(w - w0) / 4 + 1];
            f32x4 wa, wb;
            if (w < WR) {
                wa = f32x4{aw[w < WR ? w + 0 : 0], aw[w < WR ? w + 1 : 0], aw[w < WR ? w + 2 : 0], aw[w < WR ? w + 3 : 0]};
                wb = f32x4{aw[w < WR ? w + 4 : 0], aw[w < WR ? w + 5 : 0], aw[w < WR ? w + 6 : 0], aw[w < WR ? w + 7 : 0]};
            } else {
                wa = awl[((w - WR) / 4) * NP + j];
                wb = awl[((w - WR) / 4 + 1) * NP + j];
            }
            const f32x2 c0_ = f32x2{da.x, da.y} + f32x2{wa.x, wa.y};
            const f32x2 c1_ = f32x2{da.z, da.w} + f32x2{wa.z, wa.w};
            const f32x2 c2_ = f32x2{db.x, db.y} + f32x2{wb.x, wb.y};
            const f32x2 c3_ = f32x2{db.z, db.w} + f32x2{wb.z, wb.w};
            m0 = fmaxf(fmaxf(m0, c0_.x), c0_.y);
            m1 = fmaxf(fmaxf(m1, c1_.x), c1_.y);
            m2 = fmaxf(fmaxf(m2, c2_.x), c2_.y);
            m3 = fmaxf(fmaxf(m3, c3_.x), c3_.y);
        }
        if (W % 8 == 4 && w0 + 32 >= W) {          // W = 84: the last four sources (one read, two packed adds)
            static_assert(W % 8 != 4 || W <= WR, "an odd float4 count only with register-resident weights");
            const f32x4 da = dw[((W - 4 - w0) / 4) & 7];
            const f32x2 c0_ = f32x2{da.x, da.y} + f32x2{aw[W - 4], aw[W - 3]};
            const f32x2 c1_ = f32x2{da.z, da.w} + f32x2{aw[W - 2], aw[W - 1]};
            m2 = fmaxf(fmaxf(m2, c0_.x), c0_.y);
            m3 = fmaxf(fmaxf(m3, c1_.x), c1_.y);
        }
        }
#pragma unroll
        for (int k = 0; k < NXL; ++k) m1 = fmaxf(m1, xd[k] + xa[k]);
        const float dn = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3)) + e_slot;
        produce(dn, WB);
        asm volatile("" ::: "memory");   // keep the global store / prefetch behind the wave-max publication: they fill the
                                         // wait for the LDS write acknowledgement before the barrier (-2 %)
        // Unconditional store + prefetch: exact in-order vmcnt accounting (see banded_forward_kernel).  Row bases are
        // scalar index arithmetic on purpose: the SALU is idle, the VALU is not (running 64-bit per-lane pointers
        // measured 3.5% slower).
        const int tn = t + PF < Tb ? t + PF : Tb - 1;
        hist[(size_t)(t - 1) * SD + hoff] = is_fm ? M : dn;
        e_slot = load_e<ET>(E + (size_t)tn * S + jld);
        __syncthreads();
    };
#ifdef VIT_TIMING_HOOKS
    const bool probe = (a.debug & 48) != 0;
#else
    constexpr bool probe = false;   // cycle probe: VIT_TIMING_HOOKS builds only; it writes the per-song scratch, never an output
#endif
    const unsigned long long clk0 = probe ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long rt0 = probe ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int t = 1;
    for (; t + PF - 1 < Tb; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) frame(t + k, er[k], k & 1);
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t + k < Tb) frame(t + k, er[k], k & 1);

    const int fb = (Tb - 1) & 1;                                          // buffer holding delta_{Tb-1}
    terminal_argmax(tvalid ? dls[4 + sh + fb * BUF + j] : -INFINITY, j, tvalid, tot, NWT, a.last_state, a.loglik, song);
    if (probe && tid == 0) {  // timing experiments only: cycles (16) or 100 MHz ticks (32) per frame -> scratch slot 63
        const unsigned long long d = (a.debug & 16) ? __builtin_amdgcn_s_memtime() - clk0 : __builtin_amdgcn_s_memrealtime() - rt0;
        a.fmax[(size_t)song * 64 + 63] = (float)d / (float)(Tb > 1 ? Tb - 1 : 1);
    }
}

// ---------------------------------------------------------------------------------------
// Floor-max banded forward kernel, two targets per lane (plan.pair_ok && plan.floor_ok; what the bench runs).
//
// What bounds a frame of banded_floor_forward_kernel is the LDS return path: every target pulls its own W floats
// into registers (46 KB per frame at S = 361) and ds_read_b128 delivers ~128 B/clk per CU, ~380 of the ~870 cycles.
// The plan proves that the exception spans of targets 2p and 2p+1 together fit one window [lo2_p, lo2_p + W); entries
// of that window outside a row's own span are that row's constant (or an extra column), i.e. still >= c_j, so the
// floor-max identity holds for the common window.  One lane therefore evaluates BOTH targets from one set of W/4
// window reads: half the LDS traffic and half the waves (three at S = 361, one per SIMD), the same packed adds and
// max3 per target.  Everything else is as in banded_floor_forward_kernel.
// ---------------------------------------------------------------------------------------
template <int W, int NPW, int NXT, int PF, typename ET>
__global__ void __launch_bounds__(NPW * 64) banded_floor_pair_forward_kernel(FwdArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NP = NPW * 128;                 // padded state count
    constexpr int DC = NP + 16;                   // copy stride (see banded_forward_kernel)
    constexpr int BUF = 4 * DC;                   // floats per delta buffer
    constexpr int NWM = (NPW + 3) / 4 * 4;        // wave maxima per buffer, whole float4s (slots >= NPW hold -inf)
    float* dls = reinterpret_cast<float*>(smem);  // [2][4][DC]
    float* wm = dls + 2 * BUF;                    // [2][NWM]
    float* dump = wm + 2 * NWM;                   // [64 + NWM]
    VI* tot = reinterpret_cast<VI*>(dump + 64 + NWM);
    const int S = a.S, SP = a.SP, T = a.T, SD = a.SD;
    constexpr bool GEN = NXT < 0;
    constexpr int NXL = GEN ? kMaxExtras : NXT;
    const int nx = GEN ? a.n_extras : NXT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int song = blockIdx.x;
    const int Tb = song_length(a.lengths, song, T);
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* __restrict__ hist = a.hist + (size_t)song * T * SD;

    // ---------------- per-lane constants.  Slots past S carry -inf tables: their delta stays -inf.
    const int j0 = 2 * tid, j1 = 2 * tid + 1;
    const bool v0 = j0 < S, v1 = j1 < S;
    const int jc0 = v0 ? j0 : 0, jc1 = v1 ? j1 : jc0;
    const int jl0 = v0 ? j0 : S - 1, jl1 = v1 ? j1 : S - 1;               // emission columns (idle slots load a valid one)
    // history stores of frame t, relative to row t-1: own column of row t | slot S: M into pad column S of row t-1
    // | other idle slots: pad column S+1 of row t (never read)
    const unsigned hoff0 = v0 ? (unsigned)(SD + j0) : (j0 == S ? (unsigned)S : (unsigned)(SD + S + 1));
    const unsigned hoff1 = v1 ? (unsigned)(SD + j1) : (j1 == S ? (unsigned)S : (unsigned)(SD + S + 1));
    const bool fm0 = j0 == S, fm1 = j1 == S;
    const int lo2 = v0 ? reinterpret_cast<const int32_t*>(a.image + a.off_lo2)[jc0 >> 1] : 0;
    const float* __restrict__ rc = reinterpret_cast<const float*>(a.image + a.off_rowc);
    f32x2 cj = f32x2{v0 ? rc[jc0] : -INFINITY, v1 ? rc[jc1] : -INFINITY};
    float aw0[W], aw1[W];
    f32x2 xa[NXL > 0 ? NXL : 1];
    int xcol[NXL > 0 ? NXL : 1];
    bool x0 = false, x1 = false;                                          // slot is an extra column: not part of M
    {
        const float* __restrict__ tab = reinterpret_cast<const float*>(a.image + a.off_tabP);
        const float* __restrict__ xaT = reinterpret_cast<const float*>(a.image + a.off_extraA);
#pragma unroll
        for (int w = 0; w < W; ++w) {
            aw0[w] = v0 ? tab[(size_t)w * SP + jc0] : -INFINITY;
            aw1[w] = v1 ? tab[(size_t)w * SP + jc1] : -INFINITY;
        }
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            xcol[k] = k < nx ? a.extras[k] : 0;
            xa[k] = f32x2{(v0 && k < nx) ? xaT[(size_t)k * SP + jc0] : -INFINITY, (v1 && k < nx) ? xaT[(size_t)k * SP + jc1] : -INFINITY};
            x0 |= (k < nx && j0 == xcol[k]);
            x1 |= (k < nx && j1 == xcol[k]);
        }
    }
    // delta[i] lives at float position 4 + sh + i - c of copy c; the lane reads the common window from the copy that
    // makes delta[lo2] 16-byte aligned (sh: see banded_floor_forward_kernel)
    const int sh = a.win_shift2;
    const int lov = lo2 + sh;
    const float* rp = dls + 4 + (lov & 3) * DC + (lov & ~3);
    float* wp = dls + 4 + sh + j0;                                        // slot 0 of copy 0 (copy c: + c*DC - c), slot 1 follows
    float* wmp = lane == 63 ? wm + wv : dump + lane;

    for (int k = tid; k < 2 * BUF + 2 * NWM; k += NPW * 64) dls[k] = -INFINITY;
    __syncthreads();

    auto produce = [&](const f32x2 dn, const int WB) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            wp[WB * BUF + c * DC - c] = dn.x;
            wp[WB * BUF + c * DC - c + 1] = dn.y;
        }
        const float inc = wave_scan_max(fmaxf((NXL > 0 && x0) ? -INFINITY : dn.x, (NXL > 0 && x1) ? -INFINITY : dn.y));
        wmp[WB * NWM] = inc;
    };

    // ---------------- frame 0
    {
        const float* __restrict__ lpi = reinterpret_cast<const float*>(a.image + a.off_logpi);
        f32x2 d0 = f32x2{-INFINITY, -INFINITY};
        if (v0) { d0.x = lpi[j0] + load_e<ET>(E + j0); hist[j0] = d0.x; }
        if (v1) { d0.y = lpi[j1] + load_e<ET>(E + j1); hist[j1] = d0.y; }
        produce(d0, 0);
    }
    f32x2 er[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const ET* __restrict__ row = E + (size_t)(1 + k < Tb ? 1 + k : Tb - 1) * S;
        er[k] = f32x2{load_e<ET>(row + jl0), load_e<ET>(row + jl1)};
    }
#pragma unroll
    for (int w = 0; w < W; ++w) asm volatile("" ::"v"(aw0[w]), "v"(aw1[w]));
#pragma unroll
    for (int k = 0; k < NXL; ++k) asm volatile("" ::"v"(xa[k]));
    asm volatile("" ::"v"(cj));
    __syncthreads();

    auto frame = [&](const int t, f32x2& e_slot, const int RB) {
        const int WB = RB ^ 1;
        const f32x4* __restrict__ win = reinterpret_cast<const f32x4*>(rp + RB * BUF);
        float xd[NXL > 0 ? NXL : 1];
#pragma unroll
        for (int k = 0; k < NXL; ++k) xd[k] = dls[4 + sh + RB * BUF + xcol[k]];
        f32x4 wq[NWM / 4];
#pragma unroll
        for (int q = 0; q < NWM / 4; ++q) wq[q] = reinterpret_cast<const f32x4*>(wm + RB * NWM)[q];
        f32x4 dw[W / 4];
#pragma unroll
        for (int q = 0; q < W / 4; ++q) dw[q] = win[q];
        float m0 = -INFINITY, m1 = -INFINITY, n0 = -INFINITY, n1 = -INFINITY;   // two max3 chains per target
#pragma unroll
        for (int w = 0; w + 3 < W; w += 4) {
            const f32x4 d = dw[w / 4];
            const f32x2 a0 = f32x2{d.x, d.y} + f32x2{aw0[w + 0], aw0[w + 1]};
            const f32x2 a1 = f32x2{d.z, d.w} + f32x2{aw0[w + 2], aw0[w + 3]};
            const f32x2 b0 = f32x2{d.x, d.y} + f32x2{aw1[w + 0], aw1[w + 1]};
            const f32x2 b1 = f32x2{d.z, d.w} + f32x2{aw1[w + 2], aw1[w + 3]};
            m0 = fmaxf(fmaxf(m0, a0.x), a0.y);
            n0 = fmaxf(fmaxf(n0, a1.x), a1.y);
            m1 = fmaxf(fmaxf(m1, b0.x), b0.y);
            n1 = fmaxf(fmaxf(n1, b1.x), b1.y);
        }
        // M = max of delta_{t-1} over the non-extra sources; slots >= NPW of wq hold -inf
        float M = fmaxf(fmaxf(wq[0].x, wq[0].y), fmaxf(wq[0].z, wq[0].w));
#pragma unroll
        for (int q = 1; q < NWM / 4; ++q) M = fmaxf(fmaxf(fmaxf(M, wq[q].x), wq[q].y), fmaxf(wq[q].z, wq[q].w));
        const f32x2 fl = f32x2{M, M} + cj;
        m0 = fmaxf(m0, fl.x);
        m1 = fmaxf(m1, fl.y);
#pragma unroll
        for (int k = 0; k < NXL; ++k) {
            const f32x2 xv = f32x2{xd[k], xd[k]} + xa[k];
            n0 = fmaxf(n0, xv.x);
            n1 = fmaxf(n1, xv.y);
        }
        const f32x2 dn = f32x2{fmaxf(m0, n0), fmaxf(m1, n1)} + e_slot;
        produce(dn, WB);
        asm volatile("" ::: "memory");   // history stores / prefetch behind the wave-max publication
        const int tn = t + PF < Tb ? t + PF : Tb - 1;
        float* __restrict__ hb = hist + (size_t)(t - 1) * SD;
        const ET* __restrict__ erow = E + (size_t)tn * S;
        hb[hoff0] = fm0 ? M : dn.x;
        hb[hoff1] = fm1 ? M : dn.y;
        e_slot = f32x2{load_e<ET>(erow + jl0), load_e<ET>(erow + jl1)};
        __syncthreads();
    };
#ifdef VIT_TIMING_HOOKS
    const bool probe = (a.debug & 48) != 0;
#else
    constexpr bool probe = false;   // cycle probe: VIT_TIMING_HOOKS builds only; it writes the per-song scratch, never an output
#endif
    const unsigned long long clk0 = probe ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long rt0 = probe ? __builtin_amdgcn_s_memrealtime() : 0ull;
    int t = 1;
    for (; t + PF - 1 < Tb; t += PF) {
#pragma unroll
        for (int k = 0; k < PF; ++k) frame(t + k, er[k], k & 1);
    }
#pragma unroll
    for (int k = 0; k < PF - 1; ++k)
        if (t + k < Tb) frame(t + k, er[k], k & 1);

    // terminal state: lowest-index argmax of delta_{T-1}; a lane holds two adjacent states
    {
        const int fb = (Tb - 1) & 1;
        const float* fin = dls + 4 + sh + fb * BUF;
        VI x = vi_identity();
        if (v0) x = VI{fin[j0], j0};
        if (v1) x = op_fwd(x, VI{fin[j1], j1});
        x = wave_scan<false>(x);
        if (lane == 63) tot[wv] = x;
        __syncthreads();
        if (tid == 0) {
            VI acc = vi_identity();
            for (int b = 0; b < NPW; ++b) acc = op_fwd(acc, tot[b]);
            if (acc.i == kBig) acc.i = 0;
            a.last_state[song] = acc.i;
            if (a.loglik) a.loglik[song] = acc.v;
        }
    }
    if (probe && tid == 0) {  // timing experiments only: cycles (16) or 100 MHz ticks (32) per frame -> scratch slot 63
        const unsigned long long d = (a.debug & 16) ? __builtin_amdgcn_s_memtime() - clk0 : __builtin_amdgcn_s_memrealtime() - rt0;
        a.fmax[(size_t)song * 64 + 63] = (float)d / (float)(Tb > 1 ? Tb - 1 : 1);
    }
}

// ---------------------------------------------------------------------------------------
// Banded back-trace, lean form: banded plan without dense rows whose forward pass left the frame maximum in
// pad column S of every history row.  Same decisions as lazy_backtrace_kernel (below), organised for the
// dependent chain of one step -- state -> two LDS reads -> add -> wave max -> compare -> lowest matching lane:
//   * lane l < W holds window candidate l, lanes W.. hold the extra-column candidates, and lane 63 forms
//     fl(max_i delta_t[i] + c_j) with the same two reads (pad column S of the row, the row-constant table), so the
//     bound that admits the fast path costs no extra instructions;
//   * the candidate table is stored per target (tabX[j][.] contiguous: conflict-free), every index is
//     wave-uniform scalar arithmetic, decided states are collected in a register and written once per tile;
//   * the full evaluation (a row-constant candidate may tie or win) is a separate, rarely taken block.
// One wave per (song, chunk) in MODE 0 / per song in MODE 1, blockDim/64 waves per workgroup share the tables.
// ---------------------------------------------------------------------------------------
constexpr int kBtVec = 12;  // float4 per lane per tile: K * SD <= 12 * 256 floats

__device__ __forceinline__ void bt_fetch(f32x4 (&stage)[kBtVec], const f32x4* __restrict__ rows, int nvec, int lane) {
#pragma unroll
    for (int v = 0; v < kBtVec; ++v) {
        const int idx = lane + v * 64;
        stage[v] = rows[idx < nvec ? idx : nvec - 1];  // clamped: always inside the tile
    }
}

// KC: candidate slots per lane (slot k of lane l holds candidate c = 64k + l; candidates: W window entries, then the
// kMaxExtras extra-column entries, then the bound fl(M_t + c_j) formed from pad column S and the row constant).
// GT: the per-target candidate table [SP][W+5] is read from the plan image in global memory (L2-resident) instead of
// LDS -- at S = 722, W = 96 it is 310 KB and does not fit; a step then waits for one L2 access (~1 us) instead of an
// LDS access, still far cheaper than evaluating whole matrix rows.
template <int NWT, bool AFF, int MODE, int KC, bool GT>
__global__ void __launch_bounds__(512) banded_backtrace_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int EPL = NWT;               // sources per lane in the full evaluation, strided: i = e*64 + lane
    constexpr int TF = kBtVec * 256;       // floats per wave tile
    const int S = a.S, SP = a.SP, SD = a.SD, T = a.T, W = a.W, K = a.K;
    const int nx = a.n_extras;
    const int WX1 = W + kMaxExtras + 1;    // candidate-table row: window, extras, row constant
    const int CB = W + kMaxExtras;         // candidate index of the bound
    const int nwaves = blockDim.x >> 6;
    const float* L = reinterpret_cast<const float*>(smem);          // all LDS indices below are float indices into L
    float* tiles = reinterpret_cast<float*>(smem);                  // [nwaves][TF]
    int32_t* loL = reinterpret_cast<int32_t*>(tiles + nwaves * TF); // [SP]
    float* tabX = reinterpret_cast<float*>(loL + SP);               // [SP][WX1] (LDS form only)
    const float* __restrict__ gtab = reinterpret_cast<const float*>(a.image + a.off_tabX);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const int32_t* gl = reinterpret_cast<const int32_t*>(a.image + a.off_lo);
        const int nthr = blockDim.x;
        for (int k = tid; k < SP; k += nthr) loL[k] = gl[k];
        if (!GT)
            for (int k = tid; k < SP * WX1; k += nthr) tabX[k] = gtab[k];
    }
    __syncthreads();

    const int C = a.chunks;
    const int gw = blockIdx.x * nwaves + wv;            // global wave index
    const int song = MODE == 0 ? gw / C : gw;
    const int chunk = MODE == 0 ? gw % C : 0;
    if (song >= a.B) return;
    const int Tb = song_length(a.lengths, song, T);
    int32_t* __restrict__ states = a.states + (size_t)song * T;
    const float* __restrict__ hist = a.hist + (size_t)song * T * SD;
    float* tile = tiles + wv * TF;
    const int tile_off = wv * TF;

    // ---- per-lane constants, per candidate slot
    bool isw[KC], cand[KC];
    int pb[KC], tb[KC];
    unsigned long long wmask[KC];                                           // lanes of slot k that hold window candidates
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = 64 * k + lane;
        isw[k] = c < W;
        cand[k] = c < W + nx;
        const int xs = (c >= W && c < W + nx) ? a.extras[(c - W) & (kMaxExtras - 1)] : 0;
        pb[k] = c == CB ? a.mcol : a.col0 + (isw[k] ? c : xs);               // row entry read (window candidates: + lo)
        tb[k] = c < WX1 ? c : WX1 - 1;                                       // entry of the target's table row
        const int nwin = W - 64 * k;
        wmask[k] = nwin >= 64 ? ~0ull : (nwin <= 0 ? 0ull : ((1ull << nwin) - 1ull));
    }
    const int kb = CB >> 6, lb = CB & 63;                                    // slot / lane of the bound candidate
    const int tabX_off = (int)(tabX - tiles);
    int ic[EPL];
    bool isx[EPL], inS[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = e * 64 + lane;
        ic[e] = i < S ? a.col0 + i : a.col0;
        inS[e] = i < S;
        bool x = i >= S;
#pragma unroll
        for (int k = 0; k < kMaxExtras; ++k) x |= (k < nx && i == a.extras[k]);
        isx[e] = x;
    }
    const int rv = SD / 4;  // float4 per row

    // chase(top, bottom, cur, write): decide the states of frames top .. bottom (descending) from the delta rows
    // top .. bottom, starting from state `cur` at frame top+1; a tile holds rows [first, top].
    auto chase = [&](int top, const int bottom, int cur, const bool write) -> int {
        f32x4 stage[kBtVec];
        if (top >= bottom) {
            const int first = top - K + 1 > bottom ? top - K + 1 : bottom;
            bt_fetch(stage, reinterpret_cast<const f32x4*>(hist + (size_t)first * SD), (top - first + 1) * rv, lane);
        }
        while (top >= bottom) {
            const int first = top - K + 1 > bottom ? top - K + 1 : bottom;
            const int rows = top - first + 1;
#pragma unroll
            for (int v = 0; v < kBtVec; ++v) reinterpret_cast<f32x4*>(tile)[lane + v * 64] = stage[v];
            const int ntop = first - 1;
            if (ntop >= bottom) {
                const int nfirst = ntop - K + 1 > bottom ? ntop - K + 1 : bottom;
                bt_fetch(stage, reinterpret_cast<const f32x4*>(hist + (size_t)nfirst * SD), (ntop - nfirst + 1) * rv, lane);
            }
            int outv = 0;
            // MODE 1 re-chases a chunk whose assumed entry state was wrong: as soon as the new path meets the stored one the
            // rest of the chunk is already right (the step below a state depends on that state only)
            const int oldv = (MODE == 1 && lane < rows) ? states[first + lane] : -1;
            int rstop = -1;
            int row_off = __builtin_amdgcn_readfirstlane(tile_off + (rows - 1) * SD);
            for (int r = __builtin_amdgcn_readfirstlane(rows - 1); r >= 0; --r, row_off -= SD) {
                // row r of the tile = delta_t, t = first + r: decides the state at frame t from the state `cur` at t+1
                cur = __builtin_amdgcn_readfirstlane(cur);
                int lo;
                if (AFF) {
                    lo = cur - a.lo_off;
                    lo = lo < 0 ? 0 : (lo > S - W ? S - W : lo);
                } else {
                    lo = __builtin_amdgcn_readfirstlane(loL[cur]);
                }
                float v[KC], av[KC];
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const float dv = L[row_off + pb[k] + (isw[k] ? lo : 0)];
                    av[k] = GT ? gtab[(size_t)cur * WX1 + tb[k]] : L[tabX_off + (int)__umul24((unsigned)cur, (unsigned)WX1) + tb[k]];
                    v[k] = dv + av[k];
                }
                // the bound candidate: fl(max_i delta_t[i] + c_cur), on every row-constant candidate
                float mf = 0.f, cj = 0.f;
#pragma unroll
                for (int k = 0; k < KC; ++k)
                    if (KC == 1 || k == kb) {
                        mf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), lb));
                        cj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av[k]), lb));
                    }
                float mloc = -INFINITY;
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    v[k] = cand[k] ? v[k] : -INFINITY;
                    mloc = fmaxf(mloc, v[k]);
                }
                const float m = wave_max_all(mloc);
                // lowest source index among the window / extra-column candidates equal to `mm`
                auto lowest_candidate = [&](const float mm) -> unsigned {
                    unsigned best = 0x7fffffffu;
                    bool have_w = false;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const unsigned long long mk = __ballot(v[k] == mm && cand[k]);
                        const unsigned long long mw = mk & wmask[k];
                        if (mw && !have_w) {                                 // window candidates ascend with the source index
                            const unsigned c = lo + 64 * k + __builtin_ctzll(mw);
                            best = c < best ? c : best;
                            have_w = true;
                        }
                        unsigned long long mx = mk & ~wmask[k];              // extra-column candidates: arbitrary indices
                        while (mx) {
                            const unsigned c = __builtin_amdgcn_readlane(pb[k], __builtin_ctzll(mx)) - a.col0;   // column -> state
                            best = c < best ? c : best;
                            mx &= mx - 1;
                        }
                    }
                    return best;
                };
                unsigned idx = 0x7fffffffu;
                if (mf < m) {
                    // ---- common case: no row-constant candidate can tie or win
                    idx = lowest_candidate(m);
                } else {
                    // ---- full evaluation: every source outside the window / extras contributes fl(delta_t[i] + c_cur)
                    float vf[EPL];
                    float m2 = -INFINITY;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int i = e * 64 + lane;
                        const float d = L[row_off + ic[e]];
                        const bool excl = isx[e] || (unsigned)(i - lo) < (unsigned)W;
                        vf[e] = excl ? -INFINITY : d + cj;
                        m2 = fmaxf(m2, vf[e]);
                    }
                    const float mm = fmaxf(m, wave_max_all(m2));
                    // lowest index among the candidates equal to the max (an all -inf frame resolves to index 0
                    // like np.argmax: every in-range source then matches)
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const unsigned long long mk = __ballot(vf[e] == mm && inS[e]);
                        if (mk) { const unsigned c = e * 64 + __builtin_ctzll(mk); idx = c < idx ? c : idx; }
                    }
                    const unsigned c = lowest_candidate(mm);
                    idx = c < idx ? c : idx;
                    if (idx == 0x7fffffffu) idx = 0;
                }
                cur = (int)idx;
                outv = lane == r ? cur : outv;
                if (MODE == 1 && cur == __builtin_amdgcn_readlane(oldv, r)) { rstop = r; break; }
            }
            if (write && lane < rows && lane > rstop) states[first + lane] = outv;
            if (MODE == 1 && rstop >= 0) return __builtin_amdgcn_readfirstlane(states[bottom]);   // the stored path continues unchanged
            top = ntop;
        }
        return cur;
    };

    // Chunking, speculative warm-up and verification exactly as in lazy_backtrace_kernel.
    const int Lf = Tb - 1;
    if (MODE == 0) {
        const int lo_c = (int)((long long)Lf * chunk / C), hi_c = (int)((long long)Lf * (chunk + 1) / C);
        if (chunk == C - 1) {
            for (int t = Tb + lane; t < T; t += 64) states[t] = -1;
            if (lane == 0) states[Tb - 1] = a.last_state[song];
        }
        int top = hi_c - 1 + a.warm;
        int cur;
        if (chunk == C - 1 || top >= Lf - 1) {
            top = Lf - 1;
            cur = __builtin_amdgcn_readfirstlane(a.last_state[song]);
        } else {
            // guess: lowest-index argmax of delta row top+1
            const float* g = hist + (size_t)(top + 1) * SD + a.col0;
            float d[EPL];
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                d[e] = inS[e] ? g[e * 64 + lane] : -INFINITY;
                m = fmaxf(m, d[e]);
            }
            m = wave_max_all(m);
            unsigned idx = 0x7fffffffu;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const unsigned long long mk = __ballot(d[e] == m && inS[e]);
                if (mk) { const unsigned c = e * 64 + __builtin_ctzll(mk); idx = c < idx ? c : idx; }
            }
            cur = idx == 0x7fffffffu ? 0 : (int)idx;
        }
        if (hi_c <= lo_c) {                       // empty chunk (very short song)
            if (lane == 0) a.entry[(size_t)song * C + chunk] = cur;
            return;
        }
        cur = chase(top, hi_c, cur, false);       // warm-up: frames top .. hi_c, nothing written
        if (lane == 0) a.entry[(size_t)song * C + chunk] = cur;   // state this chunk assumed at frame hi_c
        chase(hi_c - 1, lo_c, cur, true);
    } else {
        int truth = -1;                           // verified state at frame hi_c of the chunk being checked
        for (int c = C - 2; c >= 0; --c) {
            const int lo_c = (int)((long long)Lf * c / C), hi_c = (int)((long long)Lf * (c + 1) / C);
            if (truth < 0) truth = __builtin_amdgcn_readfirstlane(states[hi_c]);
            const int assumed = __builtin_amdgcn_readfirstlane(a.entry[(size_t)song * C + c]);
            if (hi_c > lo_c && assumed != truth) {
                truth = chase(hi_c - 1, lo_c, truth, true);   // re-chase from the true state; ends at frame lo_c
            } else {
                truth = -1;                       // chunk c stands: its frame lo_c is already in `states`
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Lazy back-trace: one wave per song (kBtWaves songs per workgroup share the LDS tables).
// For frame t (descending) and the path state j at t+1 it rebuilds the candidates of target j
//   fl(delta_t[i] + logA_T[j][i])   for every source i
// from the stored delta row (window / c0 floor / extra columns / dense row, or the full matrix
// row for unstructured matrices), takes the max over the wave and picks the LOWEST index
// attaining it (v_cmp_eq lane masks + s_ff1).  Delta rows are staged through LDS in tiles of K
// frames; the next tile is in flight in registers while the current one is chased.
// ---------------------------------------------------------------------------------------
constexpr int kBtWaves = 4;

// MODE 0: speculative pass, one wave per (song, chunk).  MODE 1: verify pass, one wave per song.
template <int NWT, int MODE>
__global__ void __launch_bounds__(kBtWaves * 64) lazy_backtrace_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int EPL = NWT;               // sources per lane, strided: i = e*64 + lane
    const int S = a.S, SP = a.SP, SD = a.SD, T = a.T, W = a.W, K = a.K;
    const bool banded = a.banded != 0;
    const int nx = a.n_extras, nd = a.n_dense;
    const int WX = W + nx;                  // window candidates + extra-column candidates, one per lane
    const bool fast_ok = banded && a.have_fmax && WX <= 64;
    // LDS: [tile per wave: kBtVec*64 float4][out per wave: 64 ints]
    //      [tables: lo, kind, rowc, tabX[j][.] = the W window entries then the extra-column entries of target j]
    f32x4* tiles = reinterpret_cast<f32x4*>(smem);
    int32_t* outs = reinterpret_cast<int32_t*>(tiles + kBtWaves * kBtVec * 64);
    int32_t* loL = outs + kBtWaves * 64;
    int32_t* kindL = loL + SP;
    float* rowcL = reinterpret_cast<float*>(kindL + SP);  // [SP] row constants
    float* tabX = rowcL + SP;                             // [SP][WXS]: one target's candidates are contiguous (lane l reads entry l: no bank conflicts)
    const int WXS = W + kMaxExtras;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform for the compiler
    // step-structured dense matrix (plan.step_ok): the (step_kb+1) x SP band table replaces the matrix rows in LDS
    float* stepL = reinterpret_cast<float*>(loL);
    const bool step = !banded && a.step_ok != 0;
    if (step) {
        const float* gs = reinterpret_cast<const float*>(a.image + a.off_stepC);
        for (int k = tid; k < (a.step_kb + 1) * SP; k += kBtWaves * 64) stepL[k] = gs[k];
    }
    if (banded) {
        const int32_t* gl = reinterpret_cast<const int32_t*>(a.image + a.off_lo);
        const int32_t* gk = reinterpret_cast<const int32_t*>(a.image + a.off_kind);
        const float* gx = reinterpret_cast<const float*>(a.image + a.off_extraA);
        const float* gt = reinterpret_cast<const float*>(a.image + a.off_tabA);
        const float* gc = reinterpret_cast<const float*>(a.image + a.off_rowc);
        for (int k = tid; k < SP; k += kBtWaves * 64) { loL[k] = gl[k]; kindL[k] = gk[k]; rowcL[k] = gc[k]; }
        for (int k = tid; k < W * SP; k += kBtWaves * 64) tabX[(k % SP) * WXS + k / SP] = gt[k];
        for (int k = tid; k < kMaxExtras * SP; k += kBtWaves * 64) tabX[(k % SP) * WXS + W + k / SP] = gx[k];
    }
    __syncthreads();

    const int C = a.chunks;
    const int gw = blockIdx.x * kBtWaves + wv;          // global wave index
    const int song = MODE == 0 ? gw / C : gw;
    const int chunk = MODE == 0 ? gw % C : 0;
    if (song >= a.B) return;
    const int Tb = song_length(a.lengths, song, T);
    int32_t* __restrict__ states = a.states + (size_t)song * T;
    const float* __restrict__ hist = a.hist + (size_t)song * T * SD;
    const float* __restrict__ Arow = reinterpret_cast<const float*>(a.image + a.off_Arow);
    float* tile = reinterpret_cast<float*>(tiles + wv * kBtVec * 64);
    int32_t* out = outs + wv * 64;


    // loop invariants
    float dA[kMaxDenseRows][EPL];
    bool isx[EPL];                          // source excluded from the c0 floor: extra column or padding
    {
        const float* __restrict__ daT = reinterpret_cast<const float*>(a.image + a.off_denseA);
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int i = e * 64 + lane;
            bool x = i >= S;
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k) x |= (k < nx && i == a.extras[k]);
            isx[e] = x;
#pragma unroll
            for (int d = 0; d < kMaxDenseRows; ++d)
                dA[d][e] = (banded && d < nd && i < S) ? daT[(size_t)d * SP + i] : -INFINITY;
        }
    }
    // fast path: lane l < W evaluates window source lo + l, lane W + k evaluates extra column k
    const int xsrc = (lane >= W && lane < WX) ? a.extras[(lane - W) & (kMaxExtras - 1)] : 0;
    const unsigned long long wmask = W >= 64 ? ~0ull : ((1ull << W) - 1ull);

    // chase(top, bottom, cur, write): decide the states of frames top .. bottom (descending) from the
    // delta rows top .. bottom, starting from state `cur` at frame top+1; a tile holds rows [first, top].
    const int rv = SD / 4;  // float4 per row
    auto chase = [&](int top, const int bottom, int cur, const bool write) -> int {
    f32x4 stage[kBtVec];
    if (top >= bottom) {
        const int first = top - K + 1 > bottom ? top - K + 1 : bottom;
        bt_fetch(stage, reinterpret_cast<const f32x4*>(hist + (size_t)first * SD), (top - first + 1) * rv, lane);
    }
    while (top >= bottom) {
        const int first = top - K + 1 > bottom ? top - K + 1 : bottom;
        const int rows = top - first + 1;
#pragma unroll
        for (int v = 0; v < kBtVec; ++v) reinterpret_cast<f32x4*>(tile)[lane + v * 64] = stage[v];
        const int ntop = first - 1;
        if (ntop >= bottom) {
            const int nfirst = ntop - K + 1 > bottom ? ntop - K + 1 : bottom;
            bt_fetch(stage, reinterpret_cast<const f32x4*>(hist + (size_t)nfirst * SD), (ntop - nfirst + 1) * rv, lane);
        }
        const int oldv = (MODE == 1 && lane < rows) ? states[first + lane] : -1;   // see banded_backtrace_kernel
        int rstop = -1;
        for (int r = rows - 1; r >= 0; --r) {
            const float* row = tile + r * SD + a.col0;   // delta_t, t = first + r; decides the state at frame t
            const int jj = __builtin_amdgcn_readfirstlane(cur);  // path state at frame t+1 (wave-uniform)
            int lo = 0;
            int kd = -3;                         // -3 unstructured plan, -1 banded row, >= 0 dense row
            if (banded) {
                if (a.lo_affine) {
                    lo = jj - a.lo_off;
                    lo = lo < 0 ? 0 : (lo > S - W ? S - W : lo);
                    kd = -1;
#pragma unroll
                    for (int d = 0; d < kMaxDenseRows; ++d) kd = (d < nd && jj == a.dense_rows[d]) ? d : kd;
                } else {
                    kd = __builtin_amdgcn_readfirstlane(kindL[jj]);
                    lo = __builtin_amdgcn_readfirstlane(loL[jj]);
                }
            }
            bool done = false;
            if (fast_ok && kd == -1) {
                // ---- common case: only the window + extra-column candidates of target jj
                const int src = lane < W ? lo + lane : xsrc;
                float v = -INFINITY;
                if (lane < WX) v = row[src] + tabX[jj * WXS + lane];
                const float m = wave_max_all(v);
                const float mf = tile[r * SD + a.mcol] + rowcL[jj];  // column mcol >= max_i delta_t[i] over the row-constant sources: fl(. + c_jj) bounds every row-constant candidate
                if (mf < m) {                    // no row-constant candidate can tie or win
                    const unsigned long long mk = __ballot(v == m);
                    unsigned idx = 0x7fffffffu;
                    if (mk & wmask) idx = lo + __builtin_ctzll(mk & wmask);   // window lanes ascend with the source index
                    unsigned long long mx = W >= 64 ? 0ull : (mk >> W);       // extra-column lanes: arbitrary indices
                    while (mx) {
                        const unsigned c = a.extras[__builtin_ctzll(mx) & (kMaxExtras - 1)];
                        idx = c < idx ? c : idx;
                        mx &= mx - 1;
                    }
                    cur = (int)idx;
                    done = true;
                }
            }
            if (!done) {
                // ---- full evaluation: every source (c0 floor / window / extras / dense row / matrix row)
                float d[EPL], vf[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const int i = e * 64 + lane;
                    d[e] = i < S ? row[i] : -INFINITY;
                }
                float vw = -INFINITY;
                if (kd == -1) {
                    const float cjj = rowcL[jj];
                    const int src = lane < W ? lo + lane : xsrc;
                    if (lane < WX) vw = row[src] + tabX[jj * WXS + lane];
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int i = e * 64 + lane;
                        // window sources lo .. lo+63 are the lanes of vw; a wider window (W = 96, 128) continues here with
                        // its table entries; everything else outside the extra columns carries the row constant
                        const unsigned wi = (unsigned)(i - lo);
                        const bool in_vw = wi < (unsigned)(W < 64 ? W : 64);
                        const float wgt = (wi < (unsigned)W && !in_vw) ? tabX[jj * WXS + (int)wi] : cjj;
                        vf[e] = (isx[e] || in_vw) ? -INFINITY : d[e] + wgt;
                    }
                    if (WX > 64) {  // extras did not fit beside the window: fold them into the strided part
#pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const int i = e * 64 + lane;
#pragma unroll
                            for (int k = 0; k < kMaxExtras; ++k)
                                if (k < nx && i == a.extras[k]) vf[e] = d[e] + tabX[jj * WXS + W + k];
                        }
                    }
                } else if (kd >= 0) {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        float av = dA[0][e];
#pragma unroll
                        for (int q = 1; q < kMaxDenseRows; ++q) av = kd == q ? dA[q][e] : av;
                        vf[e] = d[e] + av;
                    }
                } else if (step && jj < S - 1) {
                    // voiced target of a step matrix: logA_T[jj][i] = stepC[min(|i-jj| / bw, kb)][i], unvoiced source: step_cn
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int i = e * 64 + lane;
                        const unsigned dist = (unsigned)(i > jj ? i - jj : jj - i);
                        unsigned band = (dist * (unsigned)a.step_mult) >> 16;          // dist / step_bw (host-checked for dist < 1024)
                        band = band < (unsigned)a.step_kb ? band : (unsigned)a.step_kb;
                        const float wgt = i < S - 1 ? stepL[band * SP + i] : (i == S - 1 ? a.step_cn : -INFINITY);
                        vf[e] = i < S ? d[e] + wgt : -INFINITY;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int i = e * 64 + lane;
                        vf[e] = i < S ? d[e] + Arow[(size_t)jj * SP + i] : -INFINITY;
                    }
                }
                float m = vw;
#pragma unroll
                for (int e = 0; e < EPL; ++e) m = fmaxf(m, vf[e]);
                m = wave_max_all(m);
                // lowest index among the candidates equal to the max (an all -inf frame resolves to
                // index 0 like np.argmax: every in-range source then matches)
                unsigned idx = 0x7fffffffu;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const unsigned long long mk = __ballot(vf[e] == m && e * 64 + lane < S);
                    if (mk) { const unsigned c = e * 64 + __builtin_ctzll(mk); idx = c < idx ? c : idx; }
                }
                if (kd == -1) {
                    const unsigned long long mk = __ballot(vw == m && lane < WX);
                    if (mk & wmask) { const unsigned c = lo + __builtin_ctzll(mk & wmask); idx = c < idx ? c : idx; }
                    unsigned long long mx = (W >= 64 || WX > 64) ? 0ull : (mk >> W);
                    while (mx) {
                        const unsigned c = a.extras[__builtin_ctzll(mx) & (kMaxExtras - 1)];
                        idx = c < idx ? c : idx;
                        mx &= mx - 1;
                    }
                }
                cur = idx == 0x7fffffffu ? 0 : (int)idx;
            }
            if (lane == 0) out[r] = cur;
            if (MODE == 1 && cur == __builtin_amdgcn_readlane(oldv, __builtin_amdgcn_readfirstlane(r))) { rstop = r; break; }
        }
        if (write)
            for (int r = lane; r < rows; r += 64)
                if (r > rstop) states[first + r] = out[r];
        if (MODE == 1 && rstop >= 0) return __builtin_amdgcn_readfirstlane(states[bottom]);
        top = ntop;
    }
    return cur;
    };

    // Frames 0 .. Tb-2 are decided (frame Tb-1 is the terminal state).  They are split into C chunks
    // [lo_c, hi_c); chunk c is chased from a warm-up point `a.warm` frames above hi_c, starting from
    // the best state of that frame (a guess); survivor paths coalesce, and MODE 1 verifies that the
    // state chunk c reached at frame hi_c equals what chunk c+1 (already verified) decided there --
    // if not, the chunk is chased again from the true state.  The result is exact either way.
    const int L = Tb - 1;
    if (MODE == 0) {
        const int lo_c = (int)((long long)L * chunk / C), hi_c = (int)((long long)L * (chunk + 1) / C);
        if (chunk == C - 1) {
            for (int t = Tb + lane; t < T; t += 64) states[t] = -1;
            if (lane == 0) states[Tb - 1] = a.last_state[song];
        }
        int top = hi_c - 1 + a.warm;
        int cur;
        if (chunk == C - 1 || top >= L - 1) {
            top = L - 1;
            cur = __builtin_amdgcn_readfirstlane(a.last_state[song]);
        } else {
            // guess: lowest-index argmax of delta row top+1
            const float* g = hist + (size_t)(top + 1) * SD + a.col0;
            float d[EPL];
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int i = e * 64 + lane;
                d[e] = i < S ? g[i] : -INFINITY;
                m = fmaxf(m, d[e]);
            }
            m = wave_max_all(m);
            unsigned idx = 0x7fffffffu;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const unsigned long long mk = __ballot(d[e] == m && e * 64 + lane < S);
                if (mk) { const unsigned c = e * 64 + __builtin_ctzll(mk); idx = c < idx ? c : idx; }
            }
            cur = idx == 0x7fffffffu ? 0 : (int)idx;
        }
        if (hi_c <= lo_c) {                       // empty chunk (very short song)
            if (lane == 0) a.entry[(size_t)song * C + chunk] = cur;
            return;
        }
        cur = chase(top, hi_c, cur, false);       // warm-up: frames top .. hi_c, nothing written
        if (lane == 0) a.entry[(size_t)song * C + chunk] = cur;   // state this chunk assumed at frame hi_c
        chase(hi_c - 1, lo_c, cur, true);
    } else {
        int truth = -1;                           // verified state at frame hi_c of the chunk being checked
        for (int c = C - 2; c >= 0; --c) {
            const int lo_c = (int)((long long)L * c / C), hi_c = (int)((long long)L * (c + 1) / C);
            if (truth < 0) truth = __builtin_amdgcn_readfirstlane(states[hi_c]);
            const int assumed = __builtin_amdgcn_readfirstlane(a.entry[(size_t)song * C + c]);
            if (hi_c > lo_c && assumed != truth) {
                truth = chase(hi_c - 1, lo_c, truth, true);   // re-chase from the true state; ends at frame lo_c
            } else {
                truth = -1;                       // chunk c stands: its frame lo_c is already in `states`
            }
        }
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

// DPP self-test: mode 0/1 = (value, index) first-max scan fwd/rev; mode 2 = value-only prefix max;
// mode 3 = wave_shift_up of the prefix max; mode 4 = wave_max_all.
__global__ void scan_selftest_kernel(const float* __restrict__ vals, int mode, float* __restrict__ out_v,
                                     int32_t* __restrict__ out_i) {
    const int j = threadIdx.x + blockIdx.x * blockDim.x;
    VI x{vals[j], (int)threadIdx.x};
    if (mode == 0) x = wave_scan<false>(x);
    else if (mode == 1) x = wave_scan<true>(x);
    else if (mode == 2) x.v = wave_scan_max(x.v);
    else if (mode == 3) x.v = wave_shift_up(wave_scan_max(x.v), -INFINITY);
    else x.v = wave_max_all(x.v);
    out_v[j] = x.v;
    out_i[j] = x.i;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <int NS, typename ET, int KT = 1>
static hipError_t launch_dense_t(const FwdArgs& a, hipStream_t st) {
    const size_t lds = sizeof(float) * (2 * NS * a.SD + (KT == 2 ? NS * a.SP : 0) + 1) + sizeof(VI) * 16;
    const int grid = (int)((a.B + NS - 1) / NS);
    hipLaunchKernelGGL((dense_forward_kernel<NS, ET, KT>), dim3(grid), dim3(KT * a.SP), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_step(const FwdArgs& a, bool f16, hipStream_t st) {
    constexpr int BW = 20, KB = 9, NWV = 12, PF = 2;
    if (!step_kernel_instantiated(a.S, a.step_bw, a.step_kb)) return hipErrorInvalidConfiguration;
    constexpr int VLEN = NWV * 64 + 2 * (KB * BW + BW);
    const size_t lds = sizeof(float) * (2 * KB * VLEN + 2 * (NWV * 64 + 64) + 2 * 16) + sizeof(VI) * 16;
    if (a.step_form == 0) {     // four targets per lane, bands split over two waves (step_form 2: one wave, 1: one target per lane)
        constexpr int VL4 = 768 + 2 * (KB * BW + BW);
        const size_t ldss = sizeof(float) * (KB * VL4 + KB * (VL4 / 4) + (768 + 64) + 8 * 192 + 8) + sizeof(VI) * 16;
        if (f16)
            hipLaunchKernelGGL((step4s_forward_kernel<BW, KB, PF, __half>), dim3((int)a.B), dim3(448), ldss, st, a);
        else
            hipLaunchKernelGGL((step4s_forward_kernel<BW, KB, PF, float>), dim3((int)a.B), dim3(448), ldss, st, a);
        return hipGetLastError();
    }
    if (a.step_form != 1) {
        constexpr int VL4 = 768 + 2 * (KB * BW + BW);
        const size_t lds4 = sizeof(float) * (2 * KB * VL4 + 2 * KB * (VL4 / 4) + 2 * (768 + 64) + 2 * 4 + 4) + sizeof(VI) * 16;
        if (f16)
            hipLaunchKernelGGL((step4_forward_kernel<BW, KB, PF, __half>), dim3((int)a.B), dim3(256), lds4, st, a);
        else
            hipLaunchKernelGGL((step4_forward_kernel<BW, KB, PF, float>), dim3((int)a.B), dim3(256), lds4, st, a);
        return hipGetLastError();
    }
    if (f16)
        hipLaunchKernelGGL((step_forward_kernel<BW, KB, NWV, PF, __half>), dim3((int)a.B), dim3((NWV + 1) * 64), lds, st, a);
    else
        hipLaunchKernelGGL((step_forward_kernel<BW, KB, NWV, PF, float>), dim3((int)a.B), dim3((NWV + 1) * 64), lds, st, a);
    return hipGetLastError();
}

// matrix-resident dense kernel: instantiated for half rows of 64, 128 and 184 sources (64 < S <= 368)
bool dense_resident_applies(const FwdArgs& a) { return a.S > 64 && a.S <= 368 && a.dense_form != 1; }

template <int HS, int WRG>
static hipError_t launch_dense_resident_t(const FwdArgs& a, bool f16, hipStream_t st) {
    constexpr int PF = 2;
    const int threads = (2 * a.S + 63) / 64 * 64;
    const size_t lds = sizeof(float) * 4 * ((HS - WRG) / 4) * (4 * HS) + sizeof(float) * 4 * HS + sizeof(VI) * 16;
    if (f16)
        hipLaunchKernelGGL((dense_resident_forward_kernel<HS, WRG, PF, __half>), dim3((int)a.B), dim3(threads), lds, st, a);
    else
        hipLaunchKernelGGL((dense_resident_forward_kernel<HS, WRG, PF, float>), dim3((int)a.B), dim3(threads), lds, st, a);
    return hipGetLastError();
}

static hipError_t launch_dense_resident(const FwdArgs& a, bool f16, hipStream_t st) {
    // half rows of 128 sources live in registers alone (eight waves); 184 sources: 132 registers + 13 float4 in LDS (twelve waves)
    if (a.S <= 128) return launch_dense_resident_t<64, 64>(a, f16, st);
    return a.S <= 256 ? launch_dense_resident_t<128, 128>(a, f16, st) : launch_dense_resident_t<184, 132>(a, f16, st);
}

hipError_t launch_dense(const FwdArgs& a, int ns, bool f16, hipStream_t st) {
    if (dense_resident_applies(a)) return launch_dense_resident(a, f16, st);
    while (ns > 1 && a.SP > dense_max_threads(ns)) ns >>= 1;
    // one song per workgroup: two threads per target when the workgroup still fits (S <= 512)
    if (ns == 1 && 2 * a.SP <= 1024 && !a.dense_kt1)
        return f16 ? launch_dense_t<1, __half, 2>(a, st) : launch_dense_t<1, float, 2>(a, st);
    if (f16) {
        if (ns >= 8) return launch_dense_t<8, __half>(a, st);
        if (ns >= 4) return launch_dense_t<4, __half>(a, st);
        if (ns >= 2) return launch_dense_t<2, __half>(a, st);
        return launch_dense_t<1, __half>(a, st);
    }
    if (ns >= 8) return launch_dense_t<8, float>(a, st);
    if (ns >= 4) return launch_dense_t<4, float>(a, st);
    if (ns >= 2) return launch_dense_t<2, float>(a, st);
    return launch_dense_t<1, float>(a, st);
}

// floor-max forms (plan.floor_ok; idle slot S stores the frame maximum: needs S < 64 * NWT)
template <int W, int NWT, typename ET>
static hipError_t launch_floor_t(const FwdArgs& a, hipStream_t st) {
    constexpr int NP = NWT * 64;
    constexpr int PF = W <= 32 ? 12 : 4;   // emission rows in flight: a row is requested PF frames (0.34 us each) before its use; under the overlapped
                                            // back-trace 4 left ~2 % on the table (B = 128: 4 -> 10.35, 8 -> 10.20, 12 -> 10.12, 16 -> 10.14 ms per forward pass)
    // Up to two songs per CU the one-target-per-lane kernel is (slightly) faster; beyond that the two-targets-per-lane
    // kernel wins because it moves half the window bytes through LDS (B = 512: 14.5 vs 15.5 ms).
    // FwdArgs::fwd_form 1 / 2 force one or the other.
    if constexpr (W <= 32 && NWT <= 8) {   // (at twelve waves, S = 722, the one-target kernel measured faster at every batch size)
        const bool pair = a.pair_ok && ((a.B > 256 && a.fwd_form != 1) || a.fwd_form == 2);
        if (pair) {
            constexpr int NPW = (NWT + 1) / 2;
            constexpr int PFP = 4;             // (two workgroups share a CU here and cover each other's waits: 12 rows in flight measured 9 % slower)
            constexpr int NWMP = (NPW + 3) / 4 * 4;
            const size_t ldsp = sizeof(float) * (8 * (NPW * 128 + 16) + 2 * NWMP + 64 + NWMP) + sizeof(VI) * 16;
            if (W == 32 && a.n_extras == 1)
                hipLaunchKernelGGL((banded_floor_pair_forward_kernel<W, NPW, (W == 32 ? 1 : -1), PFP, ET>), dim3((int)a.B), dim3(NPW * 64), ldsp, st, a);
            else
                hipLaunchKernelGGL((banded_floor_pair_forward_kernel<W, NPW, -1, PFP, ET>), dim3((int)a.B), dim3(NPW * 64), ldsp, st, a);
            return hipGetLastError();
        }
    }
    constexpr int NWM = (NWT + 3) / 4 * 4;
    const size_t ldsf = sizeof(float) * (8 * (NP + 16) + 2 * NWM + 64 + NWM) + sizeof(VI) * 16 +
                        ((W == 128 && NWT > 8) ? sizeof(f32x4) * 8 * NP : 0);
    if ((W == 32 || W >= 84) && a.n_extras == 1)   // the reference's matrices: band + unvoiced column (compile-time extras count)
        hipLaunchKernelGGL((banded_floor_forward_kernel<W, NWT, ((W == 32 || W >= 84) ? 1 : -1), PF, ET>), dim3((int)a.B), dim3(NWT * 64), ldsf, st, a);
    else
        hipLaunchKernelGGL((banded_floor_forward_kernel<W, NWT, -1, PF, ET>), dim3((int)a.B), dim3(NWT * 64), ldsf, st, a);
    return hipGetLastError();
}

// general (scan) form
template <int W, int NWT, typename ET>
static hipError_t launch_scan_t(const FwdArgs& a, hipStream_t st) {
    constexpr int NP = NWT * 64;
    const size_t lds = sizeof(float) * (4 * (NP + 16) + 2 * (NP + 1) + kMaxDenseRows) + sizeof(VI) * 16;
    // NWT + 2 waves put exactly two on each SIMD at S = 361 and let two workgroups share a CU.  Only a
    // plan with dense rows, run at one workgroup per CU, gets a separate wave for them (it would
    // otherwise lengthen the suffix wave, the critical one).
    if (a.n_dense > 0 && a.B <= 256)
        hipLaunchKernelGGL((banded_forward_kernel<W, NWT, true, false, -1, ET>), dim3((int)a.B), dim3((NWT + 3) * 64), lds, st, a);
#ifdef VIT_TIMING_HOOKS
    else if (a.debug)
        hipLaunchKernelGGL((banded_forward_kernel<W, NWT, false, true, -1, ET>), dim3((int)a.B), dim3((NWT + 2) * 64), lds, st, a);
#endif
    else if (W == 32 && a.n_dense == 0 && a.n_extras == 1)   // the reference's matrices: band + unvoiced column
        hipLaunchKernelGGL((banded_forward_kernel<W, NWT, false, false, (W == 32 ? 1 : -1), ET>), dim3((int)a.B), dim3((NWT + 2) * 64), lds, st, a);
    else if (W == 32 && a.n_dense == 0 && a.n_extras == 0)
        hipLaunchKernelGGL((banded_forward_kernel<W, NWT, false, false, (W == 32 ? 0 : -1), ET>), dim3((int)a.B), dim3((NWT + 2) * 64), lds, st, a);
    else
        hipLaunchKernelGGL((banded_forward_kernel<W, NWT, false, false, -1, ET>), dim3((int)a.B), dim3((NWT + 2) * 64), lds, st, a);
    return hipGetLastError();
}

template <int W, int NWT, typename ET>
static hipError_t launch_banded_t(const FwdArgs& a, hipStream_t st) {
    // fwd_form 3 forces the general (scan) kernel; so do the ablation bits of a VIT_TIMING_HOOKS build (48 = cycle probes
    // exist in the floor kernels as well)
    const bool floor_ok = a.floor_ok && a.S < NWT * 64 && a.fwd_form != 3 && !(a.debug & ~48);
    if constexpr (floor_form_instantiated(W, NWT)) {
        if (floor_ok) return launch_floor_t<W, NWT, ET>(a, st);
    }
    if constexpr (scan_form_instantiated(W, NWT)) return launch_scan_t<W, NWT, ET>(a, st);
    return hipErrorInvalidConfiguration;
}

template <int W, typename ET>
static hipError_t launch_banded_w(const FwdArgs& a, hipStream_t st) {
    switch (banded_waves_for(a.S)) {
        case 2: return launch_banded_t<W, 2, ET>(a, st);
        case 4: return launch_banded_t<W, 4, ET>(a, st);
        case 6: return launch_banded_t<W, 6, ET>(a, st);
        case 8: return launch_banded_t<W, 8, ET>(a, st);
        case 12: return launch_banded_t<W, 12, ET>(a, st);
        default: return hipErrorInvalidConfiguration;
    }
}

template <typename ET>
static hipError_t launch_banded_e(const FwdArgs& a, hipStream_t st) {
    switch (a.W) {
        case 16: return launch_banded_w<16, ET>(a, st);
        case 32: return launch_banded_w<32, ET>(a, st);
        case 64: return launch_banded_w<64, ET>(a, st);
        case 84: return launch_banded_w<84, ET>(a, st);
        case 96: return launch_banded_w<96, ET>(a, st);
        case 128: return launch_banded_w<128, ET>(a, st);
        default: return hipErrorInvalidConfiguration;
    }
}

hipError_t launch_banded(const FwdArgs& a, bool f16, hipStream_t st) {
    return f16 ? launch_banded_e<__half>(a, st) : launch_banded_e<float>(a, st);
}

int backtrace_tile_rows(int SD) {
    int k = (kBtVec * 256) / SD;
    return k > 64 ? 64 : (k < 1 ? 1 : k);
}

constexpr size_t kLdsBytes = 160 * 1024;

template <int NWT, bool AFF, int KC, bool GT>
static hipError_t launch_bt_lean(const BtArgs& a, int nwaves, size_t lds, hipStream_t st) {
    const long long waves0 = (long long)a.B * a.chunks;
    hipLaunchKernelGGL((banded_backtrace_kernel<NWT, AFF, 0, KC, GT>), dim3((int)((waves0 + nwaves - 1) / nwaves)), dim3(nwaves * 64), lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.chunks <= 1) return e;
    hipLaunchKernelGGL((banded_backtrace_kernel<NWT, AFF, 1, KC, GT>), dim3((int)((a.B + nwaves - 1) / nwaves)), dim3(nwaves * 64), lds, st, a);
    return hipGetLastError();
}

template <int NWT, int KC, bool GT>
static hipError_t launch_bt_lean_a(const BtArgs& a, int nwaves, size_t lds, hipStream_t st) {
    return a.lo_affine ? launch_bt_lean<NWT, true, KC, GT>(a, nwaves, lds, st) : launch_bt_lean<NWT, false, KC, GT>(a, nwaves, lds, st);
}

template <int NWT>
static hipError_t launch_bt_t(BtArgs a, hipStream_t st) {
    // lean kernel: banded plan, no dense rows, frame maxima stored by the forward pass
    if constexpr (NWT <= 12) {
        if (a.banded && a.have_fmax && a.n_dense == 0 && a.W <= 128 && a.bt_form != 1) {
            const int kc = (a.W + kMaxExtras + 1 + 63) / 64;
            const size_t tile = sizeof(f32x4) * kBtVec * 64, lo_tab = sizeof(int32_t) * a.SP;
            const size_t table = sizeof(float) * (size_t)a.SP * (a.W + kMaxExtras + 1);
            // candidate table in LDS when it leaves room for at least four waves, else read from the image (L2)
            if (kc == 1 && 4 * tile + lo_tab + table + 1024 <= kLdsBytes) {
                const int nw = 8 * tile + lo_tab + table + 1024 <= kLdsBytes ? 8 : 4;
                return launch_bt_lean_a<NWT, 1, false>(a, nw, nw * tile + lo_tab + table, st);
            }
            const size_t lds = 8 * tile + lo_tab;
            if (kc == 1) return launch_bt_lean_a<NWT, 1, true>(a, 8, lds, st);
            if (kc == 2) return launch_bt_lean_a<NWT, 2, true>(a, 8, lds, st);
            if (kc == 3) return launch_bt_lean_a<NWT, 3, true>(a, 8, lds, st);
        }
    }
    size_t lds = sizeof(f32x4) * kBtWaves * kBtVec * 64 + sizeof(int32_t) * kBtWaves * 64;
    if (!a.banded && a.step_ok) lds += sizeof(float) * (a.step_kb + 1) * a.SP;
    if (a.banded) {
        const size_t tables = sizeof(int32_t) * 2 * a.SP + sizeof(float) * (1 + kMaxExtras + a.W) * a.SP;
        if (lds + tables + 1024 > kLdsBytes) {   // tables do not fit: evaluate full matrix rows instead (exact, slower)
            a.banded = 0;
            a.have_fmax = 0;
        } else {
            lds += tables;
        }
    }
    const long long waves0 = (long long)a.B * a.chunks;
    hipLaunchKernelGGL((lazy_backtrace_kernel<NWT, 0>), dim3((int)((waves0 + kBtWaves - 1) / kBtWaves)), dim3(kBtWaves * 64),
                       lds, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || a.chunks <= 1) return e;
    hipLaunchKernelGGL((lazy_backtrace_kernel<NWT, 1>), dim3((int)((a.B + kBtWaves - 1) / kBtWaves)), dim3(kBtWaves * 64),
                       lds, st, a);
    return hipGetLastError();
}

int backtrace_chunks(int64_t B, int T) {
    // enough (song, chunk) waves to cover the chip twice over, chunks no shorter than ~8 warm-ups
    long long c = (2 * 1024 + B - 1) / (B > 0 ? B : 1);
    const long long cmax = T / (8 * kBtWarm) > 1 ? T / (8 * kBtWarm) : 1;
    c = c > cmax ? cmax : c;
    c = c > kBtMaxChunks ? kBtMaxChunks : c;
    return c < 1 ? 1 : (int)c;
}

hipError_t launch_backtrace(BtArgs a, hipStream_t st) {
    if (a.bt_form == 0 && sparse_backtrace_applies(a)) return launch_backtrace_sparse(a, st);
    a.K = backtrace_tile_rows(a.SD);
    const int nwt = (a.S + 63) / 64;
    if (nwt <= 2) return launch_bt_t<2>(a, st);
    if (nwt <= 4) return launch_bt_t<4>(a, st);
    if (nwt <= 6) return launch_bt_t<6>(a, st);
    if (nwt <= 8) return launch_bt_t<8>(a, st);
    if (nwt <= 12) return launch_bt_t<12>(a, st);
    return launch_bt_t<16>(a, st);
}

hipError_t launch_voicing_map(const int32_t* states, int64_t n, int32_t n_bins, uint8_t* voiced, int32_t* bins,
                              hipStream_t st) {
    if (n == 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(voicing_map_kernel, dim3((int)blocks), dim3(256), 0, st, states, n, n_bins, voiced, bins);
    return hipGetLastError();
}

hipError_t launch_scan_selftest(const float* vals, int n_waves, int mode, float* out_v, int32_t* out_i,
                                hipStream_t st) {
    hipLaunchKernelGGL(scan_selftest_kernel, dim3(n_waves), dim3(64), 0, st, vals, mode, out_v, out_i);
    return hipGetLastError();
}

}  // namespace vit
