// emission.hip -- GPU emission builders: pitch logits -> log observation probabilities, the step right
// upstream of the Viterbi decoder.  In the reference these are Python loops over
// the frames of a song on the host:
//   Viterbi.observation_probs_fn          tonet/for_paper.py:1733-1778  (+ find_peaks :1714-1731, expit :1703-1712)
//   SoftMaxViterbi.observation_probs_fn   tonet/for_paper.py:1911-1944  (+ find_peaks :1890-1909)
//   SoftMaxViterbi.observation_probs_fn   dcnet/softmax_viterbi.py:2530-2579  ("scaled likelihood": p / prior, unvoiced
//                                         logit = the voicing-threshold logit; values may exceed 1)
// followed by log(p + tiny) in viterbi_librosa_fn (:1846-1847).  Here one wave builds one frame and
// writes log(p + tiny) directly in the [frames, n_bins+1] layout vit_decode() reads.
//
// Parity: peak picking and the voicing decision are exact (compares; the voicing logit in float64 like
// the reference); exp/log/sum are the GPU's, so probabilities agree to a few ulp, not bit for bit
// (tests compare with a tolerance and require the structural zeros -> log(tiny) to be exact).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace vit {

constexpr int kObsWaves = 4;
constexpr float kTiny = 1.1754944e-38f;        // np.finfo(np.float32).tiny
constexpr float kLogTiny = -87.33654475f;      // float32 log(tiny) = -87.33655 (bits 0xC2AEAC50)

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}
__device__ __forceinline__ float wave_max_f(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}

// MODE 0: "shaun" (soft voicing on the strongest peak); MODE 1: softmax over the peak set; MODE 2: softmax over the
// peak set with a constant unvoiced logit (`threshold`), every probability divided by its state prior.
// logits: MODE 0, 2 [n_frames, U]; MODE 1 [n_frames, U+1] with column 0 = unvoiced.  out: [n_frames, U+1].
// prior (MODE 2): [U+1] in state order (unvoiced last), or null for no scaling.
//
// One wave per frame, lane l owns bins l, l+64, ... (coalesced loads and stores, conflict-free LDS).  A bin is a peak when
// it is the FIRST maximum of its window: c > max(row[b-spw .. b-1]) and c >= max(row[b+1 .. b+spw]).  The two window maxima
// come from a doubling table in LDS, R_p[i] = max(row[i .. i+p-1]) with p the largest power of two <= spw (two overlapping
// p-windows cover a window of spw): 3 LDS operations per element and level + 5 reads per bin instead of 2*spw + 1 reads
// per bin (spw = 15: ~90 instead of ~190 per lane and frame, and no 2-way bank conflicts).
template <int EPL, int MODE>
__global__ void __launch_bounds__(kObsWaves * 64) observation_kernel(const float* __restrict__ logits, int64_t n_frames,
                                                                     int U, int spw, double threshold, double offset,
                                                                     double scale, const float* __restrict__ prior,
                                                                     float* __restrict__ out) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int PW = U + 2 * spw;                       // reflect-padded row
    const int PWS = PW + 1;
    float* row = smem + (size_t)wv * 4 * PWS;          // [4][PWS]: the row, two doubling buffers, per-bin results
    float* bufA = row + PWS;
    float* bufB = bufA + PWS;
    float* aux = bufB + PWS;
    const int in_stride = MODE == 1 ? U + 1 : U;
    const int in_off = MODE == 1 ? 1 : 0;
    const int S = U + 1;
    int p = 1, levels = 0;
    while (2 * p <= spw) { p *= 2; ++levels; }

    const int64_t fstep = (int64_t)gridDim.x * kObsWaves;
    int64_t f = (int64_t)blockIdx.x * kObsWaves + wv;
    // the next frame's row is in flight while this one is worked on (a wave handles its frames one after another: without
    // the prefetch every frame starts with a full HBM round trip)
    float xn[EPL];
    float x0n = 0.f;
    auto fetch = [&](const int64_t fr) {
        const float* __restrict__ x = logits + fr * in_stride + in_off;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int b = lane + 64 * e;
            xn[e] = b < U ? x[b] : -INFINITY;
        }
        if (MODE == 1) x0n = logits[fr * in_stride];
    };
    if (f < n_frames) fetch(f);
    for (; f < n_frames; f += fstep) {
        float* __restrict__ o = out + f * S;
        // stage the row, then the reflect padding: row[spw + i] = x[i]
        float xv[EPL];
        const float x0f = x0n;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int b = lane + 64 * e;
            xv[e] = xn[e];
            if (b < U) row[spw + b] = xv[e];
        }
        if (f + fstep < n_frames) fetch(f + fstep);
        asm volatile("" ::: "memory");
        if (lane < spw) {
            row[spw - 1 - lane] = row[spw + lane + 1];         // x[-k] = x[k]
            row[spw + U + lane] = row[spw + U - 2 - lane];     // x[U-1+k] = x[U-1-k]
        }
        // doubling: after the loop `src` holds R_p (valid for i <= PW - p).  One wave: its LDS operations execute in program
        // order, so a level reads what the level before wrote without a barrier (the fences only pin the compiler's order).
        asm volatile("" ::: "memory");
        const float* src = row;
        int w = 1;
        for (int lv = 0; lv < levels; ++lv) {
            float* dst = (lv & 1) ? bufB : bufA;
#pragma unroll
            for (int e = 0; e < EPL + 2; ++e) {                   // PW = U + 2*spw <= 64 * (EPL + 2): no loop-carried index arithmetic
                const int i = lane + 64 * e;
                if (i + 2 * w <= PW) dst[i] = fmaxf(src[i], src[i + w]);
            }
            asm volatile("" ::: "memory");
            src = dst;
            w *= 2;
        }
        bool pk[EPL];
        float lmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int b = lane + 64 * e;
            pk[e] = false;
            if (b < U) {
                const int c0 = spw + b;                                       // position of the bin in the padded row
                const float mr = fmaxf(src[c0 + 1], src[c0 + spw - p + 1]);     // max(row[c0+1 .. c0+spw])
                const float ml = fmaxf(src[c0 - spw], src[c0 - p]);             // max(row[c0-spw .. c0-1])
                pk[e] = xv[e] > ml && xv[e] >= mr;
                if (pk[e]) lmax = fmaxf(lmax, xv[e]);
            }
        }
        // ---- the peaks, compacted: only they need exp / divide / log (a dozen or two per frame), so peak k moves to lane k
        //      (rank by ballot + popcount, bin index through LDS) and every lane does that arithmetic once per 64 peaks
        //      instead of once per owned bin
        asm volatile("" ::: "memory");
        float* list = bufA;                           // bin index of peak k (as float bits); R_p is no longer needed
        float* exb = bufB;                            // exp(x - g) of peak k
        float* res = aux;                             // log-probability by bin (peaks only)
        int npk = 0;
        const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const unsigned long long m = __ballot(pk[e]);
            if (pk[e]) list[npk + __popcll(m & lt)] = __int_as_float(lane + 64 * e);
            npk += __popcll(m);
        }
        asm volatile("" ::: "memory");
        // unvoiced logit: always in the peak set (MODE 2: the padded voicing-threshold logit, rounded to float32 like np.pad)
        const float x0 = MODE == 1 ? x0f : (MODE == 2 ? (float)threshold : -INFINITY);
        float g = wave_max_f(lmax);
        const bool any_peak = g > -INFINITY;
        if (MODE >= 1) g = fmaxf(g, x0);
        float lsum = 0.f;
        for (int k0 = 0; k0 < npk; k0 += 64) {
            const int k = k0 + lane;
            if (k < npk) {
                const int b = __float_as_int(list[k]);
                const float exv = expf(row[spw + b] - g);
                exb[k] = exv;
                lsum += exv;
            }
        }
        float tot = wave_sum(lsum);
        float last;                                                       // probability of the unvoiced state
        double t;                                                         // scale applied to exp(x - g)
        if (MODE == 0) {
            double pv = 0.0;
            if (any_peak) {
                const double gd = (double)g;
                const double s_ = gd >= threshold ? scale * (gd - threshold) + offset : scale * (gd - threshold) - offset;
                if (s_ > 0) pv = 1.0 / (1.0 + exp(-s_));
                else { const double q = exp(s_); pv = q / (1.0 + q); }
            }
            t = any_peak ? pv / (double)tot : 0.0;
            last = any_peak ? (float)(1.0 - pv) : 1.f;
        } else {
            const float e0 = expf(x0 - g);
            tot += e0;
            t = 1.0;
            last = any_peak ? e0 / tot : 1.f;
        }
        asm volatile("" ::: "memory");
        for (int k0 = 0; k0 < npk; k0 += 64) {
            const int k = k0 + lane;
            if (k < npk) {
                const int b = __float_as_int(list[k]);
                float pr;
                if (MODE == 0) pr = (float)((double)exb[k] * t);
                else pr = exb[k] / tot;
                if (MODE == 2 && prior) pr = pr / prior[b];      // two float32 divisions, like the reference
                res[b] = logf(pr + kTiny);
            }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int b = lane + 64 * e;
            if (b < U) o[b] = pk[e] ? res[b] : kLogTiny;
        }
        if (MODE == 2 && prior) last = last / prior[U];          // a peak-less frame: 1 / prior
        if (lane == 0) o[U] = logf(last + kTiny);
        asm volatile("" ::: "memory");                        // (the next frame's staging must not overtake these reads)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// observation_reg_kernel: the same builders with the frame in REGISTERS (round 4).  The LDS form above spends 393 vector +
// 213 scalar + 100 LDS instructions per frame (profiles/r02_pmc_extras.txt) -- more vector work than the 233-instruction
// forward recursion it feeds -- on staging the row, a doubling table and ballot / popcount ranks for the peak compaction.  Here
// lane l owns NPL CONTIGUOUS bins (one coalesced vector load and store per lane), the SPW neighbours on either side arrive by
// wave-wide DPP shifts of whole registers (wave_shr:1 / wave_shl:1, chained for SPW > NPL; -inf beyond the row's ends), every
// window maximum is a chain of v_max3_f32 over registers with compile-time indices, and the reflect padding is never built: what
// it adds to the windows reduces to two rules at the left end (at the kernel).  exp / log run densely on the owner lanes with the
// hardware's v_exp_f32 / v_log_f32 and a compensated range scaling (a few ulp) instead of through a compacted peak list, the
// wave reductions are DPP; the soft-voicing sigmoid stays in float64 like the reference's.  No LDS, no barrier.
// Instantiated for the reference's half-widths 5 ("shaun", dcnet's scaled likelihood) and 15 (softmax) and 5 / 6 / 8 / 12 bins per
// lane (U <= 64 NPL: the 320-, 360- and 721-bin grids and what lies between); other geometries keep the LDS form.
namespace {

typedef float ob_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float ob_f2u __attribute__((ext_vector_type(2), aligned(4)));

// a lane's NPL consecutive floats as 16- / 8- / 4-byte pieces at 4-byte alignment (coalesced across the wave)
template <int NPL>
__device__ __forceinline__ void ob_load(const float* __restrict__ p, float (&v)[NPL]) {
    int k = 0;
#pragma unroll
    for (; k + 3 < NPL; k += 4) { const ob_f4u t = *reinterpret_cast<const ob_f4u*>(p + k); v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w; }
#pragma unroll
    for (; k + 1 < NPL; k += 2) { const ob_f2u t = *reinterpret_cast<const ob_f2u*>(p + k); v[k] = t.x; v[k + 1] = t.y; }
    if (k < NPL) v[k] = p[k];
}
template <int NPL>
__device__ __forceinline__ void ob_store(float* __restrict__ p, const float (&v)[NPL]) {
    int k = 0;
#pragma unroll
    for (; k + 3 < NPL; k += 4) { ob_f4u t; t.x = v[k]; t.y = v[k + 1]; t.z = v[k + 2]; t.w = v[k + 3]; *reinterpret_cast<ob_f4u*>(p + k) = t; }
#pragma unroll
    for (; k + 1 < NPL; k += 2) { ob_f2u t; t.x = v[k]; t.y = v[k + 1]; *reinterpret_cast<ob_f2u*>(p + k) = t; }
    if (k < NPL) p[k] = v[k];
}

__device__ __forceinline__ float ob_shr1(float x) {   // lane l <- x[l-1]; lane 0 (no source) gets -inf: the bins in front of the row
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(x), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float ob_shl1(float x) {   // lane l <- x[l+1]; lane 63 gets -inf
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float ob_wave_max(float x) {
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
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float ob_wave_sum(float x) {   // inclusive scan by rows, lane 63 holds the total
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(x));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
// e^x through v_exp_f32 (2^y, 1 ulp): log2(e) = hi + lo, the rounding error of x * hi is recovered with an fma and applied as
// the factor 2^lo ~ 1 + lo ln 2 -- ~2 ulp where a plain x * log2(e) would lose |x| * 1e-7
__device__ __forceinline__ float ob_exp(float x) {
    const float hi = x * 1.44269502f;
    const float lo = __builtin_fmaf(x, 1.44269502f, -hi);              // the product's rounding error, exactly
    const float e2 = __builtin_amdgcn_exp2f(hi);
    return __builtin_fmaf(e2, lo * 0.693147182f, e2);                   // (what float(log2 e) itself is off by adds |x| * 1.3e-8: < 1 ulp up to |x| = 8)
}
// ln y through v_log_f32 (log2, 1 ulp), ln 2 = hi + lo
__device__ __forceinline__ float ob_log(float y) {
    const float l2 = __builtin_amdgcn_logf(y);
    return __builtin_fmaf(l2, 0.693147182f, l2 * -1.90465421e-9f);
}

}  // namespace

template <int NPL, int SPW, int MODE>
__global__ void __launch_bounds__(256) observation_reg_kernel(const float* __restrict__ logits, int64_t n_frames, int U, double threshold,
                                                             double offset, double scale, const float* __restrict__ prior,
                                                             float* __restrict__ out) {
    constexpr int H = (SPW + NPL - 1) / NPL;          // lanes a lane looks at on either side
    constexpr int NA = NPL + 2 * SPW;                 // local neighbourhood: bins NPL*lane - SPW .. NPL*lane + NPL + SPW - 1
    const int lane = threadIdx.x & 63;
    const int in_stride = MODE == 1 ? U + 1 : U;
    const int in_off = MODE == 1 ? 1 : 0;
    const int S = U + 1;
    // ---- per-lane geometry: slot k of lane l is bin NPL*l + k.  The reflect padding of the reference (np.pad 'reflect': x[-k] = x[k],
    //      x[U-1+k] = x[U-1-k]) never has to exist: beyond the row's ends the neighbourhood holds -inf, and what the mirrored values
    //      add to a window is already in it -- except at the left end, where the LEFT window is compared strictly:
    //        bin 0:              its left window is the mirror of its right one  ->  peak iff c > max(right window)
    //        bins 1 .. SPW / 2:  their left window contains their own mirror image ->  never a peak (c > c)
    //        every other bin:    the mirrored values are bins the window holds anyway (left end), or add c >= x for bins that the
    //                            strict left comparison covers (right end; the last bin's right window is all mirror: c >= -inf)
    bool real[NPL], never[NPL], first[NPL];
    float rprior[NPL];                                // MODE 2: 1 / prior of the bin
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
        const int b = NPL * lane + k;
        real[k] = b < U;
        never[k] = b >= 1 && 2 * b <= SPW;
        first[k] = b == 0;
        rprior[k] = (MODE == 2 && prior && real[k]) ? 1.f / prior[b] : 1.f;
    }
    const float rprior_u = (MODE == 2 && prior) ? 1.f / prior[U] : 1.f;
    // a full lane's NPL columns are one contiguous, coalesced vector access; the one partial lane of a row whose length is not a
    // multiple of NPL goes column by column; lanes beyond the row hold -inf and store nothing
    const int col0 = NPL * lane;
    const bool full = col0 + NPL <= U, partial = !full && col0 < U;
    const int64_t fstep = (int64_t)gridDim.x * (blockDim.x >> 6);
    int64_t f = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // wave-uniform: row bases in scalar registers
    // two frames in flight per wave (two register sets, the loop unrolled by two): with eight waves per SIMD one row ahead left the
    // kernel waiting on HBM latency at 4.2 TB/s
    float xq[2][NPL];
    float x0q[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int k = 0; k < NPL; ++k) xq[q][k] = -INFINITY;
    auto fetch = [&](const int64_t fr, float (&xn)[NPL], float& x0n) {
        const float* __restrict__ x = logits + fr * in_stride + in_off;
        if (full) ob_load<NPL>(x + col0, xn);
        if (partial) {
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                if (real[k]) xn[k] = x[col0 + k];
        }
        if (MODE == 1) x0n = logits[fr * in_stride];
    };
    if (f < n_frames) fetch(f, xq[0], x0q[0]);
    if (f + fstep < n_frames) fetch(f + fstep, xq[1], x0q[1]);
    auto process = [&](const int64_t f, float (&xn)[NPL], float& x0n) {
        float* __restrict__ o = out + f * S;
        float a[NA];
        const float x0f = x0n;
#pragma unroll
        for (int k = 0; k < NPL; ++k) a[SPW + k] = xn[k];
        if (f + 2 * fstep < n_frames) fetch(f + 2 * fstep, xn, x0n);
        // ---- neighbours: a[SPW - d] = bin NPL*lane - d lives in lane - ceil(d / NPL); shifted copies chained; a lane without a source
        //      keeps -inf (the DPP `old` operand)
        {
            float sl[NPL], sr[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) { sl[k] = a[SPW + k]; sr[k] = a[SPW + k]; }
#pragma unroll
            for (int h = 1; h <= H; ++h) {
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const int dl = h * NPL - k;               // sl[k] after h shifts = bin NPL*(lane-h) + k = own start - dl
                    if (dl <= SPW) { sl[k] = ob_shr1(sl[k]); a[SPW - dl] = sl[k]; }      // (a slot that is out of reach at h stays out of reach)
                    const int dr = (h - 1) * NPL + k;         // sr[k] after h shifts = bin NPL*(lane+h) + k = own end + 1 + dr
                    if (dr < SPW) { sr[k] = ob_shl1(sr[k]); a[SPW + NPL + dr] = sr[k]; }
                }
            }
        }
        // ---- peaks: the FIRST maximum of its window: c > max(left SPW) and c >= max(right SPW)
        bool pk[NPL];
        float lmax = -INFINITY;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            float ml = a[k], mr = a[SPW + k + 1];
#pragma unroll
            for (int j = 1; j + 1 < SPW; j += 2) { ml = fmaxf(fmaxf(ml, a[k + j]), a[k + j + 1]); mr = fmaxf(fmaxf(mr, a[SPW + k + 1 + j]), a[SPW + k + 2 + j]); }
            if (SPW % 2 == 0) { ml = fmaxf(ml, a[k + SPW - 1]); mr = fmaxf(mr, a[2 * SPW + k]); }
            const float c = a[SPW + k];
            const bool std_pk = c > ml && c >= mr, first_pk = c > mr;
            pk[k] = real[k] && !never[k] && (first[k] ? first_pk : std_pk);
            lmax = pk[k] ? fmaxf(lmax, c) : lmax;
        }
        const float x0 = MODE == 1 ? x0f : (MODE == 2 ? (float)threshold : -INFINITY);     // the unvoiced logit (always in the peak set)
        float g = ob_wave_max(lmax);
        const bool any_peak = g > -INFINITY;
        if (MODE >= 1) g = fmaxf(g, x0);
        float ex[NPL];
        float lsum = 0.f;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            ex[k] = ob_exp(pk[k] ? a[SPW + k] - g : -1000.f);          // (e^-1000 = 0: a select on the argument, no branch around the exp)
            lsum += ex[k];
        }
        float tot = ob_wave_sum(lsum);
        float last;                                      // probability of the unvoiced state
        float v[NPL];
        if (MODE == 0) {
            // soft voicing on the strongest peak (tonet/for_paper.py:1703-1712, :1757-1764) in float64 exactly like the reference:
            // 1 - expit(s) is formed by subtraction there, so for s > ~37 the unvoiced probability is EXACTLY 0 (-> log tiny) and below
            // that it carries the float64 cancellation noise of the reference (2e-5 relative at 1 - pv = 6e-12); a float32 expit(-s)
            // would be more accurate and would not be the reference's number
            double pv = 0.0;
            if (any_peak) {
                const double gd = (double)g;
                const double s_ = gd >= threshold ? scale * (gd - threshold) + offset : scale * (gd - threshold) - offset;
                if (s_ > 0) pv = 1.0 / (1.0 + exp(-s_));
                else { const double q = exp(s_); pv = q / (1.0 + q); }
            }
            const double t = any_peak ? pv / (double)tot : 0.0;
            last = any_peak ? (float)(1.0 - pv) : 1.f;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                float lg = ob_log((float)((double)ex[k] * t) + kTiny);
                asm volatile("" : "+v"(lg));                          // (evaluate, then select: no branch per slot)
                v[k] = pk[k] ? lg : kLogTiny;
            }
        } else {
            const float e0 = ob_exp(x0 - g);
            tot += e0;
            const float tf = 1.f / tot;
            last = any_peak ? e0 * tf : 1.f;
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                float pr = ex[k] * tf;
                if (MODE == 2) pr *= rprior[k];
                float lg = ob_log(pr + kTiny);
                asm volatile("" : "+v"(lg));                          // (evaluate, then select: no branch per slot)
                v[k] = pk[k] ? lg : kLogTiny;
            }
        }
        if (full) ob_store<NPL>(o + col0, v);
        if (partial) {
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                if (real[k]) o[col0 + k] = v[k];
        }
        if (MODE == 2) last *= rprior_u;                 // (a peak-less frame: 1 / prior)
        if (lane == 0) o[U] = ob_log(last + kTiny);
    };
    for (; f < n_frames; f += 2 * fstep) {
        process(f, xq[0], x0q[0]);
        if (f + fstep < n_frames) process(f + fstep, xq[1], x0q[1]);
    }
}

template <int NPL, int SPW, int MODE>
static hipError_t launch_obs_reg(const float* logits, int64_t n_frames, int U, double thr, double off, double sc, const float* prior,
                                 float* out, int n_cus, hipStream_t st) {
    int64_t blocks = (n_frames + 3) / 4;
    if (blocks > (int64_t)n_cus * 8) blocks = (int64_t)n_cus * 8;      // eight waves per SIMD: the resident capacity, every wave walks its frames
    hipLaunchKernelGGL((observation_reg_kernel<NPL, SPW, MODE>), dim3((int)blocks), dim3(256), 0, st, logits, n_frames, U, thr, off, sc, prior, out);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_obs(const float* logits, int64_t n_frames, int U, int spw, double thr, double off, double sc,
                             const float* prior, float* out, hipStream_t st) {
    if (n_frames <= 0) return hipSuccess;
    if (spw < 1 || spw >= U || spw > 64 || U > 768) return hipErrorInvalidValue;
    int64_t blocks = (n_frames + kObsWaves - 1) / kObsWaves;
    static int n_cus = 0;                         // (no plan at this entry point: ask the device once)
    if (n_cus == 0) {
        int dev = 0, cus = 0;
        n_cus = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) ? cus : 256;
    }
    // the register form for the reference's geometries (spw 5 / 15; 5, 6, 8 or 12 bins per lane)
    if ((spw == 5 || spw == 15) && U > 64 && U > 2 * spw) {
#define VIT_OBS_REG(N) return spw == 5 ? launch_obs_reg<N, 5, MODE>(logits, n_frames, U, thr, off, sc, prior, out, n_cus, st) \
                                      : launch_obs_reg<N, 15, MODE>(logits, n_frames, U, thr, off, sc, prior, out, n_cus, st)
        if (U <= 64 * 5) VIT_OBS_REG(5);
        if (U <= 64 * 6) VIT_OBS_REG(6);
        if (U <= 64 * 8) VIT_OBS_REG(8);
        if (U <= 64 * 12) VIT_OBS_REG(12);
#undef VIT_OBS_REG
    }
    if (blocks > (int64_t)n_cus * 6) blocks = (int64_t)n_cus * 6;   // six workgroups of four frames' LDS fit a CU: one resident wave per frame slot, no second round
    const size_t lds = sizeof(float) * kObsWaves * 4 * (U + 2 * spw + 1);
    if (U <= 384)
        hipLaunchKernelGGL((observation_kernel<6, MODE>), dim3((int)blocks), dim3(kObsWaves * 64), lds, st, logits, n_frames, U,
                           spw, thr, off, sc, prior, out);
    else
        hipLaunchKernelGGL((observation_kernel<12, MODE>), dim3((int)blocks), dim3(kObsWaves * 64), lds, st, logits, n_frames, U,
                           spw, thr, off, sc, prior, out);
    return hipGetLastError();
}

hipError_t launch_obs_shaun(const float* logits, int64_t n_frames, int U, int spw, double thr, double off, double sc,
                            float* out, hipStream_t st) {
    return launch_obs<0>(logits, n_frames, U, spw, thr, off, sc, nullptr, out, st);
}
hipError_t launch_obs_softmax(const float* logits, int64_t n_frames, int U, int spw, float* out, hipStream_t st) {
    return launch_obs<1>(logits, n_frames, U, spw, 0.0, 0.0, 0.0, nullptr, out, st);
}
hipError_t launch_obs_softmax_scaled(const float* logits, int64_t n_frames, int U, int spw, double unvoiced_logit,
                                     const float* prior, float* out, hipStream_t st) {
    return launch_obs<2>(logits, n_frames, U, spw, unvoiced_logit, 0.0, 0.0, prior, out, st);
}

// ---------------------------------------------------------------------------------------
// Device-resident hand-off from the acoustic model (tonet/for_paper.py:2282-2302 does this on the
// host after a .cpu().numpy() copy): a batch of snippets [n, C, F] (C = n_bins + 1 channel rows, channel 0 = unvoiced,
// F frames) becomes n*F time-major logit rows appended to a recording's buffer on the GPU --
//   mode 0 ("shaun"):   row[b] = snip[b+1][f] - snip[0][f]   (n_bins columns: logits relative to the unvoiced channel)
//   mode 1 ("softmax"): row[c] = snip[c][f]                  (n_bins + 1 columns, unvoiced first)
// Only the first `rows` (= n*F - padded_frames) rows are written.  A 64 x 64 tile goes through LDS so that both the
// frame-contiguous reads and the channel-contiguous writes are coalesced.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) snippets_append_kernel(const float* __restrict__ snips, int n, int C, int F, int mode,
                                                              float* __restrict__ out, int64_t rows) {
    __shared__ float tile[64][65];
    __shared__ float ch0[64];
    const int sn = blockIdx.z, c0 = blockIdx.y * 64, f0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;      // 64 x 4
    const float* __restrict__ src = snips + (size_t)sn * C * F;
    const int cols = mode == 0 ? C - 1 : C;
    const int coff = mode == 0 ? 1 : 0;                         // first channel that becomes a column
    for (int r = ty; r < 64; r += 4) {
        const int c = c0 + r + coff, f = f0 + tx;
        tile[r][tx] = (c < C && f < F) ? src[(size_t)c * F + f] : 0.f;
    }
    if (mode == 0 && threadIdx.x < 64) ch0[tx] = f0 + tx < F ? src[f0 + tx] : 0.f;
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int f = f0 + r, col = c0 + tx;
        const int64_t row = (int64_t)sn * F + f;
        if (f < F && col < cols && row < rows) out[row * cols + col] = mode == 0 ? tile[tx][r] - ch0[r] : tile[tx][r];
    }
}

hipError_t launch_snippets_append(const float* snips, int n, int C, int F, int mode, float* out, int64_t rows, hipStream_t st) {
    if (n <= 0 || rows <= 0) return hipSuccess;
    const int cols = mode == 0 ? C - 1 : C;
    hipLaunchKernelGGL(snippets_append_kernel, dim3((F + 63) / 64, (cols + 63) / 64, n), dim3(256), 0, st, snips, n, C, F, mode, out, rows);
    return hipGetLastError();
}

// voiced = state < n_bins; bins = min(state, n_bins-1); notes = note_range[bins]; notes_v = voiced ? notes : 0
// (tonet/for_paper.py:1828-1829, :2106-2115 est_notes_360_fn, :2207)
__global__ void voicing_notes_kernel(const int32_t* __restrict__ states, int64_t n, int32_t n_bins,
                                     const float* __restrict__ note_range, uint8_t* __restrict__ voiced,
                                     int32_t* __restrict__ bins, float* __restrict__ notes, float* __restrict__ notes_v) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = states[i];
        const bool v = s >= 0 && s < n_bins;
        const int32_t b = s < 0 ? -1 : (s < n_bins - 1 ? s : n_bins - 1);
        const float nt = b >= 0 ? note_range[b] : 0.f;
        if (voiced) voiced[i] = v ? 1 : 0;
        if (bins) bins[i] = b;
        if (notes) notes[i] = nt;
        if (notes_v) notes_v[i] = v ? nt : 0.f;
    }
}

hipError_t launch_voicing_notes(const int32_t* states, int64_t n, int32_t n_bins, const float* note_range, uint8_t* voiced,
                                int32_t* bins, float* notes, float* notes_v, hipStream_t st) {
    if (n == 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(voicing_notes_kernel, dim3((int)blocks), dim3(256), 0, st, states, n, n_bins, note_range, voiced, bins, notes, notes_v);
    return hipGetLastError();
}

}  // namespace vit
