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
