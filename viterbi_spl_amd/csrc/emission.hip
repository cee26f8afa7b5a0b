// emission.hip -- GPU emission builders: pitch logits -> log observation probabilities, the step right
// upstream of the Viterbi decoder.  In the reference these are Python loops over
// the frames of a song on the host:
//   Viterbi.observation_probs_fn          tonet/for_paper.py:1733-1778  (+ find_peaks :1714-1731, expit :1703-1712)
//   SoftMaxViterbi.observation_probs_fn   tonet/for_paper.py:1911-1944  (+ find_peaks :1890-1909)
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

// MODE 0: "shaun" (soft voicing on the strongest peak); MODE 1: softmax over the peak set.
// logits: MODE 0 [n_frames, U]; MODE 1 [n_frames, U+1] with column 0 = unvoiced.  out: [n_frames, U+1].
template <int EPL, int MODE>
__global__ void __launch_bounds__(kObsWaves * 64) observation_kernel(const float* __restrict__ logits, int64_t n_frames,
                                                                     int U, int spw, double threshold, double offset,
                                                                     double scale, float* __restrict__ out) {
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int PW = U + 2 * spw;                       // reflect-padded row
    float* row = smem + (size_t)wv * (PW + 1);
    const int in_stride = MODE == 0 ? U : U + 1;
    const int in_off = MODE == 0 ? 0 : 1;
    const int S = U + 1;

    for (int64_t f = (int64_t)blockIdx.x * kObsWaves + wv; f < n_frames; f += (int64_t)gridDim.x * kObsWaves) {
        const float* __restrict__ x = logits + f * in_stride + in_off;
        float* __restrict__ o = out + f * S;
        // stage the row (coalesced), then the reflect padding: row[spw + i] = x[i]
        for (int i = lane; i < U; i += 64) row[spw + i] = x[i];
        if (lane < spw) {
            row[spw - 1 - lane] = x[lane + 1];                 // x[-k] = x[k]
            row[spw + U + lane] = x[U - 2 - lane];             // x[U-1+k] = x[U-1-k]
        }
        // each lane owns bins [lane*EPL, lane*EPL+EPL)
        float xv[EPL];
        bool pk[EPL];
        float lmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int b = lane * EPL + e;
            pk[e] = false;
            xv[e] = -INFINITY;
            if (b < U) {
                const float c = row[spw + b];
                bool is = true;                               // FIRST maximum of its window (np.argmax == centre)
                for (int k = 1; k <= spw; ++k) {
                    is = is && (c > row[spw + b - k]) && (c >= row[spw + b + k]);
                }
                pk[e] = is;
                xv[e] = c;
                if (is) lmax = fmaxf(lmax, c);
            }
        }
        const float x0 = MODE == 1 ? logits[f * in_stride] : -INFINITY;   // unvoiced logit: always in the peak set
        float g = wave_max_f(lmax);
        const bool any_peak = g > -INFINITY;
        if (MODE == 1) g = fmaxf(g, x0);
        float ex[EPL];
        float lsum = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            ex[e] = pk[e] ? expf(xv[e] - g) : 0.f;
            lsum += ex[e];
        }
        float tot = wave_sum(lsum);
        float last;                                                       // probability of the unvoiced state
        double t;                                                         // scale applied to exp(x - g)
        if (MODE == 0) {
            double pv = 0.0;
            if (any_peak) {
                const double gd = (double)g;
                const double s = gd >= threshold ? scale * (gd - threshold) + offset : scale * (gd - threshold) - offset;
                if (s > 0) pv = 1.0 / (1.0 + exp(-s));
                else { const double q = exp(s); pv = q / (1.0 + q); }
            }
            t = any_peak ? pv / (double)tot : 0.0;
            last = any_peak ? (float)(1.0 - pv) : 1.f;
        } else {
            const float e0 = expf(x0 - g);
            tot += e0;
            t = 1.0;
            last = any_peak ? e0 / tot : 1.f;
        }
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            const int b = lane * EPL + e;
            if (b < U) {
                float p;
                if (MODE == 0) p = (float)((double)ex[e] * t);
                else p = any_peak ? ex[e] / tot : 0.f;
                o[b] = pk[e] ? logf(p + kTiny) : kLogTiny;
            }
        }
        if (lane == 0) o[U] = logf(last + kTiny);
    }
}

template <int MODE>
static hipError_t launch_obs(const float* logits, int64_t n_frames, int U, int spw, double thr, double off, double sc,
                             float* out, hipStream_t st) {
    if (n_frames <= 0) return hipSuccess;
    if (spw < 1 || spw >= U || spw > 64 || U > 768) return hipErrorInvalidValue;
    int64_t blocks = (n_frames + kObsWaves - 1) / kObsWaves;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const size_t lds = sizeof(float) * kObsWaves * (U + 2 * spw + 1);
    if (U <= 384)
        hipLaunchKernelGGL((observation_kernel<6, MODE>), dim3((int)blocks), dim3(kObsWaves * 64), lds, st, logits, n_frames, U,
                           spw, thr, off, sc, out);
    else
        hipLaunchKernelGGL((observation_kernel<12, MODE>), dim3((int)blocks), dim3(kObsWaves * 64), lds, st, logits, n_frames, U,
                           spw, thr, off, sc, out);
    return hipGetLastError();
}

hipError_t launch_obs_shaun(const float* logits, int64_t n_frames, int U, int spw, double thr, double off, double sc,
                            float* out, hipStream_t st) {
    return launch_obs<0>(logits, n_frames, U, spw, thr, off, sc, out, st);
}
hipError_t launch_obs_softmax(const float* logits, int64_t n_frames, int U, int spw, float* out, hipStream_t st) {
    return launch_obs<1>(logits, n_frames, U, spw, 0.0, 0.0, 0.0, out, st);
}

}  // namespace vit
