// backtrace_pair.hip -- speculative pass of the exact, time-parallel back-trace with TWO (song, chunk) streams per wavefront.
//
// sparse_backtrace_kernel spends a whole wavefront on one frame decision: ~33 candidates on 64 lanes, and every address, every
// comparison and the six-step maximum are paid per decision -- the kernel is bound by instruction issue (B = 2048: 61 M
// decisions in 6.8 ms = ~250 cycles of a SIMD each), not by the 0.5 KB per frame it fetches.  The production matrices have
// exception spans within 14 sources of the target (plan.wave_d <= kPairD), i.e. a frame decision has 29 band candidates, at
// most two extra columns and the bound: 32 lanes.  Here lanes 0..31 walk one (song, chunk) stream and lanes 32..63 another,
// in lockstep: one LDS read of the delta values, one of the weights (tabH[j][.], plan.hpp), one add, a five-step maximum
// and one ballot decide two frames.
//
// HALF = false: every delta row is in the history (the workgroup kernels, or the wave form with wave_history = 1).
// HALF = true:  only the rows of even frames are (wave.hip HM 1).  The 29 delta values an odd frame t needs are rebuilt first,
//               one per lane: delta_t[i] = fl(max(max_q fl(delta_{t-1}[i-14+q] + logA_T[i][i-14+q]), fl(M_{t-1} + c_i),
//               fl(delta_{t-1}[x] + logA_T[i][x])) + logE[t][i]) -- the floor-max form of the forward recursion
//               (plan.floor_all_ok), every value one the dense recursion forms, so the forward kernel's bits.
//
// A tile holds 8 frames per stream (spans of 64 / 96 history columns around the path, the rows' scalars, and with HALF 64
// emission columns of the odd frames); a window that leaves its span drops the tile of BOTH streams (they stay in
// lockstep) and re-centres it; a bound failure evaluates the whole row with all 64 lanes (an odd one is rebuilt in full
// first).  Exact whatever the guesses: the verify-and-repair pass (sparse / half kernel, MODE 1) follows unchanged.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace vit {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPgF = 8;                       // frames per tile and stream
constexpr int kPgPad = 32;                    // floats in front of and behind a stream's region (addresses of sources that do
                                              // not exist -- weight -inf -- may fall up to 28 floats outside a span)
// full history: [8 rows][64 span + 4 scalars]
constexpr int kPfNS = 64, kPfRS = kPfNS + 4;
constexpr int kPfRegion = kPgPad + kPgF * kPfRS + kPgPad;
// half history: [5 stored rows][96 span], [6 rows][8 scalars], [4 odd frames][64 emission columns]
constexpr int kPhND = 96, kPhRD = 5, kPhRA = 6, kPhRE = 4, kPhNE = 64;
constexpr int kPhRegion = kPgPad + kPhRD * kPhND + kPhRA * 8 + kPhRE * kPhNE + kPgPad;
constexpr int kPW = 2 * kPairD + 1;           // 29 band candidates
constexpr int kPB = 31;                       // candidate lane of the bound

__device__ __forceinline__ int pg_song_length(const int64_t* lengths, int song, int T) {
    if (!lengths) return T;
    long long v = lengths[song];
    v = v < 1 ? 1 : v;
    return v > T ? T : (int)v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float pg_dpp_max(float x) {
    const float y = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, ROW_MASK, 0xf, false));
    return fmaxf(x, y);
}
// max over each half of the wave: lanes 31 and 63 end up holding the maximum of lanes 0..31 / 32..63
__device__ __forceinline__ float pg_half_max(float x) {
    x = pg_dpp_max<0x111, 0xf>(x);   // row_shr:1
    x = pg_dpp_max<0x112, 0xf>(x);   // row_shr:2
    x = pg_dpp_max<0x114, 0xf>(x);   // row_shr:4
    x = pg_dpp_max<0x118, 0xf>(x);   // row_shr:8
    x = pg_dpp_max<0x142, 0xa>(x);   // row_bcast:15 into rows 1 and 3
    return x;
}
__device__ __forceinline__ float pg_wave_max(float x) {
    x = pg_half_max(x);
    x = pg_dpp_max<0x143, 0xc>(x);   // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}
__device__ __forceinline__ float pg_lane(float x, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l)); }
__device__ __forceinline__ int pg_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

template <typename ET>
__device__ __forceinline__ float pg_ld(const ET* p);
template <>
__device__ __forceinline__ float pg_ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float pg_ld<__half>(const __half* p) { return __half2float(*p); }

}  // namespace

// EPL: 64 * EPL >= S (sources per lane in a whole-row evaluation).
template <int EPL, bool HALF, typename ET>
__global__ void __launch_bounds__(1024) pair_backtrace_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int REGION = HALF ? kPhRegion : kPfRegion;
    const int S = a.S, SD = a.SD, T = a.T;
    const int nx = a.n_extras;
    const int nwaves = blockDim.x >> 6;
    float* tabH = reinterpret_cast<float*>(smem);                              // [S][kPairRow]
    float* regions = tabH + ((S * kPairRow + 3) & ~3);                         // [nwaves][2][REGION]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = pg_uni(tid >> 6);
    {
        const float* __restrict__ gtab = reinterpret_cast<const float*>(a.image + a.off_tabH);
        const int nthr = blockDim.x;
        for (int k = tid; k < S * kPairRow; k += nthr) tabH[k] = gtab[k];
        float* reg = regions + (size_t)wv * 2 * REGION;
        for (int k = lane; k < 2 * REGION; k += 64) reg[k] = 0.f;            // pads and unused scalars: any finite value
    }
    __syncthreads();

    const int C = a.chunks;
    const long long units = (long long)a.B * C;
    const long long u0 = 2ll * ((long long)blockIdx.x * nwaves + wv);
    if (u0 >= units) return;
    const int hv = lane >> 5, c = lane & 31;               // stream of this lane, candidate of this lane
    auto sel = [&](const int x0, const int x1) -> int { return hv ? x1 : x0; };
    auto self = [&](const float x0, const float x1) -> float { return hv ? x1 : x0; };

    // ---- per stream (wave-uniform pairs)
    bool valid[2];
    int song[2], chunk[2], Tb[2], lo_c[2], hi_c[2], f[2], cur[2], entry[2];
    const float* hist[2];
    const ET* E[2];
    int32_t* states[2];
    bool inS[EPL], xcol[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = e * 64 + lane;
        inS[e] = i < S;
        bool x = i >= S;
#pragma unroll
        for (int k = 0; k < 2; ++k) x |= (k < nx && i == a.extras[k]);
        xcol[e] = x;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const long long u = u0 + h;
        valid[h] = u < units;
        const long long uu = valid[h] ? u : units - 1;
        song[h] = pg_uni((int)(uu / C));
        chunk[h] = pg_uni((int)(uu % C));
        Tb[h] = pg_uni(pg_song_length(a.lengths, song[h], T));
        hist[h] = a.hist + (size_t)song[h] * a.hist_rows * SD;
        E[h] = reinterpret_cast<const ET*>(a.logE) + (size_t)song[h] * T * S;
        states[h] = a.states + (size_t)song[h] * T;
        const int Lf = Tb[h] - 1;
        lo_c[h] = pg_uni((int)((long long)Lf * chunk[h] / C));
        hi_c[h] = pg_uni((int)((long long)Lf * (chunk[h] + 1) / C));
        if (valid[h] && chunk[h] == C - 1) {
            for (int t = Tb[h] + lane; t < T; t += 64) states[h][t] = -1;
            if (lane == 0) states[h][Tb[h] - 1] = a.last_state[song[h]];
        }
        int top = hi_c[h] - 1 + a.warm;
        if (HALF) top += (top & 1) ? 0 : 1;       // the guess row top + 1 must be a stored (even) frame
        if (chunk[h] == C - 1 || top >= Lf - 1) {
            top = Lf - 1;
            cur[h] = pg_uni(a.last_state[song[h]]);
        } else {
            // guess: lowest-index argmax of delta row top + 1
            const float* g = hist[h] + (size_t)(HALF ? (top + 1) >> 1 : top + 1) * SD + a.col0;
            float d[EPL];
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                d[e] = inS[e] ? g[e * 64 + lane] : -INFINITY;
                m = fmaxf(m, d[e]);
            }
            m = pg_wave_max(m);
            unsigned idx = 0x7fffffffu;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const unsigned long long mk = __ballot(d[e] == m && inS[e]);
                if (mk) { const unsigned ci = e * 64 + __builtin_ctzll(mk); idx = ci < idx ? ci : idx; }
            }
            cur[h] = idx == 0x7fffffffu ? 0 : (int)idx;
        }
        f[h] = pg_uni(top);
        entry[h] = cur[h];                        // stands if there is no warm-up frame (top == hi_c - 1)
        if (hi_c[h] <= lo_c[h]) f[h] = lo_c[h] - 1;   // empty chunk (very short song): nothing to decide
    }

    float* region[2] = {regions + ((size_t)wv * 2 + 0) * REGION + kPgPad, regions + ((size_t)wv * 2 + 1) * REGION + kPgPad};
    const int c0_max = (SD - (HALF ? kPhND : kPfNS)) & ~3;
    const int ce_max = S - kPhNE;
    const int xc[2] = {a.xcol0 >= 0 ? a.xcol0 : a.col0 + a.extras[0], a.xcol0 >= 0 ? a.xcol0 + 1 : a.col0 + a.extras[1]};

    int n_tiles[2] = {0, 0}, n_miss[2] = {0, 0}, n_full[2] = {0, 0}, n_reb[2] = {0, 0};
    // ---- lockstep walk: a tile of up to kPgF frames per stream, then the next
    while ((valid[0] && f[0] >= lo_c[0]) || (valid[1] && f[1] >= lo_c[1])) {
        int ftop[2], c0[2], ce0[2];
        // ---- fetch both tiles, centred on the band of each stream's state
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            ftop[h] = f[h] < 0 ? 0 : f[h];
            n_tiles[h] += (valid[h] && f[h] >= lo_c[h]) ? 1 : 0;
            const int lo = cur[h] - kPairD;
            if (!HALF) {
                int x = ((a.col0 + lo - 17 + 8) & ~15);
                c0[h] = x < 0 ? 0 : (x > c0_max ? c0_max : x);
            } else {
                int x = ((a.col0 + lo - kPairD - 19 + 8) & ~15);
                c0[h] = x < 0 ? 0 : (x > c0_max ? c0_max : x);
                x = lo - 17;
                ce0[h] = x < 0 ? 0 : (x > ce_max ? ce_max : x);
            }
        }
        {
            const int ft = sel(ftop[0], ftop[1]), cc0 = sel(c0[0], c0[1]);
            const float* __restrict__ hb = hv ? hist[1] : hist[0];
            float* rg = hv ? region[1] : region[0];
            if (!HALF) {
                f32x4 sd[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {                     // rows 2v, 2v+1: 16 float4 each
                    const int r = 2 * v + (c >> 4), q = c & 15;
                    int fr = ft - r;
                    fr = fr < 0 ? 0 : fr;
                    sd[v] = *reinterpret_cast<const f32x4*>(hb + (size_t)fr * SD + cc0 + 4 * q);
                }
                float sa = 0.f;
                if (c < 3 * kPgF) {
                    const int r = c / 3, w = c - 3 * r;
                    int fr = ft - r;
                    fr = fr < 0 ? 0 : fr;
                    const int col = w == 2 ? a.mcol : (w < nx ? (w ? xc[1] : xc[0]) : a.mcol);
                    sa = hb[(size_t)fr * SD + col];
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) *reinterpret_cast<f32x4*>(rg + (2 * v + (c >> 4)) * kPfRS + 4 * (c & 15)) = sd[v];
                if (c < 3 * kPgF) rg[(c / 3) * kPfRS + kPfNS + (c % 3)] = sa;
            } else {
                const ET* __restrict__ eb = hv ? E[1] : E[0];
                const int tbh = sel(Tb[0], Tb[1]), ee0 = sel(ce0[0], ce0[1]);
                const int Rd = ft >> 1, Rt = (ft + 1) >> 1, last_row = (tbh - 1) >> 1;
                const int fo = (ft & 1) ? ft : ft - 1;             // first odd frame of the tile
                f32x4 sd[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {                     // 5 rows x 24 float4 = 120
                    int idx = c + 32 * v;
                    idx = idx < kPhRD * 24 ? idx : kPhRD * 24 - 1;
                    const int r = idx / 24, q = idx - 24 * r;
                    int row = Rd - r;
                    row = row < 0 ? 0 : row;
                    sd[v] = *reinterpret_cast<const f32x4*>(hb + (size_t)row * SD + cc0 + 4 * q);
                }
                f32x4 sa;
                {
                    const int cc = c < 2 * kPhRA ? c : 2 * kPhRA - 1;
                    int row = Rt - (cc >> 1);
                    row = row < 0 ? 0 : (row > last_row ? last_row : row);
                    sa = *reinterpret_cast<const f32x4*>(hb + (size_t)row * SD + 4 * (cc & 1));
                }
                float se[2 * kPhRE];
#pragma unroll
                for (int v = 0; v < 2 * kPhRE; ++v) {             // 4 odd frames x 64 columns
                    int fr = fo - 2 * (v >> 1);
                    fr = fr < 1 ? 1 : fr;
                    fr = fr > tbh - 1 ? tbh - 1 : fr;
                    se[v] = pg_ld<ET>(eb + (size_t)fr * S + ee0 + 32 * (v & 1) + c);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int idx = c + 32 * v;
                    if (idx < kPhRD * 24) *reinterpret_cast<f32x4*>(rg + (idx / 24) * kPhND + 4 * (idx % 24)) = sd[v];
                }
                if (c < 2 * kPhRA) *reinterpret_cast<f32x4*>(rg + kPhRD * kPhND + (c >> 1) * 8 + 4 * (c & 1)) = sa;
#pragma unroll
                for (int v = 0; v < 2 * kPhRE; ++v) rg[kPhRD * kPhND + kPhRA * 8 + (v >> 1) * kPhNE + 32 * (v & 1) + c] = se[v];
            }
        }
        int outv = 0;
        int ndone = 0;                  // steps of this tile whose decisions stand
        bool drop = false;              // a stream's whole-row evaluation reused its span area: fetch again
        for (int k = 0; k < kPgF && !drop; ++k) {
            bool act[2];
            int lo[2];
            bool miss = false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                act[h] = valid[h] && f[h] >= lo_c[h];
                lo[h] = cur[h] - kPairD;
                const int wl = lo[h] < 0 ? 0 : lo[h], wh = lo[h] + kPW - 1 > S - 1 ? S - 1 : lo[h] + kPW - 1;
                if (!HALF || !(f[h] & 1)) {
                    miss |= act[h] && (a.col0 + wl - c0[h] < 0 || a.col0 + wh - c0[h] >= (HALF ? kPhND : kPfNS));
                } else {
                    const int sl = wl - kPairD < 0 ? 0 : wl - kPairD, sh = wh + kPairD > S - 1 ? S - 1 : wh + kPairD;
                    miss |= act[h] && (a.col0 + sl - c0[h] < 0 || a.col0 + sh - c0[h] >= kPhND || wl < ce0[h] || wh >= ce0[h] + kPhNE);
                }
            }
            if (miss) { n_miss[0] += 1; break; }
            // ---- the candidates of both streams
            const int curv = sel(cur[0], cur[1]), lov = curv - kPairD, fv = sel(f[0] < 0 ? 0 : f[0], f[1] < 0 ? 0 : f[1]);
            const float* rg = hv ? region[1] : region[0];
            const int cc0 = sel(c0[0], c0[1]), ftv = sel(ftop[0], ftop[1]);
            float dv;
            if (!HALF) {
                const int r = ftv - fv;                                       // 0 .. 7
                const int off = c < kPW ? a.col0 + lov + c - cc0 : kPfNS + (c - kPW);   // scalars: extra 0, extra 1, frame maximum
                dv = rg[r * kPfRS + off];
            } else {
                const int Rd = ftv >> 1, Rt = (ftv + 1) >> 1;
                const float* ax = rg + kPhRD * kPhND;                         // [6][8]: even frame M, x0, x1 at mcol, xcol0, xcol0+1; odd at mcol_odd ..
                const bool oddf = fv & 1;
                // even frame: the stored row
                const int d_even = Rd - (fv >> 1);
                float dve = rg[d_even * kPhND + (c < kPW ? a.col0 + lov + c - cc0 : 0)];
                const int a_even = Rt - (fv >> 1), a_odd = Rt - ((fv + 1) >> 1);
                const int scol = c == kPB ? (oddf ? a.mcol_odd : a.mcol) : (oddf ? a.xcol0_odd : a.xcol0) + ((c - kPW) & 1);
                const float dvs = ax[(oddf ? a_odd : a_even) * 8 + scol];
                dv = dve;
                if (__ballot(oddf)) {
                    // ---- odd frame: rebuild delta_f[i] for the band source i = lo + c of this lane from row f - 1
                    int i = lov + c;
                    const bool iok = c < kPW && i >= 0 && i < S;
                    i = i < 0 ? 0 : (i > S - 1 ? S - 1 : i);
                    const int ee0 = sel(ce0[0], ce0[1]);
                    const int fo = (ftv & 1) ? ftv : ftv - 1;
                    const float* src = rg + d_even * kPhND + (a.col0 + i - kPairD - cc0);    // row (f - 1) / 2 = f >> 1
                    const float* wt = tabH + i * kPairRow;
                    const float* ar = ax + a_even * 8;                        // scalars of frame f - 1
                    float acc0 = ar[a.mcol] + wt[kPB], acc1 = ar[a.xcol0] + wt[kPW];
                    acc0 = fmaxf(acc0, ar[a.xcol0 + 1] + wt[kPW + 1]);
                    // (three rounds of eight sources, then five: all 58 LDS reads in flight at once would not fit the registers of a
                    //  sixteen-wave workgroup)
#pragma nounroll
                    for (int qb = 0; qb < 24; qb += 8) {
#pragma unroll
                        for (int q = 0; q < 8; q += 2) {
                            acc0 = fmaxf(acc0, src[qb + q] + wt[qb + q]);
                            acc1 = fmaxf(acc1, src[qb + q + 1] + wt[qb + q + 1]);
                        }
                    }
#pragma unroll
                    for (int q = 24; q + 1 < kPW; q += 2) {
                        acc0 = fmaxf(acc0, src[q] + wt[q]);
                        acc1 = fmaxf(acc1, src[q + 1] + wt[q + 1]);
                    }
                    acc0 = fmaxf(acc0, src[kPW - 1] + wt[kPW - 1]);
                    int ei = i - ee0;
                    ei = ei < 0 ? 0 : (ei > kPhNE - 1 ? kPhNE - 1 : ei);
                    const float en = rg[kPhRD * kPhND + kPhRA * 8 + ((fo - fv) >> 1) * kPhNE + ei];
                    const float dvo = iok ? fmaxf(acc0, acc1) + en : -INFINITY;
                    dv = oddf ? dvo : dve;
                }
                dv = c < kPW ? dv : dvs;
            }
            const float w = tabH[curv * kPairRow + c];
            const float v = dv + w;
            const float vred = c == kPB ? -INFINITY : v;
            const float hm = pg_half_max(vred);
            float m[2], mf[2], cj[2];
            m[0] = pg_lane(hm, 31); m[1] = pg_lane(hm, 63);
            mf[0] = pg_lane(v, kPB); mf[1] = pg_lane(v, 32 + kPB);
            cj[0] = pg_lane(w, kPB); cj[1] = pg_lane(w, 32 + kPB);
            const unsigned long long eq = __ballot(vred == self(m[0], m[1]));
            int idx[2];
            bool slow[2];
            auto lowest_candidate = [&](const unsigned bits, const int lo_h) -> unsigned {
                unsigned best = 0x7fffffffu;
                const unsigned bw = bits & ((1u << kPW) - 1u);
                if (bw) best = (unsigned)(lo_h + __builtin_ctz(bw));          // band candidates ascend with the source index
                if ((bits >> kPW) & 1u) best = (unsigned)a.extras[0] < best ? (unsigned)a.extras[0] : best;
                if ((bits >> (kPW + 1)) & 1u) best = (unsigned)a.extras[1] < best ? (unsigned)a.extras[1] : best;
                return best;
            };
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                slow[h] = act[h] && !(mf[h] < m[h]);
                idx[h] = act[h] ? (int)lowest_candidate((unsigned)(eq >> (32 * h)), lo[h]) : cur[h];
            }
            if (slow[0] || slow[1]) {
#pragma nounroll
                for (int h = 0; h < 2; ++h) {
                    if (!(h ? slow[1] : slow[0])) continue;
                    if (h) ++n_full[1]; else ++n_full[0];
                    // ---- whole-row evaluation with all 64 lanes: every source outside the band / extra columns contributes
                    //      fl(delta_f[i] + c_cur)
                    const int fh = h ? f[1] : f[0], loh = h ? lo[1] : lo[0], fth = h ? ftop[1] : ftop[0];
                    const float mh = h ? m[1] : m[0], cjh = h ? cj[1] : cj[0];
                    const float* __restrict__ hh = h ? hist[1] : hist[0];
                    float* stage = h ? region[1] : region[0];
                    float d[EPL];
                    const bool oddf = HALF && (fh & 1);
                    const float* __restrict__ grow = hh + (size_t)(HALF ? fh >> 1 : fh) * SD + a.col0;   // row f, or f - 1 of an odd f
#pragma unroll
                    for (int e = 0; e < EPL; ++e) d[e] = inS[e] ? grow[e * 64 + lane] : -INFINITY;
                    if (oddf) {
                        // rebuild the whole of delta_f: stage delta_{f-1} in this stream's span area (both tiles are fetched again afterwards)
                        const ET* __restrict__ eh = h ? E[1] : E[0];
                        const float* ax = stage + kPhRD * kPhND + (((fth + 1) >> 1) - (fh >> 1)) * 8;    // scalars of frame f - 1
#pragma unroll
                        for (int e = 0; e < EPL; ++e) stage[e * 64 + lane] = d[e];
                        const float Mp = ax[a.mcol], x0 = ax[a.xcol0], x1 = ax[a.xcol0 + 1];
#pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const int i = e * 64 + lane;
                            const int ic = inS[e] ? i : S - 1;
                            const float* wt = tabH + ic * kPairRow;
                            float acc = fmaxf(fmaxf(Mp + wt[kPB], x0 + wt[kPW]), x1 + wt[kPW + 1]);
#pragma unroll 4
                            for (int q = 0; q < kPW; ++q) {
                                int sidx = ic - kPairD + q;
                                sidx = sidx < 0 ? 0 : (sidx > S - 1 ? S - 1 : sidx);      // (a source that does not exist has weight -inf)
                                acc = fmaxf(acc, stage[sidx] + wt[q]);
                            }
                            d[e] = inS[e] ? acc + pg_ld<ET>(eh + (size_t)fh * S + ic) : -INFINITY;
                        }
                        drop = true;
                        if (h) ++n_reb[1]; else ++n_reb[0];
                    }
                    float m2 = -INFINITY;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int i = e * 64 + lane;
                        const bool excl = xcol[e] || (unsigned)(i - loh) < (unsigned)kPW;
                        d[e] = excl ? -INFINITY : d[e] + cjh;
                        m2 = fmaxf(m2, d[e]);
                    }
                    const float mm = fmaxf(mh, pg_wave_max(m2));
                    unsigned best = 0x7fffffffu;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const unsigned long long mk = __ballot(d[e] == mm && inS[e]);
                        if (mk) { const unsigned ci = e * 64 + __builtin_ctzll(mk); best = ci < best ? ci : best; }
                    }
                    if (mm == mh) { const unsigned ci = (unsigned)(h ? idx[1] : idx[0]); best = ci < best ? ci : best; }
                    const int res = best == 0x7fffffffu ? 0 : (int)best;       // an all -inf frame resolves to index 0 like np.argmax
                    if (h) idx[1] = res; else idx[0] = res;
                }
            }
            // ---- commit the step
            outv = c == k ? sel(idx[0], idx[1]) : outv;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!act[h]) continue;
                cur[h] = pg_uni(idx[h]);
                if (f[h] == hi_c[h]) entry[h] = cur[h];     // the state this chunk assumes at its upper boundary
                f[h] -= 1;
            }
            ndone = k + 1;
        }
        // ---- write the decided frames of this tile (below the chunk's upper boundary only: the warm-up writes nothing)
        {
            const int fr = sel(ftop[0], ftop[1]) - c;
            const bool ok = c < ndone && sel(valid[0], valid[1]) && fr < sel(hi_c[0], hi_c[1]) && fr >= sel(lo_c[0], lo_c[1]) &&
                            fr >= sel(f[0], f[1]) + 1;
            int32_t* sp = hv ? states[1] : states[0];
            if (ok) sp[fr] = outv;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
        if (valid[h] && lane == 0) {
            a.entry[(size_t)song[h] * C + chunk[h]] = entry[h];
            if (a.counters) {
                int32_t* ct = a.counters + (size_t)song[h] * kBtCounters;
                if (n_tiles[h]) atomicAdd(ct + kCtTiles, n_tiles[h]);
                if (n_miss[h]) atomicAdd(ct + kCtMisses, n_miss[h]);
                if (n_full[h]) atomicAdd(ct + kCtFullRows, n_full[h]);
                if (n_reb[h]) atomicAdd(ct + kCtRebuilt, n_reb[h]);
            }
        }
}

static int pair_epl(const BtArgs& a) { return (a.S + 63) / 64 <= 6 ? 6 : 12; }
static size_t pair_lds_bytes(const BtArgs& a, int nwaves) {
    const size_t region = a.hist_half ? kPhRegion : kPfRegion;
    return sizeof(float) * ((((size_t)a.S * kPairRow + 3) & ~(size_t)3) + (size_t)nwaves * 2 * region);
}
static int pair_waves(const BtArgs& a) {
    int nw = 16;
    while (nw > 2 && pair_lds_bytes(a, nw) + 512 > 160 * 1024) nw >>= 1;
    return nw;
}

// The half-wave speculative pass takes banded plans whose exception spans lie within kPairD sources of the target (the plan
// built tabH), with the frame maximum in the history, rows 16-byte aligned and wide enough for a span; with a half history
// the wave form's geometry.  The first frame of a fresh tile must fit its spans whatever the path state: checked here.
bool pair_backtrace_applies(const BtArgs& a) {
    if (!(a.banded && a.pair_ok && a.have_fmax && a.n_dense == 0 && a.n_extras <= 2 && a.SD % 4 == 0 && (a.S + 63) / 64 <= 12)) return false;
    const int ns = a.hist_half ? kPhND : kPfNS;
    if (a.SD < ns || a.S < kPhNE) return false;
    if (a.hist_half && !((a.S + 63) / 64 == 6 && a.xcol0 >= 0 && a.mcol_odd + 2 < 8 && a.xcol0_odd + 1 < 8 && kPhRD * kPhND >= 64 * 6)) return false;
    const int S = a.S, c0_max = (a.SD - ns) & ~3, ce_max = S - kPhNE;
    for (int j = 0; j < S; ++j) {
        const int lo = j - kPairD;
        const int wl = lo < 0 ? 0 : lo, wh = lo + kPW - 1 > S - 1 ? S - 1 : lo + kPW - 1;
        if (!a.hist_half) {
            int c0 = (a.col0 + lo - 17 + 8) & ~15;
            c0 = c0 < 0 ? 0 : (c0 > c0_max ? c0_max : c0);
            if (a.col0 + wl - c0 < 0 || a.col0 + wh - c0 >= kPfNS) return false;
        } else {
            int c0 = (a.col0 + lo - kPairD - 19 + 8) & ~15;
            c0 = c0 < 0 ? 0 : (c0 > c0_max ? c0_max : c0);
            int ce0 = lo - 17;
            ce0 = ce0 < 0 ? 0 : (ce0 > ce_max ? ce_max : ce0);
            const int sl = wl - kPairD < 0 ? 0 : wl - kPairD, sh = wh + kPairD > S - 1 ? S - 1 : wh + kPairD;
            if (a.col0 + sl - c0 < 0 || a.col0 + sh - c0 >= kPhND || wl < ce0 || wh >= ce0 + kPhNE) return false;
            if (a.col0 + wl - c0 < 0 || a.col0 + wh - c0 >= kPhND) return false;
        }
    }
    return pair_lds_bytes(a, 2) + 512 <= 160 * 1024;
}

template <int EPL, bool HALF, typename ET>
static hipError_t launch_pair_t(const BtArgs& a, hipStream_t st) {
    const int nw = pair_waves(a);
    const size_t lds = pair_lds_bytes(a, nw);
    const long long waves = ((long long)a.B * a.chunks + 1) / 2;
    hipLaunchKernelGGL((pair_backtrace_kernel<EPL, HALF, ET>), dim3((int)((waves + nw - 1) / nw)), dim3(nw * 64), lds, st, a);
    return hipGetLastError();
}

// speculative pass only; the caller launches the verify-and-repair pass of the sparse / half kernel behind it
hipError_t launch_backtrace_pair(const BtArgs& a, hipStream_t st) {
    if (!pair_backtrace_applies(a)) return hipErrorInvalidConfiguration;
    if (a.hist_half) return a.e_f16 ? launch_pair_t<6, true, __half>(a, st) : launch_pair_t<6, true, float>(a, st);
    return pair_epl(a) == 6 ? launch_pair_t<6, false, float>(a, st) : launch_pair_t<12, false, float>(a, st);
}

// (song, chunk) streams up to twice the resident wave capacity of the chip (sixteen waves per CU, two streams each)
int pair_backtrace_chunks(int64_t B, int T) {
    long long c = (2 * 4 * 1024) / (B > 0 ? B : 1);
    const long long cmax = T / (8 * kBtWarmSparse) > 1 ? T / (8 * kBtWarmSparse) : 1;
    c = c > cmax ? cmax : c;
    c = c > kBtMaxChunks ? kBtMaxChunks : c;
    return c < 1 ? 1 : (int)c;
}

}  // namespace vit
