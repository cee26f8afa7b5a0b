// backtrace_half.hip -- exact, time-parallel back-trace over a HALF history (wave form, wave.hip HM 1).
//
// The wave form is HBM-bound and the float32 delta row it stores per frame (1536 B) is more than the emission row it
// reads (1444 B).  With HM 1 it stores the rows of even frames only.  This kernel decides an even frame exactly like
// sparse_backtrace_kernel (candidates fl(delta_t[i] + logA_T[j][i]) over the window and the extra columns of the path
// state j at t+1, wave max, the bound fl(M_t + c_j), lowest matching index) and REBUILDS what it needs of an odd frame
// t: the path state's window holds 32 sources i, and
//     delta_t[i] = fl( max( max_w fl(delta_{t-1}[lo_i + w] + logA_T[i][lo_i + w]),  fl(M_{t-1} + c_i),
//                           fl(delta_{t-1}[x] + logA_T[i][x]) for extra columns x )  +  logE[t][i] )
// is the floor-max form of the forward recursion (plan.floor_all_ok; every value compared is one the dense recursion
// forms, so the result is the forward kernel's delta_t[i] bit for bit).  Two lanes per source i, sixteen window entries
// each, one v_permlane32_swap to join them.  M_t and delta_t of the extra columns of BOTH frames of a pair sit in the
// idle slots of the even row (lane 0 of the forward kernel), so an odd frame costs no history bytes at all; it costs
// its 32 emission values, which are fetched as a span like the delta rows.
//
// Per tile of 8 stored rows (16 frames) a wave fetches: 96 delta columns around the path of each row (3 lines), the
// first 8 floats of 9 rows (scalars), and 64 emission columns of 8 odd rows (2-3 lines each).  A window that leaves a
// span drops the tile (re-fetched around the new state, starting at the frame that missed); a bound failure evaluates
// the whole row -- from global memory at an even frame, rebuilt in full (361 x 35 candidates) at an odd one: rare (e.g.
// a voiced -> unvoiced switch), exact either way.  Chunking, speculative warm-up and the verify-and-repair pass are
// those of sparse_backtrace_kernel; a chunk's guess row is always an even frame.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace vit {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kHbR = 8;             // stored rows per tile
constexpr int kHbND = 96;           // delta span columns (a 32-wide window of 32-wide windows is 63 columns)
constexpr int kHbNE = 64;           // emission span columns
constexpr int kHbAux = 8;           // scalars per stored row (the row's first eight floats)
constexpr int kHbW = 32;            // window width (the wave form exists for half-widths up to 14)
constexpr int kHbTile = kHbR * kHbND + (kHbR + 1) * kHbAux + kHbR * kHbNE;   // floats per wave
constexpr int kHbWX1 = kHbW + kMaxExtras + 1;
constexpr int kHbWXS = (kHbWX1 + 3) / 4 * 4;      // row stride of the candidate table in LDS: 16-byte aligned rows (and halves of a window)
constexpr int kHbCB = kHbW + kMaxExtras;    // candidate lane of the bound

__device__ __forceinline__ int hb_song_length(const int64_t* lengths, int song, int T) {
    if (!lengths) return T;
    long long v = lengths[song];
    v = v < 1 ? 1 : v;
    return v > T ? T : (int)v;
}
__device__ __forceinline__ int hb_clamp(int x, int hi) {   // min(max(x, 0), hi): one v_med3_i32
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
__device__ __forceinline__ float hb_wave_max(float x) {   // kernels.hip wave_max_all
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
// max(x[l], x[l ^ 32]) in every lane: v_permlane32_swap exchanges the upper 32 lanes of its first operand with the lower 32
// of its second; fed two copies of x it leaves {x.lo, x.lo} and {x.hi, x.hi}
__device__ __forceinline__ float hb_other_half(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <typename ET>
__device__ __forceinline__ float hb_ld(const ET* p);
template <>
__device__ __forceinline__ float hb_ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float hb_ld<__half>(const __half* p) { return __half2float(*p); }

}  // namespace

// MODE 0: speculative pass, one wave per (song, chunk).  MODE 1: verify-and-repair pass, one wave per song.
template <int MODE, typename ET>
__global__ void __launch_bounds__(1024) half_backtrace_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int EPL = 6, W = kHbW, WX1 = kHbWX1, WXS = kHbWXS, CB = kHbCB;
    const int S = a.S, SP = a.SP, SD = a.SD, T = a.T;
    const int nx = a.n_extras;
    const int nwaves = blockDim.x >> 6;
    float* tiles = reinterpret_cast<float*>(smem);                            // [nwaves][kHbTile]
    float* tabX = tiles + nwaves * kHbTile;                                   // [SP][WXS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const float* __restrict__ gtab = reinterpret_cast<const float*>(a.image + a.off_tabX);
        const int nthr = blockDim.x;
        for (int k = tid; k < SP * WX1; k += nthr) tabX[(k / WX1) * WXS + k % WX1] = gtab[k];
    }
    __syncthreads();

    const int C = a.chunks;
    const int gw = blockIdx.x * nwaves + wv;
    const int song = MODE == 0 ? gw / C : gw;
    const int chunk = MODE == 0 ? gw % C : 0;
    if (song >= a.B) return;
    const int Tb = hb_song_length(a.lengths, song, T);
    int32_t* __restrict__ states = a.states + (size_t)song * T;
    const float* __restrict__ hist = a.hist + (size_t)song * a.hist_rows * SD;
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)song * T * S;
    float* td = tiles + wv * kHbTile;                 // [kHbR][kHbND]   delta spans of the stored rows
    float* ta = td + kHbR * kHbND;                    // [kHbR + 1][8]   scalars
    float* te = ta + (kHbR + 1) * kHbAux;             // [kHbR][kHbNE]   emission spans of the odd frames
    const int last_row = (Tb - 1) >> 1;               // last row the forward pass wrote for this song

    // ---- per-lane constants: candidate c = lane: 0 .. W-1 the window, W .. W+nx-1 the extra columns, CB the bound
    const bool isw = lane < W, cand = lane < W + nx;
    const int xs = (lane >= W && lane < W + nx) ? a.extras[(lane - W) & (kMaxExtras - 1)] : 0;
    const int aux_even = lane == CB ? a.mcol : a.xcol0 + ((lane - W) & (kMaxExtras - 1));
    const int aux_odd = lane == CB ? a.mcol_odd : a.xcol0_odd + ((lane - W) & (kMaxExtras - 1));
    const int tb = lane < WX1 ? lane : WX1 - 1;
    const unsigned long long cand_or_bound = (nx >= 64 - W ? ~0ull : ((1ull << (W + nx)) - 1ull)) | (1ull << CB);
    const int il = lane & 31, hh = lane >> 5;          // odd frames: source il of the window, half hh of ITS window
    bool inS[EPL], xcol[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int i = e * 64 + lane;
        inS[e] = i < S;
        bool x = i >= S;
#pragma unroll
        for (int k = 0; k < kMaxExtras; ++k) x |= (k < nx && i == a.extras[k]);
        xcol[e] = x;
    }
    const int lo_max = S - W, lo_off = a.lo_off;
    const bool fast_rows = !a.no_fast_rows && (nx == 0 || (nx == 1 && a.extras[0] == S - 1));      // no index above the one extra column
    auto lo_of = [&](const int j) -> int { const int l = j - lo_off; return l < 0 ? 0 : (l > lo_max ? lo_max : l); };
    const int c0_max = (SD - kHbND) & ~3;
    const int ce_max = S - kHbNE;

    // A tile = stored rows r_lo .. r_lo + rows - 1: their delta spans, the scalars of rows r_lo .. r_lo + rows, the emission
    // spans of the odd frames 2r + 1.  (Fetching the next tile ahead into registers was measured: no gain -- the sixteen waves of a
    // CU already hide the fetch -- at 25 more registers.)
    struct TileRegs { f32x4 sd[3]; f32x4 sa; float se[kHbR]; };
    auto tile_load = [&](const int r_lo, const int rows, const int c0, const int ce0, TileRegs& tr) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int idx = lane + 64 * v;
            int r = idx / (kHbND / 4);
            const int q = idx % (kHbND / 4);
            r = r < rows ? r : rows - 1;
            tr.sd[v] = *reinterpret_cast<const f32x4*>(hist + (size_t)(r_lo + r) * SD + c0 + 4 * q);
        }
        {
            int r = (lane >> 1) < rows ? (lane >> 1) : rows;       // row r_hi + 1 carries the scalars of an odd `top`
            r = r_lo + r > last_row ? last_row : r_lo + r;
            tr.sa = *reinterpret_cast<const f32x4*>(hist + (size_t)r * SD + 4 * (lane & 1));
        }
#pragma unroll
        for (int r = 0; r < kHbR; ++r) {
            int f = 2 * (r_lo + (r < rows ? r : rows - 1)) + 1;
            f = f > Tb - 1 ? Tb - 1 : f;
            tr.se[r] = hb_ld<ET>(E + (size_t)f * S + ce0 + lane);
        }
    };
    auto tile_store = [&](const TileRegs& tr) {
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const int idx = lane + 64 * v;
            *reinterpret_cast<f32x4*>(td + (idx / (kHbND / 4)) * kHbND + 4 * (idx % (kHbND / 4))) = tr.sd[v];
        }
        if (lane < 2 * (kHbR + 1)) *reinterpret_cast<f32x4*>(ta + (lane >> 1) * kHbAux + 4 * (lane & 1)) = tr.sa;
#pragma unroll
        for (int r = 0; r < kHbR; ++r) te[r * kHbNE + lane] = tr.se[r];
    };
    int n_tiles = 0, n_miss = 0, n_full = 0, n_reb = 0, n_rep = 0, n_repf = 0;      // event counts of this wave

    // chase(top, bottom, cur, write): decide the states of frames top .. bottom (descending), starting from state `cur`
    // at frame top+1.
    auto chase = [&](int top, const int bottom, int cur, const bool write) -> int {
        while (top >= bottom) {
            cur = __builtin_amdgcn_readfirstlane(cur);
            const int r_hi = top >> 1;
            int r_lo = r_hi - (kHbR - 1);
            r_lo = r_lo < (bottom >> 1) ? (bottom >> 1) : r_lo;
            const int first = 2 * r_lo > bottom ? 2 * r_lo : bottom;
            const int nfr = top - first + 1;               // <= 2 * kHbR
            const int rows = r_hi - r_lo + 1;              // <= kHbR
            // ---- the tile, centred on the window of `cur`
            const int lo_c = lo_of(cur);
            int c0 = (a.col0 + lo_c - lo_off - 8) & ~15;   // 64-byte aligned; the 63 source columns of an odd frame sit 8..23 columns in
            c0 = c0 < 0 ? 0 : (c0 > c0_max ? c0_max : c0);
            int ce0 = lo_c - (kHbNE - W) / 2;
            ce0 = ce0 < 0 ? 0 : (ce0 > ce_max ? ce_max : ce0);
            ++n_tiles;
            {
                TileRegs tr;
                tile_load(r_lo, rows, c0, ce0, tr);
                tile_store(tr);
            }
            int outv = 0;
            const int oldv = (MODE == 1 && lane < nfr) ? states[first + lane] : -1;
            int fstop = -1;           // MODE 1: frame at which the new path met the stored one
            int fnext = first - 1;    // where the next tile starts (a miss or a rebuilt row ends this one early)
            int fdone = first;        // lowest frame decided in this pass
            // Window starts this tile serves, as ONE interval [lo_a, lo_a + lo_span] (the odd-frame conditions: the sources of the 32
            // windows inside the delta span, the 32 sources inside the emission span; they imply the even-frame condition).  With
            // g = the state of span column 0:  lo_of(lo) >= g,  lo_of(lo + W - 1) + W <= g + ND,  ce0 <= lo <= ce0 + NE - W.
            const int g = c0 - a.col0;
            int lo_a = ce0 > g ? ce0 : g, lo_b = ce0 + kHbNE - W < g + kHbND - W ? ce0 + kHbNE - W : g + kHbND - W;
            if (g > 0) lo_a = lo_a > g + lo_off ? lo_a : g + lo_off;
            if (g + kHbND - W < lo_max) lo_b = lo_b < g + kHbND - 2 * W + 1 + lo_off ? lo_b : g + kHbND - 2 * W + 1 + lo_off;
            const unsigned lo_span = (unsigned)(lo_b - lo_a);
            // The scalar unit of a CU serves its sixteen waves and was ~70 % busy with this loop's index arithmetic (96 scalar
            // instructions per frame, profiles/r03_pmc_B1024_half.txt): the path state goes through an opaque VGPR copy, so that the
            // window start, the fit check and every LDS index are vector instructions (four pipes per CU) instead.
            const int lane_mg = lane - g;                                   // span column of window candidate `lane`, relative to lo
            for (int f = __builtin_amdgcn_readfirstlane(top); f >= first; --f) {
                cur = __builtin_amdgcn_readfirstlane(cur);
                if (fast_rows) {
                    // ---- the unexceptional frames in a loop of their own with direct exits (backtrace_sparse.hip, fast rows): the window
                    //      start inside the tile's interval, the bound candidate below the maximum, a window candidate attaining it (the one
                    //      extra column is the last state, so the lowest window match is the lowest match).  Any other frame falls through
                    //      to the general code below, which evaluates it again from scratch.
                    for (;;) {
                        int curv;
                        asm volatile("v_mov_b32 %0, %1" : "=v"(curv) : "s"(cur));
                        const int lov = hb_clamp(curv - lo_off, lo_max);
                        const int rr = (f >> 1) - r_lo;
                        const int rowd = rr * kHbND, rowa = rr * kHbAux;
                        float dv;
                        if (f & 1) {
                            const int iv = lov + il;
                            const int li = hb_clamp(iv - lo_off, lo_max);
                            const float* __restrict__ src = td + (rowd - g + 16 * hh) + li;
                            const float* __restrict__ wt = tabX + __umul24(iv, WXS);
                            // every read of the rebuild goes out before the first sum (left alone the compiler reads a pair, waits, adds: eight
                            // LDS round trips in a row); the sixteen weights of this half are four aligned quads (row stride WXS)
                            const f32x4* __restrict__ wq = reinterpret_cast<const f32x4*>(wt + 16 * hh);
                            f32x4 w4[4];
                            float sv[16];
#pragma unroll
                            for (int q = 0; q < 4; ++q) w4[q] = wq[q];
#pragma unroll
                            for (int q = 0; q < 16; ++q) sv[q] = src[q];
                            const float* __restrict__ ar = ta + rowa;
                            const float fl_ = ar[a.mcol] + wt[CB];
                            const float xt_ = nx ? ar[a.xcol0] + wt[W] : -INFINITY;
                            asm volatile("" ::: "memory");
                            float acc0 = -INFINITY, acc1 = -INFINITY;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc0 = fmaxf(fmaxf(acc0, sv[4 * q] + w4[q].x), sv[4 * q + 2] + w4[q].z);
                                acc1 = fmaxf(fmaxf(acc1, sv[4 * q + 1] + w4[q].y), sv[4 * q + 3] + w4[q].w);
                            }
                            float acc = hb_other_half(fmaxf(acc0, acc1));
                            acc = fmaxf(fmaxf(acc, fl_), xt_);
                            const float dw = acc + te[rr * kHbNE - ce0 + iv];
                            dv = isw ? dw : ta[rowa + kHbAux + aux_odd];
                        } else {
                            dv = (isw ? td + rowd + lov + lane_mg : ta + rowa + aux_even)[0];
                        }
                        const float v = dv + tabX[__umul24(curv, WXS) + tb];
                        const float m = hb_wave_max(cand ? v : -INFINITY);
                        const unsigned long long ge = __ballot(v >= m) & cand_or_bound;
                        if ((__ballot((unsigned)(lov - lo_a) > lo_span) | (ge & (1ull << CB))) != 0) break;
                        const unsigned gw = (unsigned)ge;                      // (W = 32: the window candidates are lanes 0 .. 31)
                        if (gw == 0) break;
                        cur = __builtin_amdgcn_readfirstlane(lov) + __builtin_ctz(gw);
                        asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(outv) : "s"(cur), "s"(f - first) : "m0");   // lane <- cur (instead of move, compare, select)
                        if (MODE == 1 && cur == __builtin_amdgcn_readlane(oldv, f - first)) { fstop = f; fdone = f + 1; break; }
                        if (--f < first) break;
                    }
                    if (f < first || fstop >= 0) break;
                }
                int curv;
                asm volatile("v_mov_b32 %0, %1" : "=v"(curv) : "s"(cur));
                int lov = curv - lo_off;
                lov = lov < 0 ? 0 : (lov > lo_max ? lo_max : lov);
                if (__ballot((unsigned)(lov - lo_a) > lo_span)) { fnext = f; fdone = f + 1; ++n_miss; break; }
                const int rr = (f >> 1) - r_lo;                            // tile row: frame f (even) or f - 1 (odd)
                const int rowd = rr * kHbND, rowa = rr * kHbAux;
                float dv;
                if (f & 1) {
                    // ---- odd frame: rebuild delta_f[lo .. lo + W) from row f - 1 and the emission row
                    const int iv = lov + il;                               // the source this lane rebuilds (both halves of the wave)
                    int li = iv - lo_off;
                    li = li < 0 ? 0 : (li > lo_max ? lo_max : li);         // start of ITS window
                    const float* __restrict__ src = td + (rowd - g + 16 * hh) + li;
                    const float* __restrict__ wt = tabX + iv * WXS;
                    float acc0 = -INFINITY, acc1 = -INFINITY;
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        acc0 = fmaxf(acc0, src[q] + wt[16 * hh + q]);
                        acc1 = fmaxf(acc1, src[q + 1] + wt[16 * hh + q + 1]);
                    }
                    float acc = hb_other_half(fmaxf(acc0, acc1));
                    const float* __restrict__ ar = ta + rowa;
                    acc = fmaxf(acc, ar[a.mcol] + wt[CB]);
                    for (int k = 0; k < nx; ++k) acc = fmaxf(acc, ar[a.xcol0 + k] + wt[W + k]);
                    const float dw = acc + te[rr * kHbNE - ce0 + iv];
                    dv = isw ? dw : ta[rowa + kHbAux + aux_odd];
                } else {
                    dv = (isw ? td + rowd + lov + lane_mg : ta + rowa + aux_even)[0];
                }
                const float av = tabX[curv * WXS + tb];
                float v = dv + av;
                const float vc = cand ? v : -INFINITY;
                const float m = hb_wave_max(vc);
                // one compare: a candidate lane with v >= m attains the maximum; the bound lane with fl(M_f + c_cur) >= m means a
                // row-constant candidate may tie or win
                const unsigned long long ge = __ballot(v >= m) & cand_or_bound;
                const int lo = __builtin_amdgcn_readfirstlane(lov);
                auto lowest_candidate = [&](const unsigned long long mk) -> unsigned {
                    unsigned best = 0x7fffffffu;
                    const unsigned mw = (unsigned)mk;
                    if (mw) best = lo + __builtin_ctz(mw);                     // window candidates ascend with the source index
                    unsigned mx = (unsigned)(mk >> 32) & ((1u << kMaxExtras) - 1u);   // extra columns: arbitrary indices
                    while (mx) {
                        const unsigned c = __builtin_amdgcn_readlane(xs, 32 + __builtin_ctz(mx));
                        best = c < best ? c : best;
                        mx &= mx - 1;
                    }
                    return best;
                };
                unsigned idx = 0x7fffffffu;
                bool rebuilt = false;
                if (!((ge >> CB) & 1ull)) {
                    idx = lowest_candidate(ge);
                } else {
                    ++n_full;
                    const float cj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av), CB));
                    // ---- full evaluation: every source outside the window / extra columns contributes fl(delta_f[i] + c_cur)
                    float d[EPL];
                    const float* __restrict__ grow = hist + (size_t)(f >> 1) * SD + a.col0;     // row f (even) or f - 1 (odd)
#pragma unroll
                    for (int e = 0; e < EPL; ++e) d[e] = inS[e] ? grow[e * 64 + lane] : -INFINITY;
                    if (f & 1) {
                        // rebuild the whole of delta_f: stage delta_{f-1} in the tile's span area (the tile is dropped afterwards)
#pragma unroll
                        for (int e = 0; e < EPL; ++e) td[e * 64 + lane] = d[e];
                        const float* __restrict__ ar = ta + rowa;
                        const float Mp = ar[a.mcol];
#pragma unroll
                        for (int e = 0; e < EPL; ++e) {
                            const int i = e * 64 + lane;
                            const int ic = inS[e] ? i : S - 1;
                            const float* __restrict__ src = td + lo_of(ic);
                            const float* __restrict__ wt = tabX + ic * WXS;
                            float acc = Mp + wt[CB];
#pragma unroll 8
                            for (int w = 0; w < W; ++w) acc = fmaxf(acc, src[w] + wt[w]);
                            for (int k = 0; k < nx; ++k) acc = fmaxf(acc, ar[a.xcol0 + k] + wt[W + k]);
                            d[e] = inS[e] ? acc + hb_ld<ET>(E + (size_t)f * S + ic) : -INFINITY;
                        }
                        rebuilt = true;
                        ++n_reb;
                    }
                    float vf[EPL];
                    float m2 = -INFINITY;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int i = e * 64 + lane;
                        const bool excl = xcol[e] || (unsigned)(i - lo) < (unsigned)W;
                        vf[e] = excl ? -INFINITY : d[e] + cj;
                        m2 = fmaxf(m2, vf[e]);
                    }
                    const float mm = fmaxf(m, hb_wave_max(m2));
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const unsigned long long mk = __ballot(vf[e] == mm && inS[e]);
                        if (mk) { const unsigned c = e * 64 + __builtin_ctzll(mk); idx = c < idx ? c : idx; }
                    }
                    const unsigned c = lowest_candidate(__ballot(vc == mm && cand));
                    idx = c < idx ? c : idx;
                    if (idx == 0x7fffffffu) idx = 0;        // an all -inf frame resolves to index 0 like np.argmax
                }
                cur = (int)idx;
                outv = lane == f - first ? cur : outv;
                if (MODE == 1 && cur == __builtin_amdgcn_readlane(oldv, f - first)) { fstop = f; fdone = f + 1; break; }
                if (rebuilt) { fnext = f - 1; fdone = f; break; }
            }
            if (write && lane < nfr && first + lane >= fdone) states[first + lane] = outv;
            if (MODE == 1) n_repf += top - fdone + 1;
            if (MODE == 1 && fstop >= 0) return __builtin_amdgcn_readfirstlane(states[bottom]);   // the stored path continues unchanged
            top = fnext;
        }
        return cur;
    };

    const int Lf = Tb - 1;
    if (MODE == 0) {
        const int lo_c = (int)((long long)Lf * chunk / C), hi_c = (int)((long long)Lf * (chunk + 1) / C);
        if (chunk == C - 1) {
            for (int t = Tb + lane; t < T; t += 64) states[t] = -1;
            if (lane == 0) states[Tb - 1] = a.last_state[song];
        }
        int top = hi_c - 1 + a.warm;
        top += top & 1 ? 0 : 1;                   // the guess row top + 1 must be a stored (even) frame
        int cur;
        if (chunk == C - 1 || top >= Lf - 1) {
            top = Lf - 1;
            cur = __builtin_amdgcn_readfirstlane(a.last_state[song]);
        } else {
            // guess: lowest-index argmax of delta row top+1
            const float* g = hist + (size_t)((top + 1) >> 1) * SD + a.col0;
            float d[EPL];
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                d[e] = inS[e] ? g[e * 64 + lane] : -INFINITY;
                m = fmaxf(m, d[e]);
            }
            m = hb_wave_max(m);
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
                ++n_rep;
                truth = chase(hi_c - 1, lo_c, truth, true);   // re-chase from the true state; ends at frame lo_c
            } else {
                truth = -1;                       // chunk c stands: its frame lo_c is already in `states`
            }
        }
    }
    if (lane == 0 && a.counters) {
        int32_t* ct = a.counters + (size_t)song * kBtCounters;
        if (n_tiles) atomicAdd(ct + kCtTiles, n_tiles);
        if (n_miss) atomicAdd(ct + kCtMisses, n_miss);
        if (n_full) atomicAdd(ct + kCtFullRows, n_full);
        if (n_reb) atomicAdd(ct + kCtRebuilt, n_reb);
        if (n_rep) atomicAdd(ct + kCtRepairs, n_rep);
        if (n_repf) atomicAdd(ct + kCtRepairFrames, n_repf);
    }
}

static size_t half_lds_bytes(const BtArgs& a, int nwaves) {
    return sizeof(float) * ((size_t)nwaves * kHbTile + (size_t)a.SP * kHbWXS);
}

// Half histories are written for plans whose window is 32 wide with an affine start (every matrix the wave form takes
// in practice); the first frame of a fresh tile must fit its spans for every path state, at either parity -- checked here
// for the plan's actual window starts, so that a tile always makes progress.
bool half_backtrace_applies(const BtArgs& a) {
    if (!(a.banded && a.n_dense == 0 && a.W == kHbW && a.lo_affine && a.n_extras <= 2 && a.S >= kHbNE && (a.S + 63) / 64 == 6 &&
          a.SD == 384 && a.col0 == a.SD - a.S && a.col0 >= 2 * (a.n_extras + 1)))
        return false;
    const int S = a.S, lo_max = S - kHbW, c0_max = (a.SD - kHbND) & ~3, ce_max = S - kHbNE;
    auto lo_of = [&](int j) { const int l = j - a.lo_off; return l < 0 ? 0 : (l > lo_max ? lo_max : l); };
    for (int j = 0; j < S; ++j) {
        const int lo = lo_of(j);
        int c0 = (a.col0 + lo - a.lo_off - 8) & ~15;
        c0 = c0 < 0 ? 0 : (c0 > c0_max ? c0_max : c0);
        int ce0 = lo - (kHbNE - kHbW) / 2;
        ce0 = ce0 < 0 ? 0 : (ce0 > ce_max ? ce_max : ce0);
        const int wlo = a.col0 + lo - c0;
        const int s_lo = a.col0 + lo_of(lo) - c0, s_hi = a.col0 + lo_of(lo + kHbW - 1) + kHbW - c0;
        if (wlo < 0 || wlo + kHbW > kHbND || s_lo < 0 || s_hi > kHbND || lo < ce0 || lo + kHbW > ce0 + kHbNE) return false;
    }
    return half_lds_bytes(a, 16) + 1024 <= 160 * 1024;
}

template <typename ET>
static hipError_t launch_half_t(const BtArgs& a, hipStream_t st, int phases) {
    // sixteen waves per workgroup by default; eight (or four) on request: with its 104 registers per lane a sixteen-wave workgroup needs
    // 416 registers of every SIMD and cannot start on a CU whose SIMDs each hold a 256-register forward wave (1024 songs in flight); an
    // eight-wave workgroup (208 per SIMD) can, which is what puts this back-trace UNDER the next batch's forward pass (BtArgs::block_waves)
    const int nw = a.block_waves == 8 || a.block_waves == 4 ? a.block_waves : 16;
    const size_t lds = half_lds_bytes(a, nw);
    const long long waves0 = (long long)a.B * a.chunks;
    hipError_t e = hipSuccess;
    if (phases & 1) {
        hipLaunchKernelGGL((half_backtrace_kernel<0, ET>), dim3((int)((waves0 + nw - 1) / nw)), dim3(nw * 64), lds, st, a);
        e = hipGetLastError();
    }
    if (e != hipSuccess || a.chunks <= 1 || !(phases & 2)) return e;
    hipLaunchKernelGGL((half_backtrace_kernel<1, ET>), dim3((int)((a.B + nw - 1) / nw)), dim3(nw * 64), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_backtrace_half(const BtArgs& a, hipStream_t st, int phases) {
    if (!half_backtrace_applies(a) || !a.logE) return hipErrorInvalidConfiguration;
    return a.e_f16 ? launch_half_t<__half>(a, st, phases) : launch_half_t<float>(a, st, phases);
}

}  // namespace vit
