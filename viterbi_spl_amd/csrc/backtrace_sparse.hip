// backtrace_sparse.hip -- exact, time-parallel back-trace that fetches only the part of the delta history it uses.
//
// banded_backtrace_kernel (kernels.hip) stages whole history rows through LDS: 1456 B per frame to use ~33 floats, and it
// is HBM-bound doing so (5.5 TB/s; 21 % of a step at B = 1024).  The path moves slowly -- a band transition moves it by
// at most the band half-width, and in music it moves by a bin or two per frame -- so this kernel fetches, for a tile of
// K frames, only
//   * the SPAN: NS consecutive columns around the window of the current path state (NS = 64 for windows up to 32 wide,
//     128 up to 96: the window plus a guard of at least 16 columns on either side), the same columns for every row, and
//   * the row's auxiliary values: the frame maximum (column mcol) and delta of the extra columns,
// i.e. 3-5 cache lines of a row's 12.  Per frame t (descending) the decision is the one banded_backtrace_kernel takes:
// candidates fl(delta_t[i] + logA_T[j][i]) over the window and the extra columns of the path state j at t+1, wave max,
// and if fl(M_t + c_j) < max no row-constant candidate can tie or win: lowest matching index.  Otherwise (rare: e.g.
// a voiced -> unvoiced switch, whose best source can be any voiced state) the whole row is evaluated straight from
// global memory -- exact either way.  When the path leaves the span (a jump through the floor or an extra column, or
// accumulated drift) the tile is dropped and re-fetched around the new state, starting at the frame that missed.
// Chunking, speculative warm-up and the verify-and-repair pass are those of banded_backtrace_kernel.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace vit {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSpK = 16;            // rows per tile
constexpr int kSpAux = 8;           // auxiliary floats per row: [0] frame maximum, [1 + k] extra column k
// span columns per row (multiple of 4): the window plus at least 16 columns of guard on either side
constexpr int sp_span(int kc) { return kc == 1 ? 64 : (kc == 2 ? 128 : 160); }

__device__ __forceinline__ int sp_song_length(const int64_t* lengths, int song, int T) {
    if (!lengths) return T;
    long long v = lengths[song];
    v = v < 1 ? 1 : v;
    return v > T ? T : (int)v;
}
__device__ __forceinline__ int sp_clamp(int x, int hi) {   // min(max(x, 0), hi): one v_med3_i32
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
}
__device__ __forceinline__ float sp_wave_max(float x) {   // kernels.hip wave_max_all
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

}  // namespace

// NWT: 64 * NWT >= S (sources per lane in the full evaluation).  AFF: window start affine in the target.
// MODE 0: speculative pass, one wave per (song, chunk).  MODE 1: verify-and-repair pass, one wave per song.
// KC: candidate slots per lane (slot k of lane l holds candidate 64k + l: the W window entries, then the extra columns, then
// the bound): 1 for windows up to 59 wide, 2 up to 123 (the jdc band on the 722-state grid: W = 96), 3 for W = 128 (imm).  GT: the per-target
// candidate table is read from the plan image (L2) instead of LDS -- at S = 722, W = 96 it is 310 KB.
template <int NWT, bool AFF, int MODE, int KC, bool GT>
__global__ void __launch_bounds__(1024) sparse_backtrace_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int EPL = NWT;
    constexpr int kSpNS = sp_span(KC);
    constexpr int kSpRS = kSpNS + kSpAux;             // floats per tile row
    constexpr int kSpVec = kSpK * kSpNS / 4 / 64;     // float4 per lane per tile
    const int S = a.S, SP = a.SP, SD = a.SD, T = a.T, W = a.W;
    const int nx = a.n_extras;
    const int WX1 = W + kMaxExtras + 1;    // candidate-table row: window, extras, row constant
    const int CB = W + kMaxExtras;         // candidate index of the bound (<= 63: checked by the launcher)
    const int nwaves = blockDim.x >> 6;
    float* tiles = reinterpret_cast<float*>(smem);                          // [nwaves][kSpK * kSpRS]
    int32_t* loL = reinterpret_cast<int32_t*>(tiles + nwaves * kSpK * kSpRS);   // [SP]
    float* tabX = reinterpret_cast<float*>(loL + SP);                       // [SP][WX1] (LDS form only)
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
    const int gw = blockIdx.x * nwaves + wv;
    const int song = MODE == 0 ? gw / C : gw;
    const int chunk = MODE == 0 ? gw % C : 0;
    if (song >= a.B) return;
    if (a.skip_nonpositive && a.lengths[song] < 1) return;        // segment of a checkpointed decode this song does not reach
    const int Tb = sp_song_length(a.lengths, song, T);
    int32_t* __restrict__ states = a.states + (size_t)song * a.states_stride;
    const float* __restrict__ hist = a.hist + (size_t)song * a.hist_rows * SD;
    float* tile = tiles + wv * kSpK * kSpRS;

    // ---- per-lane constants, per candidate slot: candidates 0 .. W-1 the window, W .. W+nx-1 the extra columns, CB the bound
    bool isw[KC], cand[KC];
    int xs[KC], auxi[KC], tb[KC];
    unsigned long long wmask[KC];                                          // lanes of slot k that hold window candidates
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const int c = 64 * k + lane;
        isw[k] = c < W;
        cand[k] = c < W + nx;
        xs[k] = (c >= W && c < W + nx) ? a.extras[(c - W) & (kMaxExtras - 1)] : 0;     // state of an extra-column candidate
        auxi[k] = c == CB ? 0 : 1 + ((c - W) & (kMaxExtras - 1));                      // aux entry read by a non-window candidate
        tb[k] = c < WX1 ? c : WX1 - 1;
        const int nwin = W - 64 * k;
        wmask[k] = nwin >= 64 ? ~0ull : (nwin <= 0 ? 0ull : ((1ull << nwin) - 1ull));
    }
    const int kb = CB >> 6, lb = CB & 63;                                  // slot / lane of the bound candidate
    const unsigned long long cand_or_bound = (W + nx >= 64 ? ~0ull : ((1ull << (W + nx)) - 1ull)) | (1ull << (CB & 63));   // (KC == 1)
    // aux loads: entry e of row r by lane r * 8 + e (two halves of eight rows)
    const int aux_e = lane & 7;
    const int aux_col = aux_e == 0 ? a.mcol : (aux_e <= nx ? (a.xcol0 >= 0 ? a.xcol0 + aux_e - 1 : a.col0 + a.extras[(aux_e - 1) & (kMaxExtras - 1)]) : a.mcol);
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
    const int lo_max = S - W;
    // at most one extra column and it is the last state (the unvoiced state of the reference's matrices): no index is above it
    const bool fast_rows = KC == 1 && !a.no_fast_rows && (nx == 0 || (nx == 1 && a.extras[0] == S - 1));
    // fast rows: byte offset of this lane's candidate in a tile row = wlo * fr_mul + fr_off (window lanes move with the window start)
    const unsigned fr_mul = isw[0] ? 4u : 0u, fr_off = 4u * (unsigned)(isw[0] ? lane : kSpNS + auxi[0]);
    const int c0_max = (SD - kSpNS) & ~3;    // (rows are 16-byte aligned; the clamp may leave the last span 16-byte aligned only)

    int n_tiles = 0, n_miss = 0, n_full = 0, n_rep = 0, n_repf = 0;      // event counts of this wave (vit_backtrace_counters)

    // chase(top, bottom, cur, write): decide the states of frames top .. bottom (descending) from the delta rows
    // top .. bottom, starting from state `cur` at frame top+1.
    auto chase = [&](int top, const int bottom, int cur, const bool write) -> int {
        while (top >= bottom) {
            cur = __builtin_amdgcn_readfirstlane(cur);
            const int first = top - kSpK + 1 > bottom ? top - kSpK + 1 : bottom;
            const int rows = top - first + 1;
            // ---- fetch the tile: span columns [c0, c0 + NS) of rows first .. top, centred on the window of `cur`
            int lo_c;
            if (AFF) { lo_c = cur - a.lo_off; lo_c = lo_c < 0 ? 0 : (lo_c > lo_max ? lo_max : lo_c); }
            else lo_c = __builtin_amdgcn_readfirstlane(loL[cur]);
            int c0 = (a.col0 + lo_c - (kSpNS - W) / 4) & ~15;      // 64-byte aligned: the span touches 2.5 lines of 128 B on average instead of 2.9
            c0 = c0 < 0 ? 0 : (c0 > c0_max ? c0_max : c0);
            ++n_tiles;
            {
                f32x4 stage[kSpVec];
                float auxv[2];
#pragma unroll
                for (int v = 0; v < kSpVec; ++v) {
                    const int idx = lane + 64 * v;
                    int r = idx / (kSpNS / 4);
                    const int q = idx % (kSpNS / 4);
                    r = r < rows ? r : rows - 1;
                    stage[v] = *reinterpret_cast<const f32x4*>(hist + (size_t)(first + r) * SD + c0 + 4 * q);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int r = (lane >> 3) + 8 * h;
                    r = r < rows ? r : rows - 1;
                    // (aux_frames == 3: the scalars of frame t from its carrier row -- one line per three frames)
                    const int t = first + r, ta = a.aux_frames == 3 ? wave_aux_row(t, Tb - 1) : t;
                    auxv[h] = hist[(size_t)ta * SD + aux_col + 2 * (ta - t)];
                }
#pragma unroll
                for (int v = 0; v < kSpVec; ++v) {
                    const int idx = lane + 64 * v;
                    *reinterpret_cast<f32x4*>(tile + (idx / (kSpNS / 4)) * kSpRS + 4 * (idx % (kSpNS / 4))) = stage[v];
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) tile[((lane >> 3) + 8 * h) * kSpRS + kSpNS + aux_e] = auxv[h];
            }
            int outv = 0;
            const int oldv = (MODE == 1 && lane < rows) ? states[first + lane] : -1;
            int rstop = -1;          // MODE 1: row at which the new path met the stored one
            int rmiss = -1;          // row whose window left the span: the tile is re-fetched from there
            for (int r = __builtin_amdgcn_readfirstlane(rows - 1); r >= 0; --r) {
                cur = __builtin_amdgcn_readfirstlane(cur);
                constexpr bool kLean = AFF && KC == 1 && !GT;          // the reference's S = 321 / 361 matrices
                if constexpr (kLean) {
                    if (fast_rows) {
                        // ---- the unexceptional rows, in a loop of their own with ONE early exit: the window fits the span, the bound
                        //      candidate stays below the maximum and a window candidate attains it.  The one extra column is the last state
                        //      (fast_rows), so the lowest window match is the lowest match whatever the extra candidate holds.  Any
                        //      other row falls through to the general code below, which evaluates it again from scratch.
                        for (;;) {
                            int curv;
                            asm volatile("v_mov_b32 %0, %1" : "=v"(curv) : "s"(cur));
                            const int lov = sp_clamp(curv - a.lo_off, lo_max);
                            const int wlov = lov + (a.col0 - c0);
                            // (a window that left the span reads entries of the tile that are not its own -- or nothing: LDS reads beyond the
                            //  allocation return zero -- and the row is an exceptional one whatever they hold)
                            const float dv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(tile + r * kSpRS) + __umul24(wlov, fr_mul) + fr_off);
                            const float vv = dv + tabX[__umul24(curv, WX1) + tb[0]];
                            const float mx = sp_wave_max(cand[0] ? vv : -INFINITY);
                            const unsigned long long ge = __ballot(vv >= mx) & cand_or_bound;
                            if ((__ballot((unsigned)wlov > (unsigned)(kSpNS - W)) | (ge & (1ull << CB))) != 0) break;
                            const unsigned long long gw = ge & wmask[0];
                            if (gw == 0) break;
                            cur = __builtin_amdgcn_readfirstlane(lov) + __builtin_ctzll(gw);
                            asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(outv) : "s"(cur), "s"(r) : "m0");   // lane <- cur (instead of move, compare, select)
                            if (MODE == 1 && cur == __builtin_amdgcn_readlane(oldv, r)) { rstop = r; break; }
                            if (--r < 0) break;
                        }
                        if (r < 0 || rstop >= 0) break;
                    }
                }
                int lo;
                float v[KC], av[KC];
                float m;
                bool fail;                                             // a row-constant candidate may tie or win: full evaluation
                unsigned long long ge = 0;                             // kLean: candidate lanes that attain the maximum
                if constexpr (kLean) {
                    // The scalar unit of a CU serves its sixteen waves and was ~60 % busy here (42 scalar instructions per frame,
                    // profiles/r03_pmc_B2048_full.txt): the path state goes through an opaque VGPR copy so that the window start,
                    // the fit check and the LDS indices are vector instructions, and ONE compare yields both the matching
                    // candidates and the bound: lane CB holds fl(M_t + c_cur), and fl(M_t + c_cur) >= max <=> not (mf < max).
                    int curv;
                    asm volatile("v_mov_b32 %0, %1" : "=v"(curv) : "s"(cur));
                    int lov = curv - a.lo_off;
                    lov = lov < 0 ? 0 : (lov > lo_max ? lo_max : lov);
                    const int wlov = lov + (a.col0 - c0);              // window start inside the span
                    if (__ballot((unsigned)wlov > (unsigned)(kSpNS - W))) { rmiss = r; ++n_miss; break; }
                    const float* trow = tile + r * kSpRS;
                    const float dv = trow[isw[0] ? wlov + lane : kSpNS + auxi[0]];
                    av[0] = tabX[curv * WX1 + tb[0]];
                    const float vv = dv + av[0];
                    v[0] = cand[0] ? vv : -INFINITY;
                    m = sp_wave_max(v[0]);
                    ge = __ballot(vv >= m) & cand_or_bound;
                    fail = (ge >> CB) & 1ull;
                    lo = __builtin_amdgcn_readfirstlane(lov);
                } else {
                    if (AFF) { lo = cur - a.lo_off; lo = lo < 0 ? 0 : (lo > lo_max ? lo_max : lo); }
                    else lo = __builtin_amdgcn_readfirstlane(loL[cur]);
                    const int wlo = a.col0 + lo - c0;                       // window start inside the span
                    if (wlo < 0 || wlo + W > kSpNS) { rmiss = r; ++n_miss; break; }
                    const float* trow = tile + r * kSpRS;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const float dv = trow[isw[k] ? wlo + 64 * k + lane : kSpNS + auxi[k]];
                        av[k] = GT ? gtab[(size_t)cur * WX1 + tb[k]] : tabX[cur * WX1 + tb[k]];
                        v[k] = dv + av[k];
                    }
                    float mf = 0.f;                                         // fl(M_t + c_cur), from the bound candidate
#pragma unroll
                    for (int k = 0; k < KC; ++k)
                        if (KC == 1 || k == kb) mf = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[k]), lb));
                    float mloc = -INFINITY;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        v[k] = cand[k] ? v[k] : -INFINITY;
                        mloc = fmaxf(mloc, v[k]);
                    }
                    m = sp_wave_max(mloc);
                    fail = !(mf < m);
                }
                auto lowest_candidate = [&](const float mm) -> unsigned {
                    unsigned best = 0x7fffffffu;
                    bool have_w = false;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const unsigned long long mk = (kLean && mm == m) ? (ge & ~(1ull << CB)) : __ballot(v[k] == mm && cand[k]);
                        const unsigned long long mw = mk & wmask[k];
                        if (mw && !have_w) {                                 // window candidates ascend with the source index
                            const unsigned c = lo + 64 * k + __builtin_ctzll(mw);
                            best = c < best ? c : best;
                            have_w = true;
                        }
                        unsigned long long mx = mk & ~wmask[k];              // extra columns: arbitrary indices
                        while (mx) {
                            const unsigned c = __builtin_amdgcn_readlane(xs[k], __builtin_ctzll(mx));
                            best = c < best ? c : best;
                            mx &= mx - 1;
                        }
                    }
                    return best;
                };
                unsigned idx = 0x7fffffffu;
                if (!fail) {
                    idx = lowest_candidate(m);
                } else {
                    float cj = 0.f;                                         // c_cur, from the bound candidate's table entry
#pragma unroll
                    for (int k = 0; k < KC; ++k)
                        if (KC == 1 || k == kb) cj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(av[k]), lb));
                    ++n_full;
                    // ---- full evaluation, straight from the history row in global memory: every source outside the
                    //      window / extra columns contributes fl(delta_t[i] + c_cur)
                    const float* __restrict__ grow = hist + (size_t)(first + r) * SD + a.col0;
                    float vf[EPL];
                    float m2 = -INFINITY;
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const int i = e * 64 + lane;
                        const float d = inS[e] ? grow[i] : -INFINITY;
                        const bool excl = xcol[e] || (unsigned)(i - lo) < (unsigned)W;
                        vf[e] = excl ? -INFINITY : d + cj;
                        m2 = fmaxf(m2, vf[e]);
                    }
                    const float mm = fmaxf(m, sp_wave_max(m2));
#pragma unroll
                    for (int e = 0; e < EPL; ++e) {
                        const unsigned long long mk = __ballot(vf[e] == mm && inS[e]);
                        if (mk) { const unsigned c = e * 64 + __builtin_ctzll(mk); idx = c < idx ? c : idx; }
                    }
                    const unsigned c = lowest_candidate(mm);
                    idx = c < idx ? c : idx;
                    if (idx == 0x7fffffffu) idx = 0;        // an all -inf frame resolves to index 0 like np.argmax
                }
                cur = (int)idx;
                outv = lane == r ? cur : outv;
                if (MODE == 1 && cur == __builtin_amdgcn_readlane(oldv, r)) { rstop = r; break; }
            }
            const int rkeep = rstop > rmiss ? rstop : rmiss;     // rows above rkeep were decided in this pass
            if (write && lane < rows && lane > rkeep) states[first + lane] = outv;
            if (MODE == 1) n_repf += rows - 1 - rkeep;
            if (MODE == 1 && rstop >= 0) return __builtin_amdgcn_readfirstlane(states[bottom]);   // the stored path continues unchanged
            top = rmiss >= 0 ? first + rmiss : first - 1;
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
            m = sp_wave_max(m);
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
        if (n_rep) atomicAdd(ct + kCtRepairs, n_rep);
        if (n_repf) atomicAdd(ct + kCtRepairFrames, n_repf);
    }
}

static int sparse_kc(const BtArgs& a) { return (a.W + kMaxExtras + 1 + 63) / 64; }

// LDS bytes: tiles + window starts (+ the candidate table when it fits next to at least four waves' tiles)
static size_t sparse_lds_bytes(const BtArgs& a, int nwaves, bool table_in_lds) {
    const int rs = sp_span(sparse_kc(a)) + kSpAux;
    return sizeof(float) * ((size_t)nwaves * kSpK * rs + (table_in_lds ? (size_t)a.SP * (a.W + kMaxExtras + 1) : 0)) + sizeof(int32_t) * a.SP;
}
static bool sparse_table_fits(const BtArgs& a) { return sparse_lds_bytes(a, 4, true) + 1024 <= 160 * 1024; }

// The sparse kernel takes banded plans without dense rows whose forward pass left the frame maximum in the history,
// with at most three candidates per lane and the span inside a row.
bool sparse_backtrace_applies(const BtArgs& a) {
    const int kc = sparse_kc(a);
    return a.banded && a.have_fmax && a.n_dense == 0 && kc <= 3 && a.W + 32 <= sp_span(kc) && a.SD >= sp_span(kc) &&
           a.SD % 4 == 0 && (a.S + 63) / 64 <= 12;
}

template <int NWT, bool AFF, int KC, bool GT>
static hipError_t launch_sparse_t(const BtArgs& a, hipStream_t st, int phases) {
    int nw = 16;
    while (nw > 4 && sparse_lds_bytes(a, nw, !GT) + 1024 > 160 * 1024) nw >>= 1;
    const size_t lds = sparse_lds_bytes(a, nw, !GT);
    const long long waves0 = (long long)a.B * a.chunks;
    hipError_t e = hipSuccess;
    if (phases & 1) {
        hipLaunchKernelGGL((sparse_backtrace_kernel<NWT, AFF, 0, KC, GT>), dim3((int)((waves0 + nw - 1) / nw)), dim3(nw * 64), lds, st, a);
        e = hipGetLastError();
    }
    if (e != hipSuccess || a.chunks <= 1 || !(phases & 2)) return e;
    hipLaunchKernelGGL((sparse_backtrace_kernel<NWT, AFF, 1, KC, GT>), dim3((int)((a.B + nw - 1) / nw)), dim3(nw * 64), lds, st, a);
    return hipGetLastError();
}

template <int NWT>
static hipError_t launch_sparse_a(const BtArgs& a, hipStream_t st, int phases) {
    const bool gt = !sparse_table_fits(a);
    if (sparse_kc(a) == 1) {
        if (gt) return a.lo_affine ? launch_sparse_t<NWT, true, 1, true>(a, st, phases) : launch_sparse_t<NWT, false, 1, true>(a, st, phases);
        return a.lo_affine ? launch_sparse_t<NWT, true, 1, false>(a, st, phases) : launch_sparse_t<NWT, false, 1, false>(a, st, phases);
    }
    if (sparse_kc(a) == 2) {
        if (gt) return a.lo_affine ? launch_sparse_t<NWT, true, 2, true>(a, st, phases) : launch_sparse_t<NWT, false, 2, true>(a, st, phases);
        return a.lo_affine ? launch_sparse_t<NWT, true, 2, false>(a, st, phases) : launch_sparse_t<NWT, false, 2, false>(a, st, phases);
    }
    if (gt) return a.lo_affine ? launch_sparse_t<NWT, true, 3, true>(a, st, phases) : launch_sparse_t<NWT, false, 3, true>(a, st, phases);
    return a.lo_affine ? launch_sparse_t<NWT, true, 3, false>(a, st, phases) : launch_sparse_t<NWT, false, 3, false>(a, st, phases);
}

hipError_t launch_backtrace_sparse(const BtArgs& a, hipStream_t st, int phases) {
    const int nwt = (a.S + 63) / 64;
    if (nwt <= 2) return launch_sparse_a<2>(a, st, phases);
    if (nwt <= 4) return launch_sparse_a<4>(a, st, phases);
    if (nwt <= 6) return launch_sparse_a<6>(a, st, phases);
    if (nwt <= 8) return launch_sparse_a<8>(a, st, phases);
    return launch_sparse_a<12>(a, st, phases);
}

// more, shorter chunks than the whole-row kernels: the sparse kernel hides its fetch latency with waves, not with a
// second tile in registers (sixteen waves per CU at 1024 songs and up)
int sparse_backtrace_chunks(int64_t B, int T, int n_cus) {
    // (song, chunk) waves up to the resident capacity of the chip (sixteen per CU), never beyond: one wave more starts a second round
    // and doubles the kernel's time (B = 320: thirteen chunks 2.1 ms, twelve 1.2 ms)
    long long c = (16ll * (n_cus > 0 ? n_cus : 256)) / (B > 0 ? B : 1);      // sixteen waves per CU
    const long long cmax = T / (8 * kBtWarmSparse) > 1 ? T / (8 * kBtWarmSparse) : 1;
    c = c > cmax ? cmax : c;
    c = c > kBtMaxChunks ? kBtMaxChunks : c;
    return c < 1 ? 1 : (int)c;
}

}  // namespace vit
