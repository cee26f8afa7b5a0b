// wave.hip -- "wave" form of the banded forward pass: ONE SONG PER WAVEFRONT (gfx950).
//
// The workgroup kernels of kernels.hip advance a song one frame per LDS round trip and barrier: ~830 cycles per
// frame whatever the arithmetic, which is what a small batch needs (latency) and what a large batch does not
// (throughput: four songs per CU still leave the VALU ~60 % idle).  Here a song never leaves one wavefront:
//
//   * lane l owns NPL = ceil(S/64) CONTIGUOUS states (right-aligned: slot NPL*l + k holds state NPL*l + k - (64*NPL - S));
//     delta lives in registers;
//   * the plan proved that every exception span lies within D sources of its target, so a lane needs delta only from
//     the H = ceil(D/NPL) lanes on either side: 2*H wave-wide DPP shifts of its NPL registers (wave_shr:1 / wave_shl:1; a
//     lane beyond the wave edge delivers 0, and the weights of sources that do not exist are -inf) -- no LDS, no
//     barrier, no other wave;
//   * own state k evaluates the 2*D+1 sources j-D .. j+D as D+1 even-aligned source pairs: one v_pk_add_f32 and one
//     v_max3_f32 per pair, weights register-resident (the true matrix entries: positions outside a row's exception span
//     carry the row constant, a candidate the dense recursion forms as well);
//   * everything else is the floor-max identity of banded_floor_forward_kernel with M taken over ALL sources
//     (plan.floor_all_ok: extra-column entries dominate the row constant too, so an extra column inside M only adds a
//     dominated candidate):  m_j = max( window candidates, fl(M + c_j), fl(delta_x + logA_T[j][x]) for extra columns x ).
//
// Every value compared is one the dense recursion forms, so delta is bit-identical (CPU replay: tests/plan_replay.py
// replay_wave).  Per frame and wave 253 VALU instructions (272 in all) and nothing to wait for but the emission prefetch;
// songs of different lengths simply finish at different times.  The history rows are written in slot order (row stride
// 64*NPL floats; layout at the kernel below) and vit_forward records that layout for the back-trace kernels.
// Measured (S = 361, fp32, T = 30000): 19.9 ms for 1024 songs, 35.7 ms for 2048 -- HBM-bound at 4.5-5 TB/s of real
// traffic.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "kernels.hpp"

namespace vit {

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
// element-aligned vector views: a lane's NPL columns start on a 4-byte (f32) / 2-byte (f16) boundary only
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef _Float16 f16x2_u __attribute__((ext_vector_type(2), aligned(2)));
typedef _Float16 f16x4_u __attribute__((ext_vector_type(4), aligned(2)));

constexpr int kBigI = 0x7fffffff;

__device__ __forceinline__ int song_length_of(const int64_t* lengths, int song, int T) {
    if (!lengths) return T;
    long long v = lengths[song];
    v = v < 1 ? 1 : v;
    return v > T ? T : (int)v;
}

// Wave-wide shifts by one lane.  bound_ctrl: the lane without a source reads 0 -- any finite value would do: the
// sources that lane stands for do not exist (state < 0 or >= S) and their window weights are -inf (plan.cpp), so the
// candidate is -inf whatever the shift delivers.  (A fill value would cost a v_mov per shift.)
__device__ __forceinline__ float dpp_shr1(float x) {   // lane l <- x[l-1]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_shl1(float x) {   // lane l <- x[l+1]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x130, 0xf, 0xf, true));
}
// max over the wave, wave-uniform result (six v_max_f32 with a DPP operand + v_readlane; see kernels.hip wave_scan_max)
__device__ __forceinline__ float wave_max(float x) {
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


template <int NPL, typename ET>
struct RowIO;

// float32 emissions: NPL consecutive columns as 16- and 8-byte pieces
template <int NPL>
struct RowIO<NPL, float> {
    static __device__ __forceinline__ void load(const float* __restrict__ p, float (&e)[NPL]) {
        int k = 0;
#pragma unroll
        for (; k + 3 < NPL; k += 4) { const f32x4_u v = *reinterpret_cast<const f32x4_u*>(p + k); e[k] = v.x; e[k + 1] = v.y; e[k + 2] = v.z; e[k + 3] = v.w; }
#pragma unroll
        for (; k + 1 < NPL; k += 2) { const f32x2_u v = *reinterpret_cast<const f32x2_u*>(p + k); e[k] = v.x; e[k + 1] = v.y; }
        if (k < NPL) e[k] = p[k];
    }
    static __device__ __forceinline__ float load1(const float* __restrict__ p) { return *p; }
};
template <int NPL>
struct RowIO<NPL, __half> {
    static __device__ __forceinline__ void load(const __half* __restrict__ ph, float (&e)[NPL]) {
        const _Float16* p = reinterpret_cast<const _Float16*>(ph);
        int k = 0;
#pragma unroll
        for (; k + 3 < NPL; k += 4) { const f16x4_u v = *reinterpret_cast<const f16x4_u*>(p + k); e[k] = (float)v.x; e[k + 1] = (float)v.y; e[k + 2] = (float)v.z; e[k + 3] = (float)v.w; }
#pragma unroll
        for (; k + 1 < NPL; k += 2) { const f16x2_u v = *reinterpret_cast<const f16x2_u*>(p + k); e[k] = (float)v.x; e[k + 1] = (float)v.y; }
        if (k < NPL) e[k] = (float)p[k];
    }
    static __device__ __forceinline__ float load1(const __half* __restrict__ p) { return __half2float(*p); }
};

template <int NPL>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&d)[NPL]) {
    int k = 0;
#pragma unroll
    for (; k + 3 < NPL; k += 4) { f32x4_u v; v.x = d[k]; v.y = d[k + 1]; v.z = d[k + 2]; v.w = d[k + 3]; *reinterpret_cast<f32x4_u*>(p + k) = v; }
#pragma unroll
    for (; k + 1 < NPL; k += 2) { f32x2_u v; v.x = d[k]; v.y = d[k + 1]; *reinterpret_cast<f32x2_u*>(p + k) = v; }
    if (k < NPL) p[k] = d[k];
}

}  // namespace

// NPL states per lane, D window half-width, NX extra columns, PF emission rows in flight, WPS minimum waves per SIMD the
// register budget must allow.
//
// Lane <-> state mapping, RIGHT-aligned: slot q = NPL*lane + k (k = 0 .. NPL-1) holds state q - o with o = 64*NPL - S,
// so lane 63 ends exactly at state S-1 and the o leading slots are idle (weights -inf: their delta stays -inf).  With
// that, a lane's NPL emission columns are one unconditional vector load at E_row + NPL*lane - o -- for the leading
// lanes that reaches back into the previous row of the SAME tensor (rows >= 1 only: row 0 is loaded element-wise), never
// out of bounds -- and the history row is stored in slot order: row stride 64*NPL floats, state i in column o + i, and
// M_t = max_i delta_t[i] in column 0, a copy of delta_t of extra column x in column 1 + x (idle slots) -- and, with one extra column and six idle
// slots (A3), the same two scalars of frames t-1 and t-2 in columns 2 3 | 4 5: they are still in scalar registers, four more selects per frame,
// and the back-trace kernels then find the scalars of three frames in ONE line (the carrier row t - t % 3 + 2; B = 2048: -8 % / -13 %).  No branch surrounds a memory instruction, so the in-order vmcnt
// of the emission prefetch is exact.  The back-trace is told the column offset and the column of M (BtArgs::col0, mcol).
//
// HM (history mode).  0: every delta row is stored (row t of a song at hist + t * 64*NPL).  1: only the rows of EVEN frames are
// stored (row t/2); lane 0's idle slots of row t carry, behind M_t and delta_t of the extra columns, the same scalars of the odd
// frame t-1 -- they are still in scalar registers when row t is stored, so the odd frames cost no store at all.  The back-trace
// (backtrace_half.hip) rebuilds the 32 delta values of an odd frame that it needs from the stored row before it and the
// emission row, with the very sums and maxima of this kernel's recursion: 768 instead of 1536 history bytes per frame.
// 5: checkpoints only (vit_decode_checkpointed, pass 1): row (t + 1) / K - 1 for the frames t with (t + 1) % K == 0 -- the row
// in front of every segment of K frames; every other frame's store goes to one scratch row per song (the same address over
// and over: it stays in L2), so that no branch surrounds a store.
//
// 6: a segment (vit_decode_checkpointed pass 2; a mode of its own so that the t_begin arithmetic stays out of HM 0 -- folded into
// HM 0 it cost the full-history kernel 25 %: 19.9 -> 25.1 ms at B = 1024): every row like HM 0, but FwdArgs::t_begin > 0 resumes from init_rows[song] = delta_{t_begin - 1} in
// slot order (a checkpoint row; lane 0's scalar slots are idle slots and are reset to -inf), computes frames t_begin ..
// min(t_end, T_b) - 1 and stores row t at t - t_begin; the terminal state is pass 1's business.
//
// UV >= 1 (one extra column and it is the last state, S - 1 = slot 64*NPL - 1 whatever S): delta of the extra column is a plain
// v_readlane of lane 63's last slot instead of a select chain over the lane's slots (which the compiler turned into an LDS round
// trip per frame).  UV = 2 (plan-proven, FwdArgs::wave_u5; NPL = 6) in addition: the row constant and the extra-column weight are the
// same for a lane's slots 0..4 (the 360 voiced targets of the reference's matrices: log tiny, log(sw10 / n_bins); idle slots: -inf) and
// only slot 5 -- lane 63's unvoiced target, lane 3's first state at S = 361 -- has its own.  Then fl(M + c), fl(delta_x + a_x) and
// their maximum are formed ONCE per lane and once for slot 5 (4 adds + 2 max instead of 12 adds, and six two-operand maxima
// instead of six max3), and delta of the extra column is a plain v_readlane of lane 63's slot 5.  UV = 3: the same with three groups of
// slots, {0,1,2} {3,4} {5} -- S = 321 (msnet / dcnet / ftanet), whose idle slots end in the middle of lane 10 (6 adds + 3 max).
template <int NPL, int D, int NX, int PF, int WPS, int HM, typename ET, int UV = 0>
__global__ void __launch_bounds__(256, WPS) wave_forward_kernel(FwdArgs a) {
    constexpr int H = wave_halo(NPL, D);
    constexpr int NG = 2 * H + 1;              // lane groups of the neighbourhood
    constexpr int NPM = wave_pairs(D);
    constexpr int SDW = 64 * NPL;              // history row stride of this form
    static_assert(NPL <= 8 && NPL % 2 == 0 && NX <= kWaveMaxExtras && NX + 1 <= NPL && PF >= 1, "geometry (source pairs never straddle two lanes)");
    static_assert(HM != 1 || 2 * (NX + 1) <= NPL, "half history: lane 0 carries the scalars of two frames");
    static_assert(UV == 0 || NX == 1, "last-state / uniform-lane forms: one extra column");
    static_assert(UV < 2 || NPL == 6, "uniform-lane forms: six states per lane");
    constexpr bool U5 = UV == 2, U3 = UV == 3;
    constexpr bool PK = HM == 7;               // packed batch: this wave is a SLOT that walks a list of songs back to back
    // every row stored, one extra column: row t carries the scalars of frames t, t-1 and t-2 (columns 0 1 | 2 3 | 4 5 of lane 0), so that
    // the back-trace finds the scalars of three frames in ONE line (kernels.hpp wave_aux_frames / wave_aux_row)
    constexpr bool A3 = (HM == 0 || HM == 6 || HM == 7) && NX == 1 && NPL >= 6;      // (and six idle slots: run time, l0a below)
    const int S = a.S;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (wid >= (PK ? a.n_slots : a.B)) return; // whole waves only; there is no barrier in this kernel
    // ---------------- per-lane constants
    const int o = SDW - S;                                 // idle leading slots (>= 1)
    const int j0 = NPL * lane - o;                         // state of slot 0 of this lane (negative: idle)
    const bool l0a = A3 && lane == 0 && wave_aux_frames(NPL, S, NX) == 3 && !(a.wave_flags & 4);   // this lane's slots 2 .. 5 carry the scalars of frames t-1, t-2
    f32x2 aw[NPL][NPM];
    {
        const float* __restrict__ tv = reinterpret_cast<const float*>(a.image + a.off_tabV);
#pragma unroll
        for (int k = 0; k < NPL; ++k)
#pragma unroll
            for (int m = 0; m < NPM; ++m) {
                aw[k][m].x = tv[(((size_t)k * NPM + m) * 2 + 0) * 64 + lane];
                aw[k][m].y = tv[(((size_t)k * NPM + m) * 2 + 1) * 64 + lane];
            }
    }
    float cj[NPL];
    float xa[NX > 0 ? NX : 1][NPL];
    int xl[NX > 0 ? NX : 1];                               // lane that owns extra column x
    bool xs[NX > 0 ? NX : 1][NPL];                         // this lane's slot k holds extra column x
    {
        const float* __restrict__ rc = reinterpret_cast<const float*>(a.image + a.off_rowc);
        const float* __restrict__ xaT = reinterpret_cast<const float*>(a.image + a.off_extraA);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int j = j0 + k;
            cj[k] = j >= 0 ? rc[j] : -INFINITY;
#pragma unroll
            for (int x = 0; x < NX; ++x) xa[x][k] = j >= 0 ? xaT[(size_t)x * a.SP + j] : -INFINITY;
        }
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            xl[x] = (a.extras[x] + o) / NPL;
#pragma unroll
            for (int k = 0; k < NPL; ++k) xs[x][k] = j0 + k == a.extras[x];
        }
    }
    const int k_begin = PK ? a.slot_begin[wid] : 0, k_end = PK ? a.slot_begin[wid + 1] : 1;
    for (int kk = k_begin; kk < k_end; ++kk) {
    // ---------------- this song: emission rows, history rows, length
    const int song = PK ? a.slot_songs[kk] : wid;
    const long long off = PK ? a.offsets[song] : (long long)song * a.T;           // first emission row of the song in the tensor
    const int T = PK ? (int)(a.offsets[song + 1] - off) : a.T;                     // rows the song owns (packed: its length)
    const int t0 = HM == 6 ? a.t_begin : 0;                                    // first frame of this launch
    const int Tl = PK ? T : song_length_of(a.lengths, song, T);
    const int Tb = HM == 6 && a.t_end < Tl ? a.t_end : Tl;                     // one past the last frame of this launch
    if (Tb <= t0) continue;                                                    // (segments: the song ended before this one)
    const ET* __restrict__ E = reinterpret_cast<const ET*>(a.logE) + (size_t)off * S;
    float* __restrict__ hist = a.hist + (PK ? (size_t)off : (size_t)song * a.hist_rows) * SDW;     // hist_rows = T (HM 0), (T + 1) / 2 (HM 1), segments (HM 5), K + 1 (HM 6)
    // emission columns of this lane in rows >= 1 (see above): the leading lanes reach back into the row in front -- of the same song,
    // or (packed, row 0 is never loaded this way) of the song before it in the tensor; the first row of the tensor has none
    const long ecol = (T > 1 || off > 0) ? (long)j0 : (long)(j0 < 0 ? 0 : j0);
    const int row_min = T > 1 ? 1 : 0;
    auto load_row = [&](int row, float (&e)[NPL]) {
        row = row < row_min ? row_min : row;
        RowIO<NPL, ET>::load(E + (size_t)row * S + ecol, e);
    };
    // history row t in slot order; lane 0's leading slots (always idle: o > NX is checked by the plan) carry M_t and a
    // copy of delta_t of the extra columns, so that the sparse back-trace finds its per-row scalars in ONE cache line
    // (HM 1: row t/2 of an even frame t; Mp / xp = the scalars of frame t-1, slots 1+NX .. 1+2*NX)
    auto store_hist = [&](const int t, const float (&d)[NPL], const float M, const float (&xd)[NX > 0 ? NX : 1], const float Mp,
                          const float (&xp)[NX > 0 ? NX : 1], const float Mq, const float xq) {
        float v[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) v[k] = d[k];
        v[0] = lane == 0 ? M : v[0];
#pragma unroll
        for (int x = 0; x < NX; ++x) v[1 + x] = lane == 0 ? xd[x] : v[1 + x];
        if (A3) {
            v[2] = l0a ? Mp : v[2];
            v[3] = l0a ? xp[0] : v[3];
            v[4] = l0a ? Mq : v[4];
            v[5] = l0a ? xq : v[5];
        }
        if (HM == 1) {
            v[1 + NX] = lane == 0 ? Mp : v[1 + NX];
#pragma unroll
            for (int x = 0; x < NX; ++x) v[2 + NX + x] = lane == 0 ? xp[x] : v[2 + NX + x];
        }
        size_t row = HM == 1 ? t >> 1 : t - t0;
        if (HM == 5) {
            const int q = (t + 1) / a.ckpt_every;
            row = (t + 1) - q * a.ckpt_every == 0 && q - 1 < a.hist_rows - 1 ? q - 1 : a.hist_rows - 1;     // checkpoint, else the scratch row
        }
        store_row<NPL>(hist + row * SDW + NPL * lane, v);
    };
    // delta of the extra columns, wave-uniform
    auto extra_deltas = [&](const float (&d)[NPL], float (&xd)[NX > 0 ? NX : 1]) {
        if (UV >= 1) { xd[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d[NPL - 1]), 63)); return; }   // state S-1 = lane 63, last slot
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            float v = d[0];
#pragma unroll
            for (int k = 1; k < NPL; ++k) v = xs[x][k] ? d[k] : v;   // per-lane masks: one v_cndmask each
            xd[x] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), xl[x]));
        }
    };

    // ---------------- frame 0, or the checkpoint row in front of this segment
    float d[NPL];
    if (t0 == 0) {
        const float* __restrict__ lpi = reinterpret_cast<const float*>(a.image + a.off_logpi);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            const int j = j0 + k;
            d[k] = j >= 0 ? lpi[j] + RowIO<NPL, ET>::load1(E + j) : -INFINITY;
        }
    } else {
        const float* __restrict__ ir = a.init_rows + (size_t)song * a.init_stride + NPL * lane;
#pragma unroll
        for (int k = 0; k < NPL; ++k) d[k] = j0 + k >= 0 ? ir[k] : -INFINITY;
    }
    auto frame_max = [&](const float (&v)[NPL]) -> float {
        float loc = v[0];
#pragma unroll
        for (int k = 1; k < NPL; ++k) loc = fmaxf(loc, v[k]);
        return wave_max(loc);
    };
    float M = frame_max(d);
    float xd[NX > 0 ? NX : 1] = {};
    extra_deltas(d, xd);
    if (t0 == 0) store_hist(0, d, M, xd, M, xd, M, xd[0]);
    float Mb = M, xb = xd[0];                  // (A3) the scalars of the frame before the previous one
    const int t1 = t0 == 0 ? 1 : t0;           // first frame the loop computes

    float er[PF][NPL];
#pragma unroll
    for (int q = 0; q < PF; ++q) load_row(t1 + q < Tb ? t1 + q : Tb - 1, er[q]);
#pragma unroll
    for (int k = 0; k < NPL; ++k)
#pragma unroll
        for (int m = 0; m < NPM; ++m) asm volatile("" ::"v"(aw[k][m]));

    auto frame = [&](const int t, float (&e)[NPL], auto stored) {
        // ---- neighbourhood: group g holds delta of lane l - H + g
        float nb[NG][NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) nb[H][k] = d[k];
#pragma unroll
        for (int s = 1; s <= H; ++s)
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                nb[H - s][k] = dpp_shr1(nb[H - s + 1][k]);
                nb[H + s][k] = dpp_shl1(nb[H + s - 1][k]);
            }
        auto NB = [&](const int p) -> float { return p < NG * NPL ? nb[p / NPL][p % NPL] : -INFINITY; };
        // ---- window candidates: D+1 packed adds and max3 per own state, walked source pair by source pair from the
        //      lane's own group outwards (the order the shifts deliver them) so that consecutive instructions belong to
        //      different states: independent chains, no wait state between a packed add and the max3 that reads it
        float acc[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) acc[k] = -INFINITY;
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
            const int g = gi == 0 ? H : (gi & 1 ? H - (gi + 1) / 2 : H + gi / 2);
#pragma unroll
            for (int pp = (g * NPL) / 2; 2 * pp < (g + 1) * NPL + 1; ++pp) {
                const int p = 2 * pp;
                if (p / NPL != g && !(NPL % 2 && (p + 1) / NPL == g && p / NPL == g - 1 && false)) continue;
                f32x2 c[NPL];
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const int p0 = wave_p0e(NPL, D, k);
                    if (p >= p0 && p < p0 + 2 * NPM) c[k] = f32x2{NB(p), NB(p + 1)} + aw[k][(p - p0) / 2];
                }
#pragma unroll
                for (int k = 0; k < NPL; ++k) {
                    const int p0 = wave_p0e(NPL, D, k);
                    if (p >= p0 && p < p0 + 2 * NPM) acc[k] = fmaxf(fmaxf(acc[k], c[k].x), c[k].y);
                }
            }
        }
        // ---- floor term, extra columns, emission
        if (U5) {
            const float ya = fmaxf(M + cj[0], xd[0] + xa[0][0]), yb = fmaxf(M + cj[NPL - 1], xd[0] + xa[0][NPL - 1]);
#pragma unroll
            for (int k = 0; k < NPL; ++k) d[k] = fmaxf(acc[k], k < NPL - 1 ? ya : yb) + e[k];
        } else if (U3) {
            const float ya = fmaxf(M + cj[0], xd[0] + xa[0][0]), yb = fmaxf(M + cj[3], xd[0] + xa[0][3]);
            const float yc = fmaxf(M + cj[NPL - 1], xd[0] + xa[0][NPL - 1]);
#pragma unroll
            for (int k = 0; k < NPL; ++k) d[k] = fmaxf(acc[k], k < 3 ? ya : (k < NPL - 1 ? yb : yc)) + e[k];
        } else {
#pragma unroll
            for (int k = 0; k < NPL; ++k) {
                float m = fmaxf(acc[k], M + cj[k]);
#pragma unroll
                for (int x = 0; x < NX; ++x) m = fmaxf(m, xd[x] + xa[x][k]);
                d[k] = m + e[k];
            }
        }
        const float Mp = M;                // the previous frame's scalars (wave-uniform: scalar registers)
        float xp[NX > 0 ? NX : 1];
#pragma unroll
        for (int x = 0; x < (NX > 0 ? NX : 1); ++x) xp[x] = xd[x];
        M = frame_max(d);
        extra_deltas(d, xd);               // for the next frame's candidates, and for the history row
        if (decltype(stored)::value && HM != 2 && HM != 4) store_hist(t, d, M, xd, Mp, xp, Mb, xb);
        if (A3) { Mb = Mp; xb = xp[0]; }
        if (HM != 3 && HM != 4) load_row(t + PF < Tb ? t + PF : Tb - 1, e);     // (HM 2 / 3 / 4: timing builds only -- no stores / no loads / neither)
    };
    // The loop body is a whole number of frame pairs when only even frames are stored: t is odd at its top, frame t + q is
    // even for odd q, and "store or not" is a compile-time property of each unrolled frame (no branch around a store).
    constexpr int UN = (HM == 1 && (PF & 1)) ? 2 * PF : PF;
    int t = t1;
    for (; t + UN - 1 < Tb; t += UN) {
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            if (HM != 1 || (q & 1)) frame(t + q, er[q % PF], std::true_type{});
            else frame(t + q, er[q % PF], std::false_type{});
        }
    }
#pragma unroll
    for (int q = 0; q < UN - 1; ++q)
        if (t + q < Tb) {
            if (HM != 1 || (q & 1)) frame(t + q, er[q % PF], std::true_type{});
            else frame(t + q, er[q % PF], std::false_type{});
        }

    // ---------------- terminal state: lowest-index argmax of delta_{Tb-1} (not in a segment launch)
    if (HM != 6 || a.t_end >= T) {
        float bv = -INFINITY;
        int bi = kBigI;
#pragma unroll
        for (int k = 0; k < NPL; ++k)
            if (j0 + k >= 0 && (d[k] > bv || bi == kBigI)) { bv = d[k]; bi = j0 + k; }   // first state of the lane, then strictly greater
        // lanes ascend with the state index: an ordered (value, index) scan keeps the first maximum
#define VIT_WSTEP(CTRL, MASK)                                                                                      \
    {                                                                                                              \
        const float sv = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(bv), CTRL, MASK, 0xf, false)); \
        const int si = __builtin_amdgcn_update_dpp(kBigI, bi, CTRL, MASK, 0xf, false);                               \
        const bool keep_earlier = !(bv > sv);   /* kernels.hip op_fwd: the later piece wins only if strictly greater */ \
        bv = keep_earlier ? sv : bv;                                                                               \
        bi = keep_earlier ? si : bi;                                                                               \
    }
        VIT_WSTEP(0x111, 0xf)
        VIT_WSTEP(0x112, 0xf)
        VIT_WSTEP(0x114, 0xf)
        VIT_WSTEP(0x118, 0xf)
        VIT_WSTEP(0x142, 0xa)
        VIT_WSTEP(0x143, 0xc)
#undef VIT_WSTEP
        if (lane == 63) {
            a.last_state[song] = bi == kBigI ? 0 : bi;
            if (a.loglik) a.loglik[song] = bv;
        }
    }
    }   // songs of this slot
}

template <int NPL, int D, int NX, typename ET>
static hipError_t launch_wave_x(const FwdArgs& a, hipStream_t st) {
    const int grid = (int)((a.B + 3) / 4);
    // Two instantiations: 512 registers (one wave per SIMD, PF1 emission rows in flight) and 256 registers (two waves per
    // SIMD, PF2 rows).  Round 2 took the first one up to 1024 songs (full history, B = 1024: PF 2 / 3 / 4 / 6 / 8 -> 24.6 /
    // 21.4 / 19.9 / 21.5 / 33.8 ms).  With the half history the kernel no longer waits for memory and the 256-register
    // code is the faster one even with a single wave on each SIMD (B = 1024, gpurun_out/r3a/wave_ablate.log: 512-register
    // form full / half history 19.9 / 21.3 ms, 256-register form 19.7 / 16.5 ms; without any load or store 16.0 ms), so it
    // runs at every batch size unless a second extra column pushes it into scratch.
    constexpr int PF1 = NX == 2 ? 3 : 4, PF2 = NX == 2 ? 2 : 3;
    const bool one = a.B <= 1024 && !(a.wave_flags & 1) && ((a.wave_flags & 2) || NX == 2);
#ifdef VIT_TIMING_HOOKS
    // result-breaking ablations (make TIMING=1 only): bits 0 / 1 of the timing mask drop the history stores / the emission loads
    if (a.debug & 3) {
        const int hm = (a.debug & 3) + 1;
        if (hm == 2) { if (one) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 2, ET>), dim3(grid), dim3(256), 0, st, a);
                       else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 2, ET>), dim3(grid), dim3(256), 0, st, a); }
        else if (hm == 3) { if (one) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 3, ET>), dim3(grid), dim3(256), 0, st, a);
                            else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 3, ET>), dim3(grid), dim3(256), 0, st, a); }
        else { if (one) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 4, ET>), dim3(grid), dim3(256), 0, st, a);
               else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 4, ET>), dim3(grid), dim3(256), 0, st, a); }
        return hipGetLastError();
    }
#endif
    if (a.offsets) {            // packed batch (vit_decode_packed): one wave per slot, full history, rows at the songs' offsets
        const int pgrid = (a.n_slots + 3) / 4;
        if (a.n_slots <= 1024 && ((a.wave_flags & 2) || NX == 2)) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 7, ET>), dim3(pgrid), dim3(256), 0, st, a);
        else if (NX == 1 && NPL == 6 && a.wave_u5 == 2) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 7, ET, (NX == 1 && NPL == 6) ? 2 : 0>), dim3(pgrid), dim3(256), 0, st, a);
        else if (NX == 1 && NPL == 6 && a.wave_u5 == 3) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 7, ET, (NX == 1 && NPL == 6) ? 3 : 0>), dim3(pgrid), dim3(256), 0, st, a);
        else if (NX == 1 && a.wave_u5 >= 1) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 7, ET, NX == 1 ? 1 : 0>), dim3(pgrid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 7, ET>), dim3(pgrid), dim3(256), 0, st, a);
        return hipGetLastError();
    }
    if (a.ckpt_every > 0) {
        if (one) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 5, ET>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 5, ET>), dim3(grid), dim3(256), 0, st, a);
    } else if (a.t_begin > 0 || a.t_end < a.T) {
        if (one) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 6, ET>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 6, ET>), dim3(grid), dim3(256), 0, st, a);
    } else if (a.hist_half) {
        if (one) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 1, ET>), dim3(grid), dim3(256), 0, st, a);
        else if (NX == 1 && NPL == 6 && a.wave_u5 == 2) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 1, ET, (NX == 1 && NPL == 6) ? 2 : 0>), dim3(grid), dim3(256), 0, st, a);
        else if (NX == 1 && NPL == 6 && a.wave_u5 == 3) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 1, ET, (NX == 1 && NPL == 6) ? 3 : 0>), dim3(grid), dim3(256), 0, st, a);
        else if (NX == 1 && a.wave_u5 >= 1) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 1, ET, NX == 1 ? 1 : 0>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 1, ET>), dim3(grid), dim3(256), 0, st, a);
    } else {
        if (one) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF1, 1, 0, ET>), dim3(grid), dim3(256), 0, st, a);
        else if (NX == 1 && NPL == 6 && a.wave_u5 == 2) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 0, ET, (NX == 1 && NPL == 6) ? 2 : 0>), dim3(grid), dim3(256), 0, st, a);
        else if (NX == 1 && NPL == 6 && a.wave_u5 == 3) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 0, ET, (NX == 1 && NPL == 6) ? 3 : 0>), dim3(grid), dim3(256), 0, st, a);
        else if (NX == 1 && a.wave_u5 >= 1) hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 0, ET, NX == 1 ? 1 : 0>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((wave_forward_kernel<NPL, D, NX, PF2, 2, 0, ET>), dim3(grid), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

template <int NPL, int D, typename ET>
static hipError_t launch_wave_n(const FwdArgs& a, hipStream_t st) {
    switch (a.n_extras) {
        case 0: return launch_wave_x<NPL, D, 0, ET>(a, st);
        case 1: return launch_wave_x<NPL, D, 1, ET>(a, st);
        case 2: return launch_wave_x<NPL, D, 2, ET>(a, st);
        default: return hipErrorInvalidConfiguration;
    }
}

template <typename ET>
static hipError_t launch_wave_e(const FwdArgs& a, hipStream_t st) {
    if (a.wave_dk != 14) return hipErrorInvalidConfiguration;
    switch (a.wave_npl) {
        case 6: return launch_wave_n<6, 14, ET>(a, st);
        default: return hipErrorInvalidConfiguration;
    }
}

__global__ void segment_prep_kernel(const int64_t* __restrict__ lengths, int64_t B, int T, int s0, int e0, const int32_t* __restrict__ states,
                                    const int32_t* __restrict__ last, int64_t* __restrict__ seg_len, int32_t* __restrict__ seg_last) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int Tb = song_length_of(lengths, (int)b, T);
    if (Tb > e0) {                    // the song goes on behind this segment: its state at frame e0 is decided already
        seg_len[b] = e0 - s0 + 1;
        seg_last[b] = states[(size_t)b * T + e0];
    } else if (Tb > s0) {             // the song ends inside this segment: pass 1 left its terminal state
        seg_len[b] = Tb - s0;
        seg_last[b] = last[b];
    } else {
        seg_len[b] = 0;
        seg_last[b] = 0;
    }
}

hipError_t launch_segment_prep(const int64_t* lengths, int64_t B, int T, int s0, int e0, const int32_t* states, const int32_t* last,
                               int64_t* seg_len, int32_t* seg_last, hipStream_t st) {
    hipLaunchKernelGGL(segment_prep_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, lengths, B, T, s0, e0, states, last, seg_len, seg_last);
    return hipGetLastError();
}

hipError_t launch_wave(const FwdArgs& a, bool f16, hipStream_t st) {
    if (!a.wave_ok) return hipErrorInvalidConfiguration;
    return f16 ? launch_wave_e<__half>(a, st) : launch_wave_e<float>(a, st);
}

}  // namespace vit
