// backtrace_lane.hip -- exact, time-parallel back-trace with ONE (song, chunk) STREAM PER LANE.
//
// The sparse kernels (backtrace_sparse.hip) give a whole wavefront to one stream: 32 + 2 candidates on 34 lanes, six DPP steps
// for the maximum, a ballot, scalar bookkeeping -- 57 instructions per decision, and with sixteen waves per CU the back-trace
// is bound by instruction issue (5.7 ms for 2048 songs x 30000 frames).  A decision is tiny, though: 31
// sums and comparisons.  Here every LANE chases a stream of its own and evaluates its candidates one after the other:
//
//   lo      = first source of the window of the path state j at t+1
//   m, arg  = running (max, lowest index) over fl(delta_t[lo + w] + logA_T[j][lo + w]), w ascending, strict >
//             then the extra columns (an equal value wins only with a lower index)
//   bound   = fl(M_t + c_j): if bound < m no row-constant candidate can tie or win and arg is psi_{t+1}[j]
//
// -- ~130 wave instructions for 64 decisions instead of 57 for one.  A lane reads exactly what its decision needs: W floats
// of its row at the window (4-byte-aligned 16-byte loads), the frame maximum and delta of the extra columns: ~180 B per
// frame where the sparse kernel's tiles fetch 480.  The candidate weights come from the per-target table in LDS (row stride
// W + 5 floats: odd, so lanes with different path states hit different banks).  Bound failures (rare: the frame of a
// voiced -> unvoiced switch, whose best source can be any voiced state) are evaluated by the whole wave for the lane that
// hit one: 64 lanes x ceil(S / 64) sources straight from the history row, the sparse kernel's full evaluation.
//
// Parallelism comes from chunks: every song is cut into C chunks (up to 256) chased speculatively from a warm-up point
// above their upper boundary (lane_spec_kernel); lane_verify_kernel compares what every chunk assumed at its upper boundary
// with what the chunk above it decided there, and lane_repair_kernel (one lane per song) re-chases the chunks whose guess was
// wrong until the new path meets the stored one.  Exact whatever the guesses were (the scheme of banded_backtrace_kernel).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "kernels.hpp"

namespace vit {

namespace {

typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2_u __attribute__((ext_vector_type(2), aligned(4)));
typedef int i32x4_u __attribute__((ext_vector_type(4), aligned(4)));

constexpr int kLnBig = 0x7fffffff;
constexpr int kLnThreads = 512;

__device__ __forceinline__ int ln_song_length(const BtArgs& a, int song) {
    if (!a.lengths) return a.T;
    long long v = a.lengths[song];
    v = v < 1 ? 1 : v;
    return v > a.T ? a.T : (int)v;
}
__device__ __forceinline__ float ln_wave_max(float x) {   // kernels.hip wave_max_all
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
__device__ __forceinline__ const float* ln_readlane_ptr(const float* p, int l) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)v, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void ln_chunk_bounds(int Lf, int c, int C, int& lo_c, int& hi_c) {
    lo_c = (int)((long long)Lf * c / C);
    hi_c = (int)((long long)Lf * (c + 1) / C);
}

// The decision of one frame for every lane of the wave: lane l holds the path state `cur` at frame t+1 of its stream and the
// address `row` of delta row t; step() returns true with psi_{t+1}[cur] when the lane's frame was decided in this call.  Every
// lane of the wave must call it (lanes without a stream pass any valid row and act = false).
//
// Bound failures are evaluated by the whole wave, one call LATER: the lane is marked pending (its caller keeps t and cur), and
// the next call loads the pending lanes' whole rows (up to two lanes, ceil(S / 64) coalesced loads each) TOGETHER with the other
// lanes' window loads, so that a failure costs its lane one call and the wave no memory round trip of its own (evaluated in
// place, with the rolled loops of the first version, a failure cost ~7 us and there is about one per wave and step).
template <int WQ, bool GT, int NWT>
struct LaneDecider {
    const BtArgs& a;
    const float* tabL;        // LDS copy of the per-target candidate table (!GT)
    const int32_t* loL;       // LDS copy of the window starts
    int lane;
    unsigned inb = 0;         // bit e: source 64 e + lane exists and is not an extra column
    int n_full = 0;           // bound failures of this lane's stream
    bool pend = false;        // this lane's frame waits for the full evaluation
    float m_p = 0.f;          // the pending frame's maximum over the window / extra-column candidates ...
    int arg_p = 0;            // ... and the lowest index that attains it
    // BtArgs::aux_frames == 3: a carrier row holds the scalars of three frames (kernels.hpp wave_aux_row); the lane keeps the six
    // values of the carrier row it last read, so that the scalar line is fetched once per three frames
    float q[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int qrow = -1;

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int e = 0; e < NWT; ++e) {
            const int i = 64 * e + lane;
            bool in = i < a.S;
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k) in &= !(k < a.n_extras && i == a.extras[k]);
            inb |= in ? 1u << e : 0u;
        }
    }

    // every source outside the window / the extra columns contributes fl(delta_t[i] + c_cur): lowest index attaining the maximum
    __device__ __forceinline__ int full_eval(const float (&d)[NWT], const int lo_l, const float cj_l, const float m_l, const int arg_l) {
        constexpr int W = 4 * WQ;
        float vf[NWT];
        float m2 = -INFINITY;
#pragma unroll
        for (int e = 0; e < NWT; ++e) {
            const int i = 64 * e + lane;
            const bool excl = !((inb >> e) & 1u) || (unsigned)(i - lo_l) < (unsigned)W;
            vf[e] = excl ? -INFINITY : d[e] + cj_l;
            m2 = fmaxf(m2, vf[e]);
        }
        const float mm = fmaxf(m_l, ln_wave_max(m2));
        unsigned idx = 0x7fffffffu;
#pragma unroll
        for (int e = NWT - 1; e >= 0; --e) {
            const unsigned long long mk = __ballot(vf[e] == mm && 64 * e + lane < a.S);
            if (mk) idx = 64 * e + __builtin_ctzll(mk);                      // descending e: the lowest block that matches wins
        }
        if (m_l == mm && (unsigned)arg_l < idx) idx = (unsigned)arg_l;       // the window / extra-column candidates that attain it
        return idx == 0x7fffffffu ? 0 : (int)idx;                            // an all -inf frame resolves to index 0 like np.argmax
    }

    // row = delta row t of this lane's song (hist + t * SD); last = the song's last row
    __device__ __forceinline__ bool step(const float* __restrict__ row, const int t, const int last, const int cur, const bool act, int& nxt) {
        constexpr int W = 4 * WQ, WX1 = W + kMaxExtras + 1;
        const int S = a.S, nx = a.n_extras;
        int lo;
        if (a.lo_affine) { lo = cur - a.lo_off; lo = lo < 0 ? 0 : (lo > S - W ? S - W : lo); }
        else lo = loL[cur];
        // ---- pending lanes of the previous call: their rows, coalesced
        unsigned long long pm = __ballot(pend);
        const int l0 = pm ? __builtin_ctzll(pm) : -1;
        pm = pm ? pm & (pm - 1) : 0;
        const int l1 = pm ? __builtin_ctzll(pm) : -1;
        float d0[NWT], d1[NWT];
        if (l0 >= 0) {
            const float* __restrict__ g0 = ln_readlane_ptr(row, l0) + a.col0;
#pragma unroll
            for (int e = 0; e < NWT; ++e) d0[e] = 64 * e + lane < S ? g0[64 * e + lane] : -INFINITY;
        }
        if (l1 >= 0) {
            const float* __restrict__ g1 = ln_readlane_ptr(row, l1) + a.col0;
#pragma unroll
            for (int e = 0; e < NWT; ++e) d1[e] = 64 * e + lane < S ? g1[64 * e + lane] : -INFINITY;
        }
        // ---- this call's frames: window, frame maximum, extra columns
        const float* __restrict__ dp = row + a.col0 + lo;
        f32x4_u dv[WQ];
#pragma unroll
        for (int q = 0; q < WQ; ++q) dv[q] = *reinterpret_cast<const f32x4_u*>(dp + 4 * q);
        float Mt;
        float dx[kMaxExtras];
        if (a.aux_frames == 3) {                      // (one extra column; mcol = 0, xcol0 = 1: columns 2k, 2k + 1 of the carrier row hold frame ta - k)
            const int ta = wave_aux_row(t, last < 0 ? 0 : last), kf = ta - t;
            if (ta != qrow) {
                const float* __restrict__ ar = row + (ptrdiff_t)kf * a.SD + a.mcol;
                const f32x4_u qa = *reinterpret_cast<const f32x4_u*>(ar);
                const f32x2_u qb = *reinterpret_cast<const f32x2_u*>(ar + 4);
                q[0] = qa.x; q[1] = qa.y; q[2] = qa.z; q[3] = qa.w; q[4] = qb.x; q[5] = qb.y;
                qrow = ta;
            }
            Mt = kf == 0 ? q[0] : (kf == 1 ? q[2] : q[4]);
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k) dx[k] = 0.f;
            dx[0] = kf == 0 ? q[1] : (kf == 1 ? q[3] : q[5]);
        } else {
            Mt = row[a.mcol];
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k) dx[k] = k < nx ? row[a.xcol0 >= 0 ? a.xcol0 + k : a.col0 + a.extras[k]] : 0.f;
        }
        float m = -INFINITY;
        int arg = kLnBig;
        float cj;
        if (GT) {
            const float* __restrict__ wt = reinterpret_cast<const float*>(a.image + a.off_tabX) + (size_t)cur * WX1;
            f32x4_u wv[WQ];
#pragma unroll
            for (int q = 0; q < WQ; ++q) wv[q] = *reinterpret_cast<const f32x4_u*>(wt + 4 * q);
            float xw[kMaxExtras];
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k) xw[k] = wt[W + k];
            cj = wt[W + kMaxExtras];
#pragma unroll
            for (int q = 0; q < WQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = dv[q][r] + wv[q][r];
                    const bool gt = v > m;
                    m = gt ? v : m;
                    arg = gt ? 4 * q + r : arg;
                }
            arg = arg == kLnBig ? kLnBig : arg + lo;
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k)
                if (k < nx) {
                    const float v = dx[k] + xw[k];
                    const int xk = a.extras[k];
                    const bool take = v > m || (v == m && xk < arg);
                    m = take ? v : m;
                    arg = take ? xk : arg;
                }
        } else {
            const float* wt = tabL + cur * WX1;
#pragma unroll
            for (int q = 0; q < WQ; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = dv[q][r] + wt[4 * q + r];
                    const bool gt = v > m;
                    m = gt ? v : m;
                    arg = gt ? 4 * q + r : arg;
                }
            arg = arg == kLnBig ? kLnBig : arg + lo;
#pragma unroll
            for (int k = 0; k < kMaxExtras; ++k)
                if (k < nx) {
                    const float v = dx[k] + wt[W + k];
                    const int xk = a.extras[k];
                    const bool take = v > m || (v == m && xk < arg);
                    m = take ? v : m;
                    arg = take ? xk : arg;
                }
            cj = wt[W + kMaxExtras];
        }
        const bool fresh = act && !pend;
        const bool fail = fresh && !(Mt + cj < m);
        bool have = fresh && !fail;
        nxt = arg;
        // ---- the pending lanes' frames (m_p / arg_p are still those of the previous call here)
        if (l0 >= 0) {
            const int r0 = full_eval(d0, __builtin_amdgcn_readlane(lo, l0), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj), l0)),
                                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m_p), l0)), __builtin_amdgcn_readlane(arg_p, l0));
            if (lane == l0) { nxt = r0; have = true; pend = false; }
        }
        if (l1 >= 0) {
            const int r1 = full_eval(d1, __builtin_amdgcn_readlane(lo, l1), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cj), l1)),
                                     __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m_p), l1)), __builtin_amdgcn_readlane(arg_p, l1));
            if (lane == l1) { nxt = r1; have = true; pend = false; }
        }
        if (fail) { pend = true; m_p = m; arg_p = arg; ++n_full; }
        return have;
    }
};

template <int WQ, bool GT>
__device__ __forceinline__ void ln_load_tables(const BtArgs& a, float* tabL, int32_t* loL) {
    constexpr int WX1 = 4 * WQ + kMaxExtras + 1;
    const int32_t* gl = reinterpret_cast<const int32_t*>(a.image + a.off_lo);
    for (int k = threadIdx.x; k < a.SP; k += blockDim.x) loL[k] = gl[k];
    if (!GT) {
        const float* __restrict__ gtab = reinterpret_cast<const float*>(a.image + a.off_tabX);
        for (int k = threadIdx.x; k < a.SP * WX1; k += blockDim.x) tabL[k] = gtab[k];
    }
    __syncthreads();
}

// Up to four decided states are collected per lane and written with one 16-byte store (frames tw .. tw + nb - 1).
struct LaneOut {
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0, nb = 0;
    __device__ __forceinline__ void push(int32_t* __restrict__ states, const int t, const int v) {
        s3 = s2; s2 = s1; s1 = s0; s0 = v;
        if (++nb == 4) {
            i32x4_u o; o.x = s0; o.y = s1; o.z = s2; o.w = s3;
            *reinterpret_cast<i32x4_u*>(states + t) = o;
            nb = 0;
        }
    }
    __device__ __forceinline__ void flush(int32_t* __restrict__ states, const int t) {   // t = frame of s0
        if (nb >= 1) states[t] = s0;
        if (nb >= 2) states[t + 1] = s1;
        if (nb >= 3) states[t + 2] = s2;
        nb = 0;
    }
};

}  // namespace

// Speculative pass: thread g = song * C + chunk.
template <int WQ, bool GT, int NWT>
__global__ void __launch_bounds__(kLnThreads) lane_spec_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* loL = reinterpret_cast<int32_t*>(smem);
    float* tabL = reinterpret_cast<float*>(loL + a.SP);
    ln_load_tables<WQ, GT>(a, tabL, loL);
    const int lane = threadIdx.x & 63;
    const int S = a.S, SD = a.SD;
    const bool packed = a.offsets != nullptr;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int song, chunk, C;
    bool valid;
    if (packed) {                           // (song, chunk) from the host-built stream table; a song's chunk count grows with its length
        valid = g < a.n_waves;
        song = a.wave_song[valid ? g : 0];
        const int cb = a.chunk_base[song];
        C = a.chunk_base[song + 1] - cb;
        chunk = valid ? (int)g - cb : 0;
    } else {
        C = a.chunks;
        song = (int)(g / C);
        chunk = (int)(g % C);
        valid = song < a.B;
        song = valid ? song : (int)a.B - 1;
        if (a.skip_nonpositive && a.lengths[song] < 1) valid = false;       // segment of a checkpointed decode this song does not reach
    }
    const long long off = packed ? a.offsets[song] : 0;
    const int Tb = packed ? (int)(a.offsets[song + 1] - off) : ln_song_length(a, song), Lf = Tb - 1;
    int32_t* __restrict__ states = a.states + (packed ? (size_t)off : (size_t)song * a.states_stride);
    const float* __restrict__ hist = a.hist + (packed ? (size_t)off : (size_t)song * a.hist_rows) * SD;
    int32_t* __restrict__ entry = a.entry + (packed ? (size_t)a.chunk_base[song] : (size_t)song * C);
    int lo_c, hi_c;
    ln_chunk_bounds(Lf, chunk, C, lo_c, hi_c);
    if (valid && chunk == C - 1) states[Tb - 1] = a.last_state[song];
    bool act = valid && hi_c > lo_c;
    int top = hi_c - 1 + a.warm;
    int cur = 0;
    bool guess = false;
    if (act) {
        if (chunk == C - 1 || top >= Lf - 1) { top = Lf - 1; cur = a.last_state[song]; }
        else guess = true;
    }
    if (__any(guess)) {
        // lowest-index argmax of delta row top+1 (any state would do: a wrong guess costs a repair, not the result)
        const float* __restrict__ grow = hist + (size_t)(guess ? top + 1 : 0) * SD + a.col0;
        float gm = -INFINITY;
        int gi = 0;
        int i = 0;
#pragma unroll 8
        for (; i + 3 < S; i += 4) {
            const f32x4_u v = *reinterpret_cast<const f32x4_u*>(grow + i);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (v[r] > gm) { gm = v[r]; gi = i + r; }
        }
        for (; i < S; ++i) {
            const float v = grow[i];
            if (v > gm) { gm = v; gi = i; }
        }
        if (guess) cur = gi;
    }
    LaneDecider<WQ, GT, NWT> decide{a, tabL, loL, lane};
    decide.init();
    LaneOut out;
    int t = act ? top : 0;
    int entry_v = cur;                      // the state this chunk assumed at frame hi_c (no warm-up: what it started from)
    while (__any(act)) {
        const int tr = t < 0 ? 0 : t;
        const float* __restrict__ row = hist + (size_t)tr * SD;
        int nxt;
        if (decide.step(row, tr, Lf, cur, act, nxt)) {
            if (t >= hi_c) { if (t == hi_c) entry_v = nxt; }
            else out.push(states, t, nxt);
            cur = nxt;
            --t;
            act = t >= lo_c;
        }
    }
    out.flush(states, t + 1);
    if (valid) entry[chunk] = entry_v;
    if (valid && a.counters && decide.n_full) atomicAdd(a.counters + (size_t)song * kBtCounters + kCtFullRows, decide.n_full);
}

// frames past a ragged song's end
__global__ void lane_pad_kernel(BtArgs a) {      // eight workgroups per song
    const int song = blockIdx.x >> 3;
    const int Tb = ln_song_length(a, song);
    int32_t* __restrict__ states = a.states + (size_t)song * a.states_stride;
    for (int t = Tb + (blockIdx.x & 7) * blockDim.x + threadIdx.x; t < a.T; t += 8 * blockDim.x) states[t] = -1;
}

// Verify: thread per (song, chunk c < C-1): did chunk c assume at its upper boundary what chunk c+1 decided there?
__global__ void lane_verify_kernel(BtArgs a, uint32_t* __restrict__ mask) {
    const bool packed = a.offsets != nullptr;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    int song, c, C;
    if (packed) {
        if (g >= a.n_waves) return;
        song = a.wave_song[g];
        C = a.chunk_base[song + 1] - a.chunk_base[song];
        c = (int)g - a.chunk_base[song];
    } else {
        C = a.chunks;
        song = (int)(g / C);
        c = (int)(g % C);
        if (song >= a.B) return;
        if (a.skip_nonpositive && a.lengths[song] < 1) return;
    }
    if (c >= C - 1) return;
    const long long off = packed ? a.offsets[song] : 0;
    const int Lf = (packed ? (int)(a.offsets[song + 1] - off) : ln_song_length(a, song)) - 1;
    int lo_c, hi_c;
    ln_chunk_bounds(Lf, c, C, lo_c, hi_c);
    if (hi_c <= lo_c) return;
    const int truth = a.states[(packed ? (size_t)off : (size_t)song * a.states_stride) + hi_c];
    const int32_t* entry = a.entry + (packed ? (size_t)a.chunk_base[song] : (size_t)song * C);
    if (entry[c] != truth) atomicOr(mask + (size_t)song * kLaneMaskWords + (c >> 5), 1u << (c & 31));
}

// Repair: one lane per song walks the chunks whose guess was wrong, from the last to the first, re-chasing from the true
// state until the new path meets the stored one (the step below a state depends on that state only).  A re-chase that
// reaches its chunk's lower boundary with a new state goes on into the chunk below unless that chunk assumed exactly this
// state.
template <int WQ, bool GT, int NWT>
__global__ void __launch_bounds__(kLnThreads) lane_repair_kernel(BtArgs a, const uint32_t* __restrict__ mask) {
    extern __shared__ __align__(16) unsigned char smem[];
    int32_t* loL = reinterpret_cast<int32_t*>(smem);
    float* tabL = reinterpret_cast<float*>(loL + a.SP);
    const int lane = threadIdx.x & 63;
    const int SD = a.SD;
    const bool packed = a.offsets != nullptr;
    int song = blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = song < a.B;
    song = valid ? song : (int)a.B - 1;
    const int C = packed ? a.chunk_base[song + 1] - a.chunk_base[song] : a.chunks;
    uint32_t mk[kLaneMaskWords];
    bool any_bit = false;
#pragma unroll
    for (int w = 0; w < kLaneMaskWords; ++w) { mk[w] = valid ? mask[(size_t)song * kLaneMaskWords + w] : 0u; any_bit |= mk[w] != 0u; }
    if (!__syncthreads_or(any_bit)) return;          // nothing to repair in this workgroup (the common case)
    ln_load_tables<WQ, GT>(a, tabL, loL);
    const long long off = packed ? a.offsets[song] : 0;
    const int Lf = (packed ? (int)(a.offsets[song + 1] - off) : ln_song_length(a, song)) - 1;
    int32_t* __restrict__ states = a.states + (packed ? (size_t)off : (size_t)song * a.states_stride);
    const float* __restrict__ hist = a.hist + (packed ? (size_t)off : (size_t)song * a.hist_rows) * SD;
    const int32_t* entry = a.entry + (packed ? (size_t)a.chunk_base[song] : (size_t)song * C);
    LaneDecider<WQ, GT, NWT> decide{a, tabL, loL, lane};
    decide.init();
    auto take_bit = [&](int& c_out) -> bool {        // highest chunk still flagged; clears it
#pragma unroll
        for (int w = kLaneMaskWords - 1; w >= 0; --w)
            if (mk[w]) {
                const int b = 31 - __builtin_clz(mk[w]);
                mk[w] &= ~(1u << b);
                c_out = 32 * w + b;
                return true;
            }
        return false;
    };
    auto clear_bit = [&](const int c) {
#pragma unroll
        for (int w = 0; w < kLaneMaskWords; ++w)
            if ((c >> 5) == w) mk[w] &= ~(1u << (c & 31));
    };
    int n_rep = 0, n_repf = 0;
    bool repairing = false, done = !valid || !any_bit;
    int t = 0, cur = 0, cc = 0, lo_cc = 0;
    while (!__all(done)) {
        if (!done && !repairing) {
            int c;
            if (take_bit(c)) {
                int hi_c;
                ln_chunk_bounds(Lf, c, C, lo_cc, hi_c);
                cc = c;
                cur = states[hi_c];
                t = hi_c - 1;
                repairing = true;
                ++n_rep;
            } else {
                done = true;
            }
        }
        const bool act = repairing && !done;
        const int tr = act ? t : 0;
        const float* __restrict__ row = hist + (size_t)tr * SD;
        int nxt;
        if (decide.step(row, tr, Lf, cur, act, nxt)) {
            const int old = states[t];
            states[t] = nxt;
            ++n_repf;
            if (nxt == old) {
                repairing = false;                   // the stored path continues unchanged (also across the chunks below)
            } else if (t > lo_cc) {
                cur = nxt;
                --t;
            } else {
                // frame lo_cc = the upper boundary of the next chunk below that holds frames (chunks in between are empty)
                int c2 = cc - 1, lo2 = 0, hi2 = 0;
                for (; c2 >= 0; --c2) {
                    ln_chunk_bounds(Lf, c2, C, lo2, hi2);
                    if (hi2 > lo2) break;
                }
                if (c2 < 0) {
                    repairing = false;               // frame 0 decided
                } else {
                    clear_bit(c2);
                    if (entry[c2] == nxt) {
                        repairing = false;           // chunk c2 started from exactly this state
                    } else {
                        cc = c2;
                        lo_cc = lo2;
                        cur = nxt;
                        --t;
                        ++n_rep;
                    }
                }
            }
        }
    }
    if (valid && a.counters) {
        int32_t* ct = a.counters + (size_t)song * kBtCounters;
        if (n_rep) atomicAdd(ct + kCtRepairs, n_rep);
        if (n_repf) atomicAdd(ct + kCtRepairFrames, n_repf);
        if (decide.n_full) atomicAdd(ct + kCtFullRows, decide.n_full);
    }
}

static size_t lane_lds_bytes(const BtArgs& a, bool table_in_lds) {
    return sizeof(int32_t) * a.SP + (table_in_lds ? sizeof(float) * (size_t)a.SP * (a.W + kMaxExtras + 1) : 0);
}
static bool lane_table_fits(const BtArgs& a) { return lane_lds_bytes(a, true) <= 64 * 1024; }   // two workgroups of eight waves per CU

// Banded plans without dense rows whose forward pass left the frame maximum in every history row (full history).
bool lane_backtrace_applies(const BtArgs& a) {
    return a.banded && a.have_fmax && a.n_dense == 0 && !a.hist_half && a.W >= 4 && a.W % 4 == 0 && a.W <= 128 && a.W <= a.S &&
           a.col0 + a.S <= a.SD && a.mask != nullptr;
}

template <int WQ, bool GT, int NWT>
static hipError_t launch_lane_t(const BtArgs& a, hipStream_t st, int phases) {
    const size_t lds = lane_lds_bytes(a, !GT);
    const long long streams = a.offsets ? (long long)a.n_waves : (long long)a.B * a.chunks;     // (packed: a.chunks = the largest per-song count)
    if (phases & 1) {
        if (a.lengths && !a.skip_nonpositive && !a.offsets) {
            hipLaunchKernelGGL(lane_pad_kernel, dim3((unsigned)(8 * a.B)), dim3(256), 0, st, a);     // (segments of a checkpointed decode: filled by the caller)
        }
        hipLaunchKernelGGL((lane_spec_kernel<WQ, GT, NWT>), dim3((unsigned)((streams + kLnThreads - 1) / kLnThreads)), dim3(kLnThreads), lds, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (a.chunks <= 1 || !(phases & 2)) return hipSuccess;
    hipLaunchKernelGGL(lane_verify_kernel, dim3((unsigned)((streams + 255) / 256)), dim3(256), 0, st, a, a.mask);
    hipLaunchKernelGGL((lane_repair_kernel<WQ, GT, NWT>), dim3((unsigned)((a.B + kLnThreads - 1) / kLnThreads)), dim3(kLnThreads), lds, st, a, a.mask);
    return hipGetLastError();
}

// the instantiated (window, state count) pairs: the narrow windows of the 321- / 361-state grids, the wide ones of the 722-state grid
template <int WQ, int NWT>
static hipError_t launch_lane_w(const BtArgs& a, hipStream_t st, int phases) {
    return lane_table_fits(a) ? launch_lane_t<WQ, false, NWT>(a, st, phases) : launch_lane_t<WQ, true, NWT>(a, st, phases);
}
template <int WQ>
static hipError_t launch_lane_s(const BtArgs& a, hipStream_t st, int phases) {
    const int nwt = (a.S + 63) / 64;
    if (nwt <= 2) return launch_lane_w<WQ, 2>(a, st, phases);
    if (nwt <= 4) return launch_lane_w<WQ, 4>(a, st, phases);
    if (nwt <= 6) return launch_lane_w<WQ, 6>(a, st, phases);
    if (nwt <= 8) return launch_lane_w<WQ, 8>(a, st, phases);
    if (nwt <= 12) return launch_lane_w<WQ, 12>(a, st, phases);
    return launch_lane_w<WQ, 16>(a, st, phases);
}

hipError_t launch_backtrace_lane(const BtArgs& a, hipStream_t st, int phases) {
    if (!lane_backtrace_applies(a) || a.chunks < 1 || a.chunks > kLaneMaxChunks) return hipErrorInvalidConfiguration;
    switch (a.W) {
        case 16: return launch_lane_s<4>(a, st, phases);
        case 32: return launch_lane_s<8>(a, st, phases);
        case 64: return launch_lane_s<16>(a, st, phases);
        case 84: return launch_lane_s<21>(a, st, phases);
        case 96: return launch_lane_s<24>(a, st, phases);
        case 128: return launch_lane_s<32>(a, st, phases);
        default: return hipErrorInvalidConfiguration;
    }
}

// Chunks per song: enough streams for sixteen waves per CU, chunks no shorter than two warm-ups.
int lane_backtrace_chunks(int64_t B, int T, int n_cus, int warm) {
    const long long target = 16ll * 64 * (n_cus > 0 ? n_cus : 256);
    long long c = target / (B > 0 ? B : 1);
    const long long cmax = T / (2 * (warm > 0 ? warm : 1)) > 1 ? T / (2 * (warm > 0 ? warm : 1)) : 1;
    c = c > cmax ? cmax : c;
    c = c > kLaneMaxChunks ? kLaneMaxChunks : c;
    return c < 1 ? 1 : (int)c;
}

}  // namespace vit
