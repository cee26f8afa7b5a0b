// kernels.hpp -- launch interface between the C ABI (capi.hip) and kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "plan.hpp"

namespace vit {

// Banded kernel geometry: NWT target waves (64*NWT >= S), NWT in {2,4,6,8,12}.  Which (W, NWT) combinations are
// instantiated is bounded by registers: W window entries stay register-resident per thread.
//   scan form  (banded_forward_kernel: + two scan waves)      W <= 32: NWT <= 12;  W == 64: NWT <= 6
//   floor form (banded_floor_forward_kernel, plan.floor_ok)   W <= 128: NWT <= 12 (W = 128 with twelve waves keeps 32 weights in LDS)
constexpr int banded_waves_for(int S) {
    const int need = (S + 63) / 64;
    const int opts[5] = {2, 4, 6, 8, 12};
    for (int k = 0; k < 5; ++k)
        if (opts[k] >= need) return opts[k];
    return 0;
}
constexpr bool scan_form_instantiated(int W, int nwt) { return nwt > 0 && ((W <= 32 && nwt <= 12) || (W == 64 && nwt <= 6)); }
constexpr bool floor_form_instantiated(int W, int nwt) { return nwt > 0 && W <= 128 && nwt <= 12; }
// target waves of the scan form for (S, W), or 0 when it is not instantiated
constexpr int banded_target_waves(int S, int W) {
    const int nwt = banded_waves_for(S);
    return scan_form_instantiated(W, nwt) ? nwt : 0;
}
// step-structured kernel (plan.step_ok): instantiated for the Durrieu geometry -- 20-bin bands, 9 near bands, 705..768 voiced states
constexpr bool step_kernel_instantiated(int S, int bw, int kb) { return bw == 20 && kb == 9 && S - 1 > 704 && S - 1 <= 768; }
// the dense kernel keeps NS running (best, arg) pairs per thread
constexpr int dense_max_threads(int NS) { return NS <= 2 ? 1024 : 512; }

struct FwdArgs {
    const uint8_t* image;   // device plan image
    const void* logE;       // [B,T,S] f32 or f16
    const int64_t* lengths; // [B] or null
    float* hist;            // [B,T,SD] delta history (the reference's T1), SD = ceil((S+2)/4)*4: columns [0,S) = delta_t,
                            // column S = max_i delta_t[i] over the non-extra sources (banded kernels), S+1.. = scratch
    float* fmax;            // [B,64] scratch for the timing experiments (phase stamps)
    int32_t* last_state;    // [B]
    float* loglik;          // [B] or null
    int64_t B;
    int T, S, SP, S4, SD, W;
    int n_extras, n_dense;
    int extras[kMaxExtras];
    float c0;
    int debug;              // timing-only ablation mask; always 0 unless built with -DVIT_TIMING_HOOKS
    int fwd_form;           // banded forward form: 0 by batch size | 1 one target per lane | 2 two targets per lane | 3 scan
    int dense_kt1;          // dense kernel: one thread per target even where two fit
    int dense_form;         // dense kernel: 0 matrix-resident form where it applies | 1 always the streaming form
    int step_form;          // step kernel: 1 = one target per lane
    size_t off_logpi, off_A4, off_lo, off_kind, off_tabA, off_extraA, off_denseA, off_rowc, off_lo2, off_tabP;
    int pair_ok;            // the plan proved pair windows: use the two-targets-per-lane kernel
    int floor_ok;           // the plan proved the one-maximum form (banded_floor_forward_kernel)
    int step_ok, step_bw, step_kb;   // step structure (step_forward_kernel)
    float step_cn;          // logA_T[j][S-1] for every voiced target j
    size_t off_stepC, off_Arow;
    int win_shift2;         // the same for the pair windows of banded_floor_pair_forward_kernel
    size_t off_tabV;        // wave form (wave.hip)
    int wave_ok, wave_npl, wave_dk;
    int wave_u5;            // wave.hip UV: 1 = the one extra column is state S-1; 2 = also: row constant and extra-column weight uniform over a
                            // lane's slots 0..4; 3 = uniform over slots {0,1,2} and {3,4}
    int wave_flags;         // bit 0: force the 256-register (two waves per SIMD) instantiation, bit 1: the 512-register one up to 1024 songs
    int hist_half;          // wave form: 1 = only the delta rows of even frames are stored (wave.hip, HM 1)
    int64_t hist_rows;      // history rows per song: T, or (T + 1) / 2 with hist_half; checkpoint pass: segments + 1
    // wave form, vit_decode_checkpointed: ckpt_every > 0 = pass 1 (checkpoint rows only, wave.hip HM 5); t_begin > 0 / t_end < T = a
    // segment resumed from init_rows [B][64 * npl] (row t stored at t - t_begin)
    int ckpt_every, t_begin, t_end;
    const float* init_rows;
    int64_t init_stride;    // floats from one song's init row to the next
    // packed batch (vit_decode_packed; wave form): emission rows of song b are rows offsets[b] .. offsets[b+1]-1 of logE, its history rows
    // sit at the same offsets; wave w (slot w < n_slots) decodes songs slot_songs[slot_begin[w] .. slot_begin[w+1]) back to back
    const int64_t* offsets; // device [B+1], or null
    int n_slots;
    const int32_t* slot_begin;   // device [n_slots + 1]
    const int32_t* slot_songs;   // device [B]
    int win_shift;          // 0..3: delta is stored shifted by this many floats in LDS so that the window starts of a
                            // 16-lane group are 16-byte aligned in the SAME copy order (bank-conflict-free b128 reads)
};

struct BtArgs {
    const uint8_t* image;
    const float* hist;      // [B,T,SD]
    const int32_t* last_state;
    const int64_t* lengths;
    int32_t* states;        // [B,T]
    int32_t* entry;         // [B,chunks] state each chunk assumed at its upper boundary
    int64_t B;
    int T, S, SP, SD, W, K;
    int col0, mcol;         // history row layout: state i in column col0 + i, the frame maximum in column mcol
    int xcol0;              // >= 0: column xcol0 + k holds a copy of delta of extra column k (next to the frame maximum); -1: none
    int aux_frames;         // 1, or 3 (wave form with every row stored and one extra column, wave_aux_frames): row t also carries the scalars of
                            // frames t-1 and t-2, at columns mcol + 2k / xcol0 + 2k -- the back-trace reads the scalars of frame t from the
                            // carrier row wave_aux_row(t, last row) and touches one scalar line per three frames instead of one per frame
    int chunks, warm;       // time-parallel back-trace: chunks per song, warm-up frames
    int banded;             // 1: row structure (window / c0 / extras / dense rows) proven by the plan
    int have_fmax;          // the forward pass was a banded kernel (it fills pad column S of the history rows)
    int bt_form;            // 0 auto (sparse fetch where it applies) | 1 generic (lazy) kernel | 2 whole-row kernels
    int lo_affine, lo_off;  // lo[j] == clamp(j - lo_off, 0, S - W)
    int dense_rows[kMaxDenseRows];
    int n_extras, n_dense;
    int extras[kMaxExtras];
    float c0;
    size_t off_lo, off_kind, off_tabA, off_extraA, off_denseA, off_Arow, off_rowc, off_tabX, off_stepC;
    int step_ok, step_kb, step_mult;   // step-structured dense matrix: band = min((dist * step_mult) >> 16, step_kb)
    float step_cn;
    // half history (wave form, HM 1): row r of a song holds frame 2r; its aux slots mcol / xcol0 + k carry the frame maximum and the
    // extra-column deltas of frame 2r, slots mcol_odd / xcol0_odd + k those of frame 2r - 1.  The back-trace reads the emissions again.
    int hist_half;
    int64_t hist_rows;
    int mcol_odd, xcol0_odd;
    const void* logE;       // [B,T,S] the tensor vit_forward decoded (f32 or f16)
    int e_f16;
    int64_t states_stride;  // states of song b start at states + b * states_stride (T; a segment of a checkpointed decode: the whole song's T)
    int block_waves;        // half back-trace: waves per workgroup (0 / 16 default | 8 | 4: small enough to start beside resident forward waves)
    int no_fast_rows;       // sparse / half back-trace: 1 = every row through the general code (vit_plan_set_option "bt_fast_rows" 1; tests)
    int skip_nonpositive;   // sparse kernel: a song whose lengths[] entry is < 1 is skipped (segments; vit_decode clamps to 1 instead)
    int32_t* counters;      // [B][kBtCounters] per-song event counts of the sparse / half / half-wave kernels (zeroed by vit_backtrace)
    // packed batch (vit_decode_packed): history rows / states of song b at offsets[b] (its length: offsets[b+1] - offsets[b]); the
    // speculative pass runs one wave per entry of wave_song (song b owns waves chunk_base[b] .. chunk_base[b+1]-1 = its chunks;
    // chunk entries are indexed the same way)
    const int64_t* offsets; // device [B+1], or null
    const int32_t* wave_song;    // device [n_waves]
    const int32_t* chunk_base;   // device [B+1]
    int n_waves;
    uint32_t* mask;         // [B][kLaneMaskWords] lane form: bit c = chunk c assumed the wrong state at its upper boundary (zeroed by vit_backtrace)
};

hipError_t launch_dense(const FwdArgs& a, int songs_per_group, bool f16, hipStream_t st);
hipError_t launch_step(const FwdArgs& a, bool f16, hipStream_t st);
hipError_t launch_banded(const FwdArgs& a, bool f16, hipStream_t st);
hipError_t launch_wave(const FwdArgs& a, bool f16, hipStream_t st);   // wave.hip: one song per wavefront
// per song, for the segment [s0, e0) of a checkpointed decode: the sub-problem's length (0: the song ends before s0) and the state
// its back-trace starts from (the state already decided at frame e0, or the song's terminal state)
hipError_t launch_segment_prep(const int64_t* lengths, int64_t B, int T, int s0, int e0, const int32_t* states, const int32_t* last,
                               int64_t* seg_len, int32_t* seg_last, hipStream_t st);
// history layout of the wave form: row stride 64*npl floats, state i in column 64*npl - S + i, the frame maximum in column 0,
// a copy of delta of extra column k in column 1 + k
constexpr int wave_hist_stride(int npl) { return 64 * npl; }
// frames whose scalars a full-history row of the wave form carries in lane 0's idle slots (the row's own and the two before it)
constexpr int wave_aux_frames(int npl, int S, int n_extras) { return n_extras == 1 && npl >= 6 && 64 * npl - S >= 6 ? 3 : 1; }   // (six idle slots)
// the row the back-trace takes the scalars of frame t from: the next row with t % 3 == 2, or the last row written
__host__ __device__ constexpr int wave_aux_row(int t, int last) { return t - t % 3 + 2 < last ? t - t % 3 + 2 : last; }
hipError_t launch_backtrace(BtArgs a, hipStream_t st);
// backtrace_sparse.hip: fetches only the span of each history row around the path (banded plans, candidates on one lane)
bool sparse_backtrace_applies(const BtArgs& a);
// phases: bit 0 the speculative pass (one wave per (song, chunk)), bit 1 the verify-and-repair pass (one wave per song)
hipError_t launch_backtrace_sparse(const BtArgs& a, hipStream_t st, int phases = 3);
// backtrace_half.hip: the same for a half history (wave form, even rows only): odd frames are rebuilt from the row before them
bool half_backtrace_applies(const BtArgs& a);
hipError_t launch_backtrace_half(const BtArgs& a, hipStream_t st, int phases = 3);
int sparse_backtrace_chunks(int64_t B, int T, int n_cus);
// backtrace_lane.hip: one (song, chunk) stream per LANE (banded plans, full history): ~130 wave instructions per 64 decisions
bool lane_backtrace_applies(const BtArgs& a);
hipError_t launch_backtrace_lane(const BtArgs& a, hipStream_t st, int phases = 3);
int lane_backtrace_chunks(int64_t B, int T, int n_cus, int warm);
hipError_t launch_voicing_map(const int32_t* states, int64_t n, int32_t n_bins, uint8_t* voiced, int32_t* bins,
                              hipStream_t st);
hipError_t launch_scan_selftest(const float* vals, int n_waves, int mode, float* out_v, int32_t* out_i,
                                hipStream_t st);
hipError_t launch_obs_shaun(const float* logits, int64_t n_frames, int U, int spw, double thr, double off, double sc,
                            float* out, hipStream_t st);
hipError_t launch_obs_softmax(const float* logits, int64_t n_frames, int U, int spw, float* out, hipStream_t st);
hipError_t launch_obs_softmax_scaled(const float* logits, int64_t n_frames, int U, int spw, double unvoiced_logit,
                                     const float* prior, float* out, hipStream_t st);
hipError_t launch_snippets_append(const float* snips, int n, int C, int F, int mode, float* out, int64_t rows, hipStream_t st);
hipError_t launch_voicing_notes(const int32_t* states, int64_t n, int32_t n_bins, const float* note_range, uint8_t* voiced,
                                int32_t* bins, float* notes, float* notes_v, hipStream_t st);
int backtrace_tile_rows(int SD);
constexpr int kBtWarm = 128;       // warm-up frames of a speculative chunk (survivor paths coalesce within tens of frames)
constexpr int kBtWarmSparse = 64;  // the sparse kernel runs many short chunks: a shorter warm-up (a wrong guess only costs a repair)
constexpr int kBtMaxChunks = 32;    // one stream per wavefront (sparse / half / whole-row kernels)
constexpr int kLaneMaxChunks = 256; // one stream per lane (backtrace_lane.hip); the workspace holds [B][kLaneMaxChunks] chunk entries
constexpr int kLaneMaskWords = kLaneMaxChunks / 32;
// per-song event counters (include/viterbi_hip.h vit_backtrace_counters): tiles fetched, span misses, whole-row evaluations
// (bound failures), of those: odd rows rebuilt in full, chunks repaired, frames rewritten by repairs
constexpr int kBtCounters = 16;
enum { kCtTiles = 0, kCtMisses = 1, kCtFullRows = 2, kCtRebuilt = 3, kCtRepairs = 4, kCtRepairFrames = 5 };
int backtrace_chunks(int64_t B, int T);

}  // namespace vit
