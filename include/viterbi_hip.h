/*
 * viterbi_hip.h -- C ABI of libviterbi_hip.so (MI355X / gfx950 Viterbi decoder).
 *
 * This is the drop-in boundary for the hot path of drwangxian/viterbi_spl: the
 * float32 log-domain Viterbi forward recursion + argmax back-trace.  Plain
 * pointers and sizes only; no torch types; no exceptions cross this ABI (every
 * entry point returns a vit_status).  All device buffers are caller-owned; the
 * library never allocates device memory and never synchronises the host inside
 * vit_decode().
 *
 * Reference interfaces replaced (paths relative to the reference repo):
 *   - viterbi_numba.core(B, prob_init, probs) -> int64[T]
 *       dcnet/aot_viterbi_core.py:8-54, call site dcnet/tf_viterbi_decoding.py:147-151
 *       ('i8[:](f4[:, ::1], f4[:], f4[:, ::1])': B is [S,S] "target <- source",
 *        C-contiguous; probs is [T,S] C-contiguous).
 *   - viterbi_librosa_fn(*, log_transition_matrix_T, log_prob_init, log_probs_st)
 *       imm/tf_viterbi.py:75-109 (the log-domain core this ABI mirrors).
 *   - the in-class copies Viterbi.viterbi_librosa_fn / SoftMaxViterbi.viterbi_librosa_fn
 *       tonet/for_paper.py:1833-1870, :1999-2037 (after their host-side log step).
 *
 * Differences from the reference boundary, all deliberate:
 *   - inputs are NEVER mutated (the Numba core logs its arguments in place,
 *     dcnet/aot_viterbi_core.py:23-25);
 *   - the contract starts at log-domain tensors: the prob -> log step stays on
 *     the host in the Python adapters, because NumPy's float32 log is the one
 *     operation whose last bit is not reproducible on a GPU;
 *   - songs are batched ([B,T,S]) and may be ragged (lengths);
 *   - states come back as int32 (the Python adapters widen to int64);
 *   - the terminal log-likelihood delta_{T-1}[s_{T-1}] is returned as well
 *     (the unused `p` at dcnet/tf_viterbi_decoding.py:255).
 *
 * NaN inputs are outside the contract.  -inf entries are accepted.
 */
#ifndef VITERBI_HIP_H_
#define VITERBI_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIT_ABI_VERSION 4

typedef enum vit_status {
    VIT_OK = 0,
    VIT_EINVAL = -1,       /* bad argument (null pointer, size out of range) */
    VIT_ENOMEM = -2,       /* host allocation failed */
    VIT_EHIP = -3,         /* a HIP runtime call failed (see vit_last_hip_error) */
    VIT_EWORKSPACE = -4,   /* workspace smaller than vit_workspace_bytes() */
    VIT_EUNSUPPORTED = -5, /* shape/algorithm combination not supported */
    VIT_ENOTUPLOADED = -6, /* vit_decode() before vit_plan_upload() */
    VIT_ENOFORWARD = -7    /* vit_backtrace() on a workspace without a matching vit_forward() on record */
} vit_status;

/* storage type of the emission tensor (arithmetic is always float32) */
typedef enum vit_dtype { VIT_F32 = 0, VIT_F16 = 1 } vit_dtype;

/* forward-kernel selection */
typedef enum vit_algo {
    VIT_ALGO_AUTO = 0,   /* banded if the plan proved the structure, else dense */
    VIT_ALGO_DENSE = 1,  /* S*S max-plus per frame, any matrix */
    VIT_ALGO_BANDED = 2, /* exact row-constant + window + extra-column decomposition; the form (one song per
                            workgroup / one song per wavefront) is chosen from the batch size */
    VIT_ALGO_WAVE = 3,   /* banded, one song per wavefront (throughput form; VIT_EUNSUPPORTED if the plan lacks it) */
    VIT_ALGO_GROUP = 4   /* banded, one song per workgroup (latency form) */
} vit_algo;

typedef struct vit_plan vit_plan; /* opaque: analysed transition matrix + prior */
typedef void *vit_stream;         /* hipStream_t */

typedef struct vit_plan_info {
    int64_t S;
    int32_t banded_ok;      /* 1 if the banded kernel may be used for this matrix */
    int32_t n_consts;       /* distinct row constants */
    int32_t n_extras;       /* extra exception columns shared by most rows */
    int32_t max_window;     /* widest per-row exception window */
    int32_t group_window;   /* window width the banded kernel evaluates per target */
    int32_t reserved[3];    /* [0] dense rows, [1] one-maximum ("floor") form proven, [2] bit 0: window start affine in the target, bit 1: pair windows proven, bit 2: step structure (dense matrix, piecewise-constant columns), bit 3: wave form (one song per wavefront) available */
    float consts[4];
    int32_t extras[4];
} vit_plan_info;

int vit_abi_version(void);
const char *vit_status_string(int status);
/* last hipError_t seen by this library on the calling thread (0 = hipSuccess) */
int vit_last_hip_error(void);

/*
 * Analyse a transition matrix on the host.
 *   logA_T : host, [S,S] float32 C-order, row j = log-probabilities INTO target j
 *            (same orientation the reference passes to its core,
 *            dcnet/tf_viterbi_decoding.py:144, imm/tf_viterbi.py:77-80)
 *   log_pi : host, [S] float32
 * 1 <= S <= 1024.
 */
int vit_plan_create(const float *logA_T, const float *log_pi, int64_t S, vit_plan **out);
void vit_plan_destroy(vit_plan *plan);
int vit_plan_query(const vit_plan *plan, vit_plan_info *info);

/*
 * Kernel-selection overrides, per plan.  Every setting decodes the same bits -- they exist so that tests and timing
 * scripts can reach each kernel form (the library reads NO environment variables).  Keys:
 *   "forward_form"     banded plans: 0 by batch size | 1 one target per lane | 2 two targets per lane | 3 scan form |
 *                      4 wave form | 5 never the wave form
 *   "backtrace_form"   0 auto | 1 generic kernel | 2 whole-row kernels (no sparse fetch) | 4 one (song, chunk) stream per LANE
 *                      instead of per wavefront (banded plans, full history; up to 256 chunks per song; VIT_EUNSUPPORTED elsewhere)
 *   "dense_songs"      songs per workgroup of the dense kernel (0 auto); "dense_one_thread" 1 = one thread per target;
 *                      "dense_form" 0 = matrix-resident dense kernel where it applies (64 < S <= 368), 1 = always stream the matrix
 *   "step_form"        step-structured kernel: 0 four targets per lane, bands split over two waves | 1 one target per lane |
 *                      2 off (plain dense kernel) | 3 four targets per lane, one wave per lane group
 *   "bt_chunks", "bt_warm"   time-parallel back-trace: chunks per song (0 auto), warm-up frames (-1 default); "bt_fast_rows" 1 = every
 *                      row through the general code of the sparse kernels (0: the unexceptional rows run in a loop of their own)
 *   "bt_block_waves"   half history's back-trace: waves per workgroup, 0 = 16 | 8 | 4.  Eight-wave workgroups (208 registers per SIMD) can start
 *                      on a CU whose SIMDs each hold one 256-register forward wave, i.e. beside the next batch's forward pass at up to
 *                      4 x CUs songs in flight (the two-stream schedule); sixteen-wave ones cannot
 *   "win_shift"        LDS window shift 0..3 (-1 from the plan); "wave_min_batch" (0 default), "wave_two" 1 | 2 (register budget of
 *                      the wave kernel) + 4 = a full-history row carries its own scalars only (default: also those of the two frames
 *                      before it, so that the back-trace touches one scalar line per three frames; A/B and tests)
 *   "wave_history"     wave form: 0 / 1 = store every delta row | 2 = store the rows of even frames only (the back-trace rebuilds
 *                      the 32 values an odd frame needs from the row before it and the emissions): half the workspace and a
 *                      faster forward pass for a slower back-trace; VIT_EUNSUPPORTED where the plan does not
 *                      allow it (window 32 wide with an affine start, <= 2 extra columns, S <= 378)
 *   "wave_uniform"     wave form: 0 = specialised variants where the plan proves them -- the one extra column is the last state (every
 *                      matrix the reference builds), and a lane's slots 0..4 share the row constant and the extra-column weight
 *                      (its 361-state matrices; for its 321-state ones three groups of slots do) | 1 = neither | 2 = the first only |
 *                      3 = the three-group form also where two groups would do
 *   "timing"           ablation / probe mask: accepted only by a -DVIT_TIMING_HOOKS build (VIT_EUNSUPPORTED otherwise;
 *                      those bits change results)
 *   "reset"            back to the defaults
 * Not thread-safe against concurrent decodes on the same plan.
 */
int vit_plan_set_option(vit_plan *plan, const char *key, int64_t value);

/* Device image of the plan: the caller allocates vit_plan_image_bytes() bytes of
 * device memory (256-byte aligned) and uploads once; the copy is enqueued on `stream`. */
size_t vit_plan_image_bytes(const vit_plan *plan);
int vit_plan_upload(vit_plan *plan, void *device_image, size_t bytes, vit_stream stream);

/* Bytes of device workspace vit_decode() needs for a [B,T,S] batch (float32 delta history
 * [B,T,SD] + per-song terminals; SD = ceil((S+2)/4)*4 with the per-frame maximum in pad column S, or 64*ceil(S/64) in slot
 * order for the wave form -- the size covers whichever form runs).  256-byte aligned base required. */
size_t vit_workspace_bytes(const vit_plan *plan, int64_t B, int64_t T);
/* The same for ONE algo (what vit_forward / vit_decode check a workspace against): smaller than vit_workspace_bytes() where
 * the chosen kernel keeps a narrower or a half history -- the wave form stores [B, ceil(T/2), 64*ceil(S/64)] floats.  The
 * answer depends on B (VIT_ALGO_AUTO / _BANDED pick the form by batch size) and on the plan's options; 0 = algo unsupported. */
size_t vit_workspace_bytes_for(const vit_plan *plan, int64_t B, int64_t T, int algo);

/*
 * Decode B songs.
 *   logE      : device, [B,T,S] C-order, float32 or float16 (emis_dtype)
 *   lengths   : device, [B] int64 or NULL; song b uses frames [0, clamp(lengths[b],1,T))
 *   workspace : device, >= vit_workspace_bytes(plan,B,T)
 *   states    : device, [B,T] int32; frames past a song's length are set to -1
 *   loglik    : device, [B] float32 or NULL
 * Enqueues kernels on `stream` and returns; no host synchronisation.
 */
int vit_decode(const vit_plan *plan, const void *logE, int emis_dtype, int64_t B, int64_t T,
               const int64_t *lengths, void *workspace, size_t workspace_bytes,
               int32_t *states, float *loglik, int algo, vit_stream stream);

/* Forward pass only / back-trace only; used by bench.py to time the two kernels separately and to run the back-trace
 * of one batch on another stream.  vit_decode() == forward then backtrace.  vit_forward records, per plan and workspace
 * pointer, which kernel family filled the workspace and how its history rows are laid out; vit_backtrace reads that
 * record (its `algo` argument is ignored) and returns VIT_ENOFORWARD when this workspace has no forward pass of the same
 * (B, T) on record.  The caller orders the two calls (same stream, or an event).  A plan keeps the records of the 64
 * workspaces most recently written; a failed vit_forward leaves none for its workspace.
 *
 * LIFETIME RULE: the emission tensor handed to vit_forward must stay valid and UNCHANGED until vit_backtrace has run -- a half
 * history (wave form, "wave_history" 2) re-reads 32 emission values of every odd frame through the pointer vit_forward
 * recorded.  vit_backtrace_checked() takes the emission pointer and storage type again and returns VIT_EINVAL when they are
 * not the ones on record (a double-buffered caller that refilled or swapped its emission buffer between the two phases);
 * the Python host always calls that form. */
int vit_forward(const vit_plan *plan, const void *logE, int emis_dtype, int64_t B, int64_t T,
                const int64_t *lengths, void *workspace, size_t workspace_bytes,
                float *loglik, int algo, vit_stream stream);
int vit_backtrace(const vit_plan *plan, int64_t B, int64_t T, const int64_t *lengths,
                  void *workspace, size_t workspace_bytes, int32_t *states, int algo, vit_stream stream);
int vit_backtrace_checked(const vit_plan *plan, const void *logE, int emis_dtype, int64_t B, int64_t T,
                          const int64_t *lengths, void *workspace, size_t workspace_bytes, int32_t *states,
                          int algo, vit_stream stream);

/* Which forward kernel family vit_forward() would launch for (plan, options, algo, batch size): 1 dense / step-structured,
 * 2 banded with one song per workgroup, 3 banded with one song per wavefront; a negative vit_status when the algo is not
 * available for this plan.  (The thresholds scale with the device's compute units; callers should ask, not guess.) */
int vit_forward_family(const vit_plan *plan, int64_t B, int algo);

/*
 * Bounded-workspace decode: the same result as vit_decode() with a workspace of about (T / segment_frames + segment_frames)
 * delta rows per song instead of T (the reference keeps its work buffers T1 / T2 for one song at a time,
 * tonet/for_paper.py:1852-1853; vit_decode keeps a history for the whole batch).  Pass 1 runs the forward recursion and keeps
 * one row per segment of segment_frames frames; pass 2 re-runs it segment by segment from the last to the first and
 * back-traces each.  Exact by construction; about twice the forward work.  Plans with the wave form only (vit_plan_info
 * reserved[2] bit 3; VIT_EUNSUPPORTED otherwise); 64 <= segment_frames (values above T act like T).  Of the plan's options
 * only "bt_fast_rows" and "wave_two" are honoured: the passes run the general wave kernel with a full history of the segment
 * and the sparse back-trace ("wave_history", "wave_uniform", "backtrace_form", "bt_chunks" are ignored).  The library does
 * not record this call for vit_backtrace().
 */
size_t vit_workspace_bytes_checkpointed(const vit_plan *plan, int64_t B, int64_t T, int64_t segment_frames);
int vit_decode_checkpointed(const vit_plan *plan, const void *logE, int emis_dtype, int64_t B, int64_t T,
                            const int64_t *lengths, void *workspace, size_t workspace_bytes, int32_t *states,
                            float *loglik, int64_t segment_frames, vit_stream stream);

/*
 * Packed (ragged) decode: B songs of DIFFERENT lengths without padding.  The reference decodes every recording whole with its
 * own T (tonet/for_paper.py:2304-2309, one viterbi(logits) call per recording).
 *   logE    : device, [offsets[B], S] C-order: the emission rows of song b are rows offsets[b] .. offsets[b+1]-1
 *   offsets : HOST, [B+1] int64, offsets[0] = 0, strictly increasing (every song holds at least one frame)
 *   states  : device, [offsets[B]] int32, packed like the emission rows
 *   loglik  : device, [B] float32 or NULL
 * Plans with the wave form only (vit_plan_info reserved[2] bit 3; VIT_EUNSUPPORTED otherwise): the forward pass runs
 * min(B, 8 x compute units) wavefronts, each decoding a host-packed list of songs back to back (longest-first greedy bins by
 * frame count), so a launch costs (total frames / wavefronts), not its longest song, and neither memory nor time is spent on
 * padding; the back-trace cuts every song into chunks of about equal length, one chunk per lane ("backtrace_form" 4's kernels).  The library builds the slot and chunk tables on
 * the host from `offsets` and uploads them through a pinned staging buffer it owns (it waits for the previous call's upload
 * before reusing it; otherwise no host synchronisation).  Bit-identical to vit_decode() of each song alone.  Not thread-safe against
 * concurrent packed decodes on the same plan (one staging buffer per plan); `lengths`-style padding does not exist here, so states
 * carries no -1 entries.
 */
size_t vit_workspace_bytes_packed(const vit_plan *plan, int64_t B, int64_t total_frames);
int vit_decode_packed(const vit_plan *plan, const void *logE, int emis_dtype, int64_t B, const int64_t *offsets,
                      void *workspace, size_t workspace_bytes, int32_t *states, float *loglik, vit_stream stream);

/* Event counts of the last vit_backtrace() on this workspace (banded plans; all zero for the kernels that do not count):
 * *offset = byte offset, inside the workspace, of an int32 [B][*n_per_song] device array (valid once the back-trace has run
 * on its stream): per song [0] tiles fetched, [1] span misses (the path left the fetched columns), [2] whole-row evaluations
 * (the bound fl(M_t + c_j) could not exclude the row constant), [3] of those: odd rows of a half history rebuilt in full,
 * [4] chunks repaired by the verify pass, [5] frames rewritten by repairs.  The data-dependent part of the back-trace's cost;
 * bench.py reports it per 1000 frames. */
int vit_backtrace_counters(const vit_plan *plan, int64_t B, int64_t T, const void *workspace, size_t *offset,
                           int32_t *n_per_song);

/* Fused epilogue of Viterbi.__call__ (tonet/for_paper.py:1828-1829):
 * voiced = state < n_bins ; bins = min(state, n_bins-1).  n entries, device pointers.
 * Negative states (ragged padding) give voiced = 0, bins = -1. */
int vit_voicing_map(const int32_t *states, int64_t n, int32_t n_bins, uint8_t *voiced, int32_t *bins,
                    vit_stream stream);

/* The same map plus the bin -> note lookup the metrics consume (tonet/for_paper.py:2106-2115 est_notes_360_fn, :2207):
 * notes = note_range[bins], notes_voiced = voiced ? notes : 0.  note_range: device, [n_bins] float32.  Any of the four
 * outputs may be NULL. */
int vit_voicing_notes(const int32_t *states, int64_t n, int32_t n_bins, const float *note_range, uint8_t *voiced,
                      int32_t *bins, float *notes, float *notes_voiced, vit_stream stream);

/*
 * Device-resident hand-off from the acoustic model (replaces the host round trip at tonet/for_paper.py:2282-2302): a
 * batch of snippets [n_snippets, n_channels, n_frames] float32 (channel 0 = unvoiced) is transposed into time-major logit
 * rows appended at rows_out (device; the caller advances the pointer recording by recording):
 *   mode 0 ("shaun"):   n_channels-1 columns, channel 0 subtracted (:2296-2297);  mode 1 ("softmax"): n_channels columns.
 * Only the first n_rows <= n_snippets * n_frames rows are written (the last batch of a recording is padded, :2299-2300).
 */
int vit_snippets_append(const float *snippets, int32_t n_snippets, int32_t n_channels, int32_t n_frames, int32_t mode,
                        float *rows_out, int64_t n_rows, vit_stream stream);

/*
 * Emission builders (the step upstream of the decoder): pitch logits -> log(p + tiny)
 * observation log-probabilities in the [n_frames, n_bins+1] layout vit_decode() reads (unvoiced state last).
 *   vit_obs_shaun   : Viterbi.observation_probs_fn, tonet/for_paper.py:1733-1778 -- logits [n_frames, n_bins];
 *                     threshold_logit = log(th/(1-th)), offset = log(p/(1-p)), scale (:1697-1699, :1743-1745);
 *                     spw = single-side peak width (5).
 *   vit_obs_softmax : SoftMaxViterbi.observation_probs_fn, tonet/for_paper.py:1911-1944 -- logits
 *                     [n_frames, n_bins+1] with column 0 = unvoiced; spw = 15.
 *   vit_obs_softmax_scaled : dcnet's SoftMaxViterbi.observation_probs_fn, dcnet/softmax_viterbi.py:2530-2579 -- logits
 *                     [n_frames, n_bins]; the unvoiced logit is the constant unvoiced_logit = log(vth/(1-vth)) (:2548-2550);
 *                     every softmax probability is divided by its state prior ("scaled likelihood", values may exceed 1,
 *                     i.e. positive log-emissions); prior: device, [n_bins+1] float32 in state order (unvoiced last), or NULL
 *                     for the unscaled variant; a frame without peaks gets 1 / prior[unvoiced] (:2562-2565); spw = 5.
 * exp/log run on the GPU: probabilities agree with the reference to a few ulp, structural zeros are exact.
 */
int vit_obs_shaun(const float *logits, int64_t n_frames, int32_t n_bins, int32_t spw, double threshold_logit,
                  double offset, double scale, float *logE, vit_stream stream);
int vit_obs_softmax(const float *logits, int64_t n_frames, int32_t n_bins, int32_t spw, float *logE,
                    vit_stream stream);
int vit_obs_softmax_scaled(const float *logits, int64_t n_frames, int32_t n_bins, int32_t spw, double unvoiced_logit,
                           const float *prior, float *logE, vit_stream stream);

/* Self-test hook of the wave-wide DPP scan primitives the kernels are built on (tests/test_gpu_parity.py): vals [n_waves*64]
 * device float32; mode 0 / 1 ordered (value, index) first-maximum scan forward / reverse, 2 value-only prefix maximum, 3 the
 * prefix maximum shifted up one lane, 4 wave-wide maximum; out_v / out_i [n_waves*64]. */
int vit_debug_scan(const float *vals, int n_waves, int mode, float *out_v, int32_t *out_i, vit_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* VITERBI_HIP_H_ */
