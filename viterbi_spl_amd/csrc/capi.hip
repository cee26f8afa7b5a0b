// capi.hip -- extern "C" entry points declared in include/viterbi_hip.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <queue>
#include <new>
#include <utility>
#include <vector>

#include "../../include/viterbi_hip.h"
#include "kernels.hpp"
#include "plan.hpp"

// Kernel-selection overrides (vit_plan_set_option).  Every combination decodes the same bits; they exist so that tests and
// timing scripts can reach each kernel form.  `timing` carries the ablation / probe mask of a -DVIT_TIMING_HOOKS build and
// is refused by a release build (those bits DO change results).
struct Tuning {
    int forward_form = 0;      // banded plans: 0 by batch size | 1 one target per lane | 2 two targets per lane | 3 scan form
                               //               | 4 wave form (one song per wavefront) | 5 never the wave form
    int backtrace_form = 0;    // 0 auto (sparse fetch, one stream per wavefront, where it applies) | 1 generic (lazy) kernel | 2 whole-row kernels |
                               // 4 one stream per LANE (backtrace_lane.hip; VIT_EUNSUPPORTED where it does not apply)
    int dense_songs = 0;       // songs per workgroup of the dense kernel (0 = by batch size)
    int dense_one_thread = 0;  // 1: one thread per target in the dense kernel even where two fit
    int dense_form = 0;        // 0: matrix-resident dense kernel where it applies (64 < S <= 368) | 1: always the streaming kernel
    int step_form = 0;         // step-structured kernel: 0 four targets per lane, split | 1 one | 2 never (plain dense kernel) | 3 four, one wave
    int bt_chunks = 0;         // time-parallel back-trace: chunks per song (0 = auto)
    int bt_warm = -1;          //                            warm-up frames (-1 = default)
    int win_shift = -1;        // LDS window shift of the floor kernels (-1 = from the plan)
    int wave_min_batch = 0;    // batch size from which banded plans take the wave form (0 = default)
    int wave_two = 0;          // wave kernel: 1 always the 256-register instantiation (the default but for two extra columns) | 2 the 512-register
                               // one up to 1024 songs
    int wave_uniform = 0;      // wave form: 0 = the last-state / uniform-lane variants where the plan proves them (wave.hip UV) | 1 = neither |
                               // 2 = the last-state variant only | 3 = the three-group form where the two-group one would run (it is valid there too)
    int bt_fast_rows = 0;      // sparse / half back-trace: 0 = the unexceptional rows in their own loop | 1 = every row through the general code
    int bt_block_waves = 0;    // half back-trace: waves per workgroup, 0 = 16 | 8 | 4 (to start beside the next batch's resident forward waves)
    int wave_history = 0;      // wave form: 0 / 1 every delta row | 2 the rows of even frames only (VIT_EUNSUPPORTED where the plan does not allow it)
    int timing = 0;
};

// What the last vit_forward left in a workspace: vit_backtrace reads the layout from here, not from its arguments.
struct FwdStamp {
    const void* ws = nullptr;
    int64_t B = 0, T = 0;
    int family = 0;            // 1 dense / step, 2 banded (one song per workgroup), 3 wave
    int SD = 0, col0 = 0, mcol = 0, xcol0 = -1, aux_frames = 1;
    int have_fmax = 0;         // column mcol of every history row holds a bound on max_i delta_t[i]
    int half = 0;              // wave form, even rows only: the back-trace re-reads the emissions
    const void* logE = nullptr;
    int e_f16 = 0;
};

struct vit_plan {
    int S = 0;
    vit::BandedPlan bp;
    vit::ImageLayout L;
    std::vector<uint8_t> host_image;
    const uint8_t* dev_image = nullptr;
    int wave_u5 = 0;           // the wave kernel's uniform-lane form applies (wave.hip U5)
    int n_cus = 256;           // compute units of the device the image was uploaded to (vit_plan_upload): kernel-selection thresholds scale with it
    Tuning tune;
    mutable std::mutex mu;
    mutable std::vector<FwdStamp> stamps;   // most recent first, at most kMaxStamps
    // vit_decode_packed: pinned host staging of the slot / chunk tables, and the event of the copy that last read it
    mutable void* pk_host = nullptr;
    mutable size_t pk_host_bytes = 0;
    mutable hipEvent_t pk_event = nullptr;
    ~vit_plan() {
        if (pk_event) (void)hipEventDestroy(pk_event);
        if (pk_host) (void)hipHostFree(pk_host);
    }
};

namespace {

thread_local int g_last_hip_error = 0;

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
inline int hist_stride(int S) { return (S + 5) / 4 * 4; }   // rows 16-byte aligned, at least two pad columns (frame max, scratch)
// widest history row any forward kernel of this plan writes (the wave form stores rows in slot order: 64 * npl floats)
inline int hist_stride_ws(const vit_plan* p) {
    const int sd = hist_stride(p->S);
    const int sw = (p->bp.ok && p->bp.wave_ok) ? vit::wave_hist_stride(p->bp.wave_npl) : 0;
    return sd > sw ? sd : sw;
}
constexpr size_t kMaxStamps = 64;    // workspaces with a forward pass on record per plan (include/viterbi_hip.h, vit_forward)
// From two workgroups per CU + 1 songs on, one song per wavefront beats one song per workgroup (256 CUs: two workgroups per CU hold 512
// songs, B = 512: 18.8 vs 14.2 ms forward; the 513th starts a second round, B = 576: 18.8 vs 19.8 ms, B = 1024: 20.0 vs 25-27 ms;
// DESIGN.md 6).
inline int64_t wave_min_batch(const vit_plan* p) { return 2 * (int64_t)p->n_cus + 1; }

void stamp_erase(const vit_plan* p, const void* ws) {
    std::lock_guard<std::mutex> g(p->mu);
    for (size_t k = 0; k < p->stamps.size(); ++k)
        if (p->stamps[k].ws == ws) { p->stamps.erase(p->stamps.begin() + k); break; }
}
void stamp_put(const vit_plan* p, const FwdStamp& st) {
    stamp_erase(p, st.ws);
    std::lock_guard<std::mutex> g(p->mu);
    p->stamps.insert(p->stamps.begin(), st);
    if (p->stamps.size() > kMaxStamps) p->stamps.pop_back();
}
bool stamp_get(const vit_plan* p, const void* ws, FwdStamp* out) {
    std::lock_guard<std::mutex> g(p->mu);
    for (const FwdStamp& st : p->stamps)
        if (st.ws == ws) { *out = st; return true; }
    return false;
}

int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return VIT_EHIP;
}

struct WsLayout {
    size_t off_hist, off_fmax, off_cnt, off_mask, off_last, off_entry, bytes;
};

// history floats per song: `rows` rows of `stride` floats
WsLayout ws_layout_hist(int64_t B, size_t rows, size_t stride) {
    WsLayout w;
    w.off_hist = 0;
    w.off_fmax = align256((size_t)B * rows * stride * sizeof(float));
    w.off_cnt = w.off_fmax + align256((size_t)B * 64 * sizeof(float));    // (off_fmax: timing-experiment scratch of the forward kernels)
    w.off_mask = w.off_cnt + align256((size_t)B * vit::kBtCounters * sizeof(int32_t));   // the back-trace's event counters, then its chunk flags:
    w.off_last = w.off_mask + align256((size_t)B * vit::kLaneMaskWords * sizeof(uint32_t));   // one memset covers both
    w.off_entry = w.off_last + align256((size_t)B * sizeof(int32_t));
    w.bytes = w.off_entry + align256((size_t)B * vit::kLaneMaxChunks * sizeof(int32_t));
    return w;
}
// the layout that covers every forward kernel of the plan (vit_workspace_bytes)
WsLayout ws_layout(const vit_plan* p, int64_t B, int64_t T) { return ws_layout_hist(B, (size_t)T, (size_t)hist_stride_ws(p)); }

// Does the wave form of this plan store a half history (even frames only)?  The back-trace kernel for it
// (backtrace_half.hip) is asked with the layout the forward kernel would write.
bool wave_half_applies(const vit_plan* p, int64_t T) {
    if (!(p->bp.ok && p->bp.wave_ok) || p->tune.wave_history != 2 || T < 2) return false;
    vit::BtArgs b{};
    b.S = p->S;
    b.SP = p->L.SP;
    b.SD = vit::wave_hist_stride(p->bp.wave_npl);
    b.col0 = b.SD - p->S;
    b.W = p->bp.W;
    b.banded = 1;
    b.n_extras = p->bp.n_extras;
    b.n_dense = p->bp.n_dense;
    b.lo_affine = p->bp.lo_affine ? 1 : 0;
    b.lo_off = p->bp.lo_off;
    return vit::half_backtrace_applies(b);
}
// the layout of one forward family (1 dense / step, 2 banded workgroup kernels, 3 wave)
WsLayout ws_layout_family(const vit_plan* p, int family, int64_t B, int64_t T) {
    if (family == 3) {
        const size_t sd = (size_t)vit::wave_hist_stride(p->bp.wave_npl);
        return wave_half_applies(p, T) ? ws_layout_hist(B, (size_t)((T + 1) / 2), sd) : ws_layout_hist(B, (size_t)T, sd);
    }
    return ws_layout_hist(B, (size_t)T, (size_t)hist_stride(p->S));
}

int check_common(const vit_plan* plan, int64_t B, int64_t T, const void* ws) {
    if (!plan || !ws) return VIT_EINVAL;
    if (B < 0 || T < 1 || T > (int64_t)1 << 30 || B > (int64_t)1 << 30) return VIT_EINVAL;
    if (!plan->dev_image) return VIT_ENOTUPLOADED;
    if (((uintptr_t)ws & 255) != 0) return VIT_EINVAL;
    return VIT_OK;
}

}  // namespace

extern "C" {

int vit_abi_version(void) { return VIT_ABI_VERSION; }

const char* vit_status_string(int status) {
    switch (status) {
        case VIT_OK: return "ok";
        case VIT_EINVAL: return "invalid argument";
        case VIT_ENOMEM: return "out of host memory";
        case VIT_EHIP: return "HIP runtime error";
        case VIT_EWORKSPACE: return "workspace too small";
        case VIT_EUNSUPPORTED: return "unsupported shape or algorithm";
        case VIT_ENOTUPLOADED: return "plan image not uploaded";
        case VIT_ENOFORWARD: return "no forward pass on record for this workspace";
        default: return "unknown status";
    }
}

int vit_last_hip_error(void) { return g_last_hip_error; }

int vit_plan_create(const float* logA_T, const float* log_pi, int64_t S, vit_plan** out) {
    if (!logA_T || !log_pi || !out) return VIT_EINVAL;
    if (S < 1 || S > 1024) return VIT_EINVAL;
    vit_plan* p = new (std::nothrow) vit_plan();
    if (!p) return VIT_ENOMEM;
    try {
        p->S = (int)S;
        p->bp = vit::analyze_banded(logA_T, (int)S);
        if (!p->bp.ok) vit::analyze_step(logA_T, (int)S, p->bp);
        p->L = vit::make_layout((int)S, p->bp);
        p->host_image.resize(p->L.bytes);
        vit::fill_image(logA_T, log_pi, p->bp, p->L, p->host_image.data());
        // wave form, uniform-lane variant: six states per lane, ONE extra column = the last state, and for every lane the row constant
        // and the extra-column weight of its slots 0..4 agree bit for bit (idle slots count as -inf)
        if (p->bp.ok && p->bp.wave_ok && p->bp.n_extras == 1 && p->bp.extras[0] == (int)S - 1) p->wave_u5 = 1;     // (the extra column is the last state)
        if (p->wave_u5 == 1 && p->bp.wave_npl == 6) {
            const float* xa = reinterpret_cast<const float*>(p->host_image.data() + p->L.off_extraA);
            const int o = 384 - (int)S;
            auto uniform = [&](const int k0, const int k1) {       // slots k0 .. k1-1 of every lane agree in row constant and extra-column weight
                for (int l = 0; l < 64; ++l) {
                    uint32_t c_ref = 0, x_ref = 0;
                    for (int k = k0; k < k1; ++k) {
                        const int j = 6 * l + k - o;
                        const float cv = j >= 0 ? p->bp.rowc[j] : -INFINITY, xv = j >= 0 ? xa[j] : -INFINITY;
                        uint32_t cb, xb;
                        std::memcpy(&cb, &cv, 4);
                        std::memcpy(&xb, &xv, 4);
                        if (k == k0) { c_ref = cb; x_ref = xb; }
                        if (cb != c_ref || xb != x_ref) return false;
                    }
                }
                return true;
            };
            if (uniform(0, 5)) p->wave_u5 = 2;                      // S = 361: the idle slots end at a lane boundary + 5
            else if (uniform(0, 3) && uniform(3, 5)) p->wave_u5 = 3;   // S = 321: they end in the middle of a lane
        }
    } catch (const std::bad_alloc&) {
        delete p;
        return VIT_ENOMEM;
    }
    *out = p;
    return VIT_OK;
}

void vit_plan_destroy(vit_plan* plan) { delete plan; }

int vit_plan_query(const vit_plan* plan, vit_plan_info* info) {
    if (!plan || !info) return VIT_EINVAL;
    std::memset(info, 0, sizeof(*info));
    info->S = plan->S;
    info->banded_ok = plan->bp.ok ? 1 : 0;
    info->n_consts = plan->bp.ok ? 1 : 0;
    info->n_extras = plan->bp.n_extras;
    info->max_window = plan->bp.max_window;
    info->group_window = plan->bp.W;
    info->reserved[0] = plan->bp.n_dense;
    info->reserved[1] = plan->bp.ok && plan->bp.floor_ok ? 1 : 0;
    info->reserved[2] = (plan->bp.ok && plan->bp.lo_affine ? 1 : 0) | (plan->bp.ok && plan->bp.pair_ok ? 2 : 0) |
                        (plan->bp.step_ok && vit::step_kernel_instantiated(plan->S, plan->bp.step_bw, plan->bp.step_kb) ? 4 : 0) |
                        (plan->bp.ok && plan->bp.wave_ok ? 8 : 0);
    info->consts[0] = plan->bp.c0;
    for (int k = 0; k < vit::kMaxExtras; ++k) info->extras[k] = k < plan->bp.n_extras ? plan->bp.extras[k] : -1;
    return VIT_OK;
}

size_t vit_plan_image_bytes(const vit_plan* plan) { return plan ? plan->L.bytes : 0; }

int vit_plan_upload(vit_plan* plan, void* device_image, size_t bytes, vit_stream stream) {
    if (!plan || !device_image) return VIT_EINVAL;
    if (bytes < plan->L.bytes || ((uintptr_t)device_image & 255) != 0) return VIT_EINVAL;
    hipError_t e = hipMemcpyAsync(device_image, plan->host_image.data(), plan->L.bytes, hipMemcpyHostToDevice,
                                  (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(e);
    plan->dev_image = static_cast<const uint8_t*>(device_image);
    int dev = 0, cus = 0;      // the thresholds below are "how many workgroups / waves fit the chip at once": from the device, not from "256"
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
        plan->n_cus = cus;
    return VIT_OK;
}

size_t vit_workspace_bytes(const vit_plan* plan, int64_t B, int64_t T) {
    if (!plan || B < 0 || T < 1) return 0;
    return ws_layout(plan, B, T).bytes;
}

// options of vit_plan_set_option
static int* tuning_field(Tuning& t, const char* key) {
    struct { const char* k; int Tuning::*f; } const tab[] = {
        {"forward_form", &Tuning::forward_form}, {"backtrace_form", &Tuning::backtrace_form},
        {"dense_songs", &Tuning::dense_songs}, {"dense_one_thread", &Tuning::dense_one_thread}, {"dense_form", &Tuning::dense_form},
        {"step_form", &Tuning::step_form}, {"bt_chunks", &Tuning::bt_chunks}, {"bt_warm", &Tuning::bt_warm},
        {"win_shift", &Tuning::win_shift}, {"wave_min_batch", &Tuning::wave_min_batch}, {"wave_two", &Tuning::wave_two},
        {"bt_fast_rows", &Tuning::bt_fast_rows}, {"bt_block_waves", &Tuning::bt_block_waves}, {"wave_history", &Tuning::wave_history}, {"wave_uniform", &Tuning::wave_uniform},
        {"timing", &Tuning::timing},
    };
    for (const auto& e : tab)
        if (std::strcmp(e.k, key) == 0) return &(t.*(e.f));
    return nullptr;
}

int vit_plan_set_option(vit_plan* plan, const char* key, int64_t value) {
    if (!plan || !key) return VIT_EINVAL;
    if (std::strcmp(key, "reset") == 0) { plan->tune = Tuning(); return VIT_OK; }
    int* f = tuning_field(plan->tune, key);
    if (!f || value < -1 || value > (int64_t)1 << 30) return VIT_EINVAL;
#ifndef VIT_TIMING_HOOKS
    if (f == &plan->tune.timing && value != 0) return VIT_EUNSUPPORTED;   // result-breaking ablations: timing builds only
#endif
    *f = (int)value;
    return VIT_OK;
}

// forward kernel family for (plan, algo, batch): 1 dense / step, 2 banded one song per workgroup, 3 wave; < 0 = status
static int resolve_family(const vit_plan* plan, int algo, int64_t B) {
    const int nwt = vit::banded_waves_for(plan->S);
    const bool group_ok = plan->bp.ok && (vit::scan_form_instantiated(plan->bp.W, nwt) ||
                                          (plan->bp.floor_ok && plan->S < nwt * 64 && vit::floor_form_instantiated(plan->bp.W, nwt)));
    const bool wave_ok = plan->bp.ok && plan->bp.wave_ok;
    const int ff = plan->tune.forward_form;
    const int64_t wmin = plan->tune.wave_min_batch > 0 ? plan->tune.wave_min_batch : wave_min_batch(plan);
    if (algo == VIT_ALGO_DENSE) return 1;
    if (algo == VIT_ALGO_WAVE) return wave_ok ? 3 : VIT_EUNSUPPORTED;
    if (algo == VIT_ALGO_GROUP) return group_ok ? 2 : VIT_EUNSUPPORTED;
    if (algo != VIT_ALGO_AUTO && algo != VIT_ALGO_BANDED) return VIT_EINVAL;
    const bool want_wave = wave_ok && ff != 5 && (ff == 4 || (ff == 0 && B >= wmin) || !group_ok);
    if (want_wave) return 3;
    if (group_ok) return 2;
    return algo == VIT_ALGO_AUTO ? 1 : VIT_EUNSUPPORTED;
}

// the fields of FwdArgs that depend on the plan and its options only
static void fwd_args_from_plan(const vit_plan* plan, vit::FwdArgs& a) {
    const Tuning& tn = plan->tune;
    a.image = plan->dev_image;
    a.S = plan->S;
    a.SP = plan->L.SP;
    a.S4 = plan->L.S4;
    a.SD = hist_stride(plan->S);
    a.W = plan->bp.W;
    a.n_extras = plan->bp.ok ? plan->bp.n_extras : 0;
    a.n_dense = plan->bp.ok ? plan->bp.n_dense : 0;
    for (int k = 0; k < vit::kMaxExtras; ++k) a.extras[k] = plan->bp.extras[k];
    a.c0 = plan->bp.c0;
    a.debug = tn.timing;          // 0 unless built with -DVIT_TIMING_HOOKS (vit_plan_set_option refuses it otherwise)
    a.fwd_form = tn.forward_form >= 1 && tn.forward_form <= 3 ? tn.forward_form : 0;
    a.dense_kt1 = tn.dense_one_thread;
    a.dense_form = tn.dense_form;
    a.step_form = tn.step_form;
    a.off_logpi = plan->L.off_logpi;
    a.off_A4 = plan->L.off_A4;
    a.off_lo = plan->L.off_lo;
    a.off_kind = plan->L.off_kind;
    a.off_tabA = plan->L.off_tabA;
    a.off_extraA = plan->L.off_extraA;
    a.off_denseA = plan->L.off_denseA;
    a.off_rowc = plan->L.off_rowc;
    a.off_lo2 = plan->L.off_lo2;
    a.off_tabP = plan->L.off_tabP;
    a.pair_ok = plan->bp.ok && plan->bp.pair_ok ? 1 : 0;
    a.floor_ok = plan->bp.ok && plan->bp.floor_ok ? 1 : 0;
    a.step_ok = plan->bp.step_ok ? 1 : 0;
    a.step_bw = plan->bp.step_bw;
    a.step_kb = plan->bp.step_kb;
    a.step_cn = plan->bp.step_cn;
    a.off_stepC = plan->L.off_stepC;
    a.off_Arow = plan->L.off_Arow;
    a.off_tabV = plan->L.off_tabV;
    a.wave_ok = plan->bp.ok && plan->bp.wave_ok ? 1 : 0;
    a.wave_npl = plan->bp.wave_npl;
    a.wave_dk = plan->bp.wave_dk;
    a.wave_u5 = tn.wave_uniform == 1 ? 0 : (tn.wave_uniform == 2 ? (plan->wave_u5 >= 1 ? 1 : 0) : (tn.wave_uniform == 3 && plan->wave_u5 == 2 ? 3 : plan->wave_u5));
    a.wave_flags = ((tn.wave_two & 3) == 1 ? 1 : ((tn.wave_two & 3) == 2 ? 2 : 0)) | (tn.wave_two & 4);   // (bit 2: a row carries its own scalars only -- A/B)
    a.win_shift = (plan->bp.ok && plan->bp.lo_affine) ? (plan->bp.lo_off & 3) : 0;
    a.win_shift2 = (plan->bp.ok && plan->bp.pair_ok && plan->bp.lo2_affine) ? (plan->bp.lo2_off & 3) : 0;
    if (tn.win_shift >= 0) a.win_shift = a.win_shift2 = tn.win_shift & 3;   // every value is functionally correct
}

size_t vit_workspace_bytes_for(const vit_plan* plan, int64_t B, int64_t T, int algo) {
    if (!plan || B < 0 || T < 1) return 0;
    const int family = resolve_family(plan, algo, B);
    if (family < 0) return 0;
    return ws_layout_family(plan, family, B, T).bytes;
}

int vit_forward(const vit_plan* plan, const void* logE, int emis_dtype, int64_t B, int64_t T,
                const int64_t* lengths, void* workspace, size_t workspace_bytes, float* loglik, int algo,
                vit_stream stream) {
    int rc = check_common(plan, B, T, workspace);
    if (rc != VIT_OK) return rc;
    if (!logE) return VIT_EINVAL;
    if (emis_dtype != VIT_F32 && emis_dtype != VIT_F16) return VIT_EINVAL;
    const int family = resolve_family(plan, algo, B);
    if (family < 0) return family;
    const Tuning& tn = plan->tune;
    const bool half = family == 3 && wave_half_applies(plan, T);
    if (family == 3 && tn.wave_history == 2 && !half && T >= 2) return VIT_EUNSUPPORTED;
    const WsLayout w = ws_layout_family(plan, family, B, T);
    if (workspace_bytes < w.bytes) return VIT_EWORKSPACE;
    if (B == 0) return VIT_OK;
    stamp_erase(plan, workspace);      // whatever this workspace held is gone once the kernel below starts; re-stamped on success

    uint8_t* ws = static_cast<uint8_t*>(workspace);
    vit::FwdArgs a{};
    fwd_args_from_plan(plan, a);
    a.logE = logE;
    a.lengths = lengths;
    a.hist = reinterpret_cast<float*>(ws + w.off_hist);
    a.fmax = reinterpret_cast<float*>(ws + w.off_fmax);
    a.last_state = reinterpret_cast<int32_t*>(ws + w.off_last);
    a.loglik = loglik;
    a.B = B;
    a.T = (int)T;
    a.hist_half = half ? 1 : 0;
    a.hist_rows = half ? (T + 1) / 2 : T;
    a.t_begin = 0;
    a.t_end = (int)T;

    FwdStamp st;
    st.ws = workspace;
    st.B = B;
    st.T = T;
    st.family = family;
    st.SD = a.SD;
    st.col0 = 0;
    st.mcol = plan->S;
    st.have_fmax = family == 2 ? 1 : 0;
    st.logE = logE;
    st.e_f16 = emis_dtype == VIT_F16 ? 1 : 0;
    hipError_t e;
    if (family == 3) {
        st.SD = vit::wave_hist_stride(plan->bp.wave_npl);
        st.col0 = st.SD - plan->S;
        st.mcol = 0;
        st.xcol0 = 1;
        st.aux_frames = half || (a.wave_flags & 4) ? 1 : vit::wave_aux_frames(plan->bp.wave_npl, plan->S, a.n_extras);
        st.have_fmax = 1;
        st.half = half ? 1 : 0;
        e = vit::launch_wave(a, emis_dtype == VIT_F16, (hipStream_t)stream);
    } else if (family == 2) {
        e = vit::launch_banded(a, emis_dtype == VIT_F16, (hipStream_t)stream);
    } else if (algo == VIT_ALGO_AUTO && a.step_ok && vit::step_kernel_instantiated(a.S, a.step_bw, a.step_kb) &&
               tn.step_form != 2) {
        // dense matrix with step structure (Durrieu): VIT_ALGO_DENSE still means the plain dense kernel
        e = vit::launch_step(a, emis_dtype == VIT_F16, (hipStream_t)stream);
    } else {
        // songs per workgroup share the streamed matrix (measured at S = 361, B = 1024: 1 -> 57, 2 -> 64, 4 -> 48 Mframes/s)
        const int ns = tn.dense_songs > 0 ? tn.dense_songs : (B >= 512 ? 2 : 1);
        e = vit::launch_dense(a, ns, emis_dtype == VIT_F16, (hipStream_t)stream);
    }
    if (e != hipSuccess) return hip_fail(e);
    stamp_put(plan, st);
    return VIT_OK;
}

// the fields of BtArgs that depend on the plan and its options only
static void bt_args_from_plan(const vit_plan* plan, vit::BtArgs& b) {
    b.image = plan->dev_image;
    b.S = plan->S;
    b.SP = plan->L.SP;
    b.W = plan->bp.ok ? plan->bp.W : 0;
    b.banded = plan->bp.ok ? 1 : 0;
    b.n_extras = plan->bp.ok ? plan->bp.n_extras : 0;
    b.n_dense = plan->bp.ok ? plan->bp.n_dense : 0;
    for (int k = 0; k < vit::kMaxExtras; ++k) b.extras[k] = plan->bp.extras[k];
    b.c0 = plan->bp.c0;
    b.bt_form = plan->tune.backtrace_form;
    b.no_fast_rows = plan->tune.bt_fast_rows == 1 ? 1 : 0;
    b.block_waves = plan->tune.bt_block_waves;
    b.lo_affine = plan->bp.lo_affine ? 1 : 0;
    b.lo_off = plan->bp.lo_off;
    for (int d = 0; d < vit::kMaxDenseRows; ++d) b.dense_rows[d] = d < plan->bp.n_dense ? plan->bp.dense_rows[d] : -1;
    b.off_lo = plan->L.off_lo;
    b.off_kind = plan->L.off_kind;
    b.off_tabA = plan->L.off_tabA;
    b.off_extraA = plan->L.off_extraA;
    b.off_tabX = plan->L.off_tabX;
    b.off_stepC = plan->L.off_stepC;
    b.step_ok = 0;
    if (plan->bp.step_ok) {   // band = dist / bw as a multiply-shift, verified here for every distance that can occur
        const int bw = plan->bp.step_bw;
        const int mult = (65536 + bw - 1) / bw;
        bool exact = true;
        for (int d = 0; d < 1024 && exact; ++d) exact = ((unsigned)(d * mult) >> 16) == (unsigned)(d / bw);
        if (exact) { b.step_ok = 1; b.step_kb = plan->bp.step_kb; b.step_mult = mult; b.step_cn = plan->bp.step_cn; }
    }
    b.off_denseA = plan->L.off_denseA;
    b.off_Arow = plan->L.off_Arow;
    b.off_rowc = plan->L.off_rowc;
}

int vit_forward_family(const vit_plan* plan, int64_t B, int algo) {
    if (!plan || B < 0) return VIT_EINVAL;
    return resolve_family(plan, algo, B);
}

static int backtrace_impl(const vit_plan* plan, const void* logE, int emis_dtype, bool check_e, int64_t B, int64_t T, const int64_t* lengths,
                          void* workspace, size_t workspace_bytes, int32_t* states, vit_stream stream);

int vit_backtrace(const vit_plan* plan, int64_t B, int64_t T, const int64_t* lengths, void* workspace,
                  size_t workspace_bytes, int32_t* states, int algo, vit_stream stream) {
    (void)algo;   // kept for ABI compatibility: the layout comes from what vit_forward recorded for this workspace
    return backtrace_impl(plan, nullptr, 0, false, B, T, lengths, workspace, workspace_bytes, states, stream);
}

int vit_backtrace_checked(const vit_plan* plan, const void* logE, int emis_dtype, int64_t B, int64_t T, const int64_t* lengths,
                          void* workspace, size_t workspace_bytes, int32_t* states, int algo, vit_stream stream) {
    (void)algo;
    if (!logE || (emis_dtype != VIT_F32 && emis_dtype != VIT_F16)) return VIT_EINVAL;
    return backtrace_impl(plan, logE, emis_dtype, true, B, T, lengths, workspace, workspace_bytes, states, stream);
}

static int backtrace_impl(const vit_plan* plan, const void* logE, int emis_dtype, bool check_e, int64_t B, int64_t T, const int64_t* lengths,
                          void* workspace, size_t workspace_bytes, int32_t* states, vit_stream stream) {
    int rc = check_common(plan, B, T, workspace);
    if (rc != VIT_OK) return rc;
    if (!states) return VIT_EINVAL;
    if (B == 0) return VIT_OK;
    FwdStamp st;
    if (!stamp_get(plan, workspace, &st) || st.B != B || st.T != T) return VIT_ENOFORWARD;   // no matching vit_forward
    // the emission tensor must be the one the forward pass decoded (a half history reads it again through the recorded pointer)
    if (check_e && (st.logE != logE || st.e_f16 != (emis_dtype == VIT_F16 ? 1 : 0))) return VIT_EINVAL;
    const Tuning& tn = plan->tune;
    const WsLayout w = st.half ? ws_layout_hist(B, (size_t)((T + 1) / 2), (size_t)st.SD) : ws_layout_hist(B, (size_t)T, (size_t)st.SD);
    if (workspace_bytes < w.bytes) return VIT_EWORKSPACE;
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    vit::BtArgs b{};
    b.hist = reinterpret_cast<const float*>(ws + w.off_hist);
    b.last_state = reinterpret_cast<const int32_t*>(ws + w.off_last);
    b.lengths = lengths;
    b.states = states;
    b.entry = reinterpret_cast<int32_t*>(ws + w.off_entry);
    b.B = B;
    b.T = (int)T;
    bt_args_from_plan(plan, b);
    b.SD = st.SD;
    b.col0 = st.col0;
    b.mcol = st.mcol;
    b.xcol0 = st.xcol0;
    b.aux_frames = st.aux_frames;
    b.have_fmax = st.have_fmax;
    b.states_stride = T;
    if (st.family == 3 && !(b.banded && b.n_dense == 0)) return VIT_EINVAL;   // (cannot happen: wave_ok implies both)
    b.hist_rows = T;
    b.counters = reinterpret_cast<int32_t*>(ws + w.off_cnt);          // event counts of this back-trace, chunk flags of the lane form
    b.mask = reinterpret_cast<uint32_t*>(ws + w.off_mask);
    {
        hipError_t ez = hipMemsetAsync(ws + w.off_cnt, 0, w.off_last - w.off_cnt, (hipStream_t)stream);
        if (ez != hipSuccess) return hip_fail(ez);
    }
    if (st.half) {
        b.hist_half = 1;
        b.hist_rows = (T + 1) / 2;
        b.mcol_odd = 1 + b.n_extras;
        b.xcol0_odd = 2 + b.n_extras;
        b.logE = st.logE;
        b.e_f16 = st.e_f16;
    }
    if (st.half) {
        b.chunks = vit::sparse_backtrace_chunks(B, (int)T, plan->n_cus);
        b.warm = vit::kBtWarmSparse;
        if (tn.bt_chunks >= 1 && tn.bt_chunks <= vit::kBtMaxChunks) b.chunks = tn.bt_chunks;
        if (tn.bt_warm >= 0) b.warm = tn.bt_warm;
        hipError_t eh = vit::launch_backtrace_half(b, (hipStream_t)stream);
        return eh == hipSuccess ? VIT_OK : hip_fail(eh);
    }
    if (b.bt_form == 4) {                                             // one (song, chunk) stream per lane
        if (!vit::lane_backtrace_applies(b)) return VIT_EUNSUPPORTED;
        b.warm = tn.bt_warm >= 0 ? tn.bt_warm : vit::kBtWarmSparse;
        b.chunks = vit::lane_backtrace_chunks(B, (int)T, plan->n_cus, b.warm);
        if (tn.bt_chunks >= 1 && tn.bt_chunks <= vit::kLaneMaxChunks) b.chunks = tn.bt_chunks;
        hipError_t el = vit::launch_backtrace_lane(b, (hipStream_t)stream);
        return el == hipSuccess ? VIT_OK : hip_fail(el);
    }
    const bool sparse = b.bt_form == 0 && vit::sparse_backtrace_applies(b);
    b.chunks = sparse ? vit::sparse_backtrace_chunks(B, (int)T, plan->n_cus) : vit::backtrace_chunks(B, (int)T);
    b.warm = sparse ? vit::kBtWarmSparse : vit::kBtWarm;
    // test hooks (vit_plan_set_option): force the chunking / warm-up so that the verify-and-repair pass is exercised
    if (tn.bt_chunks >= 1 && tn.bt_chunks <= vit::kBtMaxChunks) b.chunks = tn.bt_chunks;
    if (tn.bt_warm >= 0) b.warm = tn.bt_warm;
    hipError_t e = vit::launch_backtrace(b, (hipStream_t)stream);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_backtrace_counters(const vit_plan* plan, int64_t B, int64_t T, const void* workspace, size_t* offset, int32_t* n_per_song) {
    if (!plan || !workspace || !offset || !n_per_song) return VIT_EINVAL;
    FwdStamp st;
    if (!stamp_get(plan, workspace, &st) || st.B != B || st.T != T) return VIT_ENOFORWARD;
    const WsLayout w = st.half ? ws_layout_hist(B, (size_t)((T + 1) / 2), (size_t)st.SD) : ws_layout_hist(B, (size_t)T, (size_t)st.SD);
    *offset = w.off_cnt;
    *n_per_song = vit::kBtCounters;
    return VIT_OK;
}

int vit_decode(const vit_plan* plan, const void* logE, int emis_dtype, int64_t B, int64_t T,
               const int64_t* lengths, void* workspace, size_t workspace_bytes, int32_t* states, float* loglik,
               int algo, vit_stream stream) {
    if (!states) return VIT_EINVAL;
    int rc = vit_forward(plan, logE, emis_dtype, B, T, lengths, workspace, workspace_bytes, loglik, algo, stream);
    if (rc != VIT_OK) return rc;
    return vit_backtrace(plan, B, T, lengths, workspace, workspace_bytes, states, algo, stream);
}


// ---------------------------------------------------------------------------------------------------------------------
// Bounded-workspace decode (wave-form plans).  The reference keeps T1 / T2 for ONE song (tonet/for_paper.py:1852-1853); the
// batched decode above keeps a delta history for the whole batch (46 MB per song at T = 30000).  Here the history never
// exists at once: pass 1 runs the forward recursion over all T frames and keeps one delta row per segment of K frames (the row
// in front of the segment) plus the terminal state; pass 2 walks the segments from the last to the first, re-runs the
// forward kernel over one segment from its checkpoint row into a K-row buffer and back-traces it from the state the segment
// behind it decided at its first frame.  Exact by construction (the same kernels, the same sums); twice the forward work.
namespace {

struct CkLayout {
    int64_t nseg;
    size_t off_ckpt, off_seg, off_cnt, off_last, off_entry, off_slen, off_slast, bytes;
};
CkLayout ck_layout(const vit_plan* p, int64_t B, int64_t T, int64_t K) {
    CkLayout c;
    K = K > T ? T : K;                                                           // (a segment longer than the songs: one segment of T frames)
    const size_t sd = (size_t)vit::wave_hist_stride(p->bp.wave_npl) * sizeof(float);
    c.nseg = (T + K - 1) / K;
    c.off_ckpt = 0;                                                              // [B][nseg] rows: nseg - 1 checkpoints + the scratch row
    c.off_seg = align256((size_t)B * (size_t)c.nseg * sd);                       // [B][K + 1] rows of the segment being walked
    c.off_cnt = c.off_seg + align256((size_t)B * (size_t)(K + 1) * sd);
    c.off_last = c.off_cnt + align256((size_t)B * 64 * sizeof(float));
    c.off_entry = c.off_last + align256((size_t)B * sizeof(int32_t));
    c.off_slen = c.off_entry + align256((size_t)B * vit::kBtMaxChunks * sizeof(int32_t));
    c.off_slast = c.off_slen + align256((size_t)B * sizeof(int64_t));
    c.bytes = c.off_slast + align256((size_t)B * sizeof(int32_t));
    return c;
}
bool ck_supported(const vit_plan* p, int64_t K) { return p->bp.ok && p->bp.wave_ok && K >= 64 && K <= (int64_t)1 << 24; }

}  // namespace

size_t vit_workspace_bytes_checkpointed(const vit_plan* plan, int64_t B, int64_t T, int64_t segment_frames) {
    if (!plan || B < 0 || T < 1 || !ck_supported(plan, segment_frames)) return 0;
    return ck_layout(plan, B, T, segment_frames).bytes;
}

int vit_decode_checkpointed(const vit_plan* plan, const void* logE, int emis_dtype, int64_t B, int64_t T, const int64_t* lengths,
                            void* workspace, size_t workspace_bytes, int32_t* states, float* loglik, int64_t segment_frames,
                            vit_stream stream) {
    int rc = check_common(plan, B, T, workspace);
    if (rc != VIT_OK) return rc;
    if (!logE || !states) return VIT_EINVAL;
    if (emis_dtype != VIT_F32 && emis_dtype != VIT_F16) return VIT_EINVAL;
    if (!ck_supported(plan, segment_frames)) return segment_frames < 64 || segment_frames > (int64_t)1 << 24 ? VIT_EINVAL : VIT_EUNSUPPORTED;
    const int64_t K = segment_frames > T ? T : segment_frames;
    const CkLayout c = ck_layout(plan, B, T, K);
    if (workspace_bytes < c.bytes) return VIT_EWORKSPACE;
    if (B == 0) return VIT_OK;
    stamp_erase(plan, workspace);
    hipStream_t st = (hipStream_t)stream;
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    const bool f16 = emis_dtype == VIT_F16;
    const int SDW = vit::wave_hist_stride(plan->bp.wave_npl);
    hipError_t e;
    if (lengths) {     // frames past a song's end: -1 (segments a song does not reach are skipped, not written)
        e = hipMemsetAsync(states, 0xff, (size_t)B * (size_t)T * sizeof(int32_t), st);
        if (e != hipSuccess) return hip_fail(e);
    }
    int32_t* counters = reinterpret_cast<int32_t*>(ws + c.off_cnt);
    e = hipMemsetAsync(counters, 0, (size_t)B * vit::kBtCounters * sizeof(int32_t), st);
    if (e != hipSuccess) return hip_fail(e);

    // ---- pass 1: checkpoint rows + terminal state
    vit::FwdArgs a{};
    fwd_args_from_plan(plan, a);
    a.logE = logE;
    a.lengths = lengths;
    a.hist = reinterpret_cast<float*>(ws + c.off_ckpt);
    a.fmax = reinterpret_cast<float*>(ws + c.off_cnt);
    a.last_state = reinterpret_cast<int32_t*>(ws + c.off_last);
    a.loglik = loglik;
    a.B = B;
    a.T = (int)T;
    a.hist_rows = c.nseg;
    a.ckpt_every = (int)K;
    a.t_begin = 0;
    a.t_end = (int)T;
    e = vit::launch_wave(a, f16, st);
    if (e != hipSuccess) return hip_fail(e);

    // ---- pass 2: segments, last to first
    vit::BtArgs b{};
    bt_args_from_plan(plan, b);
    b.SD = SDW;
    b.col0 = SDW - plan->S;
    b.mcol = 0;
    b.xcol0 = 1;
    b.aux_frames = 1;       // (a segment's sub-problem ends one frame behind the rows its forward pass wrote: every frame's scalars from its own row)
    b.have_fmax = 1;
    b.hist = reinterpret_cast<const float*>(ws + c.off_seg);
    b.hist_rows = K + 1;
    b.last_state = reinterpret_cast<const int32_t*>(ws + c.off_slast);
    b.lengths = reinterpret_cast<const int64_t*>(ws + c.off_slen);
    b.entry = reinterpret_cast<int32_t*>(ws + c.off_entry);
    b.B = B;
    b.states_stride = T;
    b.skip_nonpositive = 1;
    b.counters = counters;
    b.bt_form = 0;
    if (!vit::sparse_backtrace_applies(b)) return VIT_EUNSUPPORTED;
    for (int64_t sgm = c.nseg - 1; sgm >= 0; --sgm) {
        const int64_t s0 = sgm * K, e0 = s0 + K < T ? s0 + K : T;
        vit::FwdArgs f = a;
        f.ckpt_every = 0;
        f.hist = reinterpret_cast<float*>(ws + c.off_seg);
        f.hist_rows = K + 1;
        f.loglik = nullptr;
        f.t_begin = (int)s0;
        f.t_end = (int)e0;
        f.init_rows = sgm > 0 ? reinterpret_cast<const float*>(ws + c.off_ckpt) + (size_t)(sgm - 1) * SDW : nullptr;
        f.init_stride = (int64_t)c.nseg * SDW;
        e = vit::launch_wave(f, f16, st);
        if (e != hipSuccess) return hip_fail(e);
        e = vit::launch_segment_prep(lengths, B, (int)T, (int)s0, (int)e0, states, reinterpret_cast<const int32_t*>(ws + c.off_last),
                                     reinterpret_cast<int64_t*>(ws + c.off_slen), reinterpret_cast<int32_t*>(ws + c.off_slast), st);
        if (e != hipSuccess) return hip_fail(e);
        b.T = (int)(e0 < T ? e0 - s0 + 1 : e0 - s0);      // the frame behind the segment is the sub-problem's terminal frame
        b.states = states + s0;
        b.chunks = vit::sparse_backtrace_chunks(B, b.T, plan->n_cus);
        b.warm = vit::kBtWarmSparse;
        e = vit::launch_backtrace_sparse(b, st);
        if (e != hipSuccess) return hip_fail(e);
    }
    return VIT_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Packed (ragged) decode.  The reference decodes every recording whole with its own T (tonet/for_paper.py:2304-2309); a padded
// [B, T_max, S] tensor wastes memory on the padding and a launch with one wavefront per song costs its LONGEST song.  Here the
// emissions of B songs are one [sum T_b, S] buffer, the history and the states are packed the same way, and the forward pass
// runs n_slots <= 8 waves per CU, each walking a host-packed list of songs back to back (longest-first greedy bins by frames:
// the rule of sharded.shard_by_length), so that every wave carries about the same number of frames.  The back-trace is the lane
// form (backtrace_lane.hip): every song is cut into chunks of about equal length (total frames / resident LANES), one lane per
// chunk -- the one-stream-per-wavefront kernels hold 16 streams per CU, fewer than a ragged batch of thousands of songs needs
// (3250 songs: a second round of waves, 14 instead of 6 ms).
namespace {

struct PkLayout {
    int64_t n_slots, max_waves;
    size_t off_hist, off_cnt, off_mask, off_last, off_entry, off_offsets, off_slot_begin, off_slot_songs, off_wave_song, off_chunk_base, bytes;
    size_t tables_bytes;      // offsets .. chunk_base: one contiguous upload
};
inline int64_t pk_slots(const vit_plan* p, int64_t B) { const int64_t cap = 8 * (int64_t)p->n_cus; return B < cap ? B : cap; }
inline int64_t pk_max_waves(const vit_plan* p, int64_t B) { return B + 16 * 64 * (int64_t)p->n_cus; }     // (song, chunk) streams of the back-trace: one per lane
PkLayout pk_layout(const vit_plan* p, int64_t B, int64_t N) {
    PkLayout k;
    const size_t sd = (size_t)vit::wave_hist_stride(p->bp.wave_npl) * sizeof(float);
    k.n_slots = pk_slots(p, B);
    k.max_waves = pk_max_waves(p, B);
    k.off_hist = 0;
    k.off_cnt = align256((size_t)N * sd);
    k.off_mask = k.off_cnt + align256((size_t)B * vit::kBtCounters * sizeof(int32_t));
    k.off_last = k.off_mask + align256((size_t)B * vit::kLaneMaskWords * sizeof(uint32_t));
    k.off_entry = k.off_last + align256((size_t)B * sizeof(int32_t));
    k.off_offsets = k.off_entry + align256((size_t)k.max_waves * sizeof(int32_t));
    k.off_slot_begin = k.off_offsets + align256((size_t)(B + 1) * sizeof(int64_t));
    k.off_slot_songs = k.off_slot_begin + align256((size_t)(k.n_slots + 1) * sizeof(int32_t));
    k.off_wave_song = k.off_slot_songs + align256((size_t)B * sizeof(int32_t));
    k.off_chunk_base = k.off_wave_song + align256((size_t)k.max_waves * sizeof(int32_t));
    k.bytes = k.off_chunk_base + align256((size_t)(B + 1) * sizeof(int32_t));
    k.tables_bytes = k.bytes - k.off_offsets;
    return k;
}

}  // namespace

size_t vit_workspace_bytes_packed(const vit_plan* plan, int64_t B, int64_t total_frames) {
    if (!plan || B < 0 || total_frames < 0 || !(plan->bp.ok && plan->bp.wave_ok)) return 0;
    return pk_layout(plan, B, total_frames).bytes;
}

int vit_decode_packed(const vit_plan* plan, const void* logE, int emis_dtype, int64_t B, const int64_t* offsets, void* workspace,
                      size_t workspace_bytes, int32_t* states, float* loglik, vit_stream stream) {
    if (!plan || !workspace || !offsets) return VIT_EINVAL;
    if (B < 0 || B > (int64_t)1 << 30) return VIT_EINVAL;
    if (!plan->dev_image) return VIT_ENOTUPLOADED;
    if (((uintptr_t)workspace & 255) != 0) return VIT_EINVAL;
    if (emis_dtype != VIT_F32 && emis_dtype != VIT_F16) return VIT_EINVAL;
    if (!(plan->bp.ok && plan->bp.wave_ok)) return VIT_EUNSUPPORTED;
    if (offsets[0] != 0) return VIT_EINVAL;
    for (int64_t b = 0; b < B; ++b) {
        const int64_t tb = offsets[b + 1] - offsets[b];
        if (tb < 1 || tb > (int64_t)1 << 30) return VIT_EINVAL;      // every song holds at least one frame
    }
    const int64_t N = offsets[B];
    if (B == 0) return VIT_OK;
    if (!logE || !states) return VIT_EINVAL;
    PkLayout k = pk_layout(plan, B, N);
    if (workspace_bytes < k.bytes) return VIT_EWORKSPACE;
    {   // fewer slots than songs when the batch is short of frames: a slot's load should not fall below the longest song, which bounds
        // the launch anyway (1623 songs of 7500..30000 frames: 1024 slots of ~30000 frames, one wave per SIMD, instead of 1623 waves
        // of which the longest share their SIMDs to the end)
        int64_t tmax = 1;
        for (int64_t b = 0; b < B; ++b) tmax = std::max<int64_t>(tmax, offsets[b + 1] - offsets[b]);
        const int64_t by_frames = std::max<int64_t>(1, N / tmax);
        k.n_slots = std::min<int64_t>(k.n_slots, by_frames);
    }
    stamp_erase(plan, workspace);
    hipStream_t st = (hipStream_t)stream;
    uint8_t* ws = static_cast<uint8_t*>(workspace);

    // ---- host tables in the plan's pinned staging buffer (the previous call's upload must have read it)
    {
        std::lock_guard<std::mutex> g(plan->mu);
        if (plan->pk_event) { hipError_t ew = hipEventSynchronize(plan->pk_event); if (ew != hipSuccess) return hip_fail(ew); }
        else { hipError_t ec = hipEventCreateWithFlags(&plan->pk_event, hipEventDisableTiming); if (ec != hipSuccess) return hip_fail(ec); }
        if (plan->pk_host_bytes < k.tables_bytes) {
            if (plan->pk_host) (void)hipHostFree(plan->pk_host);
            plan->pk_host = nullptr;
            plan->pk_host_bytes = 0;
            hipError_t ea = hipHostMalloc(&plan->pk_host, k.tables_bytes, hipHostMallocDefault);
            if (ea != hipSuccess) return hip_fail(ea);
            plan->pk_host_bytes = k.tables_bytes;
        }
    }
    uint8_t* hb = static_cast<uint8_t*>(plan->pk_host);
    std::memset(hb, 0, k.tables_bytes);
    int64_t* h_off = reinterpret_cast<int64_t*>(hb + (k.off_offsets - k.off_offsets));
    int32_t* h_slot_begin = reinterpret_cast<int32_t*>(hb + (k.off_slot_begin - k.off_offsets));
    int32_t* h_slot_songs = reinterpret_cast<int32_t*>(hb + (k.off_slot_songs - k.off_offsets));
    int32_t* h_wave_song = reinterpret_cast<int32_t*>(hb + (k.off_wave_song - k.off_offsets));
    int32_t* h_chunk_base = reinterpret_cast<int32_t*>(hb + (k.off_chunk_base - k.off_offsets));
    std::memcpy(h_off, offsets, (size_t)(B + 1) * sizeof(int64_t));
    int64_t max_chunks = 1;
    try {
        // forward slots: longest song first into the slot with the fewest frames (ties: fewest songs, lowest slot)
        std::vector<int32_t> order((size_t)B);
        for (int64_t b = 0; b < B; ++b) order[(size_t)b] = (int32_t)b;
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return offsets[x + 1] - offsets[x] > offsets[y + 1] - offsets[y]; });
        typedef std::pair<std::pair<int64_t, int32_t>, int32_t> Load;       // ((frames, songs), slot)
        std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
        for (int32_t sl = 0; sl < (int32_t)k.n_slots; ++sl) heap.push(Load{{0, 0}, sl});
        std::vector<int32_t> slot_of((size_t)B);
        std::vector<int32_t> count((size_t)k.n_slots, 0);
        for (int32_t sng : order) {
            Load l = heap.top();
            heap.pop();
            slot_of[(size_t)sng] = l.second;
            ++count[(size_t)l.second];
            l.first.first += offsets[sng + 1] - offsets[sng];
            ++l.first.second;
            heap.push(l);
        }
        h_slot_begin[0] = 0;
        for (int64_t sl = 0; sl < k.n_slots; ++sl) h_slot_begin[sl + 1] = h_slot_begin[sl] + count[(size_t)sl];
        std::vector<int32_t> fill(h_slot_begin, h_slot_begin + k.n_slots);
        for (int32_t sng : order) h_slot_songs[fill[(size_t)slot_of[(size_t)sng]]++] = sng;     // a slot walks its songs longest first
        // back-trace streams (one per lane: backtrace_lane.hip): chunks of about (total frames / resident lanes) frames, never shorter
        // than four warm-ups, at most kLaneMaxChunks per song
        const int64_t resident = 16 * 64 * (int64_t)plan->n_cus;
        int64_t cf = (N + resident - 1) / resident, tmax = 0;
        for (int64_t b = 0; b < B; ++b) tmax = std::max<int64_t>(tmax, offsets[b + 1] - offsets[b]);
        cf = std::max<int64_t>(cf, 4 * vit::kBtWarmSparse);
        cf = std::max<int64_t>(cf, (tmax + vit::kLaneMaxChunks - 1) / vit::kLaneMaxChunks);
        int64_t w = 0;
        for (int64_t b = 0; b < B; ++b) {
            const int64_t tb = offsets[b + 1] - offsets[b];
            int64_t c = (tb + cf / 2) / cf;
            c = c < 1 ? 1 : (c > vit::kLaneMaxChunks ? vit::kLaneMaxChunks : c);
            h_chunk_base[b] = (int32_t)w;
            for (int64_t q = 0; q < c; ++q) h_wave_song[w + q] = (int32_t)b;
            w += c;
            max_chunks = c > max_chunks ? c : max_chunks;
        }
        h_chunk_base[B] = (int32_t)w;
        if (w > k.max_waves) return VIT_EINVAL;      // (cannot happen: sum of round(T_b / cf) <= B + N / cf <= B + 16 n_cus)
    } catch (const std::bad_alloc&) {
        return VIT_ENOMEM;
    }
    const int n_waves = h_chunk_base[B];
    hipError_t e = hipMemcpyAsync(ws + k.off_offsets, hb, k.tables_bytes, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return hip_fail(e);
    e = hipEventRecord(plan->pk_event, st);
    if (e != hipSuccess) return hip_fail(e);
    e = hipMemsetAsync(ws + k.off_cnt, 0, k.off_last - k.off_cnt, st);
    if (e != hipSuccess) return hip_fail(e);

    // ---- forward: one wave per slot
    vit::FwdArgs a{};
    fwd_args_from_plan(plan, a);
    a.logE = logE;
    a.lengths = nullptr;
    a.hist = reinterpret_cast<float*>(ws + k.off_hist);
    a.fmax = nullptr;
    a.last_state = reinterpret_cast<int32_t*>(ws + k.off_last);
    a.loglik = loglik;
    a.B = B;
    a.T = 1;                      // (unused by the packed kernel: a song's rows come from the offsets)
    a.hist_rows = 0;
    a.t_begin = 0;
    a.t_end = 1;
    a.offsets = reinterpret_cast<const int64_t*>(ws + k.off_offsets);
    a.n_slots = (int)k.n_slots;
    a.slot_begin = reinterpret_cast<const int32_t*>(ws + k.off_slot_begin);
    a.slot_songs = reinterpret_cast<const int32_t*>(ws + k.off_slot_songs);
    e = vit::launch_wave(a, emis_dtype == VIT_F16, st);
    if (e != hipSuccess) return hip_fail(e);

    // ---- back-trace: one wave per (song, chunk)
    vit::BtArgs b{};
    bt_args_from_plan(plan, b);
    b.SD = vit::wave_hist_stride(plan->bp.wave_npl);
    b.col0 = b.SD - plan->S;
    b.mcol = 0;
    b.xcol0 = 1;
    b.aux_frames = (a.wave_flags & 4) ? 1 : vit::wave_aux_frames(plan->bp.wave_npl, plan->S, b.n_extras);
    b.have_fmax = 1;
    b.hist = reinterpret_cast<const float*>(ws + k.off_hist);
    b.hist_rows = 0;
    b.last_state = reinterpret_cast<const int32_t*>(ws + k.off_last);
    b.lengths = nullptr;
    b.states = states;
    b.states_stride = 0;
    b.entry = reinterpret_cast<int32_t*>(ws + k.off_entry);
    b.B = B;
    b.T = 1;
    b.counters = reinterpret_cast<int32_t*>(ws + k.off_cnt);
    b.mask = reinterpret_cast<uint32_t*>(ws + k.off_mask);
    b.offsets = a.offsets;
    b.wave_song = reinterpret_cast<const int32_t*>(ws + k.off_wave_song);
    b.chunk_base = reinterpret_cast<const int32_t*>(ws + k.off_chunk_base);
    b.n_waves = n_waves;
    b.chunks = (int)max_chunks;
    b.warm = plan->tune.bt_warm >= 0 ? plan->tune.bt_warm : vit::kBtWarmSparse;
    b.bt_form = 4;
    if (!vit::lane_backtrace_applies(b)) return VIT_EUNSUPPORTED;
    e = vit::launch_backtrace_lane(b, st);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_voicing_map(const int32_t* states, int64_t n, int32_t n_bins, uint8_t* voiced, int32_t* bins,
                    vit_stream stream) {
    if (n < 0 || n_bins < 1 || (n > 0 && (!states || !voiced || !bins))) return VIT_EINVAL;
    hipError_t e = vit::launch_voicing_map(states, n, n_bins, voiced, bins, (hipStream_t)stream);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_obs_shaun(const float* logits, int64_t n_frames, int32_t n_bins, int32_t spw, double threshold_logit,
                  double offset, double scale, float* logE, vit_stream stream) {
    if (n_frames < 0 || n_bins < 2 || (n_frames > 0 && (!logits || !logE))) return VIT_EINVAL;
    hipError_t e = vit::launch_obs_shaun(logits, n_frames, n_bins, spw, threshold_logit, offset, scale, logE, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return VIT_EUNSUPPORTED;
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_obs_softmax(const float* logits, int64_t n_frames, int32_t n_bins, int32_t spw, float* logE, vit_stream stream) {
    if (n_frames < 0 || n_bins < 2 || (n_frames > 0 && (!logits || !logE))) return VIT_EINVAL;
    hipError_t e = vit::launch_obs_softmax(logits, n_frames, n_bins, spw, logE, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return VIT_EUNSUPPORTED;
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_obs_softmax_scaled(const float* logits, int64_t n_frames, int32_t n_bins, int32_t spw, double unvoiced_logit,
                           const float* prior, float* logE, vit_stream stream) {
    if (n_frames < 0 || n_bins < 2 || (n_frames > 0 && (!logits || !logE))) return VIT_EINVAL;
    hipError_t e = vit::launch_obs_softmax_scaled(logits, n_frames, n_bins, spw, unvoiced_logit, prior, logE, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return VIT_EUNSUPPORTED;
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_snippets_append(const float* snippets, int32_t n_snippets, int32_t n_channels, int32_t n_frames, int32_t mode,
                        float* rows_out, int64_t n_rows, vit_stream stream) {
    if (n_snippets < 0 || n_channels < 2 || n_frames < 1 || (mode != 0 && mode != 1) || n_rows < 0 ||
        n_rows > (int64_t)n_snippets * n_frames || (n_rows > 0 && (!snippets || !rows_out)))
        return VIT_EINVAL;
    hipError_t e = vit::launch_snippets_append(snippets, n_snippets, n_channels, n_frames, mode, rows_out, n_rows, (hipStream_t)stream);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_voicing_notes(const int32_t* states, int64_t n, int32_t n_bins, const float* note_range, uint8_t* voiced,
                      int32_t* bins, float* notes, float* notes_voiced, vit_stream stream) {
    if (n < 0 || n_bins < 1 || (n > 0 && (!states || !note_range))) return VIT_EINVAL;
    hipError_t e = vit::launch_voicing_notes(states, n, n_bins, note_range, voiced, bins, notes, notes_voiced, (hipStream_t)stream);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

/* DPP scan self-test used by tests/test_gpu_parity.py */
int vit_debug_scan(const float* vals, int n_waves, int mode, float* out_v, int32_t* out_i, vit_stream stream) {
    if (!vals || !out_v || !out_i || n_waves < 1) return VIT_EINVAL;
    hipError_t e = vit::launch_scan_selftest(vals, n_waves, mode, out_v, out_i, (hipStream_t)stream);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

}  // extern "C"
