// capi.hip -- extern "C" entry points declared in include/viterbi_hip.h.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/viterbi_hip.h"
#include "kernels.hpp"
#include "plan.hpp"

struct vit_plan {
    int S = 0;
    vit::BandedPlan bp;
    vit::ImageLayout L;
    std::vector<uint8_t> host_image;
    const uint8_t* dev_image = nullptr;
};

namespace {

thread_local int g_last_hip_error = 0;

inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
inline int hist_stride(int S) { return (S + 5) / 4 * 4; }   // rows 16-byte aligned, at least two pad columns (frame max, scratch)

int hip_fail(hipError_t e) {
    g_last_hip_error = (int)e;
    return VIT_EHIP;
}

struct WsLayout {
    size_t off_hist, off_fmax, off_last, off_entry, bytes;
};

WsLayout ws_layout(int S, int64_t B, int64_t T) {
    WsLayout w;
    w.off_hist = 0;
    w.off_fmax = align256((size_t)B * (size_t)T * hist_stride(S) * sizeof(float));
    w.off_last = w.off_fmax + align256((size_t)B * 64 * sizeof(float));   // timing-experiment scratch
    w.off_entry = w.off_last + align256((size_t)B * sizeof(int32_t));
    w.bytes = w.off_entry + align256((size_t)B * vit::kBtMaxChunks * sizeof(int32_t));
    return w;
}

int check_common(const vit_plan* plan, int64_t B, int64_t T, const void* ws, size_t ws_bytes) {
    if (!plan || !ws) return VIT_EINVAL;
    if (B < 0 || T < 1 || T > (int64_t)1 << 30 || B > (int64_t)1 << 30) return VIT_EINVAL;
    if (!plan->dev_image) return VIT_ENOTUPLOADED;
    if (ws_bytes < ws_layout(plan->S, B, T).bytes) return VIT_EWORKSPACE;
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
                        (plan->bp.step_ok && vit::step_kernel_instantiated(plan->S, plan->bp.step_bw, plan->bp.step_kb) ? 4 : 0);
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
    return VIT_OK;
}

size_t vit_workspace_bytes(const vit_plan* plan, int64_t B, int64_t T) {
    if (!plan || B < 0 || T < 1) return 0;
    return ws_layout(plan->S, B, T).bytes;
}

static int resolve_algo(const vit_plan* plan, int algo) {
    const int nwt = vit::banded_waves_for(plan->S);
    const bool banded_possible = plan->bp.ok && (vit::scan_form_instantiated(plan->bp.W, nwt) ||
                                                 (plan->bp.floor_ok && plan->S < nwt * 64 && vit::floor_form_instantiated(plan->bp.W, nwt)));
    if (algo == VIT_ALGO_AUTO) return banded_possible ? VIT_ALGO_BANDED : VIT_ALGO_DENSE;
    if (algo == VIT_ALGO_BANDED) return banded_possible ? VIT_ALGO_BANDED : VIT_EUNSUPPORTED;
    if (algo == VIT_ALGO_DENSE) return VIT_ALGO_DENSE;
    return VIT_EINVAL;
}

int vit_forward(const vit_plan* plan, const void* logE, int emis_dtype, int64_t B, int64_t T,
                const int64_t* lengths, void* workspace, size_t workspace_bytes, float* loglik, int algo,
                vit_stream stream) {
    int rc = check_common(plan, B, T, workspace, workspace_bytes);
    if (rc != VIT_OK) return rc;
    if (!logE) return VIT_EINVAL;
    if (emis_dtype != VIT_F32 && emis_dtype != VIT_F16) return VIT_EINVAL;
    if (B == 0) return VIT_OK;
    const int requested = algo;
    algo = resolve_algo(plan, algo);
    if (algo < 0) return algo;

    const WsLayout w = ws_layout(plan->S, B, T);
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    vit::FwdArgs a{};
    a.image = plan->dev_image;
    a.logE = logE;
    a.lengths = lengths;
    a.hist = reinterpret_cast<float*>(ws + w.off_hist);
    a.fmax = reinterpret_cast<float*>(ws + w.off_fmax);
    a.last_state = reinterpret_cast<int32_t*>(ws + w.off_last);
    a.loglik = loglik;
    a.B = B;
    a.T = (int)T;
    a.S = plan->S;
    a.SP = plan->L.SP;
    a.S4 = plan->L.S4;
    a.SD = hist_stride(plan->S);
    a.W = plan->bp.W;
    a.n_extras = plan->bp.ok ? plan->bp.n_extras : 0;
    a.n_dense = plan->bp.ok ? plan->bp.n_dense : 0;
    for (int k = 0; k < vit::kMaxExtras; ++k) a.extras[k] = plan->bp.extras[k];
    a.c0 = plan->bp.c0;
    if (const char* dbg = std::getenv("VIT_DEBUG_FLAGS")) a.debug = std::atoi(dbg);  // timing experiments only
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
    a.win_shift = (plan->bp.ok && plan->bp.lo_affine) ? (plan->bp.lo_off & 3) : 0;
    a.win_shift2 = (plan->bp.ok && plan->bp.pair_ok && plan->bp.lo2_affine) ? (plan->bp.lo2_off & 3) : 0;
    if (const char* e = std::getenv("VIT_WIN_SHIFT")) a.win_shift = a.win_shift2 = std::atoi(e) & 3;   // timing experiments only

    hipError_t e;
    if (algo == VIT_ALGO_BANDED) {
        e = vit::launch_banded(a, emis_dtype == VIT_F16, (hipStream_t)stream);
    } else if (requested == VIT_ALGO_AUTO && a.step_ok && vit::step_kernel_instantiated(a.S, a.step_bw, a.step_kb) &&
               !(a.debug & 16384)) {
        // dense matrix with step structure (Durrieu): VIT_ALGO_DENSE still means the plain dense kernel
        e = vit::launch_step(a, emis_dtype == VIT_F16, (hipStream_t)stream);
    } else {
        int ns = B >= 512 ? 2 : 1;   // songs per workgroup share the streamed matrix (measured at S = 361, B = 1024: 1 -> 57, 2 -> 64, 4 -> 48 Mframes/s)
        if (const char* e = std::getenv("VIT_DENSE_NS")) ns = std::atoi(e);   // timing experiments only
        e = vit::launch_dense(a, ns, emis_dtype == VIT_F16, (hipStream_t)stream);
    }
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_backtrace(const vit_plan* plan, int64_t B, int64_t T, const int64_t* lengths, void* workspace,
                  size_t workspace_bytes, int32_t* states, int algo, vit_stream stream) {
    int rc = check_common(plan, B, T, workspace, workspace_bytes);
    if (rc != VIT_OK) return rc;
    if (!states) return VIT_EINVAL;
    algo = resolve_algo(plan, algo);
    if (algo < 0) return algo;
    if (B == 0) return VIT_OK;
    const WsLayout w = ws_layout(plan->S, B, T);
    uint8_t* ws = static_cast<uint8_t*>(workspace);
    vit::BtArgs b{};
    b.image = plan->dev_image;
    b.hist = reinterpret_cast<const float*>(ws + w.off_hist);
    b.last_state = reinterpret_cast<const int32_t*>(ws + w.off_last);
    b.lengths = lengths;
    b.states = states;
    b.entry = reinterpret_cast<int32_t*>(ws + w.off_entry);
    b.chunks = vit::backtrace_chunks(B, (int)T);
    b.warm = vit::kBtWarm;
    // test hooks: force the chunking / warm-up so that the verify-and-repair pass is exercised
    if (const char* e = std::getenv("VIT_BT_CHUNKS")) { int c = std::atoi(e); if (c >= 1 && c <= vit::kBtMaxChunks) b.chunks = c; }
    if (const char* e = std::getenv("VIT_BT_WARM")) { int g = std::atoi(e); if (g >= 0) b.warm = g; }
    b.B = B;
    b.T = (int)T;
    b.S = plan->S;
    b.SP = plan->L.SP;
    b.SD = hist_stride(plan->S);
    b.W = plan->bp.ok ? plan->bp.W : 0;
    b.banded = plan->bp.ok ? 1 : 0;
    b.n_extras = plan->bp.ok ? plan->bp.n_extras : 0;
    b.n_dense = plan->bp.ok ? plan->bp.n_dense : 0;
    for (int k = 0; k < vit::kMaxExtras; ++k) b.extras[k] = plan->bp.extras[k];
    b.c0 = plan->bp.c0;
    b.have_fmax = algo == VIT_ALGO_BANDED ? 1 : 0;
    if (const char* dbg = std::getenv("VIT_DEBUG_FLAGS")) b.debug = std::atoi(dbg);  // timing experiments only
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
    hipError_t e = vit::launch_backtrace(b, (hipStream_t)stream);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

int vit_decode(const vit_plan* plan, const void* logE, int emis_dtype, int64_t B, int64_t T,
               const int64_t* lengths, void* workspace, size_t workspace_bytes, int32_t* states, float* loglik,
               int algo, vit_stream stream) {
    if (!states) return VIT_EINVAL;
    int rc = vit_forward(plan, logE, emis_dtype, B, T, lengths, workspace, workspace_bytes, loglik, algo, stream);
    if (rc != VIT_OK) return rc;
    return vit_backtrace(plan, B, T, lengths, workspace, workspace_bytes, states, algo, stream);
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

/* not part of the public header: DPP scan self-test used by tests/test_gpu_parity.py */
int vit_debug_scan(const float* vals, int n_waves, int mode, float* out_v, int32_t* out_i, vit_stream stream) {
    if (!vals || !out_v || !out_i || n_waves < 1) return VIT_EINVAL;
    hipError_t e = vit::launch_scan_selftest(vals, n_waves, mode, out_v, out_i, (hipStream_t)stream);
    return e == hipSuccess ? VIT_OK : hip_fail(e);
}

}  // extern "C"
