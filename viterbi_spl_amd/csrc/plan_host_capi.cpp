// plan_host_capi.cpp -- C entry points over plan.cpp for CPU-only tests (no HIP).
// Exposes the analysis result and the packed device image so that tests can replay
// the banded decomposition on the host and compare it with the dense oracle.
#include <cstring>
#include <vector>

#include "plan.hpp"

extern "C" {

struct vph_plan {
    vit::BandedPlan bp;
    vit::ImageLayout L;
    std::vector<uint8_t> image;
};

vph_plan* vph_create(const float* logA_T, const float* log_pi, int S) {
    vph_plan* p = new vph_plan();
    p->bp = vit::analyze_banded(logA_T, S);
    if (!p->bp.ok) vit::analyze_step(logA_T, S, p->bp);
    p->L = vit::make_layout(S, p->bp);
    p->image.resize(p->L.bytes);
    vit::fill_image(logA_T, log_pi, p->bp, p->L, p->image.data());
    return p;
}
void vph_destroy(vph_plan* p) { delete p; }

// info[0..15]: ok, S, SP, W, n_extras, n_dense, max_window, extras[4], dense_rows[4], S4
void vph_info(const vph_plan* p, int* info, float* c0) {
    info[0] = p->bp.ok; info[1] = p->bp.S; info[2] = p->bp.SP; info[3] = p->bp.W;
    info[4] = p->bp.n_extras; info[5] = p->bp.n_dense; info[6] = p->bp.max_window;
    for (int k = 0; k < 4; ++k) { info[7 + k] = p->bp.extras[k]; info[11 + k] = p->bp.dense_rows[k]; }
    info[15] = p->L.S4 | (p->bp.pair_ok ? 0x10000 : 0) | (p->bp.floor_ok ? 0x20000 : 0) | (p->bp.step_ok ? 0x40000 : 0) |
               ((p->bp.step_ok ? p->bp.step_kb : 0) << 20) | ((p->bp.step_ok ? p->bp.step_bw : 0) << 24);
    *c0 = p->bp.c0;
}
// offsets[0..11]: logpi, A4, lo, kind, tabA, extraA, denseA, total bytes, Arow, rowc, lo2, tabP
void vph_offsets(const vph_plan* p, long long* off) {
    off[0] = p->L.off_logpi; off[1] = p->L.off_A4; off[2] = p->L.off_lo; off[3] = p->L.off_kind;
    off[4] = p->L.off_tabA; off[5] = p->L.off_extraA; off[6] = p->L.off_denseA; off[7] = p->L.bytes;
    off[8] = p->L.off_Arow; off[9] = p->L.off_rowc; off[10] = p->L.off_lo2; off[11] = p->L.off_tabP;
}
// wave form: w[0..5] = wave_ok, npl, proven half-width, instantiated half-width, floor_all_ok, offset of tabV
void vph_wave(const vph_plan* p, long long* w) {
    w[0] = p->bp.wave_ok; w[1] = p->bp.wave_npl; w[2] = p->bp.wave_d; w[3] = p->bp.wave_dk;
    w[4] = p->bp.floor_all_ok; w[5] = (long long)p->L.off_tabV;
}
void vph_image(const vph_plan* p, unsigned char* out) { std::memcpy(out, p->image.data(), p->image.size()); }

}  // extern "C"
