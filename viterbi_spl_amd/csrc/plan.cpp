// plan.cpp -- see plan.hpp.  Host-only C++17.
#include "plan.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>

namespace vit {

static inline uint32_t f2u(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

static inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

static const int kWidths[] = {16, 32, 64, 84, 96, 128};  // window widths the banded kernels are instantiated for

BandedPlan analyze_banded(const float* A, int S) {
    BandedPlan bp;
    bp.S = S;
    bp.SP = (S + 63) / 64 * 64;
    bp.lo.assign(bp.SP, 0);
    bp.kind.assign(bp.SP, -2);
    bp.rowc.assign(bp.SP, 0.f);
    if (S < 32) return bp;  // nothing to gain; the dense kernel handles it

    // 1. modal bit pattern of every row = its row constant c_j
    std::vector<uint32_t> row_mode(S);
    std::vector<uint32_t> tmp(S);
    for (int j = 0; j < S; ++j) {
        for (int i = 0; i < S; ++i) tmp[i] = f2u(A[(size_t)j * S + i]);
        std::sort(tmp.begin(), tmp.end());
        int best_run = 0, run = 0;
        uint32_t best_val = tmp[0];
        for (int i = 0; i < S; ++i) {
            run = (i > 0 && tmp[i] == tmp[i - 1]) ? run + 1 : 1;
            if (run > best_run) { best_run = run; best_val = tmp[i]; }
        }
        row_mode[j] = best_val;
        if ((best_val & 0x7f800000u) == 0x7f800000u && (best_val & 0x007fffffu) != 0u) return bp;  // NaN
        std::memcpy(&bp.rowc[j], &best_val, 4);
    }
    std::map<uint32_t, int> votes;
    for (int j = 0; j < S; ++j) votes[row_mode[j]]++;
    uint32_t c0 = row_mode[0];
    int c0_rows = 0;
    for (auto& kv : votes) if (kv.second > c0_rows) { c0_rows = kv.second; c0 = kv.first; }
    std::memcpy(&bp.c0, &c0, 4);

    // 2. extra columns: exceptions (w.r.t. the row's own constant) shared by more than a quarter of the rows
    std::vector<int> col_exc(S, 0);
    for (int j = 0; j < S; ++j)
        for (int i = 0; i < S; ++i) col_exc[i] += (f2u(A[(size_t)j * S + i]) != row_mode[j]);
    std::vector<int> cand;
    for (int i = 0; i < S; ++i) if (col_exc[i] > S / 4) cand.push_back(i);
    if ((int)cand.size() > kMaxExtras) return bp;
    bp.n_extras = (int)cand.size();
    for (int k = 0; k < bp.n_extras; ++k) bp.extras[k] = cand[k];
    auto is_extra = [&](int i) {
        for (int k = 0; k < bp.n_extras; ++k) if (bp.extras[k] == i) return true;
        return false;
    };

    // 3. per-row exception spans
    std::vector<int> lo(S, 0), hi(S, 0);
    for (int j = 0; j < S; ++j) {
        int l = S, h = -1;
        for (int i = 0; i < S; ++i) {
            if (is_extra(i) || f2u(A[(size_t)j * S + i]) == row_mode[j]) continue;
            l = std::min(l, i);
            h = std::max(h, i);
        }
        if (h < l) { l = h = std::min(j, S - 1); }
        lo[j] = l;
        hi[j] = h;
    }

    // 4. evaluated window width: the narrowest instantiated width that leaves at most kMaxDenseRows rows outside
    //    (those become "dense rows": a full max over every source) and still beats the dense kernel (2W <= S)
    int W = 0;
    for (int w : kWidths) {
        int outliers = 0;
        for (int j = 0; j < S; ++j) outliers += (hi[j] - lo[j] + 1 > w);
        if (outliers <= kMaxDenseRows && w <= S && 2 * w <= S) { W = w; break; }
    }
    if (W == 0) return bp;
    bp.W = W;
    std::vector<int> dense;
    int max_window = 1;
    for (int j = 0; j < S; ++j) {
        if (hi[j] - lo[j] + 1 > W) { dense.push_back(j); continue; }
        max_window = std::max(max_window, hi[j] - lo[j] + 1);
    }
    bp.n_dense = (int)dense.size();
    for (int d = 0; d < bp.n_dense; ++d) bp.dense_rows[d] = dense[d];
    bp.max_window = max_window;

    for (int j = 0; j < S; ++j) {
        bp.kind[j] = -1;
        bp.lo[j] = std::max(0, std::min(lo[j], S - W));
    }
    for (int d = 0; d < bp.n_dense; ++d) { bp.kind[dense[d]] = d; bp.lo[dense[d]] = 0; }

    // 5. proof obligation, re-checked from the bits: outside [lo, lo+W) and the extra columns every
    //    entry of a banded row equals that row's constant.
    for (int j = 0; j < S; ++j) {
        if (bp.kind[j] != -1) continue;
        for (int i = 0; i < S; ++i) {
            const bool inside = i >= bp.lo[j] && i < bp.lo[j] + W;
            if (!inside && !is_extra(i) && f2u(A[(size_t)j * S + i]) != row_mode[j]) return bp;
        }
    }
    // near-diagonal windows: lo[j] is an affine function of j (lets the back-trace skip a table lookup)
    for (int off = 0; off <= W && !bp.lo_affine; ++off) {
        bool all = true;
        for (int j = 0; j < S && all; ++j)
            if (bp.kind[j] == -1) all = bp.lo[j] == std::max(0, std::min(j - off, S - W));
        if (all) { bp.lo_affine = true; bp.lo_off = off; }
    }
    // pair windows: the exception spans of targets 2p and 2p+1 together fit ONE window of W sources, so one lane can
    // evaluate both targets from a single set of window reads (half the LDS traffic per target)
    {
        bp.lo2.assign(bp.SP / 2, 0);
        bool ok = bp.n_dense == 0;
        for (int p = 0; p < (S + 1) / 2 && ok; ++p) {
            const int j0 = 2 * p, j1 = std::min(2 * p + 1, S - 1);
            const int l = std::min(lo[j0], lo[j1]), h = std::max(hi[j0], hi[j1]);
            if (h - l + 1 > W) { ok = false; break; }
            bp.lo2[p] = std::max(0, std::min(l, S - W));
        }
        bp.pair_ok = ok;
        for (int off = 0; off <= 2 * W && ok && !bp.lo2_affine; ++off) {
            bool all = true;
            for (int p = 0; p < (S + 1) / 2 && all; ++p) all = bp.lo2[p] == std::max(0, std::min(2 * p - off, S - W));
            if (all) { bp.lo2_affine = true; bp.lo2_off = off; }
        }
    }
    // "floor-max" form: if no in-window entry of a banded row is below the row constant, the window term already
    // dominates fl(delta_i + c_j) for every in-window source i (rounding is monotone), so the out-of-window maximum
    // may be taken over ALL non-extra sources -- one number per frame instead of a prefix and a suffix scan.
    {
        bool ok = bp.n_dense == 0;
        for (int j = 0; j < S && ok; ++j) {
            const float cj = bp.rowc[j];
            for (int w = 0; w < W && ok; ++w) {
                const int i = bp.lo[j] + w;
                if (!is_extra(i) && !(A[(size_t)j * S + i] >= cj)) ok = false;
            }
        }
        bp.floor_ok = ok;
    }
    // "wave" form: the frame maximum over ALL sources needs the extra-column entries to dominate the row constant as
    // well; the half-width bounds how many neighbouring lanes a lane must see
    {
        bool ok = bp.floor_ok;
        for (int j = 0; j < S && ok; ++j)
            for (int k = 0; k < bp.n_extras && ok; ++k)
                if (!(A[(size_t)j * S + bp.extras[k]] >= bp.rowc[j])) ok = false;
        bp.floor_all_ok = ok;
        int d = 0;
        for (int j = 0; j < S; ++j) d = std::max(d, std::max(j - lo[j], hi[j] - j));
        bp.wave_d = d;
        bp.wave_npl = (S + 63) / 64;
        bp.wave_dk = wave_table_d(bp.wave_npl, d);
        // (the idle leading slots of a history row carry the frame maximum and a copy of the extra columns' delta)
        bp.wave_ok = ok && bp.n_dense == 0 && bp.n_extras <= kWaveMaxExtras && bp.wave_dk > 0 && S + bp.n_extras < 64 * bp.wave_npl;
    }
    bp.ok = true;
    return bp;
}

void analyze_step(const float* A, int S, BandedPlan& bp) {
    bp.step_ok = false;
    const int n = S - 1, SP = bp.SP;
    if (n < 128) return;
    // band width and count from source column 0: runs of equal values over the voiced targets
    int bw = 1;
    while (bw < n && f2u(A[(size_t)bw * S]) == f2u(A[0])) ++bw;
    if (bw < 4 || bw > 64) return;
    int kb = 0;
    for (int j = 0; j < n;) {
        int r = j;
        while (r < n && f2u(A[(size_t)r * S]) == f2u(A[(size_t)j * S])) ++r;
        if (r == n) break;                     // the last run: the far value
        if (r - j != bw) return;               // a near band that is not exactly bw wide
        ++kb;
        j = r;
    }
    if (kb < 1 || kb > kMaxStepBands || (kb + 1) * bw >= n) return;
    std::vector<float> C((size_t)(kMaxStepBands + 1) * SP, -std::numeric_limits<float>::infinity());
    std::vector<char> have((size_t)(kb + 1) * n, 0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const int d = i > j ? i - j : j - i;
            const int k = std::min(d / bw, kb);
            const float v = A[(size_t)j * S + i];
            if (f2u(v) != f2u(v) || (f2u(v) & 0x7fffffffu) > 0x7f800000u) return;   // NaN
            if (!have[(size_t)k * n + i]) { have[(size_t)k * n + i] = 1; C[(size_t)k * SP + i] = v; }
            else if (f2u(C[(size_t)k * SP + i]) != f2u(v)) return;
        }
    for (int i = 0; i < n; ++i)
        for (int k = 0; k <= kb; ++k) {
            if (!have[(size_t)k * n + i]) C[(size_t)k * SP + i] = C[(size_t)kb * SP + i];   // band never reached from this source
            if (!(C[(size_t)k * SP + i] >= C[(size_t)kb * SP + i])) return;                  // near bands must dominate the far value
        }
    for (int j = 1; j < n; ++j)
        if (f2u(A[(size_t)j * S + n]) != f2u(A[n])) return;                               // unvoiced source: one value for all voiced targets
    bp.step_bw = bw;
    bp.step_kb = kb;
    bp.step_cn = A[n];
    bp.stepC = std::move(C);
    bp.step_ok = true;
}

ImageLayout make_layout(int S, const BandedPlan& bp) {
    ImageLayout L;
    L.S = S;
    L.SP = (S + 63) / 64 * 64;
    L.S4 = (S + 3) / 4;
    L.W = bp.ok ? bp.W : 0;
    L.n_extras = bp.ok ? bp.n_extras : 0;
    L.n_dense = bp.ok ? bp.n_dense : 0;
    size_t off = 0;
    L.off_logpi = off;  off = align256(off + sizeof(float) * L.SP);
    L.off_A4 = off;     off = align256(off + sizeof(float) * 4 * (size_t)L.S4 * L.SP);
    L.off_lo = off;     off = align256(off + sizeof(int32_t) * L.SP);
    L.off_kind = off;   off = align256(off + sizeof(int32_t) * L.SP);
    L.off_tabA = off;   off = align256(off + sizeof(float) * (size_t)std::max(L.W, 1) * L.SP);
    L.off_extraA = off; off = align256(off + sizeof(float) * kMaxExtras * L.SP);
    L.off_denseA = off; off = align256(off + sizeof(float) * kMaxDenseRows * L.SP);
    L.off_Arow = off;   off = align256(off + sizeof(float) * (size_t)S * L.SP);
    L.off_rowc = off;   off = align256(off + sizeof(float) * L.SP);
    L.off_lo2 = off;    off = align256(off + sizeof(int32_t) * L.SP);
    L.off_tabP = off;   off = align256(off + sizeof(float) * (size_t)std::max(L.W, 1) * L.SP);
    L.off_tabX = off;   off = align256(off + sizeof(float) * (size_t)(std::max(L.W, 1) + kMaxExtras + 1) * L.SP);
    L.off_stepC = off;  off = align256(off + sizeof(float) * (size_t)(kMaxStepBands + 1) * L.SP);
    L.off_tabV = off;
    if (bp.ok && bp.wave_ok) off = align256(off + sizeof(float) * (size_t)bp.wave_npl * wave_pairs(bp.wave_dk) * 2 * 64);
    L.bytes = off;
    return L;
}

void fill_image(const float* A, const float* log_pi, const BandedPlan& bp, const ImageLayout& L,
                uint8_t* image) {
    const float ninf = -std::numeric_limits<float>::infinity();
    const int S = L.S, SP = L.SP;
    std::memset(image, 0, L.bytes);
    float* pi = reinterpret_cast<float*>(image + L.off_logpi);
    for (int j = 0; j < SP; ++j) pi[j] = j < S ? log_pi[j] : ninf;

    float* A4 = reinterpret_cast<float*>(image + L.off_A4);
    for (int q = 0; q < L.S4; ++q)
        for (int j = 0; j < SP; ++j)
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                A4[((size_t)q * SP + j) * 4 + r] = (j < S && i < S) ? A[(size_t)j * S + i] : ninf;
            }

    float* Arow = reinterpret_cast<float*>(image + L.off_Arow);
    for (int j = 0; j < S; ++j)
        for (int i = 0; i < SP; ++i) Arow[(size_t)j * SP + i] = i < S ? A[(size_t)j * S + i] : ninf;

    int32_t* lo = reinterpret_cast<int32_t*>(image + L.off_lo);
    int32_t* kind = reinterpret_cast<int32_t*>(image + L.off_kind);
    float* tab = reinterpret_cast<float*>(image + L.off_tabA);
    float* xa = reinterpret_cast<float*>(image + L.off_extraA);
    float* da = reinterpret_cast<float*>(image + L.off_denseA);
    for (int j = 0; j < SP; ++j) { lo[j] = 0; kind[j] = -2; }
    for (int k = 0; k < kMaxExtras; ++k) for (int j = 0; j < SP; ++j) xa[(size_t)k * SP + j] = ninf;
    for (int d = 0; d < kMaxDenseRows; ++d) for (int j = 0; j < SP; ++j) da[(size_t)d * SP + j] = ninf;
    float* rc = reinterpret_cast<float*>(image + L.off_rowc);
    for (int j = 0; j < SP; ++j) rc[j] = 0.f;
    if (bp.step_ok) std::memcpy(image + L.off_stepC, bp.stepC.data(), sizeof(float) * bp.stepC.size());
    if (!bp.ok) return;
    for (int j = 0; j < S; ++j) { lo[j] = bp.lo[j]; kind[j] = bp.kind[j]; rc[j] = bp.rowc[j]; }
    for (int w = 0; w < L.W; ++w)
        for (int j = 0; j < SP; ++j)
            tab[(size_t)w * SP + j] = j < S ? A[(size_t)j * S + bp.lo[j] + w] : ninf;
    for (int k = 0; k < bp.n_extras; ++k)
        for (int j = 0; j < S; ++j) xa[(size_t)k * SP + j] = A[(size_t)j * S + bp.extras[k]];
    for (int d = 0; d < bp.n_dense; ++d)
        for (int i = 0; i < S; ++i) da[(size_t)d * SP + i] = A[(size_t)bp.dense_rows[d] * S + i];
    {   // per-target candidate rows for the back-trace: window, extra columns (-inf beyond n_extras), row constant
        const int WX1 = L.W + kMaxExtras + 1;
        float* tx = reinterpret_cast<float*>(image + L.off_tabX);
        for (int j = 0; j < SP; ++j) {
            float* row = tx + (size_t)j * WX1;
            for (int c = 0; c < WX1; ++c) row[c] = ninf;
            if (j >= S) continue;
            for (int w = 0; w < L.W; ++w) row[w] = A[(size_t)j * S + bp.lo[j] + w];
            for (int k = 0; k < bp.n_extras; ++k) row[L.W + k] = A[(size_t)j * S + bp.extras[k]];
            row[L.W + kMaxExtras] = bp.rowc[j];
        }
    }
    if (bp.wave_ok) {   // weights in the order wave_forward_kernel loads them: [own state k][pair m][half h][lane]
        const int npl = bp.wave_npl, dk = bp.wave_dk, H = wave_halo(npl, dk), NPM = wave_pairs(dk);
        float* tv = reinterpret_cast<float*>(image + L.off_tabV);
        for (int k = 0; k < npl; ++k)
            for (int m = 0; m < NPM; ++m)
                for (int h = 0; h < 2; ++h)
                    for (int l = 0; l < 64; ++l) {
                        // right-aligned slots: slot q holds state q - o, o = 64*npl - S (wave.hip)
                        const int o = 64 * npl - S;
                        const int j = npl * l + k - o;
                        const int i = npl * (l - H) + wave_p0e(npl, dk, k) + 2 * m + h - o;
                        // always the true matrix entry: positions outside the exception span are the row constant,
                        // a candidate the dense recursion forms as well; idle targets (j < 0) get -inf so that their
                        // delta stays -inf; so do sources outside [0, S): whatever a shift delivers there, the
                        // candidate is -inf
                        const float v = (j >= 0 && i >= 0 && i < S) ? A[(size_t)j * S + i] : ninf;
                        tv[(((size_t)k * NPM + m) * 2 + h) * 64 + l] = v;
                    }
    }
    if (bp.pair_ok) {
        int32_t* lo2 = reinterpret_cast<int32_t*>(image + L.off_lo2);
        float* tp = reinterpret_cast<float*>(image + L.off_tabP);
        for (int p = 0; p < SP / 2; ++p) lo2[p] = p < (int)bp.lo2.size() ? bp.lo2[p] : 0;
        for (int w = 0; w < L.W; ++w)
            for (int j = 0; j < SP; ++j)
                tp[(size_t)w * SP + j] = j < S ? A[(size_t)j * S + bp.lo2[j / 2] + w] : ninf;
    }
}

}  // namespace vit
