// plan.hpp -- host-side analysis of a log-domain transition matrix.
//
// Pure C++ (no HIP): compiled into libviterbi_hip.so by hipcc and into
// libviterbi_plan_host.so by g++ so that the analysis and the exactness of the
// banded decomposition can be tested on a machine without a GPU.
//
// The decomposition: the production matrices built by the
// reference's */viterbi_transition_post_processing.py are
//   band(+/- d_max) + one dense row + one dense column, every other entry the
//   single constant c = log(0 + tiny).
// For a target row j whose entries outside a window [lo_j, lo_j+W) and outside a
// few shared "extra" columns all equal one constant c_j, rounding is monotone, so
//   max_{i outside} fl(delta_i + c_j) = fl( max_{i outside} delta_i + c_j ):
// one prefix-max and one suffix-max scan over the raw delta vector, evaluated at
// the window edges, replace S - W candidates per target.  Every value compared
// is one the dense recursion also computes, so the result is bit-identical.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace vit {

constexpr int kMaxExtras = 4;
constexpr int kMaxDenseRows = 4;
constexpr int kMaxWindow = 128;  // widest window the banded kernels are instantiated for (jdc: d_max 40 -> 82, imm: 56 -> 114)

struct BandedPlan {
    bool ok = false;          // the decomposition was proven for this matrix
    int S = 0;
    int SP = 0;               // S rounded up to a multiple of 64 (threads per workgroup)
    float c0 = 0.f;           // the most common row constant (informational)
    std::vector<float> rowc;  // [SP] row constant c_j: every entry of row j outside its window and the extra columns
    int n_extras = 0;
    int extras[kMaxExtras] = {0, 0, 0, 0};
    int n_dense = 0;
    int dense_rows[kMaxDenseRows] = {0, 0, 0, 0};
    int max_window = 0;       // widest proven exception window among banded rows
    int W = 0;                // window width the kernel evaluates (16/32/64/96/128)
    std::vector<int32_t> lo;  // [SP] first source of the evaluated window (lo + W <= S)
    std::vector<int32_t> kind;// [SP] -1 banded row, d >= 0 dense row d, -2 padding
    bool lo_affine = false;   // lo[j] == clamp(j - lo_off, 0, S - W) for every banded row
    bool floor_ok = false;    // every non-extra window entry of every banded row is >= that row's constant, and no dense rows:
                              // then max_{i outside window} fl(delta_i + c_j) can be replaced by fl(max_{all non-extra i} delta_i + c_j)
    bool pair_ok = false;     // targets (2p, 2p+1) share one window [lo2[p], lo2[p]+W) that covers both exception spans
    std::vector<int32_t> lo2; // [SP/2]
    bool lo2_affine = false;  // lo2[p] == clamp(2p - lo2_off, 0, S - W)
    int lo2_off = 0;
    int lo_off = 0;

    // "Step" structure (the Durrieu matrix of imm's own decoder; analyze_step): with n = S-1 voiced states, for voiced
    // source i and voiced target j  logA_T[j][i] == stepC[min(|i-j| / step_bw, step_kb)][i]  -- column i is piecewise
    // constant in distance bands of step_bw bins, constant from distance step_kb*step_bw on --, no near band of a column
    // is below its far value, and the unvoiced source column is one value for every voiced target (step_cn).  Then
    // fl(delta_i + logA_T[j][i]) takes only step_kb+1 values per source and the far sources reduce to ONE maximum.
    bool step_ok = false;
    int step_bw = 0, step_kb = 0;
    std::vector<float> stepC;   // [(kMaxStepBands+1)][SP], rows > step_kb unused, -inf for i >= n
    float step_cn = 0.f;

    // "Wave" form (wave_forward_kernel: one song per wavefront, no LDS, no barrier).  Lane l owns the wave_npl =
    // ceil(S/64) contiguous states wave_npl*l ..; every exception span lies within wave_d sources of its target
    // (|i - j| <= wave_d), so a lane needs delta only from the lanes within ceil(wave_d / wave_npl) of itself.
    bool floor_all_ok = false;  // floor_ok, and every extra-column entry of a banded row is >= that row's constant too:
                                // then the frame maximum M may be taken over ALL sources (a dominated candidate more)
    bool wave_ok = false;       // floor_all_ok, no dense rows, <= kWaveMaxExtras extra columns, an instantiated geometry
    int wave_npl = 0;
    int wave_d = 0;             // proven half-width
    int wave_dk = 0;            // half-width the kernel (and tabV) is instantiated for (>= wave_d)
};

constexpr int kWaveMaxExtras = 2;
// half-width the wave kernel is instantiated for, given the states per lane and the proven half-width (0: none)
constexpr int wave_table_d(int npl, int d) { return (npl == 6 && d <= 14) ? 14 : 0; }
constexpr int wave_pairs(int dk) { return dk + 1; }   // packed source pairs per target: 2*dk + 1 sources, even-aligned
constexpr int wave_halo(int npl, int dk) { return (dk + npl - 1) / npl; }   // lanes a lane looks at on each side
// A lane's neighbourhood is the npl * (2*halo + 1) sources of lanes l-halo .. l+halo, position p <-> source
// npl * (l - halo) + p.  Own state k evaluates the even-aligned positions p0e .. p0e + 2*(dk+1) - 1, which cover
// its sources j - dk .. j + dk; pair m, half h sits at p0e + 2m + h.
constexpr int wave_p0e(int npl, int dk, int k) { return (k + npl * wave_halo(npl, dk) - dk) & ~1; }

constexpr int kMaxStepBands = 15;
// Fills the step_* fields of bp (bp.SP must be set: call after analyze_banded).
void analyze_step(const float* logA_T, int S, BandedPlan& bp);

// Analyse logA_T ([S,S] row-major, row j = into target j).
BandedPlan analyze_banded(const float* logA_T, int S);

// Byte layout of the device image (all offsets in bytes from the image base,
// every section 256-byte aligned).
struct ImageLayout {
    int S = 0, SP = 0, S4 = 0, W = 0, n_extras = 0, n_dense = 0;
    size_t off_logpi = 0;    // float [SP]          (-inf padded)
    size_t off_A4 = 0;       // float [S4][SP][4]   A4[q][j][r] = logA_T[j][4q+r] (-inf padded)
    size_t off_lo = 0;       // int32 [SP]
    size_t off_kind = 0;     // int32 [SP]
    size_t off_tabA = 0;     // float [W][SP]       tabA[w][j] = logA_T[j][lo_j + w]
    size_t off_extraA = 0;   // float [4][SP]       extraA[k][j] = logA_T[j][extras[k]]
    size_t off_denseA = 0;   // float [4][SP]       denseA[d][i] = logA_T[dense_rows[d]][i]
    size_t off_Arow = 0;     // float [S][SP]       row-major copy (row j = into target j) for the back-trace
    size_t off_rowc = 0;     // float [SP]          row constants c_j
    size_t off_lo2 = 0;      // int32 [SP/2]        pair windows (pair_ok)
    size_t off_tabP = 0;     // float [W][SP]       tabP[w][j] = logA_T[j][lo2[j/2] + w]
    size_t off_tabX = 0;     // float [SP][W+5]     per target: W window entries, 4 extra-column entries, row constant (back-trace)
    size_t off_stepC = 0;    // float [16][SP]      step-structure band values per source (step_ok)
    size_t off_tabV = 0;     // float [npl][dk+1][2][64]  wave form: weights of lane l, own state k, source pair m, half h
    size_t bytes = 0;
};

ImageLayout make_layout(int S, const BandedPlan& bp);
void fill_image(const float* logA_T, const float* log_pi, const BandedPlan& bp,
                const ImageLayout& L, uint8_t* image);

}  // namespace vit
