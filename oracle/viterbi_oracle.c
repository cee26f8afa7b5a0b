/*
 * oracle/viterbi_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement (plain C) of the float32 log-domain Viterbi recursion
 * that drwangxian/viterbi_spl runs on the host.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this; the product path
 * (viterbi_spl_amd/) never does.
 *
 * Semantics restated from (paths relative to /root/reference):
 *   imm/tf_viterbi.py:91-107            forward loop + backtrace (log-domain core)
 *   dcnet/tf_viterbi_decoding.py:189-205 same loop in the prob-domain wrapper
 *   dcnet/aot_viterbi_core.py:8-54      C-style (B, prob_init, probs) signature
 *
 *   delta_0[j] = fl32(log_pi[j] + logE[0][j])
 *   for t >= 1:  m_j   = max_i fl32(delta_{t-1}[i] + logA_T[j][i])
 *                psi_t[j] = LOWEST i attaining m_j      (np.argmax first hit)
 *                delta_t[j] = fl32(m_j + logE[t][j])
 *   s_{T-1} = lowest argmax_j delta_{T-1}[j];  s_t = psi_{t+1}[s_{t+1}]
 *
 * The order of the two additions and the strict '>' tie-break are load
 * bearing.  Build with -ffp-contract=off.
 *
 * NaN inputs are outside the contract (np.argmax would return the first NaN).
 *
 * Pinning: validated bit-for-bit against the imported reference functions by
 * tests/golden/make_goldens.py; goldens are committed under tests/golden/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* status codes shared with the tests */
#define VO_OK 0
#define VO_EINVAL -1
#define VO_ENOMEM -2

/*
 * One song.  logA_T: [S,S] row j = "into target j" (imm/tf_viterbi.py:77-80).
 * logE: [T,S] C-order.  psi_out (optional): [T,S] uint16 back-pointers, row 0
 * unused (zeroed).  delta_out (optional): [S] final delta row.
 * states: [T] int32.  loglik (optional): delta_{T-1}[s_{T-1}].
 *
 * Implementation note: the inner loops run source-major over a transposed
 * copy of logA_T so that gcc can vectorise over targets; visiting sources in
 * increasing order with a strict '>' keeps the lowest source on ties, i.e. the
 * exact np.argmax result.
 */
static int decode_one(const float *A_sm /* [S,S] source-major: A_sm[i*S+j] = logA_T[j][i] */,
                      const float *log_pi, const float *logE, int64_t T, int64_t S,
                      uint16_t *psi, /* [T,S] scratch, required */
                      float *delta_out, int32_t *states, float *loglik)
{
    float *cur = (float *)malloc(sizeof(float) * (size_t)S * 2);
    int32_t *arg = (int32_t *)malloc(sizeof(int32_t) * (size_t)S);
    if (!cur || !arg) { free(cur); free(arg); return VO_ENOMEM; }
    float *best = cur + S;

    for (int64_t j = 0; j < S; ++j) cur[j] = log_pi[j] + logE[j];
    memset(psi, 0, sizeof(uint16_t) * (size_t)S);

    for (int64_t t = 1; t < T; ++t) {
        const float *e = logE + t * S;
        uint16_t *p = psi + t * S;
        const float d0 = cur[0];
        for (int64_t j = 0; j < S; ++j) { best[j] = d0 + A_sm[j]; arg[j] = 0; }
        for (int64_t i = 1; i < S; ++i) {
            const float di = cur[i];
            const float *row = A_sm + i * S;
            for (int64_t j = 0; j < S; ++j) {
                const float v = di + row[j];
                const int gt = v > best[j];
                best[j] = gt ? v : best[j];
                arg[j] = gt ? (int32_t)i : arg[j];
            }
        }
        for (int64_t j = 0; j < S; ++j) { cur[j] = best[j] + e[j]; p[j] = (uint16_t)arg[j]; }
    }

    int64_t s = 0;
    float top = cur[0];
    for (int64_t j = 1; j < S; ++j) if (cur[j] > top) { top = cur[j]; s = j; }
    if (loglik) *loglik = top;
    if (delta_out) memcpy(delta_out, cur, sizeof(float) * (size_t)S);
    states[T - 1] = (int32_t)s;
    for (int64_t t = T - 2; t >= 0; --t) { s = psi[(t + 1) * S + s]; states[t] = (int32_t)s; }

    free(cur); free(arg);
    return VO_OK;
}

/*
 * Batched entry point.  logE: [B,T,S] float32.  lengths: NULL or [B] with
 * 1 <= lengths[b] <= T; states beyond a song's length are set to -1.
 * delta_out: NULL or [B,S]; loglik: NULL or [B].  threads <= 0 -> all cores.
 */
int vo_decode_f32(const float *logA_T, const float *log_pi, const float *logE,
                  int64_t B, int64_t T, int64_t S, const int64_t *lengths,
                  int32_t *states, float *loglik, float *delta_out, int threads)
{
    if (!logA_T || !log_pi || !logE || !states) return VO_EINVAL;
    if (B < 0 || T < 1 || S < 1 || S > 65535) return VO_EINVAL;
    if (lengths) for (int64_t b = 0; b < B; ++b) if (lengths[b] < 1 || lengths[b] > T) return VO_EINVAL;

    float *A_sm = (float *)malloc(sizeof(float) * (size_t)S * (size_t)S);
    if (!A_sm) return VO_ENOMEM;
    for (int64_t j = 0; j < S; ++j)
        for (int64_t i = 0; i < S; ++i) A_sm[i * S + j] = logA_T[j * S + i];

    int status = VO_OK;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#else
    (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t b = 0; b < B; ++b) {
        const int64_t Tb = lengths ? lengths[b] : T;
        uint16_t *psi = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)Tb * (size_t)S);
        int rc = psi ? decode_one(A_sm, log_pi, logE + b * T * S, Tb, S, psi,
                                  delta_out ? delta_out + b * S : NULL,
                                  states + b * T, loglik ? loglik + b : NULL)
                     : VO_ENOMEM;
        for (int64_t t = Tb; t < T; ++t) states[b * T + t] = -1;
        free(psi);
        if (rc != VO_OK) {
#pragma omp critical
            status = rc;
        }
    }
    free(A_sm);
    return status;
}

int vo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
