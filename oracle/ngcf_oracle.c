/*
 * Plain-C CPU oracle for the NGCF propagation hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Scalar restatement of the reference's arithmetic (file:line into /root/reference/model/).
 * It is the checker for the HIP kernels at sizes where a single core finishes in seconds;
 * nothing under seoul_tourism_recommendation_ngcf_amd/ links or loads it.
 *
 * Parity pin: `spmm_coo` reproduces `torch.mm(sparse_coo, dense)` on CPU bit for bit
 * (sequential fmaf per stored entry, in stored order) - checked in
 * tests/test_oracle_golden.py against fixtures captured from the imported reference.
 * The dense part (`layer_dense`) follows NGCF.py:131-144 op for op but sums each dot product
 * in plain k order, which MKL does not promise, so it is pinned to tolerance, not bits.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; fmaf() is called explicitly)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NGCF_LEAKY 0.2f      /* NGCF.py:140 */
#define NGCF_EPS   1e-12f    /* F.normalize eps, NGCF.py:144 */

/* L.E, NGCF.py:130.  out[r,:] = fma(v, E[c,:], out[r,:]) per stored entry, stored order. */
void ngcf_oracle_spmm_coo_f32(const int64_t *rows, const int64_t *cols, const float *vals,
                              int64_t nnz, int64_t n_rows, const float *E, int64_t ldE, int d,
                              float *out, int64_t ldo)
{
    for (int64_t r = 0; r < n_rows; ++r)
        memset(out + r * ldo, 0, sizeof(float) * (size_t)d);
    for (int64_t e = 0; e < nnz; ++e) {
        const float v = vals[e];
        const float *src = E + cols[e] * ldE;
        float *dst = out + rows[e] * ldo;
        for (int j = 0; j < d; ++j)
            dst[j] = fmaf(v, src[j], dst[j]);
    }
}

/* Same product from CSR (int32 columns), the layout the HIP engine consumes. */
void ngcf_oracle_spmm_csr_f32(const int64_t *rowptr, const int32_t *colidx, const float *vals,
                              int64_t n_rows, const float *E, int64_t ldE, int d,
                              float *out, int64_t ldo)
{
    for (int64_t r = 0; r < n_rows; ++r) {
        float *dst = out + r * ldo;
        memset(dst, 0, sizeof(float) * (size_t)d);
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e) {
            const float v = vals[e];
            const float *src = E + (int64_t)colidx[e] * ldE;
            for (int j = 0; j < d; ++j)
                dst[j] = fmaf(v, src[j], dst[j]);
        }
    }
}

static float dotf(const float *a, const float *b, int n)
{
    float s = 0.f;
    for (int k = 0; k < n; ++k)
        s += a[k] * b[k];
    return s;
}

/*
 * Dense half of one layer, NGCF.py:131-146 (eval: mess-dropout is the identity):
 *   M     = (LE.W1^T + b1) + (E.W1^T + b1) + ((LE*E).W2^T + b2)      NGCF.py:131-138
 *   carry = leaky_relu(M, 0.2)                                        NGCF.py:140
 *   norm  = carry / max(||carry||_2, 1e-12)                           NGCF.py:144
 * W1, W2 are [d_out, d_in] row-major (nn.Linear layout).  `carry` may be NULL.
 */
void ngcf_oracle_layer_dense_f32(const float *LE, int64_t ldLE, const float *E, int64_t ldE,
                                 int64_t n_rows, int d_in, int d_out,
                                 const float *W1, const float *b1, const float *W2, const float *b2,
                                 float *carry, int64_t ldc, float *norm, int64_t ldn)
{
    float *lee = (float *)malloc(sizeof(float) * (size_t)d_in);
    float *m = (float *)malloc(sizeof(float) * (size_t)d_out);
    for (int64_t r = 0; r < n_rows; ++r) {
        const float *le = LE + r * ldLE, *e = E + r * ldE;
        for (int k = 0; k < d_in; ++k)
            lee[k] = le[k] * e[k];                                     /* NGCF.py:135 */
        float ss = 0.f;
        for (int j = 0; j < d_out; ++j) {
            const float t1 = dotf(le, W1 + (int64_t)j * d_in, d_in) + b1[j];
            const float t2 = dotf(e, W1 + (int64_t)j * d_in, d_in) + b1[j];
            const float t3 = dotf(lee, W2 + (int64_t)j * d_in, d_in) + b2[j];
            float v = (t1 + t2) + t3;
            v = v >= 0.f ? v : NGCF_LEAKY * v;
            m[j] = v;
            ss += v * v;
        }
        float nrm = sqrtf(ss);
        if (nrm < NGCF_EPS)
            nrm = NGCF_EPS;
        for (int j = 0; j < d_out; ++j) {
            if (carry)
                carry[r * ldc + j] = m[j];
            norm[r * ldn + j] = m[j] / nrm;
        }
    }
    free(lee);
    free(m);
}

/* Row gather, NGCF.py:151-155: out[b,:] = table[row_off + idx[b], :] (bit-exact copy). */
void ngcf_oracle_gather_rows_f32(const float *table, int64_t ld, int d, const int64_t *idx,
                                 int64_t n_idx, int64_t row_off, float *out)
{
    for (int64_t b = 0; b < n_idx; ++b)
        memcpy(out + b * d, table + (row_off + idx[b]) * ld, sizeof(float) * (size_t)d);
}

static float logsigmoidf_(float x)
{
    /* min(x,0) - log1p(exp(-|x|)), the form torch uses */
    const float mn = x < 0.f ? x : 0.f;
    return mn - log1pf(expf(-fabsf(x)));
}

/*
 * BPR, bprloss.py:15-22.  Bu/Bp/Bn are the row counts of u/p/n; each is 1 (broadcast) or R,
 * R = max of the three.  The squared norms run over each tensor's own rows.
 */
float ngcf_oracle_bpr_f32(const float *u, int64_t Bu, const float *p, int64_t Bp,
                          const float *n, int64_t Bn, int D, float weight_decay, float batch_size)
{
    int64_t R = Bu;
    if (Bp > R) R = Bp;
    if (Bn > R) R = Bn;
    float log_prob = 0.f;
    for (int64_t r = 0; r < R; ++r) {
        const float *ur = u + (Bu == 1 ? 0 : r) * (int64_t)D;
        const float *pr = p + (Bp == 1 ? 0 : r) * (int64_t)D;
        const float *nr = n + (Bn == 1 ? 0 : r) * (int64_t)D;
        const float x = fabsf(dotf(ur, pr, D)) - fabsf(dotf(ur, nr, D));   /* bprloss.py:16-18 */
        log_prob += logsigmoidf_(x);                                       /* bprloss.py:19 */
    }
    float reg = 0.f;
    for (int64_t r = 0; r < Bu; ++r) reg += dotf(u + r * D, u + r * D, D);
    for (int64_t r = 0; r < Bp; ++r) reg += dotf(p + r * D, p + r * D, D);
    for (int64_t r = 0; r < Bn; ++r) reg += dotf(n + r * D, n + r * D, D);
    return (-log_prob + weight_decay * reg) / batch_size;                  /* bprloss.py:20-22 */
}
