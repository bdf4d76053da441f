// backward.hip - kernels of the backward pass.
#include "spmm_device.h"

// =============================================================================================
// Backward pass (`loss.backward()` in experiment.py:57).
// Everything the backward runs besides L^T . dLE on the SpMM kernels: BPR gradient, the gathers' gradient rows summed per
// distinct row, normalise/dropout/LeakyReLU backward, weight and input gradients on the fp32 matrix cores, and L^T . dLE for
// a row-sparse dLE.  No library GEMM, no atomics: every sum has a fixed order (bit-identical gradients from run to run).
// =============================================================================================

// ---- BPR backward (bprloss.py:15-22) ----------------------------------------------------------
// loss = (-sum logsig(|u.p| - |u.n|) + wd (|u|^2 + |p|^2 + |n|^2)) / bs ; one wave per row
__global__ __launch_bounds__(256) void bpr_backward_kernel(const float *__restrict__ u, int64_t Bu,
                                                           const float *__restrict__ p, int64_t Bp,
                                                           const float *__restrict__ n, int64_t Bn, int64_t R, int D,
                                                           float wd, float batch_size, const float *__restrict__ gout,
                                                           float *__restrict__ du, float *__restrict__ dp,
                                                           float *__restrict__ dn)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int lane = threadIdx.x & 63;
    const float g = gout[0] / batch_size;
    const int64_t ru = Bu == 1 ? 0 : r, rp = Bp == 1 ? 0 : r, rn = Bn == 1 ? 0 : r;
    const float *ur = u + ru * D, *pr = p + rp * D, *nr = n + rn * D;
    float up = 0.f, un = 0.f;
    for (int j = lane; j < D; j += 64) {
        up = fmaf(ur[j], pr[j], up);
        un = fmaf(ur[j], nr[j], un);
    }
    up = wave_sum(up);
    un = wave_sum(un);
    const float x = fabsf(up) - fabsf(un);
    const float s = -1.f / (1.f + expf(x));                  // d(-logsigmoid(x))/dx = -sigmoid(-x)
    const float sp = up > 0.f ? 1.f : (up < 0.f ? -1.f : 0.f);   // d|t|/dt, 0 at 0 like torch.abs
    const float sn = un > 0.f ? 1.f : (un < 0.f ? -1.f : 0.f);
    const float two_wd = 2.f * wd;
    for (int j = lane; j < D; j += 64) {
        const float a = ur[j], b = pr[j], c = nr[j];
        // the weight-decay term of a broadcast row is counted once (its own single row)
        const float gu = g * (s * (sp * b - sn * c) + ((Bu == 1 && r > 0) ? 0.f : two_wd * a));
        const float gp = g * (s * sp * a + ((Bp == 1 && r > 0) ? 0.f : two_wd * b));
        const float gn = g * (-s * sn * a + ((Bn == 1 && r > 0) ? 0.f : two_wd * c));
        if (Bu == R) du[ru * D + j] = gu;          // a broadcast operand's gradient is a sum over the rows: bpr_backward_bcast_kernel
        if (Bp == R) dp[rp * D + j] = gp;
        if (Bn == R) dn[rn * D + j] = gn;
    }
}

// The gradient of a BROADCAST operand (a [1, D] row against R > 1 rows: bprloss.py broadcasts like torch): the sum of the per-row
// terms in ROW ORDER, by one workgroup - the per-row coefficients of 256 rows at a time into LDS (a wave per row), then a thread
// per column adds them up.  (r04: this was three float atomicAdds per element - the last ones in the library; a rare path, R * D
// multiply-adds on one CU.)
__global__ __launch_bounds__(256) void bpr_backward_bcast_kernel(const float *__restrict__ u, int64_t Bu, const float *__restrict__ p,
                                                                 int64_t Bp, const float *__restrict__ n, int64_t Bn, int64_t R, int D,
                                                                 float wd, float batch_size, const float *__restrict__ gout,
                                                                 float *__restrict__ du, float *__restrict__ dp, float *__restrict__ dn)
{
    __shared__ float c_pos[256], c_neg[256];                   // s * sign(u.p), s * sign(u.n) of the rows of a chunk
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float g = gout[0] / batch_size, two_wd = 2.f * wd;
    for (int j0 = 0; j0 < D; j0 += 256) {
        const int j = j0 + threadIdx.x;
        float au = 0.f, ap = 0.f, an = 0.f;
        for (int64_t r0 = 0; r0 < R; r0 += 256) {
            const int cnt = (int)(R - r0 < 256 ? R - r0 : 256);
            __syncthreads();
            for (int rr = wave; rr < cnt; rr += 4) {
                const int64_t r = r0 + rr;
                const float *ur = u + (Bu == 1 ? 0 : r) * D, *pr = p + (Bp == 1 ? 0 : r) * D, *nr = n + (Bn == 1 ? 0 : r) * D;
                float up = 0.f, un = 0.f;
                for (int k = lane; k < D; k += 64) {
                    up = fmaf(ur[k], pr[k], up);
                    un = fmaf(ur[k], nr[k], un);
                }
                up = wave_sum(up);
                un = wave_sum(un);
                const float sg = -1.f / (1.f + expf(fabsf(up) - fabsf(un)));
                if (lane == 0) {
                    c_pos[rr] = sg * (up > 0.f ? 1.f : (up < 0.f ? -1.f : 0.f));
                    c_neg[rr] = sg * (un > 0.f ? 1.f : (un < 0.f ? -1.f : 0.f));
                }
            }
            __syncthreads();
            if (j < D)
                for (int rr = 0; rr < cnt; ++rr) {
                    const int64_t r = r0 + rr;
                    const float a = u[(Bu == 1 ? 0 : r) * D + j], b = p[(Bp == 1 ? 0 : r) * D + j], c = n[(Bn == 1 ? 0 : r) * D + j];
                    au += c_pos[rr] * b - c_neg[rr] * c;
                    ap = fmaf(c_pos[rr], a, ap);
                    an = fmaf(-c_neg[rr], a, an);
                }
        }
        if (j < D) {                                           // the weight-decay term of a broadcast row is counted once
            if (Bu == 1) du[j] = g * (au + two_wd * u[j]);
            if (Bp == 1) dp[j] = g * (ap + two_wd * p[j]);
            if (Bn == 1) dn[j] = g * (an + two_wd * n[j]);
        }
    }
}

extern "C" int ngcf_bpr_backward_f32(const float *u, int64_t Bu, const float *p, int64_t Bp, const float *n, int64_t Bn,
                                     int D, float wd, float batch_size, const float *grad_out, float *du, float *dp,
                                     float *dn, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!u || !p || !n || !grad_out || !du || !dp || !dn || D <= 0) return fail(NGCF_ERR_ARG, "bpr_backward: null argument");
    const int64_t R = std::max(Bu, std::max(Bp, Bn));
    if (R < 1 || (Bu != 1 && Bu != R) || (Bp != 1 && Bp != R) || (Bn != 1 && Bn != R))
        return fail(NGCF_ERR_ARG, "bpr_backward: row counts %lld/%lld/%lld do not broadcast", (long long)Bu, (long long)Bp, (long long)Bn);
    bpr_backward_kernel<<<dim3((unsigned)((R + 3) / 4)), 256, 0, stream>>>(u, Bu, p, Bp, n, Bn, R, D, wd, batch_size, grad_out,
                                                                            du, dp, dn);
    LAUNCH_CHECK();
    if (Bu != R || Bp != R || Bn != R) {                        // (only with R > 1: an operand of one row against several)
        bpr_backward_bcast_kernel<<<1, 256, 0, stream>>>(u, Bu, p, Bp, n, Bn, R, D, wd, batch_size, grad_out, du, dp, dn);
        LAUNCH_CHECK();
    }
    return NGCF_OK;
}

// ---- gather backward: the gradient rows of the (users, positive items, negative items) gathers (NGCF.py:151-155), summed per
// distinct row of all_E in a FIXED order: out[r, :] = sum over j in [segptr[r], segptr[r+1]) of g[order[j], :], in that order
// (the caller sorts the gathered positions by row, stable, so duplicates add up in batch order).  No atomics: two runs give
// the same bits.  One wave per output row.
__global__ __launch_bounds__(256) void segment_sum_rows_kernel(const float *__restrict__ g, int64_t ldg, int d,
                                                               const int64_t *__restrict__ order, const int64_t *__restrict__ segptr,
                                                               int64_t n_seg, const int64_t *__restrict__ dst_rows,
                                                               const int64_t *__restrict__ n_seg_dev, float *__restrict__ out, int64_t ldo,
                                                               int64_t n_out_rows)
{
    int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_seg || (n_seg_dev && r >= *n_seg_dev)) return;       // n_seg_dev: the number of segments lives on the device
    const int64_t j0 = segptr[r], j1 = segptr[r + 1];
    if (dst_rows) {                                                 // scatter form: segment r is row dst_rows[r] of a larger matrix
        r = dst_rows[r];
        if (r < 0 || r >= n_out_rows) return;                       // (an id the forward gather clamped and flagged: nothing to add, nothing written)
    }
    // four chains over the segment (positions j0 + 4 i + q; the last (j1 - j0) % 4 positions go to chain 0), combined as
    // (s0 + s1) + (s2 + s3): a fixed order.  r04: the positions `order[j]` of 64 entries are fetched by ONE load (a lane each) and
    // handed round with readlane, a lane keeps the sums of five column blocks and sixteen gradient rows are requested before the
    // first is added - the loads depend on nothing but that one load (before: order -> row -> next order, twice per group of four,
    // for every column block: a popular item is gathered 40 times in a batch of 1 024 on the Seoul graph's 100 items - 22 us).
    const int lane = threadIdx.x & 63;
    const int64_t n4 = (j1 - j0) / 4 * 4;                           // entries in full groups of four
    constexpr int kQ = 5;                                           // column blocks of 64 per pass (the Seoul model's 260 columns: one pass)
    for (int c0 = 0; c0 < d; c0 += 64 * kQ) {
        float s[kQ][4];
#pragma unroll
        for (int q = 0; q < kQ; ++q) s[q][0] = s[q][1] = s[q][2] = s[q][3] = 0.f;
        for (int64_t jb = j0; jb < j1; jb += 64) {                  // (jb - j0 is a multiple of 64: groups of four never straddle)
            const int cnt = (int)(j1 - jb < 64 ? j1 - jb : 64);
            const long long ord = order[jb + (lane < cnt ? lane : 0)];
            const int lo = (int)(ord & 0xffffffffll), hi = (int)(ord >> 32);
            const auto row_of = [&](int t) {
                return g + (((long long)__builtin_amdgcn_readlane(hi, t) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, t)) * ldg + c0 + lane;
            };
            int t = 0;
            for (; t + 16 <= cnt && jb - j0 + t + 16 <= n4; t += 16) {     // four groups at once: sixteen rows in flight, added in group order
                float v[16][kQ];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float *gp = row_of(t + u);
#pragma unroll
                    for (int q = 0; q < kQ; ++q) v[u][q] = c0 + lane + 64 * q < d ? gp[64 * q] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int q = 0; q < kQ; ++q) s[q][u & 3] += v[u][q];
            }
            for (; t + 4 <= cnt && jb - j0 + t + 4 <= n4; t += 4) {
                float v[4][kQ];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float *gp = row_of(t + u);
#pragma unroll
                    for (int q = 0; q < kQ; ++q) v[u][q] = c0 + lane + 64 * q < d ? gp[64 * q] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < kQ; ++q) s[q][u] += v[u][q];
            }
            for (; t < cnt; ++t) {
                const float *gp = row_of(t);
#pragma unroll
                for (int q = 0; q < kQ; ++q)
                    if (c0 + lane + 64 * q < d) s[q][0] += gp[64 * q];
            }
        }
#pragma unroll
        for (int q = 0; q < kQ; ++q)
            if (c0 + lane + 64 * q < d) out[r * ldo + c0 + lane + 64 * q] = (s[q][0] + s[q][1]) + (s[q][2] + s[q][3]);
    }
}

extern "C" int ngcf_segment_sum_rows_f32(const float *g, int64_t ldg, int d, const int64_t *order, const int64_t *segptr,
                                         int64_t n_seg, const int64_t *dst_rows, const int64_t *n_seg_dev, float *out, int64_t ldo,
                                         int64_t n_out_rows, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n_seg == 0) return NGCF_OK;
    if (!g || !order || !segptr || !out || d <= 0 || ldg < d || ldo < d || n_seg < 0) return fail(NGCF_ERR_ARG, "segment_sum_rows: bad argument");
    if (dst_rows && n_out_rows <= 0) return fail(NGCF_ERR_ARG, "segment_sum_rows: the scatter form needs the row count of out");
    if ((n_seg + 3) / 4 >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "segment_sum_rows: too many rows");
    segment_sum_rows_kernel<<<dim3((unsigned)((n_seg + 3) / 4)), 256, 0, stream>>>(g, ldg, d, order, segptr, n_seg, dst_rows, n_seg_dev, out, ldo,
                                                                                   n_out_rows);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---- normalise + dropout + LeakyReLU backward: (dN, dC, C) -> dM, one wave per row ------------
// forward: A = leaky(M); C = keep ? A/(1-p) : 0; N = C / max(|C|, eps)    (NGCF.py:140-144)
// VEC = 2 (r04): a lane owns the column pairs 2 lane, 2 lane + 128, .. (8-byte accesses: at d = 128 one load per operand and row
// instead of two; same arithmetic per element and the same order in the two row sums - lanes hold other columns, the wave sum is a
// sum over all of them either way - so only the association of those two sums differs from VEC = 1).
template <int VEC>
__global__ __launch_bounds__(256) void layer_bwd_pre_kernel(const float *__restrict__ dN, int64_t ldn,
                                                            const float *__restrict__ dC, int64_t ldc,
                                                            const float *__restrict__ C, int64_t ldC, int64_t n_rows,
                                                            int d, float leaky, float drop_p, uint64_t seed_in,
                                                            const float *__restrict__ drop_mask, int64_t ldk,
                                                            const int64_t *__restrict__ row_ids,
                                                            float *__restrict__ dM, int64_t ldm)
{
    const uint64_t seed = drop_p > 0.f ? resolve_seed(seed_in) : seed_in;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int64_t r_hash = row_ids ? row_ids[r] : r;         // compacted rows: the hash stream is indexed by the row of the matrix
    const int lane = threadIdx.x & 63;
    const float *c = C + r * ldC, *g = dN ? dN + r * ldn : nullptr;   // dN == nullptr: no gradient through the normalised block
    float den = 1.f, ydot = 0.f;
    if (g) {
        float ss = 0.f, dot = 0.f;
        for (int j = lane * VEC; j < d; j += 64 * VEC) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) {
                ss = fmaf(c[j + q], c[j + q], ss);
                dot = fmaf(c[j + q], g[j + q], dot);
            }
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float nrm = sqrtf(ss);
        const bool clamped = nrm < 1e-12f;                   // F.normalize's clamp_min: N = C / eps there
        den = clamped ? 1e-12f : nrm;
        ydot = clamped ? 0.f : dot / (den * den);            // (y.dy)/|x| with y = x/|x|
    }
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    using V = typename VecT<VEC>::type;
    for (int j = lane * VEC; j < d; j += 64 * VEC) {
        float cv[VEC], gv[VEC], dc[VEC], mk[VEC], out[VEC];
        *reinterpret_cast<V *>(cv) = *reinterpret_cast<const V *>(c + j);
        if (g) *reinterpret_cast<V *>(gv) = *reinterpret_cast<const V *>(g + j);
        if (dC) *reinterpret_cast<V *>(dc) = *reinterpret_cast<const V *>(dC + r * ldc + j);
        if (drop_mask) *reinterpret_cast<V *>(mk) = *reinterpret_cast<const V *>(drop_mask + r * ldk + j);
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float t = g ? gv[q] / den - cv[q] * (ydot / den) : 0.f;
            if (dC) t += dc[q];
            if (drop_mask) {
                t *= mk[q];                                   // the host-drawn noise tensor of the forward (0 or 1/(1-p))
            } else if (drop_p > 0.f) {
                const uint32_t h = mix32(seed ^ ((uint64_t)r_hash * 0x9E3779B97F4A7C15ULL + (uint64_t)(j + q)));
                t = h < thr ? 0.f : t * keep_scale;
            }
            out[q] = t * (cv[q] > 0.f ? 1.f : leaky);         // sign(C) == sign(M) wherever C was kept
        }
        *reinterpret_cast<V *>(dM + r * ldm + j) = *reinterpret_cast<const V *>(out);
    }
}

extern "C" int ngcf_layer_bwd_pre_f32(const float *dN, int64_t ldn, const float *dC, int64_t ldc, const float *C, int64_t ldC,
                                      int64_t n_rows, int d, float leaky, float drop_p, uint64_t seed, const float *drop_mask,
                                      int64_t ld_mask, const int64_t *row_ids, float *dM, int64_t ldm, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows == 0) return NGCF_OK;
    if ((!dN && !dC) || !C || !dM || d <= 0) return fail(NGCF_ERR_ARG, "layer_bwd_pre: bad argument");
    auto even = [](const float *p, int64_t ld) { return !p || (ld % 2 == 0 && ((uintptr_t)p & 7) == 0); };
    if (d % 2 == 0 && even(dN, ldn) && even(dC, ldc) && even(C, ldC) && even(drop_mask, ld_mask) && even(dM, ldm))
        layer_bwd_pre_kernel<2><<<dim3((unsigned)((n_rows + 3) / 4)), 256, 0, stream>>>(dN, ldn, dC, ldc, C, ldC, n_rows, d, leaky, drop_p,
                                                                                        seed, drop_mask, ld_mask, row_ids, dM, ldm);
    else
        layer_bwd_pre_kernel<1><<<dim3((unsigned)((n_rows + 3) / 4)), 256, 0, stream>>>(dN, ldn, dC, ldc, C, ldC, n_rows, d, leaky, drop_p,
                                                                                        seed, drop_mask, ld_mask, row_ids, dM, ldm);
    LAUNCH_CHECK();
    return NGCF_OK;
}


// out[r, :] += add[r, :]   (dE = dE_direct + L^T.dLE accumulation)
__global__ void add_rows_kernel(float *__restrict__ out, int64_t ldo, const float *__restrict__ add, int64_t lda, int64_t n_rows, int d)
{
    const int64_t total = n_rows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const int j = (int)(i % d);
        out[r * ldo + j] += add[r * lda + j];
    }
}

// the same on 16-byte pieces (widths and leading dimensions that are multiples of 4, aligned rows: every padded matrix of the backward)
__global__ void add_rows4_kernel(float *__restrict__ out, int64_t ldo, const float *__restrict__ add, int64_t lda, int64_t n_rows, int d4)
{
    const int64_t total = n_rows * d4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d4;
        const int j = (int)(i - r * d4) * 4;
        float4 a = *reinterpret_cast<const float4 *>(out + r * ldo + j);
        const float4 b = *reinterpret_cast<const float4 *>(add + r * lda + j);
        a.x += b.x, a.y += b.y, a.z += b.z, a.w += b.w;
        *reinterpret_cast<float4 *>(out + r * ldo + j) = a;
    }
}

extern "C" int ngcf_add_rows_f32(float *out, int64_t ldo, const float *add, int64_t lda, int64_t n_rows, int d, void *stream_)
{
    if (n_rows == 0) return NGCF_OK;
    if (!out || !add || d <= 0) return fail(NGCF_ERR_ARG, "add_rows: bad argument");
    if (d % 4 == 0 && ldo % 4 == 0 && lda % 4 == 0 && aligned16(out) && aligned16(add)) {
        add_rows4_kernel<<<grid_for(n_rows * (d / 4), 256), 256, 0, (hipStream_t)stream_>>>(out, ldo, add, lda, n_rows, d / 4);
        LAUNCH_CHECK();
        return NGCF_OK;
    }
    add_rows_kernel<<<grid_for(n_rows * d, 256), 256, 0, (hipStream_t)stream_>>>(out, ldo, add, lda, n_rows, d);
    LAUNCH_CHECK();
    return NGCF_OK;
}


// ---- weight gradients of a layer on the fp32 matrix cores --------------------------------------
//   gW[o][c]        = sum_rows dM[row][o] * (LE + E)[row][c]     (d loss / d W1, NGCF.py:131-133)
//   gW[o][d_in + c] = sum_rows dM[row][o] * (LE * E)[row][c]     (d loss / d W2, NGCF.py:135-136)
// A [d_out x 2 d_in] result with the 1.1 M rows as the reduction dimension: a library GEMM picks a 32x32 macro
// tile for this shape (2.3 ms at C3) and needs the [LE+E | LE*E] operand materialised (1.1 GB).  Here 256
// persistent workgroups stream blocks of 32 rows through LDS (the sum/product operand is formed on the way in),
// 8 waves each keep 4 of the 32 output tiles in registers (v_mfma_f32_32x32x2_f32: A = dM^T, B = [S|P], k = row),
// and write one partial result per workgroup; a second kernel adds the partials in workgroup order (fixed order,
// no atomics).  Widths up to 128 (padded to multiples of 32 inside LDS); wider layers use the library GEMM.
typedef float bw_f32x16 __attribute__((ext_vector_type(16)));
typedef float bw_f32x4 __attribute__((ext_vector_type(4)));
static constexpr int kBwRows = 32;        // rows per LDS block
static constexpr int kBwWGs = 256;
static constexpr int kBwM = 128, kBwN = 256;

template <bool ALIGNED>
__global__ __launch_bounds__(512) void bwd_weight_kernel(const float *__restrict__ dM, int64_t ldM,
                                                         const float *__restrict__ LE, int64_t ldLE,
                                                         const float *__restrict__ E, int64_t ldE, int64_t n_rows, int d_in,
                                                         int d_out, int P, float *__restrict__ partial,
                                                         float *__restrict__ partial_bias)
{
    __shared__ float As[2][kBwRows][kBwM];     // dM rows, columns >= d_out stay zero
    __shared__ float Bs[2][kBwRows][kBwN];     // [LE+E (P columns) | LE*E (P columns)], the rest stays zero
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * kBwRows * kBwM; i += 512) (&As[0][0][0])[i] = 0.f;
    for (int i = tid; i < 2 * kBwRows * kBwN; i += 512) (&Bs[0][0][0])[i] = 0.f;
    bw_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    // staging: 2 float4 slots per thread and operand: slot s -> row s / 32, columns 4 * (s % 32) ..
    bw_f32x4 rm[2], rl[2], re[2];
    const int64_t n_blocks = (n_rows + kBwRows - 1) / kBwRows;
    auto load4 = [&](const float *base, int64_t ld, int64_t row, int c, int width) -> bw_f32x4 {
        bw_f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < n_rows && c < width) {
            const float *p = base + row * ld + c;
            if (ALIGNED && c + 4 <= width) {
                v = *reinterpret_cast<const bw_f32x4 *>(p);
            } else {
                v.x = p[0];
                if (c + 1 < width) v.y = p[1];
                if (c + 2 < width) v.z = p[2];
                if (c + 3 < width) v.w = p[3];
            }
        }
        return v;
    };
    auto load_block = [&](int64_t b) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int s = tid + s2 * 512, r = s >> 5, c = (s & 31) * 4;
            const int64_t row = b * kBwRows + r;
            rm[s2] = load4(dM, ldM, row, c, d_out);
            rl[s2] = load4(LE, ldLE, row, c, d_in);
            re[s2] = load4(E, ldE, row, c, d_in);
        }
    };
    auto store_block = [&](int buf) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int s = tid + s2 * 512, r = s >> 5, c = (s & 31) * 4;
            *reinterpret_cast<bw_f32x4 *>(&As[buf][r][c]) = rm[s2];
            if (c < P) {
                *reinterpret_cast<bw_f32x4 *>(&Bs[buf][r][c]) = rl[s2] + re[s2];
                *reinterpret_cast<bw_f32x4 *>(&Bs[buf][r][P + c]) = rl[s2] * re[s2];
            }
        }
    };
    __syncthreads();                           // LDS zeroed
    int64_t b = blockIdx.x;
    if (b < n_blocks) {
        load_block(b);
        store_block(0);
    }
    __syncthreads();
    int buf = 0;
    for (; b < n_blocks; b += gridDim.x, buf ^= 1) {
        const bool more = b + gridDim.x < n_blocks;
        if (more) load_block(b + gridDim.x);   // the next block's global loads fly under the MFMAs
        // bias gradient = column sums of dM (NGCF.py:131-136: b1 enters twice, b2 once - the caller scales): thread (o, q)
        // adds rows 8 q .. 8 q + 7 of column o of the block that is in LDS anyway
#pragma unroll
        for (int r = 0; r < kBwRows / 4; ++r) bsum += As[buf][(tid >> 7) * (kBwRows / 4) + r][tid & (kBwM - 1)];
#pragma unroll
        for (int j = 0; j < kBwRows / 2; ++j) {
            const float bv = Bs[buf][2 * j + lh][wave * 32 + li];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[buf][2 * j + lh][t * 32 + li], bv, acc[t], 0, 0, 0);
        }
        if (more) store_block(buf ^ 1);
        __syncthreads();
    }
    // the partial leaves COMPACT: [d_out][2 d_in] contiguous per workgroup (r03: the padded [128][256] tile made the reduction
    // read a 4-byte word out of every 128 KB per workgroup and output element)
    float *out = partial + (int64_t)blockIdx.x * d_out * 2 * d_in;
    const int col = wave * 32 + li;                       // column of the [P | P] operand: sums in [0, P), products in [P, 2 P)
    const int cc = col < P ? col : d_in + (col - P);      // its place in [d_in | d_in]
    const bool col_ok = col < P ? col < d_in : (col - P) < d_in;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (col_ok && o < d_out) out[o * 2 * d_in + cc] = acc[t][r];
        }
    // the four row-quarters of a column, added in a fixed order (the last loop iteration ended with a barrier: As is free)
    float *bs = &As[0][0][0];
    bs[tid] = bsum;
    __syncthreads();
    if (tid < kBwM) partial_bias[(int64_t)blockIdx.x * kBwM + tid] = ((bs[tid] + bs[kBwM + tid]) + bs[2 * kBwM + tid]) + bs[3 * kBwM + tid];
}

// 64 output elements per workgroup; its four waves each add a quarter of the partials (eight independent load chains, combined in
// a fixed order), the quarters are added in wave order: the result does not depend on anything but n_wg.  (r03: one thread per
// element walking all partials - and one per bias element walking them as a single dependent chain - took 26 us of load latency
// on the Seoul graph's 93 partials.)
__global__ __launch_bounds__(256) void bwd_weight_reduce_kernel(const float *__restrict__ partial, const float *__restrict__ partial_bias,
                                                                int n_wg, int d_in, int d_out, float *__restrict__ gW1, int64_t ld1,
                                                                float *__restrict__ gW2, int64_t ld2, float *__restrict__ gb1,
                                                                float *__restrict__ gb2)
{
    __shared__ float quarter[4][64];
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int n_w = d_out * 2 * d_in;
    const int i = blockIdx.x * 64 + e;
    const bool is_w = i < n_w, is_b = !is_w && (gb1 || gb2) && i - n_w < d_out;
    float r = 0.f;
    if (is_w || is_b) {
        const float *p = is_w ? partial + i : partial_bias + (i - n_w);
        const int64_t step = is_w ? (int64_t)n_w : (int64_t)kBwM;
        const int per = (n_wg + 3) / 4, w0 = q * per, w1 = min(n_wg, w0 + per);
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
        int w = w0;
        for (; w + 8 <= w1; w += 8) {
            s0 += p[(w + 0) * step];
            s1 += p[(w + 1) * step];
            s2 += p[(w + 2) * step];
            s3 += p[(w + 3) * step];
            s4 += p[(w + 4) * step];
            s5 += p[(w + 5) * step];
            s6 += p[(w + 6) * step];
            s7 += p[(w + 7) * step];
        }
        if (w + 0 < w1) s0 += p[(w + 0) * step];
        if (w + 1 < w1) s1 += p[(w + 1) * step];
        if (w + 2 < w1) s2 += p[(w + 2) * step];
        if (w + 3 < w1) s3 += p[(w + 3) * step];
        if (w + 4 < w1) s4 += p[(w + 4) * step];
        if (w + 5 < w1) s5 += p[(w + 5) * step];
        if (w + 6 < w1) s6 += p[(w + 6) * step];
        r = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    }
    quarter[q][e] = r;
    __syncthreads();
    if (q != 0) return;
    r = ((quarter[0][e] + quarter[1][e]) + quarter[2][e]) + quarter[3][e];
    if (is_w) {
        const int o = i / (2 * d_in), c = i % (2 * d_in);
        if (c < d_in) gW1[(int64_t)o * ld1 + c] = r;
        else gW2[(int64_t)o * ld2 + (c - d_in)] = r;
    } else if (is_b) {
        const int o = i - n_w;
        if (gb2) gb2[o] = r;
        if (gb1) gb1[o] = 2.0f * r;          // b1 enters the layer twice (NGCF.py:131,133)
    }
}

// A NARROW block of input columns (d_in <= 4: the 1..3 columns the reference's widths leave beyond a multiple of 128 - 130 -> 2,
// 515 -> 3; autograd._bwd_weight cuts the layer into blocks of 128 input columns).  The matrix-core kernel above multiplies the
// zero padding of such a block along at the full block's price (0.78 ms for the two last columns of C3's 130-wide first layer);
// this is a pass over dM with the 2 x d_in operand values of a row as scalars: thread o of a 128-thread group adds
// dM[row][o] * (LE + E)[row][c] and dM[row][o] * (LE * E)[row][c] over its workgroup's rows (coalesced rows of dM, a fixed order),
// two row groups per workgroup combined through LDS; the same compact per-workgroup partials, so the same reduction follows.
__global__ __launch_bounds__(256) void bwd_weight_narrow_kernel(const float *__restrict__ dM, int64_t ldM, const float *__restrict__ LE,
                                                                int64_t ldLE, const float *__restrict__ E, int64_t ldE, int64_t n_rows,
                                                                int d_in, int d_out, float *__restrict__ partial,
                                                                float *__restrict__ partial_bias)
{
    __shared__ float comb[9][kBwM];
    const int o = threadIdx.x & (kBwM - 1), half = threadIdx.x >> 7;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, p[4] = {0.f, 0.f, 0.f, 0.f}, b = 0.f;
    const int64_t per = (n_rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
    // eight rows per step and row group: eight independent loads of dM in flight per thread (rows past the end contribute zeros)
    for (int64_t row = r0 + 8 * half; row < r1; row += 16) {
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = (o < d_out && row + q < r1) ? dM[(row + q) * ldM + o] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t rq = row + q < r1 ? row + q : r1 - 1;                 // (clamped: m[q] is zero there)
            b += m[q];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < d_in) {
                    const float le = LE[rq * ldLE + c], e = E[rq * ldE + c];    // the same address in every lane: one broadcast load
                    s[c] = fmaf(m[q], le + e, s[c]);
                    p[c] = fmaf(m[q], le * e, p[c]);
                }
        }
    }
    if (half == 1) {
#pragma unroll
        for (int c = 0; c < 4; ++c) comb[c][o] = s[c], comb[4 + c][o] = p[c];
        comb[8][o] = b;
    }
    __syncthreads();
    if (half == 1 || o >= d_out) return;
    float *out = partial + (int64_t)blockIdx.x * d_out * 2 * d_in;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < d_in) {
            out[o * 2 * d_in + c] = s[c] + comb[c][o];
            out[o * 2 * d_in + d_in + c] = p[c] + comb[4 + c][o];
        }
    partial_bias[(int64_t)blockIdx.x * kBwM + o] = b + comb[8][o];
}

extern "C" int64_t ngcf_bwd_weight_workspace_bytes(void)
{
    return (int64_t)kBwWGs * kBwM * kBwN * sizeof(float) + (int64_t)kBwWGs * kBwM * sizeof(float) + 256;
}

extern "C" int ngcf_layer_bwd_weight_f32(const float *dM, int64_t ldM, const float *LE, int64_t ldLE, const float *E,
                                         int64_t ldE, int64_t n_rows, int d_in, int d_out, float *gW1, int64_t ld1, float *gW2,
                                         int64_t ld2, float *gb1, float *gb2, void *workspace, int64_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!gW1 || !gW2 || ld1 < d_in || ld2 < d_in || (n_rows > 0 && (!dM || !LE || !E)))
        return fail(NGCF_ERR_ARG, "layer_bwd_weight: null argument or leading dimension too small");
    if (n_rows < 0 || d_in < 1 || d_out < 1 || d_in > 128 || d_out > 128)
        return fail(NGCF_ERR_ARG, "layer_bwd_weight: widths d_in=%d d_out=%d not in 1..128", d_in, d_out);
    if (ldM < d_out || ldLE < d_in || ldE < d_in) return fail(NGCF_ERR_ARG, "layer_bwd_weight: leading dimension too small");
    if (!workspace || workspace_bytes < ngcf_bwd_weight_workspace_bytes())
        return fail(NGCF_ERR_WORKSPACE, "layer_bwd_weight: workspace %lld B < %lld B", (long long)workspace_bytes,
                    (long long)ngcf_bwd_weight_workspace_bytes());
    float *partial = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
    float *partial_bias = partial + (int64_t)kBwWGs * kBwM * kBwN;
    const int P = (int)align_up(d_in, 32);
    const bool al = ldM % 4 == 0 && ldLE % 4 == 0 && ldE % 4 == 0 && aligned16(dM) && aligned16(LE) && aligned16(E);
    // workgroups: one per CU on a large matrix; on a small one (the Seoul graph's 5 940 rows are 186 blocks, a compacted last
    // layer a few dozen) every workgroup should still see >= 2 blocks - each writes a 128 KB partial that the reduction reads
    // back (256 of them: 33 MB and 116 us for a 65 x 130 gradient)
    const int64_t n_blocks = (n_rows + kBwRows - 1) / kBwRows;
    const int n_wg = (int)std::min<int64_t>(kBwWGs, std::max<int64_t>(1, (n_blocks + 1) / 2));
    if (d_in <= 4 && n_rows >= 65536) {       // a narrow remainder block of a large matrix: no matrix cores for 2 x d_in columns
        const int n_nwg = kBwWGs * 8;         // memory-bound: eight workgroups per CU
        float *nbias = partial + (int64_t)n_nwg * d_out * 2 * d_in;          // (behind the weight partials: 1024 x 128 floats fit easily)
        bwd_weight_narrow_kernel<<<n_nwg, 256, 0, stream>>>(dM, ldM, LE, ldLE, E, ldE, n_rows, d_in, d_out, partial, nbias);
        LAUNCH_CHECK();
        const int total_n = d_out * 2 * d_in + ((gb1 || gb2) ? d_out : 0);
        bwd_weight_reduce_kernel<<<(total_n + 63) / 64, 256, 0, stream>>>(partial, nbias, n_nwg, d_in, d_out, gW1, ld1, gW2, ld2, gb1, gb2);
        LAUNCH_CHECK();
        return NGCF_OK;
    }
    if (al)
        bwd_weight_kernel<true><<<n_wg, 512, 0, stream>>>(dM, ldM, LE, ldLE, E, ldE, n_rows, d_in, d_out, P, partial, partial_bias);
    else
        bwd_weight_kernel<false><<<n_wg, 512, 0, stream>>>(dM, ldM, LE, ldLE, E, ldE, n_rows, d_in, d_out, P, partial, partial_bias);
    LAUNCH_CHECK();
    const int total = d_out * 2 * d_in + ((gb1 || gb2) ? d_out : 0);
    bwd_weight_reduce_kernel<<<(total + 63) / 64, 256, 0, stream>>>(partial, partial_bias, n_wg, d_in, d_out, gW1, ld1, gW2, ld2, gb1, gb2);
    LAUNCH_CHECK();
    return NGCF_OK;
}


// =============================================================================================
// Input gradients of a layer's dense half in one kernel (r02; replaces a library GEMM + the combine kernel and the
// [N, 2 d_in] intermediate between them):
//   dS = dM . W1,  dP = dM . W2                     ([n_rows, d_in] each; W1, W2 are nn.Linear weights [d_out, d_in])
//   dLE = dS + dP * E,   dE = dS + dP * LE          (NGCF.py:131-136 differentiated)
// Same structure as layer_dense_kernel (dense.hip): a workgroup of 4 waves owns 128 rows, each wave a 32 x 128 output
// panel as four 32x32 tiles of v_mfma_f32_32x32x2_f32; K = d_out is walked in chunks of 32 through double-buffered LDS
// (A: rows of dM, B: the weight rows W[k, col0 .. col0+128), which need no transpose).  The K loop runs twice over the same
// rows of dM (second read from L2), once per weight matrix, so that both 64-register accumulators are live only in the
// epilogue, where a lane holds dS and dP of the same (row, column) and forms both outputs.
// =============================================================================================
#define NGCF_BI_KC 32
static constexpr int kBiRows = 128, kBiMaxCols = 160;

// Wp[half][chunk][k][c] = W_half[chunk * 32 + k][col0 + c], c < wcols   (zero outside the matrix)
__global__ void bwd_input_pack_kernel(const float *__restrict__ W1, const float *__restrict__ W2, int d_out, int d_in, int col0,
                                      int wcols, int n_chunks, float *__restrict__ Wp)
{
    const int per_half = n_chunks * NGCF_BI_KC * wcols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * per_half; i += gridDim.x * blockDim.x) {
        const int half = i / per_half, x = i % per_half;
        const int c = x % wcols, k = x / wcols;
        const int col = col0 + c;
        Wp[i] = (k < d_out && col < d_in) ? (half ? W2 : W1)[(int64_t)k * d_in + col] : 0.f;
    }
}

// NT 32x32 tiles per wave: a panel of 32*NT input columns (NT = 4: 128; NT = 5: 160, which takes the reference's 130-wide
// first layer in one panel instead of two)
// SMALL (r03): a workgroup owns 32 rows and its four waves split the panel's columns (one 32x32 tile each at NT = 4) instead of
// 128 rows with a wave per 32 of them: on a matrix of a few thousand rows (the Seoul graph's 5 940: 47 workgroups of the tall shape
// on 256 CUs, 34 us) the panel is spread over 186 workgroups.
template <int NT, bool SMALL>
__global__ __launch_bounds__(256, 2) void layer_bwd_input_kernel(const float *__restrict__ dM, int64_t ldM, int64_t n_rows, int d_out,
                                                              const float *__restrict__ Wp, int n_chunks,
                                                              const float *__restrict__ LE, int64_t ldLE,
                                                              const float *__restrict__ E, int64_t ldE, int d_in, int col0,
                                                              float *__restrict__ dLE, int64_t ldd, float *__restrict__ dE, int64_t lde)
{
    constexpr int BM = SMALL ? 32 : kBiRows, WCOLS = 32 * NT, XLD = NGCF_BI_KC + 4;
    constexpr int NTW = SMALL ? NT / 4 : NT;          // tiles per wave
    constexpr int XJ = BM * 8 / 256;                  // float4 of a dM chunk per thread
    static_assert(!SMALL || NT % 4 == 0, "the small shape splits the panel's tiles over four waves");
    __shared__ float Xs[2 * BM * XLD];
    __shared__ float Ws[2 * NGCF_BI_KC * WCOLS];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * BM;
    const int d4 = (d_out + 3) & ~3;
    bw_f32x16 acc[2][NTW];
    bw_f32x4 xreg[XJ], wreg[NT];
    const int tile0 = SMALL ? wave * NTW : 0;         // first tile (32 columns) of this wave
    const int wrow = SMALL ? 0 : wave * 32;           // first row of this wave inside the workgroup's rows

    auto load_chunk = [&](int half, int chunk) {      // global -> registers (rows past the end re-read the last row, never stored)
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int f = tid + 256 * j;
            int64_t grow = row0 + f / 8;
            grow = grow < n_rows ? grow : n_rows - 1;
            const int c0 = chunk * NGCF_BI_KC + (f % 8) * 4;
            const int cc = c0 < d4 ? c0 : d4 - 4;
            bw_f32x4 a = *reinterpret_cast<const bw_f32x4 *>(dM + grow * ldM + cc);
            a.x = c0 < d_out ? a.x : 0.f;
            a.y = c0 + 1 < d_out ? a.y : 0.f;
            a.z = c0 + 2 < d_out ? a.z : 0.f;
            a.w = c0 + 3 < d_out ? a.w : 0.f;
            xreg[j] = a;
        }
        const bw_f32x4 *src = reinterpret_cast<const bw_f32x4 *>(Wp + ((int64_t)half * n_chunks + chunk) * NGCF_BI_KC * WCOLS);
#pragma unroll
        for (int j = 0; j < NT; ++j) wreg[j] = src[tid + 256 * j];       // 32 x WCOLS floats = NT float4 per thread
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            const int f = tid + 256 * j;
            *reinterpret_cast<bw_f32x4 *>(Xs + buf * (BM * XLD) + (f / 8) * XLD + (f % 8) * 4) = xreg[j];
        }
        bw_f32x4 *dst = reinterpret_cast<bw_f32x4 *>(Ws + buf * (NGCF_BI_KC * WCOLS));
#pragma unroll
        for (int j = 0; j < NT; ++j) dst[tid + 256 * j] = wreg[j];
    };
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[half][t][r] = 0.f;
        load_chunk(half, 0);
        store_chunk(0);                                // (the previous half's loop ended with a barrier)
        __syncthreads();
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const bool more = chunk + 1 < n_chunks;
            if (more) load_chunk(half, chunk + 1);      // global loads fly under the MFMAs
            const int buf = chunk & 1;
            const float *X = Xs + buf * (BM * XLD) + (wrow + li) * XLD + lh * 4;
            const float *W = Ws + buf * (NGCF_BI_KC * WCOLS) + tile0 * 32 + li;
#pragma unroll
            for (int kb = 0; kb < NGCF_BI_KC / 8; ++kb) {
                const bw_f32x4 a4 = *reinterpret_cast<const bw_f32x4 *>(X + kb * 8);
                const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const float *wk = W + (kb * 8 + lh * 4 + sx) * WCOLS;
                    float bv[NTW];
#pragma unroll
                    for (int t = 0; t < NTW; ++t) bv[t] = wk[t * 32];
#pragma unroll
                    for (int t = 0; t < NTW; ++t) acc[half][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], bv[t], acc[half][t], 0, 0, 0);
                }
            }
            if (more) store_chunk((chunk + 1) & 1);
            __syncthreads();
        }
    }
    // epilogue: dLE = dS + dP * E, dE = dS + dP * LE
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t grow = row0 + wrow + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (grow >= n_rows) continue;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int col = col0 + (tile0 + t) * 32 + li;
            if (col < d_in) {
                const float ds = acc[0][t][r], dp = acc[1][t][r];
                dLE[grow * ldd + col] = fmaf(dp, E[grow * ldE + col], ds);
                dE[grow * lde + col] = fmaf(dp, LE[grow * ldLE + col], ds);
            }
        }
    }
}


// ---- r04: the same product with the weights RESIDENT in LDS (the structure of layer_dense_resident_kernel, dense.hip) --------
// The staged kernel above runs at 37 % of the fp32 matrix peak at C3 (1.23 ms per 1.1 M-row layer): its two K loops over staged
// chunks (8 barriers per 128 rows, dM read twice, the W chunks re-staged by every workgroup) and its epilogue (2 loads + 2 stores per
// output element) run one after the other, so matrix time and memory time add up.  Here [W1 | W2] for one panel of 128 input
// columns - K x 256 floats, 128 KB at K = 128 - is loaded into LDS ONCE per persistent workgroup (8 waves, one workgroup per CU);
// a wave owns 32 rows outright: each lane reads the 16-byte pieces of ITS row of dM straight from global memory into the MFMA A
// layout, two chunks (of 16 k) ahead in two fixed register sets, and every k-pair feeds EIGHT accumulator tiles (dS and dP of the
// four 32-column tiles) - dM is read once, there is no barrier after the prologue, and the next tile's first chunks are requested
// before the epilogue of the current one.  The k order per output element is the staged kernel's (ascending k-pairs), so the
// results are bit-identical to it.
static constexpr int kBiResWaves = 8, kBiResWGs = 256, kBiResDC = 16;
#ifndef NGCF_BI_EPI_DEPTH
#define NGCF_BI_EPI_DEPTH 2
#endif

// Wr[k][0..127] = W1[k][t * 32 + li] at [li * 4 + t], Wr[k][128..255] the same of W2 (zero outside the matrices): a lane's four
// tile values of one k are 16 contiguous bytes, a wave's reads of one k are 512 contiguous bytes (conflict-free ds_read_b128)
__global__ void bwd_input_pack_resident_kernel(const float *__restrict__ W1, const float *__restrict__ W2, int d_out, int d_in, int col0,
                                               int k_pad, float *__restrict__ Wr)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < k_pad * 256; i += gridDim.x * blockDim.x) {
        const int k = i >> 8, x = i & 255, half = x >> 7, y = x & 127;
        const int col = col0 + (y & 3) * 32 + (y >> 2);
        Wr[i] = (k < d_out && col < d_in) ? (half ? W2 : W1)[(int64_t)k * d_in + col] : 0.f;
    }
}

__global__ __launch_bounds__(kBiResWaves * 64) void layer_bwd_input_resident_kernel(
    const float *__restrict__ dM, int64_t ldM, int64_t n_rows, int d_out, const float *__restrict__ Wr, int n_chunks,
    const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ E, int64_t ldE, int d_in, int col0, float *__restrict__ dLE,
    int64_t ldd, float *__restrict__ dE, int64_t lde)
{
    extern __shared__ float Wres[];                // [n_chunks * 16][256]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    {
        const bw_f32x4 *src = reinterpret_cast<const bw_f32x4 *>(Wr);
        bw_f32x4 *dst = reinterpret_cast<bw_f32x4 *>(Wres);
        const int n4 = n_chunks * kBiResDC * 256 / 4;
        for (int i = tid; i < n4; i += kBiResWaves * 64) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t n_tiles = (n_rows + 31) / 32;
    const int d4 = (d_out + 3) & ~3;
    const float *W = Wres + li * 4;
    const int last = n_chunks - 1;
    const int64_t tile_step = (int64_t)gridDim.x * kBiResWaves;
    auto row_of = [&](int64_t t) {                 // the lane's row of tile t (rows past the end re-read the last row, never stored)
        const int64_t g = t * 32 + li;
        return g < n_rows ? g : n_rows - 1;
    };
    // the lane's two 16-byte pieces of a chunk: dM at columns c*16 + lh*4 (a) and c*16 + 8 + lh*4 (b); columns past d_out are
    // re-read from the row's last float4 and zeroed at use
    auto fetch = [&](const float *row, int c, bw_f32x4 &a, bw_f32x4 &b) {
        const int ca = c * kBiResDC + lh * 4, cb = ca + 8;
        a = *reinterpret_cast<const bw_f32x4 *>(row + (ca < d4 ? ca : d4 - 4));
        b = *reinterpret_cast<const bw_f32x4 *>(row + (cb < d4 ? cb : d4 - 4));
    };
    bw_f32x4 a0, b0, a1, b1;                        // two chunks of look-ahead in two fixed register sets (every prefetch unconditional)
    int64_t tile = (int64_t)blockIdx.x * kBiResWaves + wave;
    {
        const float *r0 = dM + row_of(tile < n_tiles ? tile : 0) * ldM;
        fetch(r0, 0, a0, b0);
        fetch(r0, last < 1 ? last : 1, a1, b1);
    }
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t row0 = tile * 32;
        const float *m_row = dM + row_of(tile) * ldM;
        bw_f32x16 accS[4], accP[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) accS[t][r] = 0.f, accP[t][r] = 0.f;
        auto chunk_mfma = [&](int c, bw_f32x4 a, bw_f32x4 b) {
            const int ca = c * kBiResDC + lh * 4, cb = ca + 8;
            if (cb + 4 > d_out) {                   // only the last chunk of a width that is not a multiple of 16
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ca + q >= d_out) a[q] = 0.f;
                    if (cb + q >= d_out) b[q] = 0.f;
                }
            }
            const float *wc = W + (int64_t)c * kBiResDC * 256;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const bw_f32x4 av = kb ? b : a;
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const float *wk = wc + (kb * 8 + lh * 4 + sx) * 256;
                    const bw_f32x4 v1 = *reinterpret_cast<const bw_f32x4 *>(wk);
                    const bw_f32x4 v2 = *reinterpret_cast<const bw_f32x4 *>(wk + 128);
                    accS[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v1.x, accS[0], 0, 0, 0);
                    accS[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v1.y, accS[1], 0, 0, 0);
                    accS[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v1.z, accS[2], 0, 0, 0);
                    accS[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v1.w, accS[3], 0, 0, 0);
                    accP[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v2.x, accP[0], 0, 0, 0);
                    accP[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v2.y, accP[1], 0, 0, 0);
                    accP[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v2.z, accP[2], 0, 0, 0);
                    accP[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sx], v2.w, accP[3], 0, 0, 0);
                }
            }
        };
        int c = 0;
        for (; c + 1 < n_chunks; c += 2) {
            {
                const bw_f32x4 ua = a0, ub = b0;
                fetch(m_row, c + 2 < last ? c + 2 : last, a0, b0);          // in flight under two chunks of MFMAs
                __builtin_amdgcn_sched_barrier(0);
                chunk_mfma(c, ua, ub);
            }
            {
                const bw_f32x4 ua = a1, ub = b1;
                fetch(m_row, c + 3 < last ? c + 3 : last, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                chunk_mfma(c + 1, ua, ub);
            }
        }
        if (c < n_chunks) chunk_mfma(c, a0, b0);
        {   // the next tile's first two chunks, ahead of this tile's epilogue (the last tile of a wave re-reads its own)
            const float *rn = dM + row_of(tile + tile_step < n_tiles ? tile + tile_step : tile) * ldM;
            fetch(rn, 0, a0, b0);
            fetch(rn, last < 1 ? last : 1, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: dLE = dS + dP * E, dE = dS + dP * LE.  Full tiles take a path without per-element tests: behind a branch the
        // compiler cannot move a load ahead of the stores of the row before it, and an epilogue of 16 dependent load -> FMA -> store
        // rounds is 16 memory latencies per tile (the first version: 1.16 ms per layer, matrix pipe 43 % busy); here the E / LE values
        // of two row groups are in flight while the previous group is combined and stored.
        if (row0 + 32 <= n_rows && col0 + 128 <= d_in) {
            const int64_t rbase = row0 + 4 * lh;
            const float *pe = E + rbase * ldE + col0 + li, *pl = LE + rbase * ldLE + col0 + li;
            float *ple = dLE + rbase * ldd + col0 + li, *pde = dE + rbase * lde + col0 + li;
            constexpr int DEPTH = NGCF_BI_EPI_DEPTH;   // row groups of E / LE values in flight
            float ev[DEPTH][4], lv[DEPTH][4];
            auto ld_r = [&](int r, float (&e)[4], float (&l)[4]) {
                const int64_t ro = (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int t = 0; t < 4; ++t) e[t] = pe[ro * ldE + t * 32], l[t] = pl[ro * ldLE + t * 32];
            };
#pragma unroll
            for (int r = 0; r < DEPTH - 1; ++r) ld_r(r, ev[r], lv[r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (r + DEPTH - 1 < 16) ld_r(r + DEPTH - 1, ev[(r + DEPTH - 1) % DEPTH], lv[(r + DEPTH - 1) % DEPTH]);
                const int64_t ro = (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float ds = accS[t][r], dp = accP[t][r];
                    ple[ro * ldd + t * 32] = fmaf(dp, ev[r % DEPTH][t], ds);
                    pde[ro * lde + t * 32] = fmaf(dp, lv[r % DEPTH][t], ds);
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (grow >= n_rows) continue;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int col = col0 + t * 32 + li;
                if (col < d_in) {
                    const float ds = accS[t][r], dp = accP[t][r];
                    dLE[grow * ldd + col] = fmaf(dp, E[grow * ldE + col], ds);
                    dE[grow * lde + col] = fmaf(dp, LE[grow * ldLE + col], ds);
                }
            }
        }
    }
}

// The 1..4 input columns a 130- / 515-wide first layer leaves beyond its panels of 128: per row the dot products of its dM row with
// columns c of W1 and W2 (K values each), one wave per four rows - lane l holds k = l and l + 64 of the weight columns in registers -
// and the same epilogue.  A pass over dM (0.56 GB at C3) instead of a 160-column panel of the staged kernel.
__global__ __launch_bounds__(256) void layer_bwd_input_narrow_kernel(const float *__restrict__ dM, int64_t ldM, int64_t n_rows, int d_out,
                                                                     const float *__restrict__ W1, const float *__restrict__ W2,
                                                                     const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ E,
                                                                     int64_t ldE, int d_in, int col0, int ncols, float *__restrict__ dLE,
                                                                     int64_t ldd, float *__restrict__ dE, int64_t lde)
{
    const int lane = threadIdx.x & 63;
    float w1[2][4], w2[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = lane + 64 * h;
            const bool ok = k < d_out && c < ncols;
            w1[h][c] = ok ? W1[(int64_t)k * d_in + col0 + c] : 0.f;
            w2[h][c] = ok ? W2[(int64_t)k * d_in + col0 + c] : 0.f;
        }
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    // four rows per step: eight independent loads of dM in flight per lane (one row per step is one memory latency per row: 0.64 ms
    // at C3 for a pass that moves 0.56 GB)
    for (int64_t row4 = wave0 * 4; row4 < n_rows; row4 += n_waves * 4) {
        float m0[4], m1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = row4 + q < n_rows ? row4 + q : n_rows - 1;
            m0[q] = lane < d_out ? dM[row * ldM + lane] : 0.f;
            m1[q] = lane + 64 < d_out ? dM[row * ldM + lane + 64] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = row4 + q;
            // v[0..3] = dS of columns 0..3, v[4..7] = dP: eight partial dot products per lane, reduced over the wave with a halving
            // butterfly - at every step a lane keeps half of its values and takes the partner's for those (10 exchanges, not 48)
            float v[8];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v[c] = fmaf(m1[q], w1[1][c], m0[q] * w1[0][c]);
                v[4 + c] = fmaf(m1[q], w2[1][c], m0[q] * w2[0][c]);
            }
            float u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {              // lanes with bit 5 clear keep v[0..3], the others v[4..7]
                const float mine = (lane & 32) ? v[4 + i] : v[i], theirs = (lane & 32) ? v[i] : v[4 + i];
                u[i] = mine + __shfl_xor(theirs, 32);
            }
            float t2[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {              // bit 4 clear: u[0..1], set: u[2..3]
                const float mine = (lane & 16) ? u[2 + i] : u[i], theirs = (lane & 16) ? u[i] : u[2 + i];
                t2[i] = mine + __shfl_xor(theirs, 16);
            }
            float x;
            {
                const float mine = (lane & 8) ? t2[1] : t2[0], theirs = (lane & 8) ? t2[0] : t2[1];
                x = mine + __shfl_xor(theirs, 8);
            }
            x += __shfl_xor(x, 4);
            x += __shfl_xor(x, 2);
            x += __shfl_xor(x, 1);
            // lane's value: index = (bit5 ? 4 : 0) + (bit4 ? 2 : 0) + (bit3 ? 1 : 0) of v; lanes 0, 8, 16, 24 hold dS of columns 0..3 and
            // lanes 32, 40, 48, 56 hold dP of the same columns: pair them up through one more exchange
            const float dp = __shfl(x, (lane & 31) + 32);
            const int c = ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1);
            if (row < n_rows && lane < 32 && (lane & 7) == 0 && c < ncols) {
                const int col = col0 + c;
                dLE[row * ldd + col] = fmaf(dp, E[row * ldE + col], x);
                dE[row * lde + col] = fmaf(dp, LE[row * ldLE + col], x);
            }
        }
    }
}

extern "C" int64_t ngcf_layer_bwd_input_workspace_bytes(int d_out)
{
    if (d_out <= 0) return -1;
    const int64_t n_chunks = (d_out + NGCF_BI_KC - 1) / NGCF_BI_KC;
    return align_up(2 * n_chunks * NGCF_BI_KC * kBiMaxCols * (int64_t)sizeof(float), 256) + 256;
}

extern "C" int ngcf_layer_bwd_input_f32(const float *dM, int64_t ldM, int64_t n_rows, int d_out, const float *W1, const float *W2,
                                        int d_in, const float *LE, int64_t ldLE, const float *E, int64_t ldE, float *dLE,
                                        int64_t ldd, float *dE, int64_t lde, void *workspace, int64_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows == 0) return NGCF_OK;
    if (!dM || !W1 || !W2 || !LE || !E || !dLE || !dE || d_in <= 0 || d_out < 1 || ldM < d_out || ldLE < d_in || ldE < d_in ||
        ldd < d_in || lde < d_in)
        return fail(NGCF_ERR_ARG, "layer_bwd_input: bad argument");
    if (ldM % 4 != 0 || !aligned16(dM) || ldM < align_up(d_out, 4))
        return fail(NGCF_ERR_ARG, "layer_bwd_input: dM needs 16-byte aligned rows padded to a multiple of 4 floats");
    const int64_t need = ngcf_layer_bwd_input_workspace_bytes(d_out);
    if (!workspace || workspace_bytes < need)
        return fail(NGCF_ERR_WORKSPACE, "layer_bwd_input: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    float *Wp = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
    const int n_chunks = (d_out + NGCF_BI_KC - 1) / NGCF_BI_KC;
    const int64_t blocks = (n_rows + kBiRows - 1) / kBiRows;
    if (blocks >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "layer_bwd_input: too many rows");
    // r04: large matrices at K <= 128 - panels of 128 input columns on the weights-resident kernel, a remainder of 1..4 columns
    // (the reference's 130- / 515-wide first layers) on the narrow kernel
    const int k_chunks = (d_out + kBiResDC - 1) / kBiResDC;
    const bool resident = ngcf_opts().bwd_input_resident && d_out <= 128 && d_out >= 4 && n_rows >= 2 * 32 * kBiResWaves * kBiResWGs &&
                          need >= (int64_t)k_chunks * kBiResDC * 256 * (int64_t)sizeof(float) + 256;
    if (resident && d_out <= 128) {
        static bool attr_set[kMaxDevices] = {};      // the attribute is per device
        const int dev_i = current_device_slot();
        if (!attr_set[dev_i]) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(layer_bwd_input_resident_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        160 * 1024));
            attr_set[dev_i] = true;
        }
        const size_t lds = (size_t)k_chunks * kBiResDC * 256 * sizeof(float);
        int col0 = 0;
        for (; col0 < d_in && d_in - col0 > 4; col0 += 128) {
            bwd_input_pack_resident_kernel<<<64, 256, 0, stream>>>(W1, W2, d_out, d_in, col0, k_chunks * kBiResDC, Wp);
            LAUNCH_CHECK();
            layer_bwd_input_resident_kernel<<<dim3(kBiResWGs), kBiResWaves * 64, lds, stream>>>(dM, ldM, n_rows, d_out, Wp, k_chunks, LE, ldLE, E,
                                                                                               ldE, d_in, col0, dLE, ldd, dE, lde);
            LAUNCH_CHECK();
        }
        if (col0 < d_in) {
            layer_bwd_input_narrow_kernel<<<dim3(kBiResWGs * 8), 256, 0, stream>>>(dM, ldM, n_rows, d_out, W1, W2, LE, ldLE, E, ldE, d_in, col0,
                                                                                  d_in - col0, dLE, ldd, dE, lde);
            LAUNCH_CHECK();
        }
        return NGCF_OK;
    }
    for (int col0 = 0; col0 < d_in;) {      // panels of 128 input columns, the last one up to 160 (stream-ordered re-use of Wp)
        const int left = d_in - col0;
        const int wcols = left > 128 && left <= 160 ? 160 : 128;
        bwd_input_pack_kernel<<<64, 256, 0, stream>>>(W1, W2, d_out, d_in, col0, wcols, n_chunks, Wp);
        LAUNCH_CHECK();
        if (wcols == 160)
            layer_bwd_input_kernel<5, false><<<dim3((unsigned)blocks), 256, 0, stream>>>(dM, ldM, n_rows, d_out, Wp, n_chunks, LE, ldLE, E, ldE,
                                                                                         d_in, col0, dLE, ldd, dE, lde);
        else if (n_rows <= 16384)         // fewer than 128 tall tiles: 32-row tiles, the waves side by side (see the kernel)
            layer_bwd_input_kernel<4, true><<<dim3((unsigned)((n_rows + 31) / 32)), 256, 0, stream>>>(dM, ldM, n_rows, d_out, Wp, n_chunks, LE, ldLE,
                                                                                                    E, ldE, d_in, col0, dLE, ldd, dE, lde);
        else
            layer_bwd_input_kernel<4, false><<<dim3((unsigned)blocks), 256, 0, stream>>>(dM, ldM, n_rows, d_out, Wp, n_chunks, LE, ldLE, E, ldE,
                                                                                         d_in, col0, dLE, ldd, dE, lde);
        LAUNCH_CHECK();
        col0 += wcols;
    }
    return NGCF_OK;
}


// =============================================================================================
// L^T . X for a ROW-SPARSE X.  The gradient that reaches the last layer comes from the three row gathers only
// (NGCF.py:151-155): dLE of that layer is non-zero on R <= 3 B rows, given compacted as X [R, d] with a table slot[N]
// (slot[r] = row of X that holds matrix row r, -1 = zero row).  out = init + L^T . dLE is formed on the CSR of L^T, row by
// row: a wave walks the stored entries (c, r, v) of output row c 64 at a time, looks slot[r] up (4 bytes per entry out of a
// table that lives in L2), compacts the hits with a ballot and adds v * X[slot[r], :] for each hit in entry order - so every
// output element is a sum in a FIXED order (r03: this replaces a scatter with float atomics, whose gradients differed from run
// to run).  Rows cut by the CSR's segment plan go through partial sums + spmm_fixup_kernel like every other product.  The whole
// of `out` is written: rows without a hit get init or zero, so no zero-fill and no index_put precede it.
// Cost: one pass over colidx (400 MB on C3, ~0.15 ms) + R-proportional work, instead of a full L^T . dLE SpMM (1.5 ms).
// =============================================================================================
template <int NQ>      // this lane's columns lane, lane + 64, ..: d <= 64 * NQ
__global__ __launch_bounds__(256) void spmm_t_rows_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                          const float *__restrict__ vals, int64_t n_rows,
                                                          const int32_t *__restrict__ seg_row, const int64_t *__restrict__ seg_begin,
                                                          int64_t n_seg, int64_t seg_blocks, int seg_len,
                                                          const int32_t *__restrict__ slot, const float *__restrict__ X, int64_t ldx, int d,
                                                          const float *__restrict__ init, int64_t ldi, float *__restrict__ out, int64_t ldo,
                                                          float *__restrict__ partial, int dp, EdgeDrop dr_in)
{
    const EdgeDropR dr = resolve_drop(dr_in);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t begin, end, row;
    float *dst;
    bool first;                 // this unit starts its row: it carries the row's init term
    if ((int64_t)blockIdx.x < seg_blocks) {
        const int64_t s = (int64_t)blockIdx.x * 4 + wave;
        if (s >= n_seg) return;
        begin = seg_begin[s];
        row = seg_row[s];
        const int64_t row_end = rowptr[row + 1];
        end = begin + seg_len < row_end ? begin + seg_len : row_end;
        first = begin == rowptr[row];
        dst = partial + s * (int64_t)dp;
    } else {
        row = ((int64_t)blockIdx.x - seg_blocks) * 4 + wave;
        if (row >= n_rows) return;
        begin = rowptr[row];
        end = rowptr[row + 1];
        if (end - begin > seg_len) return;   // cut row: produced from its segments
        first = true;
        dst = out + row * ldo;
    }
    float acc[NQ];
    const int s_self = first && init ? slot[row] : -1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = (s_self >= 0 && lane + 64 * q < d) ? init[(int64_t)s_self * ldi + lane + 64 * q] : 0.f;
    for (int64_t base = begin; base < end; base += 64) {
        const int cnt = (int)(end - base < 64 ? end - base : 64);
        int sl = -1;
        float v = 0.f;
        if (lane < cnt) {
            const int c = colidx[base + lane];
            sl = slot[c];
            if (sl >= 0) {
                v = vals[base + lane];
                if (dr.n > 0 && !edge_keep(dr, row, c)) sl = -1;
            }
        }
        unsigned long long hits = __ballot(sl >= 0);
        while (hits) {                                         // wave-uniform: hits in entry order
            const int j = __builtin_ctzll(hits);
            hits &= hits - 1;
            const int sj = __shfl(sl, j);
            const float vj = __shfl(v, j);
            const float *x = X + (int64_t)sj * ldx;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (lane + 64 * q < d) acc[q] = fmaf(vj, x[lane + 64 * q], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < d) dst[lane + 64 * q] = acc[q];
}


// ---- r04: the same sums, 16 rows to a wave, the MEMBERSHIP test out of LDS, loads and stores kept apart -----------------------
// The kernel above is bound by LATENCY, not by bytes (C3: 0.93 ms for 0.4 GB of indices in and 0.56 GB of rows out): a wave owns
// one row and runs rowptr -> colidx -> slot -> X -> store as a chain of dependent round trips, one 64-entry load in flight, 1.1 M
// times.  What r04 measured on the way here (tools/t_rows_lab.py, per-wave clocks):
//   * moving only the slot test into LDS: 0.93 -> 0.96 ms - the chain was the cost, not the gather;
//   * 16 / 32 rows to a wave with the hits applied as they are found: 0.80 - 0.98 ms, of which 0.57 - 0.64 ms in the row WRITES:
//     on gfx9 one counter (vmcnt) covers loads and stores and retires in order, so every load that follows a store waits for that
//     store's acknowledge (~2 us under load) - a store per row and a load per hit made a chain again, and a load that MAY happen in
//     a block (`hit ? init[..] : 0`) makes the compiler wait for everything at the join even when it did not happen.
// So: a wave takes 16 CONSECUTIVE rows - one contiguous stretch of colidx, scanned 512 entries at a time with eight independent
// loads in flight; whether column c is one of the R rows is ONE BIT of a bitmap (N / 8 bytes: 137 KB at C3) that a persistent
// workgroup (16 waves, one per CU) holds in LDS, so the 99 % of entries that miss cost an LDS read; the hits (entry, column, slot,
// value) are parked, in entry order, in a small per-wave list in LDS while the scan goes on (loads only), and the list is drained
// eight hits at a time: eight X rows loaded together, then applied in order to ONE running accumulator that is written when the
// row changes - the row an entry belongs to is a ballot over the 17 row pointers held in lanes - with the rows between two hits
// written as init / zero from a loop that loads nothing.  Units are handed out through counters (two levels, see the kernel), the
// long (item) rows at the end of the range first.  Per row the same entries are added in
// the same order with the same fmaf: bit-identical to the kernel above.
__global__ void slot_bitmap_kernel(const int32_t *__restrict__ slot, int64_t n, uint32_t *__restrict__ bm, int64_t n_words,
                                   unsigned long long *__restrict__ counters, int n_counters)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_counters) counters[i] = 0ull;
    const unsigned long long b = __ballot(i < n && slot[i] >= 0);
    if ((threadIdx.x & 63) == 0) {
        const int64_t w = i >> 5;
        if (w < n_words) bm[w] = (uint32_t)b;
        if (w + 1 < n_words) bm[w + 1] = (uint32_t)(b >> 32);
    }
}

static constexpr int kTrWaves = 16, kTrWGs = 256, kTrRows = 16, kTrU = 8, kTrPanels = 8;
static constexpr int kTrList = 84;                 // parked hits per wave (16 bytes each: 1 344 B a wave, 21 KB a workgroup)
static constexpr int kTrGroup = 8;                 // hits whose X rows are loaded together
static constexpr int kTrChunk = 32;                // units a workgroup takes from the device-wide counter at a time
static constexpr int64_t kTrLdsBytes = 160 * 1024;

// one row (or one segment of a cut row), 64 entries at a time: the walk of spmm_t_rows_kernel with the bitmap in front of the slot table
template <int NQ>
__device__ __forceinline__ void t_rows_walk(const uint32_t *bm, const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                            int64_t begin, int64_t end, int64_t row, bool first, const int32_t *__restrict__ slot,
                                            const float *__restrict__ X, int64_t ldx, int d, const float *__restrict__ init, int64_t ldi,
                                            float *__restrict__ dst, const EdgeDropR &dr, int lane)
{
    float acc[NQ];
    const int s_self = first && init && ((bm[row >> 5] >> (row & 31)) & 1u) ? slot[row] : -1;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = (s_self >= 0 && lane + 64 * q < d) ? init[(int64_t)s_self * ldi + lane + 64 * q] : 0.f;
    for (int64_t base = begin; base < end; base += 64) {
        const int cnt = (int)(end - base < 64 ? end - base : 64);
        int sl = -1;
        float v = 0.f;
        if (lane < cnt) {
            const int c = colidx[base + lane];
            if ((bm[c >> 5] >> (c & 31)) & 1u) {
                sl = slot[c];
                v = vals[base + lane];
                if (dr.n > 0 && !edge_keep(dr, row, c)) sl = -1;
            }
        }
        unsigned long long hits = __ballot(sl >= 0);
        while (hits) {                                         // wave-uniform: hits in entry order
            const int j = __builtin_ctzll(hits);
            hits &= hits - 1;
            const int sj = __shfl(sl, j);
            const float vj = __shfl(v, j);
            const float *x = X + (int64_t)sj * ldx;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (lane + 64 * q < d) acc[q] = fmaf(vj, x[lane + 64 * q], acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < d) dst[lane + 64 * q] = acc[q];
}

template <int NQ>
__global__ __launch_bounds__(kTrWaves * 64) void spmm_t_rows_bm_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                                    const float *__restrict__ vals, int64_t n_rows,
                                                                    const int32_t *__restrict__ seg_row, const int64_t *__restrict__ seg_begin,
                                                                    int64_t n_seg, int seg_len, const int32_t *__restrict__ slot,
                                                                    const uint32_t *__restrict__ bm_g, int n_words,
                                                                    unsigned long long *__restrict__ counter,
                                                                    const float *__restrict__ X, int64_t ldx, int d,
                                                                    const float *__restrict__ init, int64_t ldi, float *__restrict__ out,
                                                                    int64_t ldo, float *__restrict__ partial, int dp, EdgeDrop dr_in)
{
    extern __shared__ uint32_t bm[];                            // [n_words bitmap][kTrWaves lists of kTrList int4]
    {                                                          // the bitmap: 16 bytes a lane, four loads in flight
        const uint4 *src = reinterpret_cast<const uint4 *>(bm_g);
        uint4 *dst4 = reinterpret_cast<uint4 *>(bm);
        const int n4 = n_words / 4;                             // (n_words is a multiple of 4)
        for (int i0 = threadIdx.x; i0 < n4; i0 += 4 * kTrWaves * 64) {
            uint4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = i0 + k * kTrWaves * 64 < n4 ? src[i0 + k * kTrWaves * 64] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i0 + k * kTrWaves * 64 < n4) dst4[i0 + k * kTrWaves * 64] = r[k];
        }
        if (threadIdx.x == 0)                                   // the unit pool of this workgroup: "chunk used up" (see below)
            *reinterpret_cast<unsigned long long *>(reinterpret_cast<int4 *>(bm + n_words) + kTrWaves * kTrList) = (unsigned long long)kTrChunk;
    }
    __syncthreads();
    const EdgeDropR dr = resolve_drop(dr_in);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int4 *list = reinterpret_cast<int4 *>(bm + n_words) + wave * kTrList;       // this wave's parked hits: (entry - begin, column, slot, value)
    // the segments of the cut rows (a few thousand at most): one wave each, into the partial sums
    for (int64_t s = (int64_t)blockIdx.x * kTrWaves + wave; s < n_seg; s += (int64_t)gridDim.x * kTrWaves) {
        const int64_t begin = seg_begin[s], row = seg_row[s], row_end = rowptr[row + 1];
        t_rows_walk<NQ>(bm, colidx, vals, begin, begin + seg_len < row_end ? begin + seg_len : row_end, row, begin == rowptr[row], slot, X, ldx, d,
                        init, ldi, partial + s * (int64_t)dp, dr, lane);
    }
    const int64_t n_units = (n_rows + kTrRows - 1) / kTrRows;
    // Units are handed out in two levels.  A ticket per unit from ONE device-wide counter was the whole cost of an earlier version:
    // returning atomics on one address retire at ~13 ns each (69 000 units: 0.9 ms), and as vector-memory operations they hold up,
    // in order, every load issued behind them.  So the device-wide counter hands out chunks of kTrChunk units to a workgroup
    // (4 300 atomics at C3) and the waves of the workgroup draw from the chunk with an LDS atomic: `pool` = (chunk base << 24 | units
    // drawn); the wave that draws number kTrChunk exactly is the one that fetches the next chunk, later ones wait for it.
    unsigned long long *pool = reinterpret_cast<unsigned long long *>(reinterpret_cast<int4 *>(bm + n_words) + kTrWaves * kTrList);
    for (;;) {
        unsigned long long t = 0;
        if (lane == 0) {
            for (;;) {
                const unsigned long long old = atomicAdd(pool, 1ull);
                const unsigned long long drawn = old & 0xffffffull;
                if (drawn < (unsigned long long)kTrChunk) {
                    t = (old >> 24) + drawn;
                    break;
                }
                if (drawn == (unsigned long long)kTrChunk) {                      // this wave fetches the next chunk and takes its first unit
                    t = atomicAdd(counter, (unsigned long long)kTrChunk);
                    __hip_atomic_store(pool, (t << 24) | 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                while ((__hip_atomic_load(pool, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & 0xffffffull) > (unsigned long long)kTrChunk)
                    __builtin_amdgcn_s_sleep(8);                                   // a chunk is on its way (microseconds)
            }
        }
        t = __shfl(t, 0);
        if (t >= (unsigned long long)n_units) break;           // every wave gets here: the counters only grow
        const int64_t row0 = (n_units - 1 - (int64_t)t) * kTrRows;
        const int nr = (int)(n_rows - row0 < kTrRows ? n_rows - row0 : kTrRows);
        const long long rp = rowptr[row0 + (lane < nr ? lane : nr)];          // lanes 0 .. nr: the row pointers; the rest repeat the last
        const long long rp_next = __shfl_down(rp, 1);
        // rows longer than seg_len are cut: their sums come from the segments above and the fix-up, their entries are skipped here
        unsigned long long live = __ballot(lane < nr && rp_next - rp <= seg_len);
        int self = -1;                                         // lane k: the row of `init` that output row row0 + k starts from
        if (init && lane < nr && ((bm[(row0 + lane) >> 5] >> ((row0 + lane) & 31)) & 1u)) self = slot[row0 + lane];
        while (live) {                                         // maximal runs [ka, kb) of rows that are not cut: one contiguous stretch of entries
            const int ka = __builtin_ctzll(live);
            const int kb = ka + __builtin_ctzll(~(live >> ka));  // (bits 16 .. 63 of live are clear)
            live &= ~0ull << kb;
            const int64_t begin = __shfl(rp, ka), end = __shfl(rp, kb);
            int cur = ka;                                      // the row the accumulator belongs to
            float acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
            // the accumulator of row r starts from its init row.  The load happens in a block of its own and is SETTLED there (an
            // asm that reads the value, so the wait for it is placed inside the block): otherwise the compiler waits, at the join,
            // for a load that mostly did not happen - and on gfx9 that wait (vmcnt(0)) is also a wait for every store before it
            auto start_row = [&](int r) {
                const int sr = __shfl(self, r);
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = 0.f;
                if (sr >= 0) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (lane + 64 * q < d) acc[q] = init[(int64_t)sr * ldi + lane + 64 * q];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) asm volatile("v_mov_b32 %0, %0" : "+v"(acc[q]));
                }
            };
            // write row `cur`, then the rows up to (not including) `upto`, which have no hit: zeros (a loop of stores and nothing
            // else - merged with the init rows the compiler shares the store and waits before it) and, rarely, their init row
            auto advance = [&](int upto) {
                float *dst = out + (row0 + cur) * ldo;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    if (lane + 64 * q < d) dst[lane + 64 * q] = acc[q];
                unsigned long long with_init = __ballot(self >= 0) & (upto >= 64 ? ~0ull : (1ull << upto) - 1ull) & (~0ull << (cur + 1));
                for (int r = cur + 1; r < upto; ++r) {
                    if ((with_init >> r) & 1ull) continue;
                    dst = out + (row0 + r) * ldo;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (lane + 64 * q < d) dst[lane + 64 * q] = 0.f;
                }
                while (with_init) {
                    const int r = __builtin_ctzll(with_init);
                    with_init &= with_init - 1;
                    const int sr = __shfl(self, r);
                    dst = out + (row0 + r) * ldo;
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        if (lane + 64 * q < d) dst[lane + 64 * q] = init[(int64_t)sr * ldi + lane + 64 * q];
                }
            };
            // one hit, its X row in x[]: close the rows before it, add it
            auto apply = [&](int e_rel, int c, float vj, const float (&x)[NQ]) {
                const long long e = begin + e_rel;
                const int rj = __popcll(__ballot(lane >= 1 && lane <= nr && rp <= e));   // rows that begin at or before e, minus one
                if (dr.n > 0 && !edge_keep(dr, row0 + rj, c)) return;
                if (rj != cur) {
                    advance(rj);
                    cur = rj;
                    start_row(rj);
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = fmaf(vj, x[q], acc[q]);
            };
            start_row(ka);
            int n_list = 0;
            for (int64_t base = begin;; base += 64 * kTrU) {
                const bool more = base < end;
                int c[kTrU], sl[kTrU];
                float v[kTrU];
                unsigned long long hm[kTrU];
                int n_new = 0;
                if (more) {
#pragma unroll
                    for (int u = 0; u < kTrU; ++u) {
                        const int64_t e = base + 64 * u + lane;
                        c[u] = e < end ? colidx[e] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < kTrU; ++u) {
                        sl[u] = -1;
                        v[u] = 0.f;
                        if (c[u] >= 0 && ((bm[c[u] >> 5] >> (c[u] & 31)) & 1u)) {
                            sl[u] = slot[c[u]];
                            v[u] = vals[base + 64 * u + lane];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kTrU; ++u) {
                        hm[u] = __ballot(sl[u] >= 0);
                        n_new += __popcll(hm[u]);
                    }
                }
                if (n_list > 0 && (!more || n_list + n_new > kTrList)) {          // drain the list: eight X rows at a time
                    __builtin_amdgcn_wave_barrier();
                    for (int i = 0; i < n_list; i += kTrGroup) {
                        int4 ent[kTrGroup];
                        float x[kTrGroup][NQ];
#pragma unroll
                        for (int g = 0; g < kTrGroup; ++g) {
                            ent[g] = list[i + g < n_list ? i + g : i];
                            const float *xr = X + (int64_t)__builtin_amdgcn_readfirstlane(ent[g].z) * ldx;
#pragma unroll
                            for (int q = 0; q < NQ; ++q) x[g][q] = lane + 64 * q < d ? xr[lane + 64 * q] : 0.f;
                        }
#pragma unroll
                        for (int g = 0; g < kTrGroup; ++g)
                            if (i + g < n_list)
                                apply(__builtin_amdgcn_readfirstlane(ent[g].x), __builtin_amdgcn_readfirstlane(ent[g].y),
                                      __int_as_float(__builtin_amdgcn_readfirstlane(ent[g].w)), x[g]);
                    }
                    __builtin_amdgcn_wave_barrier();
                    n_list = 0;
                }
                if (!more) break;
                if (n_new > kTrList) {                          // more hits in one window than the list holds (a dense X): one by one
#pragma unroll
                    for (int u = 0; u < kTrU; ++u) {
                        unsigned long long hits = hm[u];
                        while (hits) {
                            const int j = __builtin_ctzll(hits);
                            hits &= hits - 1;
                            const float *xr = X + (int64_t)__shfl(sl[u], j) * ldx;
                            float x[NQ];
#pragma unroll
                            for (int q = 0; q < NQ; ++q) x[q] = lane + 64 * q < d ? xr[lane + 64 * q] : 0.f;
                            apply((int)(base - begin) + 64 * u + j, __shfl(c[u], j), __shfl(v[u], j), x);
                        }
                    }
                } else if (n_new > 0) {                         // park them behind the ones already there, in entry order
#pragma unroll
                    for (int u = 0; u < kTrU; ++u) {
                        if (sl[u] >= 0) {
                            const int pos = n_list + __builtin_amdgcn_mbcnt_hi((uint32_t)(hm[u] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm[u], 0));
                            list[pos] = make_int4((int)(base - begin) + 64 * u + lane, c[u], sl[u], __float_as_int(v[u]));
                        }
                        n_list += __popcll(hm[u]);
                    }
                }
            }
            advance(kb);
        }
    }
}

extern "C" int ngcf_spmm_t_rows_f32(const ngcf_csr_t *c, const int32_t *slot, const float *X, int64_t ldx, int d, const float *init,
                                    int64_t ldi, float *out, int64_t ldo, float drop_p, const uint64_t *seeds, int n_seeds,
                                    void *workspace, int64_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!c) return fail(NGCF_ERR_ARG, "spmm_t_rows: null csr");
    if (c->n_rows == 0) return NGCF_OK;
    if (!slot || !X || !out || d <= 0 || ldx < d || ldo < d || (init && ldi < d)) return fail(NGCF_ERR_ARG, "spmm_t_rows: bad argument");
    if (n_seeds < 0 || n_seeds > 4 || (n_seeds > 0 && !seeds)) return fail(NGCF_ERR_ARG, "spmm_t_rows: 0..4 seeds expected");
    if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(NGCF_ERR_ARG, "spmm_t_rows: drop_p=%f not in [0,1)", drop_p);
    EdgeDrop dr{drop_p > 0.f ? n_seeds : 0, (uint32_t)((double)drop_p * 4294967296.0), {0, 0, 0, 0}, 1};   // the CSR walked is L^T
    for (int q = 0; q < n_seeds; ++q) dr.seed[q] = seeds[q];
    const int64_t seg_blocks = (c->n_seg + 3) / 4, row_blocks = (c->n_rows + 3) / 4;
    if (seg_blocks + row_blocks >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "spmm_t_rows: too many rows for one launch");
    // r04: the membership bitmap of the R rows in LDS (see spmm_t_rows_bm_kernel) where it fits (N / 8 bytes + 21 KB of hit lists <= 160 KB: N <= 1.13 M) and the matrix
    // is large enough for a persistent grid to pay; it lives behind the partial sums in the workspace
    const int64_t n_slot = std::max(c->n_cols, c->n_rows);      // (square in every use: the slot table covers rows and columns)
    const int64_t n_words = (n_slot + 127) / 128 * 4;             // 32-bit words, a multiple of 4
    const int64_t part_bytes = c->n_seg > 0 ? align_up(c->n_seg * align_up(std::min(d, 512), 4) * (int64_t)sizeof(float), 256) + 256 : 256;
    uint32_t *bitmap = nullptr;
    unsigned long long *counters = nullptr;
    if (ngcf_opts().t_rows_bitmap && c->n_rows == c->n_cols && n_words * 4 + kTrWaves * kTrList * 16 + 16 <= kTrLdsBytes && c->nnz >= (1 << 22) && workspace &&
        d <= 512 * kTrPanels && workspace_bytes >= part_bytes + n_words * 4 + 1024) {
        bitmap = reinterpret_cast<uint32_t *>(align_up((int64_t)(uintptr_t)workspace + part_bytes, 256));
        counters = reinterpret_cast<unsigned long long *>(align_up((int64_t)(uintptr_t)(bitmap + n_words), 256));   // one per panel
        slot_bitmap_kernel<<<(unsigned)((n_slot + 255) / 256), 256, 0, stream>>>(slot, n_slot, bitmap, n_words, counters, kTrPanels);
        LAUNCH_CHECK();
    }
    for (int col0 = 0; col0 < d; col0 += 512) {                 // panels of 512 columns (a lane holds 8)
        const int w = std::min(512, d - col0);
        const int dp = (int)align_up(w, 4);
        float *partial = nullptr;
        if (c->n_seg > 0) {
            const int64_t need = align_up(c->n_seg * dp * (int64_t)sizeof(float), 256) + 256;
            if (!workspace || workspace_bytes < need)
                return fail(NGCF_ERR_WORKSPACE, "spmm_t_rows: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
            partial = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
        }
        if (bitmap) {
#define NGCF_TROWS_BM(NQ)                                                                                                             \
    do {                                                                                                                              \
        static bool attr_set[kMaxDevices] = {};                                                                                       \
        const int dev_i = current_device_slot();                                                                                      \
        if (!attr_set[dev_i]) {                                                                                                       \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_t_rows_bm_kernel<NQ>),                                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTrLdsBytes));                                     \
            attr_set[dev_i] = true;                                                                                                   \
        }                                                                                                                             \
        spmm_t_rows_bm_kernel<NQ><<<dim3(kTrWGs), kTrWaves * 64, (size_t)n_words * 4 + kTrWaves * kTrList * 16 + 16, stream>>>(                                       \
            c->rowptr, c->colidx, c->vals, c->n_rows, c->seg_row, c->seg_begin, c->n_seg, c->seg_len, slot, bitmap, (int)n_words,     \
            counters + col0 / 512, X + col0, ldx, w, init ? init + col0 : nullptr, ldi, out + col0, ldo, partial, dp, dr);                                   \
    } while (0)
            if (w <= 64) NGCF_TROWS_BM(1);
            else if (w <= 128) NGCF_TROWS_BM(2);
            else if (w <= 256) NGCF_TROWS_BM(4);
            else NGCF_TROWS_BM(8);
#undef NGCF_TROWS_BM
            LAUNCH_CHECK();
            if (c->n_heavy > 0) {
                spmm_fixup_kernel<1><<<dim3((unsigned)((c->n_heavy + 3) / 4)), 256, 0, stream>>>(c->heavy_row, c->heavy_seg_ptr, c->n_heavy,
                                                                                              partial, dp, w, out + col0, ldo);
                LAUNCH_CHECK();
            }
            continue;
        }
#define NGCF_TROWS(NQ)                                                                                                                \
    spmm_t_rows_kernel<NQ><<<dim3((unsigned)(seg_blocks + row_blocks)), 256, 0, stream>>>(                                            \
        c->rowptr, c->colidx, c->vals, c->n_rows, c->seg_row, c->seg_begin, c->n_seg, seg_blocks, c->seg_len, slot, X + col0, ldx, w, \
        init ? init + col0 : nullptr, ldi, out + col0, ldo, partial, dp, dr)
        if (w <= 64) NGCF_TROWS(1);
        else if (w <= 128) NGCF_TROWS(2);
        else if (w <= 256) NGCF_TROWS(4);
        else NGCF_TROWS(8);
#undef NGCF_TROWS
        LAUNCH_CHECK();
        if (c->n_heavy > 0) {
            spmm_fixup_kernel<1><<<dim3((unsigned)((c->n_heavy + 3) / 4)), 256, 0, stream>>>(c->heavy_row, c->heavy_seg_ptr, c->n_heavy, partial,
                                                                                          dp, w, out + col0, ldo);
            LAUNCH_CHECK();
        }
    }
    return NGCF_OK;
}


// =============================================================================================
// The distinct rows of a small index vector, in ONE launch (r03).  The gradient of all_E is non-zero on the rows the three
// gathers of a forward touched (NGCF.py:151-155): M = |u_id| + |pos_item| + |neg_item| <= 3 B positions, with duplicates.  The
// backward needs them sorted, distinct, and - for a summation in a fixed order - the gathered positions grouped by row in batch
// order.  torch.unique(return_inverse, return_counts) + a stable sort of the inverse + a cumsum are ~12 library launches of
// 5-40 us each on a launch-bound training step; for M <= 8 192 one workgroup does all of it in LDS: a bitonic sort of the 64-bit
// keys (row << 13 | position: equal rows keep their batch order), head flags, a scan.  Outputs: order[M] (positions, sorted by
// row), rows[<= M] (distinct, ascending), segptr[<= M + 1] (group bounds inside `order`), n_rows[1].
// =============================================================================================
static constexpr int kSortMax = 8192, kSortThreads = 1024;

template <typename K>     // key type: 32 bits when row << 13 | position fits (rows below 2^19: the Seoul graph), else 64
__global__ __launch_bounds__(kSortThreads) void rows_sort_unique_kernel(const int64_t *__restrict__ idx, int M, int64_t max_row,
                                                                        int64_t *__restrict__ order, int64_t *__restrict__ rows,
                                                                        int64_t *__restrict__ segptr, int64_t *__restrict__ n_rows)
{
    __shared__ K key[kSortMax];
    __shared__ int wsum[kSortThreads / 64];
    __shared__ int carry_s;
    const int tid = threadIdx.x;
    int P = 64;
    while (P < M) P <<= 1;                                          // power of two >= M
    // An id outside [0, max_row] (the forward gather clamped it and set the sticky status word; with deferred index checks the
    // backward may still run) becomes the one sentinel row max_row + 1: it sorts behind every valid row, forms the last segment
    // and is NOT counted in n_rows - nothing downstream ever indexes all_E with it.
    for (int i = tid; i < P; i += kSortThreads) {
        K k = (K)~(K)0;
        if (i < M) {
            int64_t r = idx[i];
            if (max_row >= 0 && (r < 0 || r > max_row)) r = max_row + 1;
            k = (K)(((K)r << 13) | (K)i);
        }
        key[i] = k;
    }
    __syncthreads();
    // one comparator per thread and pass: pair t exchanges i = (t with a zero bit inserted at log2 j) and i + j.  At j <= 64 the 64
    // pairs of a wave stay inside one 128-key block, so those passes need no workgroup barrier - a wave's LDS operations complete in
    // order - only the wait for its own outstanding ones (r03: 78 barriers for M = 3 072 before, 27 now).
    for (int k = 2; k <= P; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += kSortThreads) {
                const int i = 2 * t - (t & (j - 1)), l = i + j;
                const K a = key[i], b = key[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    key[i] = b;
                    key[l] = a;
                }
            }
            if (j > 64 || j == 1) __syncthreads();
            else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        }
    // head flags + exclusive scan (chunks of kSortThreads, running carry)
    if (tid == 0) carry_s = 0;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int base = 0; base < M; base += kSortThreads) {
        const int i = base + tid;
        int head = 0;
        K kv = 0;
        if (i < M) {
            kv = key[i];
            order[i] = (int64_t)(kv & 8191u);
            head = i == 0 || (key[i - 1] >> 13) != (kv >> 13);
        }
        int incl = head;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (head) {
            const int r = before + incl - 1;
            rows[r] = (int64_t)(kv >> 13);
            segptr[r] = i;
        }
        __syncthreads();
        if (tid == kSortThreads - 1) carry_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) {
        segptr[carry_s] = M;
        const bool sentinel = max_row >= 0 && M > 0 && (int64_t)(key[M - 1] >> 13) == max_row + 1;
        n_rows[0] = carry_s - (sentinel ? 1 : 0);
    }
}

extern "C" int ngcf_rows_sort_unique(const int64_t *idx, int64_t M, int64_t max_row, int64_t *order, int64_t *rows, int64_t *segptr,
                                     int64_t *n_rows, void *stream)
{
    if (M < 0 || M > kSortMax) return fail(NGCF_ERR_ARG, "rows_sort_unique: M=%lld not in [0, %d]", (long long)M, kSortMax);
    if (!order || !rows || !segptr || !n_rows || (M > 0 && !idx)) return fail(NGCF_ERR_ARG, "rows_sort_unique: null argument");
    if (max_row >= ((int64_t)1 << 50)) return fail(NGCF_ERR_ARG, "rows_sort_unique: rows must be below 2^50");
    if (max_row >= 0 && max_row < ((int64_t)1 << 19) - 2)
        rows_sort_unique_kernel<unsigned><<<1, kSortThreads, 0, (hipStream_t)stream>>>(idx, (int)M, max_row, order, rows, segptr, n_rows);
    else
        rows_sort_unique_kernel<unsigned long long><<<1, kSortThreads, 0, (hipStream_t)stream>>>(idx, (int)M, max_row, order, rows, segptr, n_rows);
    LAUNCH_CHECK();
    return NGCF_OK;
}
