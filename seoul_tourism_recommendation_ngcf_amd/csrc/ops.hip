// ops.hip - row copy, feature injection, gathers, BPR, top-k, row partition helper.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// strided row copy (E0 -> its block of all_E), NGCF.py:120-121,147
// ---------------------------------------------------------------------------------------------
// dst2 (optional): a second copy of the same rows with another leading dimension (the 16-byte aligned E0 the first layer
// gathers from when the rows of all_E are not: embed_size 65/130/515) - the source is read once
template <int VEC>
__global__ void copy_rows_kernel(const float *__restrict__ src, int64_t lds, float *__restrict__ dst, int64_t ldd,
                                 float *__restrict__ dst2, int64_t ldd2, int64_t n_rows, int d)
{
    using V = typename VecT<VEC>::type;
    const int per_row = d / VEC;
    const int64_t total = n_rows * per_row;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t r = i / per_row;
        const int q = (int)(i % per_row) * VEC;
        const V v = *reinterpret_cast<const V *>(src + r * lds + q);
        *reinterpret_cast<V *>(dst + r * ldd + q) = v;
        if (dst2) *reinterpret_cast<V *>(dst2 + r * ldd2 + q) = v;
    }
}

static int copy_rows_impl(const float *src, int64_t lds, float *dst, int64_t ldd, float *dst2, int64_t ldd2, int64_t n_rows, int d,
                          hipStream_t stream)
{
    if (!src || !dst || d <= 0 || n_rows < 0 || lds < d || ldd < d || (dst2 && ldd2 < d)) return fail(NGCF_ERR_ARG, "copy_rows: bad argument");
    if (n_rows == 0) return NGCF_OK;
    auto al = [&](int v) {   // every row of every operand is a whole number of v-float vectors at a v*4-byte aligned address
        const uintptr_t m = (uintptr_t)v * 4 - 1;
        return d % v == 0 && lds % v == 0 && ldd % v == 0 && !((uintptr_t)src & m) && !((uintptr_t)dst & m) &&
               (!dst2 || (ldd2 % v == 0 && !((uintptr_t)dst2 & m)));
    };
    if (al(4))
        copy_rows_kernel<4><<<grid_for(n_rows * (d / 4), 256), 256, 0, stream>>>(src, lds, dst, ldd, dst2, ldd2, n_rows, d);
    else if (al(2))
        copy_rows_kernel<2><<<grid_for(n_rows * (d / 2), 256), 256, 0, stream>>>(src, lds, dst, ldd, dst2, ldd2, n_rows, d);
    else
        copy_rows_kernel<1><<<grid_for(n_rows * d, 256), 256, 0, stream>>>(src, lds, dst, ldd, dst2, ldd2, n_rows, d);
    LAUNCH_CHECK();
    return NGCF_OK;
}

extern "C" int ngcf_copy_rows_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t n_rows, int d,
                                  void *stream_)
{
    return copy_rows_impl(src, lds, dst, ldd, nullptr, 0, n_rows, d, (hipStream_t)stream_);
}

extern "C" int ngcf_copy_rows2_f32(const float *src, int64_t lds, float *dst, int64_t ldd, float *dst2, int64_t ldd2,
                                   int64_t n_rows, int d, void *stream_)
{
    if (!dst2) return fail(NGCF_ERR_ARG, "copy_rows2: null second destination");
    return copy_rows_impl(src, lds, dst, ldd, dst2, ldd2, n_rows, d, (hipStream_t)stream_);
}

// dst[idx[b], :] = src[idx[b], :] for the rows a batch touched (r04): block 0 of a RETAINED all_E follows the <= B rows the
// feature injection rewrote (NGCF.py:114) instead of the whole table being copied again (563 MB at C3).  One wave per index;
// duplicates write the same bits; ids outside [0, n_rows) are skipped (the injection flagged them).
__global__ __launch_bounds__(256) void copy_rows_indexed_kernel(const float *__restrict__ src, int64_t lds, float *__restrict__ dst,
                                                                int64_t ldd, const int64_t *__restrict__ idx, int64_t n_idx,
                                                                int64_t n_rows, int d)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_idx) return;
    const int64_t r = idx[b];
    if (r < 0 || r >= n_rows) return;
    for (int c = threadIdx.x & 63; c < d; c += 64) dst[r * ldd + c] = src[r * lds + c];
}

extern "C" int ngcf_copy_rows_indexed_f32(const float *src, int64_t lds, float *dst, int64_t ldd, const int64_t *idx, int64_t n_idx,
                                          int64_t n_rows, int d, void *stream_)
{
    if (!src || !dst || d <= 0 || n_idx < 0 || n_rows < 0 || lds < d || ldd < d || (n_idx > 0 && !idx))
        return fail(NGCF_ERR_ARG, "copy_rows_indexed: bad argument");
    if (n_idx == 0 || n_rows == 0) return NGCF_OK;
    if ((n_idx + 3) / 4 >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "copy_rows_indexed: too many indices");
    copy_rows_indexed_kernel<<<dim3((unsigned)((n_idx + 3) / 4)), 256, 0, (hipStream_t)stream_>>>(src, lds, dst, ldd, idx, n_idx, n_rows, d);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---------------------------------------------------------------------------------------------
// feature injection, NGCF.py:103-115
// ---------------------------------------------------------------------------------------------
struct InjectTables {
    const float *table[5];
    const int64_t *idx[5];
    int64_t card[5];
};

// pass 1: winner[u] = max batch position that names user u (last occurrence wins)
__global__ void inject_claim_kernel(const int64_t *__restrict__ u_id, int64_t B, int64_t n_user, int32_t *winner,
                                    int32_t *status)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t u = u_id[b];
    if (u < 0 || u >= n_user) {
        atomicOr(status, 1);
        return;
    }
    atomicMax(&winner[u], (int32_t)b);
}

// pass 2: the winning occurrence writes row u; one wave per batch position
__global__ __launch_bounds__(256) void inject_write_kernel(float *__restrict__ user_w, int64_t ldu, int64_t n_user,
                                                           int d0, InjectTables t, int fw,
                                                           const int64_t *__restrict__ u_id, int64_t B, float ratio,
                                                           float one_minus_ratio,
                                                           const int32_t *__restrict__ winner, int32_t *status)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int64_t u = u_id[b];
    if (u < 0 || u >= n_user) return;
    if (winner[u] != (int32_t)b) return;
    bool bad = false;
    int64_t fi[5];
#pragma unroll
    for (int f = 0; f < 5; ++f) {
        fi[f] = t.idx[f][b];
        bad |= fi[f] < 0 || fi[f] >= t.card[f];
    }
    if (bad) {
        if (lane == 0) atomicOr(status, 1);
        return;
    }
    float *row = user_w + u * ldu;
    for (int j = lane; j < d0; j += 64) {
        const int f = j / fw, k = j % fw;
        const float feat = t.table[f][fi[f] * fw + k];
        // two rounded products and one rounded add, as torch evaluates NGCF.py:114-115 (no FMA contraction)
        row[j] = __fadd_rn(__fmul_rn(row[j], one_minus_ratio), __fmul_rn(feat, ratio));
    }
}

// pass 3: restore scratch to -1
__global__ void inject_reset_kernel(const int64_t *__restrict__ u_id, int64_t B, int64_t n_user, int32_t *winner)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t u = u_id[b];
    if (u >= 0 && u < n_user) winner[u] = -1;
}

extern "C" int ngcf_feature_inject_f32(float *user_w, int64_t ldu, int64_t n_user, int d0, const float *const *tables,
                                       const int64_t *const *idx, const int64_t *cards, int fw, const int64_t *u_id,
                                       int64_t B, double emb_ratio, int32_t *scratch, int32_t *status, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!user_w || !tables || !idx || !cards || !scratch || !status) return fail(NGCF_ERR_ARG, "feature_inject: null argument");
    if (B < 0 || B >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "feature_inject: bad batch size");
    if (5 * fw != d0)
        return fail(NGCF_ERR_ARG,
                    "feature_inject: shape mismatch: 5 feature tables of width %d give %d columns, user rows have %d "
                    "(embed_size must be a multiple of 5, NGCF.py:39-43,114)", fw, 5 * fw, d0);
    if (B == 0) return NGCF_OK;
    if (!u_id) return fail(NGCF_ERR_ARG, "feature_inject: null u_id");
    InjectTables t;
    for (int f = 0; f < 5; ++f) {
        if (!tables[f] || !idx[f]) return fail(NGCF_ERR_ARG, "feature_inject: null table/index %d", f);
        t.table[f] = tables[f];
        t.idx[f] = idx[f];
        t.card[f] = cards[f];
    }
    const int tb = (int)((B + 255) / 256);
    inject_claim_kernel<<<tb, 256, 0, stream>>>(u_id, B, n_user, scratch, status);
    LAUNCH_CHECK();
    // `1 - emb_ratio` is a Python double in the reference and reaches the fp32 multiply rounded once
    inject_write_kernel<<<(int)((B + 3) / 4), 256, 0, stream>>>(user_w, ldu, n_user, d0, t, fw, u_id, B, (float)emb_ratio,
                                                                 (float)(1.0 - emb_ratio), scratch, status);
    LAUNCH_CHECK();
    inject_reset_kernel<<<tb, 256, 0, stream>>>(u_id, B, n_user, scratch);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---------------------------------------------------------------------------------------------
// row gather, NGCF.py:151-155 (bit-exact copies)
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ table, int64_t ld, int d,
                                                          const int64_t *__restrict__ idx, int64_t B, int64_t row_off,
                                                          int64_t n_idx_rows, float *__restrict__ out, int64_t ldo,
                                                          int32_t *status)
{
    using V = typename VecT<VEC>::type;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int64_t i = idx[b];
    if (i < 0 || i >= n_idx_rows) {
        if (lane == 0) atomicOr(status, 1);
        return;
    }
    const float *src = table + (row_off + i) * ld;
    float *dst = out + b * ldo;
    for (int o = lane * VEC; o < d; o += 64 * VEC) *reinterpret_cast<V *>(dst + o) = *reinterpret_cast<const V *>(src + o);
}

extern "C" int ngcf_gather_rows_f32(const float *table, int64_t ld, int d, const int64_t *idx, int64_t B,
                                    int64_t row_off, int64_t n_idx_rows, float *out, int64_t ldo, int32_t *status,
                                    void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (B == 0) return NGCF_OK;
    if (!table || !out || !status || !idx || d <= 0 || B < 0 || ld < d || ldo < d)
        return fail(NGCF_ERR_ARG, "gather_rows: bad argument");
    const bool vec = d % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && aligned16(table) && aligned16(out);
    const int blocks = (int)((B + 3) / 4);
    if (vec)
        gather_rows_kernel<4><<<blocks, 256, 0, stream>>>(table, ld, d, idx, B, row_off, n_idx_rows, out, ldo, status);
    else
        gather_rows_kernel<1><<<blocks, 256, 0, stream>>>(table, ld, d, idx, B, row_off, n_idx_rows, out, ldo, status);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// The three gathers of a forward (users, positive items, negative items: NGCF.py:151-155) in one launch: at Seoul sizes a
// forward is launch-bound and each gather is a 4 us kernel.
struct GatherSet {
    const int64_t *idx;
    int64_t B, row_off, n_idx_rows;
    float *out;
};
template <int VEC>
__global__ __launch_bounds__(256) void gather_rows3_kernel(const float *__restrict__ table, int64_t ld, int d, GatherSet s0, GatherSet s1,
                                                           GatherSet s2, int64_t blocks0, int64_t blocks1, int64_t ldo,
                                                           int32_t *status)
{
    using V = typename VecT<VEC>::type;
    int64_t blk = blockIdx.x;
    GatherSet s = s0;
    if (blk >= blocks0 + blocks1) s = s2, blk -= blocks0 + blocks1;
    else if (blk >= blocks0) s = s1, blk -= blocks0;
    const int64_t b = blk * 4 + (threadIdx.x >> 6);
    if (b >= s.B) return;
    const int lane = threadIdx.x & 63;
    const int64_t i = s.idx[b];
    if (i < 0 || i >= s.n_idx_rows) {
        if (lane == 0) atomicOr(status, 1);
        return;
    }
    const float *src = table + (s.row_off + i) * ld;
    float *dst = s.out + b * ldo;
    for (int o = lane * VEC; o < d; o += 64 * VEC) *reinterpret_cast<V *>(dst + o) = *reinterpret_cast<const V *>(src + o);
}

extern "C" int ngcf_gather_rows3_f32(const float *table, int64_t ld, int d, const int64_t *idx0, int64_t B0, int64_t row_off0,
                                     int64_t n_rows0, float *out0, const int64_t *idx1, int64_t B1, int64_t row_off1,
                                     int64_t n_rows1, float *out1, const int64_t *idx2, int64_t B2, int64_t row_off2,
                                     int64_t n_rows2, float *out2, int64_t ldo, int32_t *status, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (B0 < 0 || B1 < 0 || B2 < 0) return fail(NGCF_ERR_ARG, "gather_rows3: negative batch");
    if (B0 + B1 + B2 == 0) return NGCF_OK;
    if (!table || !status || d <= 0 || ld < d || ldo < d || (B0 && (!idx0 || !out0)) || (B1 && (!idx1 || !out1)) ||
        (B2 && (!idx2 || !out2)))
        return fail(NGCF_ERR_ARG, "gather_rows3: bad argument");
    const bool vec = d % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && aligned16(table) && (!B0 || aligned16(out0)) &&
                     (!B1 || aligned16(out1)) && (!B2 || aligned16(out2));
    const int64_t k0 = (B0 + 3) / 4, k1 = (B1 + 3) / 4, k2 = (B2 + 3) / 4;
    if (k0 + k1 + k2 >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "gather_rows3: too many rows for one launch");
    const GatherSet s0{idx0, B0, row_off0, n_rows0, out0}, s1{idx1, B1, row_off1, n_rows1, out1}, s2{idx2, B2, row_off2, n_rows2, out2};
    if (vec)
        gather_rows3_kernel<4><<<dim3((unsigned)(k0 + k1 + k2)), 256, 0, stream>>>(table, ld, d, s0, s1, s2, k0, k1, ldo, status);
    else
        gather_rows3_kernel<1><<<dim3((unsigned)(k0 + k1 + k2)), 256, 0, stream>>>(table, ld, d, s0, s1, s2, k0, k1, ldo, status);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---------------------------------------------------------------------------------------------
// BPR, bprloss.py:15-22
// ---------------------------------------------------------------------------------------------
__device__ inline float log_sigmoid(float x)
{
    return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// one wave per row r < R; block partials: part[2*block + 0] = -sum logsigmoid, part[2*block + 1] = sum of squares
__global__ __launch_bounds__(256) void bpr_rows_kernel(const float *__restrict__ u, int64_t Bu, const float *__restrict__ p,
                                                       int64_t Bp, const float *__restrict__ n, int64_t Bn, int64_t R,
                                                       int D, float *__restrict__ part)
{
    __shared__ float sh[8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    float nl = 0.f, sq = 0.f;
    if (r < R) {
        const float *ur = u + (Bu == 1 ? 0 : r) * (int64_t)D;
        const float *pr = p + (Bp == 1 ? 0 : r) * (int64_t)D;
        const float *nr = n + (Bn == 1 ? 0 : r) * (int64_t)D;
        float up = 0.f, un = 0.f, uu = 0.f, pp = 0.f, nn = 0.f;
        for (int j = lane; j < D; j += 64) {
            const float a = ur[j], b = pr[j], c = nr[j];
            up = fmaf(a, b, up);
            un = fmaf(a, c, un);
            uu = fmaf(a, a, uu);
            pp = fmaf(b, b, pp);
            nn = fmaf(c, c, nn);
        }
        up = wave_sum(up);
        un = wave_sum(un);
        uu = wave_sum(uu);
        pp = wave_sum(pp);
        nn = wave_sum(nn);
        nl = -log_sigmoid(fabsf(up) - fabsf(un));                  // bprloss.py:16-19
        // each tensor's own rows are counted once (a broadcast row only at r == 0)
        sq = (r < Bu ? uu : 0.f) + (r < Bp ? pp : 0.f) + (r < Bn ? nn : 0.f);
    }
    if (lane == 0) {
        sh[wave * 2] = nl;
        sh[wave * 2 + 1] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * (int64_t)blockIdx.x] = (sh[0] + sh[2]) + (sh[4] + sh[6]);
        part[2 * (int64_t)blockIdx.x + 1] = (sh[1] + sh[3]) + (sh[5] + sh[7]);
    }
}

__global__ __launch_bounds__(256) void bpr_finish_kernel(const float *__restrict__ part, int64_t n_blocks, float wd,
                                                         float batch_size, float *__restrict__ loss)
{
    __shared__ float sh[8];
    float nl = 0.f, sq = 0.f;
    for (int64_t i = threadIdx.x; i < n_blocks; i += 256) {
        nl += part[2 * i];
        sq += part[2 * i + 1];
    }
    nl = wave_sum(nl);
    sq = wave_sum(sq);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sh[wave * 2] = nl;
        sh[wave * 2 + 1] = sq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float a = (sh[0] + sh[2]) + (sh[4] + sh[6]);
        const float b = (sh[1] + sh[3]) + (sh[5] + sh[7]);
        *loss = (a + wd * b) / batch_size;                          // bprloss.py:20-22
    }
}

extern "C" int64_t ngcf_bpr_workspace_bytes(int64_t R)
{
    if (R < 0) return -1;
    return align_up(((R + 3) / 4) * 2 * (int64_t)sizeof(float), 256) + 256;
}

extern "C" int ngcf_bpr_fused_f32(const float *u, int64_t Bu, const float *p, int64_t Bp, const float *n, int64_t Bn,
                                  int D, float wd, float batch_size, float *loss, void *workspace,
                                  int64_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!u || !p || !n || !loss || D <= 0) return fail(NGCF_ERR_ARG, "bpr: null argument");
    const int64_t R = std::max(Bu, std::max(Bp, Bn));
    if (R < 1) return fail(NGCF_ERR_ARG, "bpr: empty operand (rows %lld/%lld/%lld)", (long long)Bu, (long long)Bp, (long long)Bn);
    if ((Bu != 1 && Bu != R) || (Bp != 1 && Bp != R) || (Bn != 1 && Bn != R))
        return fail(NGCF_ERR_ARG, "bpr: row counts %lld/%lld/%lld do not broadcast", (long long)Bu, (long long)Bp, (long long)Bn);
    const int64_t need = ngcf_bpr_workspace_bytes(R);
    if (!workspace || workspace_bytes < need)
        return fail(NGCF_ERR_WORKSPACE, "bpr: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    float *part = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
    const int64_t blocks = (R + 3) / 4;
    bpr_rows_kernel<<<dim3((unsigned)blocks), 256, 0, stream>>>(u, Bu, p, Bp, n, Bn, R, D, part);
    LAUNCH_CHECK();
    bpr_finish_kernel<<<1, 256, 0, stream>>>(part, blocks, wd, batch_size, loss);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---------------------------------------------------------------------------------------------
// row partition helper (host only)
// ---------------------------------------------------------------------------------------------
extern "C" int ngcf_shard_plan(const int64_t *rowptr, int64_t row_begin, int64_t row_end, int world, int64_t *bounds)
{
    if (!rowptr || !bounds || world < 1 || row_begin < 0 || row_end < row_begin)
        return fail(NGCF_ERR_ARG, "shard_plan: bad argument");
    const int64_t total = rowptr[row_end] - rowptr[row_begin];
    bounds[0] = row_begin;
    int64_t r = row_begin;
    for (int w = 1; w < world; ++w) {
        // first row whose prefix reaches w/world of the entries
        const int64_t target = rowptr[row_begin] + (total * w) / world;
        const int64_t *it = std::lower_bound(rowptr + r, rowptr + row_end + 1, target);
        int64_t cut = it - rowptr;
        // of the two row boundaries around the target take the nearer one
        if (cut > r && cut <= row_end && target - rowptr[cut - 1] < rowptr[cut] - target) --cut;
        if (cut < r) cut = r;
        if (cut > row_end) cut = row_end;
        if (total == 0) cut = row_begin + ((row_end - row_begin) * w) / world;
        bounds[w] = cut;
        r = cut;
    }
    bounds[world] = row_end;
    return NGCF_OK;
}


// =============================================================================================
// Top-k selection per row (`torch.topk` on the score matrix, experiment.py:104-111,
// demo.py:234-235).  One workgroup per row: a 4-pass 8-bit radix select finds the k-th largest key, one more
// pass collects the k winners (ties at the threshold: lowest column first), a bitonic sort in LDS orders them
// descending (equal values: lowest column first).  The score matrix itself is a plain GEMM (u . items^T).
// =============================================================================================
#define NGCF_TOPK_MAX 1024

__device__ inline uint32_t float_key(float x)      // monotone map float -> uint32 (larger float = larger key)
{
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct TopkShared {
    uint32_t hist[256];
    uint32_t sh_prefix, sh_need, sh_cnt_gt, sh_cnt_eq;
    uint32_t skey[NGCF_TOPK_MAX];
    int32_t sidx[NGCF_TOPK_MAX];
};

// top-k of one score row by the 256 threads of a workgroup; writes out_val[0..k), out_idx[0..k)
__device__ inline void topk_one_row(const float *__restrict__ row, int64_t n_cols, int k, int kp2, float *__restrict__ out_val,
                                    int64_t *__restrict__ out_idx, TopkShared &sh)
{
    uint32_t *hist = sh.hist, *skey = sh.skey;
    int32_t *sidx = sh.sidx;
    uint32_t &sh_prefix = sh.sh_prefix, &sh_need = sh.sh_need, &sh_cnt_gt = sh.sh_cnt_gt, &sh_cnt_eq = sh.sh_cnt_eq;
    const int tid = threadIdx.x;
    // ---- radix select: after the 4 passes `prefix` is the key of the k-th largest element
    uint32_t prefix = 0, need = (uint32_t)k;      // `need` = how many of the current prefix class are still wanted
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0;
        __syncthreads();
        const uint32_t mask_hi = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int64_t j = tid; j < n_cols; j += 256) {
            const uint32_t key = float_key(row[j]);
            if ((key & mask_hi) == (prefix & mask_hi)) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t acc = 0;
            int dsel = 0;
            for (int dgt = 255; dgt >= 0; --dgt) {
                if (acc + hist[dgt] >= need) {
                    dsel = dgt;
                    break;
                }
                acc += hist[dgt];
            }
            sh_prefix = prefix | ((uint32_t)dsel << shift);
            sh_need = need - acc;
        }
        __syncthreads();
        prefix = sh_prefix;
        need = sh_need;
        __syncthreads();
    }
    // ---- collect: all keys above the threshold, then `need` keys equal to it in column order
    if (tid == 0) {
        sh_cnt_gt = 0;
        sh_cnt_eq = 0;
    }
    for (int j = tid; j < kp2; j += 256) {
        skey[j] = 0u;                     // padding sorts last
        sidx[j] = 0x7fffffff;
    }
    __syncthreads();
    const uint32_t n_gt = (uint32_t)k - need;
    for (int64_t j0 = 0; j0 < n_cols; j0 += 256) {         // block-ordered so that ties keep the lowest columns
        const int64_t j = j0 + tid;
        uint32_t key = 0;
        bool gt = false, eq = false;
        if (j < n_cols) {
            key = float_key(row[j]);
            gt = key > prefix;
            eq = key == prefix;
        }
        if (gt) {
            const uint32_t pos = atomicAdd(&sh_cnt_gt, 1u);
            skey[pos] = key;
            sidx[pos] = (int32_t)j;
        }
        // equal keys: rank inside this 256-column block by a wave/LDS-free trick - serialise through LDS counter in order
        __syncthreads();
        if (eq) hist[tid] = 1; else hist[tid] = 0;
        __syncthreads();
        if (eq) {
            uint32_t before = 0;
            for (int t = 0; t < tid; ++t) before += hist[t];
            const uint32_t pos = sh_cnt_eq + before;
            if (pos < need) {
                skey[n_gt + pos] = key;
                sidx[n_gt + pos] = (int32_t)j;
            }
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t tot = 0;
            for (int t = 0; t < 256; ++t) tot += hist[t];
            sh_cnt_eq += tot;
        }
        __syncthreads();
    }
    // ---- bitonic sort, descending by (key, then ascending column)
    for (int size = 2; size <= kp2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < kp2; i += 256) {
                const int p = i ^ stride;
                if (p > i) {
                    const bool desc = (i & size) == 0;
                    const uint32_t ka = skey[i], kb = skey[p];
                    const int32_t ia = sidx[i], ib = sidx[p];
                    const bool a_first = ka > kb || (ka == kb && ia < ib);     // a belongs before b in the final order
                    if (a_first != desc) {
                        skey[i] = kb; skey[p] = ka;
                        sidx[i] = ib; sidx[p] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k; j += 256) {
        out_idx[j] = sidx[j];
        out_val[j] = row[sidx[j]];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void topk_rows_kernel(const float *__restrict__ scores, int64_t ld, int64_t n_cols, int k,
                                                        int kp2, float *__restrict__ out_val, int64_t *__restrict__ out_idx)
{
    __shared__ TopkShared sh;
    topk_one_row(scores + (int64_t)blockIdx.x * ld, n_cols, k, kp2, out_val + (int64_t)blockIdx.x * k,
                 out_idx + (int64_t)blockIdx.x * k, sh);
}

// ---------------------------------------------------------------------------------------------
// Score-and-select in one launch (experiment.py:93,104-109: `torch.mm(u_embeds, pos_i_embeds.T)` -> topk; demo.py:233-235:
// `torch.mm(u_embeds, all_items_emb.T)` -> topk(100)).  A workgroup owns kRecRows user rows.  Phase A: their scores against
// every item, item tiles of 256 x 32 staged through LDS (coalesced 128-byte row segments), each thread one item x all the
// workgroup's user rows (the user values are LDS broadcasts), written to the workgroup's rows of the score scratch.
// Phase B: the radix select + bitonic sort above on each of those rows.  No library GEMM, no second launch.
// ---------------------------------------------------------------------------------------------
#define NGCF_REC_ROWS 8
#define NGCF_REC_TK 32

__global__ __launch_bounds__(256) void recommend_topk_kernel(const float *__restrict__ U, int64_t ldu, int64_t B,
                                                             const float *__restrict__ items, int64_t ldi, int64_t n_items, int D,
                                                             int k, int kp2, float *__restrict__ scratch, int64_t lds_,
                                                             float *__restrict__ out_val, int64_t *__restrict__ out_idx)
{
    constexpr int RB = NGCF_REC_ROWS, TK = NGCF_REC_TK, TI = 256, SLD = TK + 4;   // item rows in LDS: 36 floats, b128-aligned
    __shared__ float sI[TI * SLD];
    __shared__ float sU[RB * TK];
    __shared__ TopkShared sh;
    const int tid = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * RB;
    const int n_b = (int)((B - b0) < RB ? (B - b0) : RB);
    for (int64_t i0 = 0; i0 < n_items; i0 += TI) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < D; k0 += TK) {
            __syncthreads();
            // stage: 8 consecutive threads read the 32 floats (128 B) of one item row; zero beyond the matrix
#pragma unroll
            for (int j = 0; j < TI * TK / 256; ++j) {
                const int f = tid + 256 * j;
                const int it = f / TK, kk = f % TK;
                const int64_t gi = i0 + it;
                sI[it * SLD + kk] = (gi < n_items && k0 + kk < D) ? items[gi * ldi + k0 + kk] : 0.f;
            }
            {
                const int r = tid / TK, kk = tid % TK;           // 256 threads = RB x TK exactly
                sU[tid] = (r < n_b && k0 + kk < D) ? U[(b0 + r) * ldu + k0 + kk] : 0.f;
            }
            __syncthreads();
            const float *mine = sI + tid * SLD;
#pragma unroll
            for (int kq = 0; kq < TK; kq += 4) {
                const float4 x = *reinterpret_cast<const float4 *>(mine + kq);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const float4 u = *reinterpret_cast<const float4 *>(sU + r * TK + kq);   // broadcast
                    acc[r] = fmaf(x.x, u.x, acc[r]);
                    acc[r] = fmaf(x.y, u.y, acc[r]);
                    acc[r] = fmaf(x.z, u.z, acc[r]);
                    acc[r] = fmaf(x.w, u.w, acc[r]);
                }
            }
        }
        if (i0 + tid < n_items) {
#pragma unroll
            for (int r = 0; r < RB; ++r)
                if (r < n_b) scratch[(b0 + r) * lds_ + i0 + tid] = acc[r];
        }
    }
    __threadfence();          // the rows are re-read by other threads of this workgroup
    __syncthreads();
    for (int r = 0; r < n_b; ++r)
        topk_one_row(scratch + (b0 + r) * lds_, n_items, k, kp2, out_val + (b0 + r) * k, out_idx + (b0 + r) * k, sh);
}



extern "C" int ngcf_topk_rows_f32(const float *scores, int64_t ld, int64_t n_rows, int64_t n_cols, int k, float *out_val,
                                  int64_t *out_idx, void *stream_)
{
    if (n_rows == 0) return NGCF_OK;
    if (!scores || !out_val || !out_idx || ld < n_cols) return fail(NGCF_ERR_ARG, "topk_rows: bad argument");
    if (k < 1 || k > n_cols) return fail(NGCF_ERR_ARG, "selected index k out of range (k=%d, row length %lld)", k, (long long)n_cols);
    if (k > NGCF_TOPK_MAX) return fail(NGCF_ERR_ARG, "topk_rows: k=%d > %d is not supported", k, NGCF_TOPK_MAX);
    if (n_cols >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "topk_rows: row too long");
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    topk_rows_kernel<<<dim3((unsigned)n_rows), 256, 0, (hipStream_t)stream_>>>(scores, ld, n_cols, k, kp2, out_val, out_idx);
    LAUNCH_CHECK();
    return NGCF_OK;
}

extern "C" int ngcf_recommend_topk_f32(const float *u, int64_t ldu, int64_t B, const float *items, int64_t ldi, int64_t n_items,
                                       int D, int k, float *scratch, int64_t ld_scratch, float *out_val, int64_t *out_idx,
                                       void *stream_)
{
    if (B == 0) return NGCF_OK;
    if (!u || !items || !scratch || !out_val || !out_idx || D <= 0 || ldu < D || ldi < D || ld_scratch < n_items)
        return fail(NGCF_ERR_ARG, "recommend_topk: bad argument");
    if (k < 1 || k > n_items) return fail(NGCF_ERR_ARG, "selected index k out of range (k=%d, row length %lld)", k, (long long)n_items);
    if (k > NGCF_TOPK_MAX) return fail(NGCF_ERR_ARG, "recommend_topk: k=%d > %d is not supported", k, NGCF_TOPK_MAX);
    if (n_items >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "recommend_topk: too many items");
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    const int64_t blocks = (B + NGCF_REC_ROWS - 1) / NGCF_REC_ROWS;
    recommend_topk_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream_>>>(u, ldu, B, items, ldi, n_items, D, k, kp2, scratch,
                                                                                   ld_scratch, out_val, out_idx);
    LAUNCH_CHECK();
    return NGCF_OK;
}


// ---------------------------------------------------------------------------------------------
// Dropout seeds that live on the device (common.h, resolve_seed): every word steps to the next value of a counter-based chain.
// Launched by the mirror at the start of a training forward in "device" dropout mode - inside a captured hipGraph too, so that
// every replay draws new masks although its kernel arguments are baked in.
// ---------------------------------------------------------------------------------------------
__global__ void seeds_advance_kernel(uint64_t *seeds, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = seeds[i] + 0x9E3779B97F4A7C15ULL;          // splitmix64 step
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    x ^= x >> 31;
    seeds[i] = x >> 2;                                      // below 2^62: a value never looks like a tagged address
}

extern "C" int ngcf_seeds_advance(uint64_t *seeds, int n, void *stream)
{
    if (n <= 0) return NGCF_OK;
    if (!seeds) return fail(NGCF_ERR_ARG, "seeds_advance: null argument");
    seeds_advance_kernel<<<(n + 63) / 64, 64, 0, (hipStream_t)stream>>>(seeds, n);
    LAUNCH_CHECK();
    return NGCF_OK;
}
