// spmm.hip - row-wise CSR SpMM kernels (plain + d-sliced), dispatch, workspace sizing, launch timing.
#include "spmm_device.h"

// One launch covers the whole product: the first `seg_blocks` workgroups take the segments of the cut
// rows (the longest units, so they start first; partial sums go to the workspace [n_seg][dp]), the rest
// take one uncut row per wave and write it directly.  Cut rows are finished by spmm_fixup_kernel.
template <int VEC, int LPR, int CH, int U>
__global__ __launch_bounds__(256) void spmm_kernel(const int64_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ colidx,
                                                   const float *__restrict__ vals, int64_t row_begin,
                                                   int64_t n_rows, const int32_t *__restrict__ seg_row,
                                                   const int64_t *__restrict__ seg_begin, int64_t n_seg,
                                                   int64_t seg_blocks, int seg_len,
                                                   const float *__restrict__ E, int64_t ldE, int d,
                                                   float *__restrict__ out, int64_t ldo,
                                                   float *__restrict__ partial, int dp, EdgeDrop dr)
{
    using V = typename VecT<VEC>::type;
    const int wave = threadIdx.x >> 6;
    int64_t begin, end;
    float *dst;
    if ((int64_t)blockIdx.x < seg_blocks) {
        const int64_t s = (int64_t)blockIdx.x * 4 + wave;
        if (s >= n_seg) return;
        begin = seg_begin[s];
        const int64_t row_end = rowptr[seg_row[s] + 1];
        end = begin + seg_len < row_end ? begin + seg_len : row_end;
        dst = partial + s * (int64_t)dp;
    } else {
        const int64_t row = row_begin + ((int64_t)blockIdx.x - seg_blocks) * 4 + wave;
        if (row >= n_rows) return;
        begin = rowptr[row];
        end = rowptr[row + 1];
        if (end - begin > seg_len) return;   // cut row: produced from its segments
        dst = out + row * ldo;
    }
    V acc[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) acc[ch] = vzero<VEC>();
    spmm_accumulate<VEC, LPR, CH, U>(colidx, vals, begin, end, E, ldE, d, acc, dr);
    spmm_store<VEC, LPR, CH>(acc, dst, d);
}

// Rows [row_begin, row_end) in slices of 32 floats, slice-major: every CU works on the same 128-B slice of the
// gathered table at a time, so for a table of a few hundred thousand rows the hot rows of that slice stay in L2.
template <int U>
__global__ __launch_bounds__(256) void spmm_sliced_kernel(const int64_t *__restrict__ rowptr,
                                                          const int32_t *__restrict__ colidx,
                                                          const float *__restrict__ vals, int64_t row_begin,
                                                          int64_t row_end, int64_t row_blocks, int seg_len,
                                                          const float *__restrict__ E, int64_t ldE,
                                                          float *__restrict__ out, int64_t ldo, EdgeDrop dr)
{
    const int64_t slice = blockIdx.x / row_blocks;
    const int64_t row = row_begin + ((int64_t)blockIdx.x % row_blocks) * 4 + (threadIdx.x >> 6);
    if (row >= row_end) return;
    const int64_t begin = rowptr[row], end = rowptr[row + 1];
    if (end - begin > seg_len) return;   // cut row: produced from its segments
    float4 acc[1];
    acc[0] = vzero4();
    spmm_accumulate<4, 8, 1, U>(colidx, vals, begin, end, E + slice * 32, ldE, 32, acc, dr);
    spmm_store<4, 8, 1>(acc, out + row * ldo + slice * 32, 32);
}


// Compact side table of a 1..4-column tail panel: T[c] = {E[c, 0..tail), 0...} as one 16-byte row per table row.  The widths
// the reference forces on the first layer (65, 130, 515: NGCF.py:39-43) leave 1..3 columns beyond the wide panel; gathered
// from the strided table each of them costs a 64-byte sector per stored entry (100 M isolated sectors on C3, ~2 ms), from
// this table (17.6 MB at C3: L2/Infinity-Cache resident) one 16-byte load per entry, 64 entries per wave instruction.
__global__ void tail_pack_kernel(const float *__restrict__ E, int64_t ldE, int64_t n, int tail, float4 *__restrict__ T)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const float *e = E + r * ldE;
        float4 t = make_float4(e[0], 0.f, 0.f, 0.f);
        if (tail > 1) t.y = e[1];
        if (tail > 2) t.z = e[2];
        if (tail > 3) t.w = e[3];
        T[r] = t;
    }
}

// the tail product lands in a compact [n_rows, 4] block and is copied into its columns of `out` (which may be a column
// slice of a wider matrix: nothing beyond the `tail` columns is written)
__global__ void tail_unpack_kernel(const float4 *__restrict__ T, int64_t n, int tail, float *__restrict__ out, int64_t ldo)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const float4 t = T[r];
        float *o = out + r * ldo;
        o[0] = t.x;
        if (tail > 1) o[1] = t.y;
        if (tail > 2) o[2] = t.z;
        if (tail > 3) o[3] = t.w;
    }
}

// The tail product itself: [n_rows, 4] = L . T[n_cols, 4].  16 lanes per row (or per <= seg_len-entry segment of a cut row),
// four of them per wave, four independent (col, val, gather) triples in flight per lane: a 50-entry user row is one pass of
// its group.  One wave per row (spmm_kernel<4,1,1,1>) is latency-bound at ~0.5 us per row: 1.0 ms per product on C3, this
// form 0.3 ms.
__global__ __launch_bounds__(256) void spmm_tail_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                        const float *__restrict__ vals, int64_t n_rows,
                                                        const int32_t *__restrict__ seg_row, const int64_t *__restrict__ seg_begin,
                                                        int64_t n_seg, int64_t seg_blocks, int seg_len, const float4 *__restrict__ T,
                                                        float4 *__restrict__ out, float4 *__restrict__ partial)
{
    const int unit = threadIdx.x >> 4, l = threadIdx.x & 15;
    int64_t begin = 0, end = 0;
    float4 *dst = nullptr;
    if ((int64_t)blockIdx.x < seg_blocks) {
        const int64_t s = (int64_t)blockIdx.x * 16 + unit;
        if (s < n_seg) {
            begin = seg_begin[s];
            const int64_t row_end = rowptr[seg_row[s] + 1];
            end = begin + seg_len < row_end ? begin + seg_len : row_end;
            dst = partial + s;
        }
    } else {
        const int64_t row = ((int64_t)blockIdx.x - seg_blocks) * 16 + unit;
        if (row < n_rows) {
            begin = rowptr[row];
            end = rowptr[row + 1];
            dst = out + row;
            if (end - begin > seg_len) dst = nullptr, end = begin;   // cut row: produced from its segments
        }
    }
    float4 a0 = vzero4(), a1 = vzero4(), a2 = vzero4(), a3 = vzero4();
    for (int64_t e = begin + l; e < end; e += 64) {
        const int64_t e1 = e + 16, e2 = e + 32, e3 = e + 48;
        const bool k1 = e1 < end, k2 = e2 < end, k3 = e3 < end;
        const int c0 = colidx[e], c1 = k1 ? colidx[e1] : 0, c2 = k2 ? colidx[e2] : 0, c3 = k3 ? colidx[e3] : 0;
        const float v0 = vals[e], v1 = k1 ? vals[e1] : 0.f, v2 = k2 ? vals[e2] : 0.f, v3 = k3 ? vals[e3] : 0.f;
        const float4 t0 = T[c0], t1 = T[c1], t2 = T[c2], t3 = T[c3];
        a0 = vfma(v0, t0, a0);
        a1 = vfma(v1, t1, a1);
        a2 = vfma(v2, t2, a2);
        a3 = vfma(v3, t3, a3);
    }
    float4 acc = vadd(vadd(a0, a1), vadd(a2, a3));
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) acc = vadd(acc, vshfl_xor(acc, m));
    if (l == 0 && dst) *dst = acc;
}

static int64_t tail_table_bytes(const ngcf_csr *c)
{
    return align_up(c->n_cols * (int64_t)sizeof(float4), 256) + align_up(c->n_rows * (int64_t)sizeof(float4), 256) + 512;
}

static int64_t partial_bytes(const ngcf_csr *c, int d)
{
    const int64_t n_part = std::max(c->n_seg, c->swept.out.n_seg + c->swept.n_partial);
    return align_up(n_part * align_up(d, 4) * (int64_t)sizeof(float), 256) + 256;
}

extern "C" int64_t ngcf_spmm_workspace_bytes(const ngcf_csr_t *c, int d)
{
    if (!c || d <= 0) return -1;
    // [partial sums of the cut rows][compact tail table] - the table sits at the end, the partial sums at the start
    return partial_bytes(c, d) + tail_table_bytes(c);
}

// ---------------------------------------------------------------------------------------------
// optional in-library timing of the dominant kernel (bench.py's roofline figure): when enabled,
// a hipEvent pair is recorded on the launch stream around every spmm_kernel launch.
// ---------------------------------------------------------------------------------------------
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_prof_events;   // pairs: begin, end
static size_t g_prof_used = 0;

void prof_mark(hipStream_t stream, int which)
{
    if (!g_prof_on) return;
    if (g_prof_used >= g_prof_events.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        g_prof_events.push_back(e);
    }
    (void)which;
    (void)hipEventRecord(g_prof_events[g_prof_used++], stream);
}

extern "C" int ngcf_prof_enable(int on)
{
    g_prof_on = on != 0;
    g_prof_used = 0;
    return NGCF_OK;
}

// Waits for the recorded events; returns the number of timed spmm launches and their summed duration.
extern "C" int ngcf_prof_collect(int64_t *n_launches, double *total_ms)
{
    if (!n_launches || !total_ms) return fail(NGCF_ERR_ARG, "prof_collect: null argument");
    double sum = 0.0;
    int64_t n = 0;
    for (size_t i = 0; i + 1 < g_prof_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(hipEventSynchronize(g_prof_events[i + 1]));
        HIP_TRY(hipEventElapsedTime(&ms, g_prof_events[i], g_prof_events[i + 1]));
        sum += ms;
        ++n;
    }
    *n_launches = n;
    *total_ms = sum;
    g_prof_used = 0;
    return NGCF_OK;
}

namespace {
struct SpmmArgs {
    bool with_swept;        // the swept parts take their row groups; the row-wise kernels take the rest
    const ngcf_csr *c;
    const float *E;
    int64_t ldE;
    int d;
    float *out;
    int64_t ldo;
    float *partial;
    int dp;
    hipStream_t stream;
    EdgeDrop dr;
};

template <int VEC, int LPR, int CH, int U>
int launch_spmm(const SpmmArgs &a)
{
    const ngcf_csr *c = a.c;
    // cut rows: all of them, or (beside the swept parts) those of the groups the parts do not cover
    const int64_t n_seg = a.with_swept ? c->swept.out.n_seg : c->n_seg, n_heavy = a.with_swept ? c->swept.out.n_heavy : c->n_heavy;
    const int32_t *seg_row = a.with_swept ? c->swept.out.seg_row : c->seg_row;
    const int64_t *seg_begin = a.with_swept ? c->swept.out.seg_begin : c->seg_begin;
    const int32_t *heavy_row = a.with_swept ? c->swept.out.heavy_row : c->heavy_row;
    const int64_t *heavy_seg_ptr = a.with_swept ? c->swept.out.heavy_seg_ptr : c->heavy_seg_ptr;
    const int64_t seg_blocks = (n_seg + 3) / 4;
    // d-slicing of the sliceable row groups needs 16-byte slices of 32 floats
    const bool can_slice = VEC == 4 && a.d % 32 == 0 && a.d >= 64 && c->mode != 1 && !getenv("NGCF_NO_SLICING");
    bool seg_done = seg_blocks == 0;
    for (size_t g = 0; g <= c->groups.size(); ++g) {
        const bool last = g == c->groups.size();
        if (last && seg_done) break;
        if (!last && a.with_swept && c->swept.group_swept[g]) continue;
        if (!last && can_slice && c->groups[g].sliceable) {
            const int64_t rb = (c->groups[g].end - c->groups[g].begin + 3) / 4;
            const int64_t blocks = rb * (a.d / 32);
            if (blocks >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "spmm: too many rows for one launch");
            spmm_sliced_kernel<8><<<dim3((unsigned)blocks), 256, 0, a.stream>>>(c->rowptr, c->colidx, c->vals, c->groups[g].begin,
                                                                               c->groups[g].end, rb, c->seg_len, a.E, a.ldE,
                                                                               a.out, a.ldo, a.dr);
            LAUNCH_CHECK();
            continue;
        }
        // unsliced group; the segments of the cut rows ride in front of the first such launch
        const int64_t rbeg = last ? 0 : c->groups[g].begin, rend = last ? 0 : c->groups[g].end;
        const int64_t sb = seg_done ? 0 : seg_blocks;
        const int64_t blocks = sb + (rend - rbeg + 3) / 4;
        if (blocks >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "spmm: too many rows for one launch");
        if (blocks > 0) {
            spmm_kernel<VEC, LPR, CH, U><<<dim3((unsigned)blocks), 256, 0, a.stream>>>(
                c->rowptr, c->colidx, c->vals, rbeg, rend, seg_row, seg_begin, seg_done ? 0 : n_seg, sb, c->seg_len,
                a.E, a.ldE, a.d, a.out, a.ldo, a.partial, a.dp, a.dr);
            LAUNCH_CHECK();
        }
        seg_done = true;
    }
    if (n_heavy > 0) {
        const int64_t fb = (n_heavy + 3) / 4;
        spmm_fixup_kernel<VEC><<<dim3((unsigned)fb), 256, 0, a.stream>>>(heavy_row, heavy_seg_ptr, n_heavy, a.partial, a.dp, a.d, a.out,
                                                                          a.ldo);
        LAUNCH_CHECK();
    }
    return NGCF_OK;
}
}  // namespace

int spmm_dispatch(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo, void *workspace,
                  int64_t workspace_bytes, hipStream_t stream, const EdgeDrop &dr)
{
    if (!c || !E || !out) return fail(NGCF_ERR_ARG, "spmm: null argument");
    if (d <= 0 || d > 8192) return fail(NGCF_ERR_ARG, "spmm: width d=%d not in [1, 8192]", d);
    if (ldE < d || ldo < d) return fail(NGCF_ERR_ARG, "spmm: leading dimension smaller than d");
    if (d > 512) {   // wider than one wave covers: column panels of 512 (Seoul's 515-wide first layer, BASELINE configs[1])
        for (int o = 0; o < d; o += 512) {
            const int rc = spmm_dispatch(c, E + o, ldE, std::min(512, d - o), out + o, ldo, workspace, workspace_bytes, stream, dr);
            if (rc != NGCF_OK) return rc;
        }
        return NGCF_OK;
    }
    // Widths the reference forces on the first layer (embed_size is a multiple of 5: 65, 130, 515; NGCF.py:39-43): with
    // 16-byte aligned rows the product runs as a wide main panel on the fast kernels plus a narrow tail panel.
    const bool rows_aligned = (ldE % 4 == 0) && (ldo % 4 == 0) && aligned16(E) && aligned16(out);
    // (measured at 130 on C3: 19.5 -> 18.7 ms per forward; with edge dropout the second pass over the entries costs
    // more than the scalar loads, so those products stay in one piece)
    if (rows_aligned && d > 4 && dr.n == 0 && !getenv("NGCF_NO_PANEL_SPLIT")) {
        int main = 0;
        if (d % 64 != 0 && d > 64 && swept_usable(c, ldE, d & ~63)) main = d & ~63;                // swept kernel + tail
        else if (d % 4 != 0) main = d & ~3;                                                        // float4 kernel + 1..3 columns
        if (main > 0) {
            const int rc = spmm_dispatch(c, E, ldE, main, out, ldo, workspace, workspace_bytes, stream, dr);
            if (rc != NGCF_OK) return rc;
            const int tail = d - main;
            if (tail <= 4 && workspace && workspace_bytes >= ngcf_spmm_workspace_bytes(c, d) && !getenv("NGCF_NO_TAIL_TABLE")) {
                // workspace: [partial sums ...            | T_in [n_cols] | T_out [n_rows]]
                uintptr_t end = reinterpret_cast<uintptr_t>(workspace) + (uintptr_t)workspace_bytes;
                float4 *Tout = reinterpret_cast<float4 *>((end - (uintptr_t)(c->n_rows * (int64_t)sizeof(float4))) & ~(uintptr_t)255);
                float4 *Tin = reinterpret_cast<float4 *>((reinterpret_cast<uintptr_t>(Tout) - (uintptr_t)(c->n_cols * (int64_t)sizeof(float4))) &
                                                         ~(uintptr_t)255);
                tail_pack_kernel<<<grid_for(c->n_cols, 256), 256, 0, stream>>>(E + main, ldE, c->n_cols, tail, Tin);
                LAUNCH_CHECK();
                // cut rows (the CSR's own segment plan): partial float4 per segment at the start of the workspace, then the fix-up
                float4 *tpart = reinterpret_cast<float4 *>(align_up((int64_t)(uintptr_t)workspace, 256));
                const int64_t seg_blocks = (c->n_seg + 15) / 16, row_blocks = (c->n_rows + 15) / 16;
                if (seg_blocks + row_blocks >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "spmm: too many rows for one launch");
                prof_mark(stream, 0);
                spmm_tail_kernel<<<dim3((unsigned)(seg_blocks + row_blocks)), 256, 0, stream>>>(
                    c->rowptr, c->colidx, c->vals, c->n_rows, c->seg_row, c->seg_begin, c->n_seg, seg_blocks, c->seg_len, Tin, Tout, tpart);
                LAUNCH_CHECK();
                if (c->n_heavy > 0) {
                    spmm_fixup_kernel<4><<<dim3((unsigned)((c->n_heavy + 3) / 4)), 256, 0, stream>>>(
                        c->heavy_row, c->heavy_seg_ptr, c->n_heavy, reinterpret_cast<const float *>(tpart), 4, 4,
                        reinterpret_cast<float *>(Tout), 4);
                    LAUNCH_CHECK();
                }
                prof_mark(stream, 1);
                tail_unpack_kernel<<<grid_for(c->n_rows, 256), 256, 0, stream>>>(Tout, c->n_rows, tail, out + main, ldo);
                LAUNCH_CHECK();
                return NGCF_OK;
            }
            return spmm_dispatch(c, E + main, ldE, tail, out + main, ldo, workspace, workspace_bytes, stream, dr);
        }
    }
    const int dp = (int)align_up(d, 4);
    const bool vec = (d % 4 == 0) && rows_aligned;
    const bool with_swept = vec && dr.n == 0 && swept_usable(c, ldE, d);
    float *partial = nullptr;
    if (with_swept ? c->swept.out.n_seg + c->swept.n_partial > 0 : c->n_seg > 0) {
        const int64_t need = partial_bytes(c, d);
        if (!workspace || workspace_bytes < need)
            return fail(NGCF_ERR_WORKSPACE, "spmm: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
        partial = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
    }
    SpmmArgs a{with_swept, c, E, ldE, d, out, ldo, partial, dp, stream, dr};
    prof_mark(stream, 0);                       // one L.E product = everything between the two marks
    int rc = NGCF_OK;
    if (with_swept) rc = launch_swept(c, E, ldE, d, out, ldo, partial, dp, stream);
    if (rc != NGCF_OK) return rc;
    if (vec) {
        const int nq = d / 4;
        if (nq <= 8) rc = launch_spmm<4, 8, 1, 4>(a);
        else if (nq <= 16) rc = launch_spmm<4, 16, 1, 8>(a);
        else if (nq <= 32) rc = launch_spmm<4, 32, 1, 8>(a);
        else if (nq <= 64) rc = launch_spmm<4, 64, 1, 8>(a);
        else rc = launch_spmm<4, 64, 2, 4>(a);
    } else if (d <= 8) rc = launch_spmm<1, 8, 1, 8>(a);       // narrow tail panels: 8 gathered rows per wave instruction
    else if (d <= 64) rc = launch_spmm<1, 64, 1, 8>(a);
    else if (d <= 128) rc = launch_spmm<1, 64, 2, 4>(a);
    else if (d <= 256) rc = launch_spmm<1, 64, 4, 2>(a);
    else rc = launch_spmm<1, 64, 8, 1>(a);
    prof_mark(stream, 1);
    return rc;
}

extern "C" int ngcf_spmm_csr_f32(const ngcf_csr_t *c, const float *E, int64_t ldE, int d, float *LE, int64_t ldLE,
                                 void *workspace, int64_t workspace_bytes, void *stream)
{
    return spmm_dispatch(c, E, ldE, d, LE, ldLE, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ngcf_spmm_csr_dropout_f32(const ngcf_csr_t *c, const float *E, int64_t ldE, int d, float *LE, int64_t ldLE,
                                         float drop_p, const uint64_t *seeds, int n_seeds, const int64_t *eid,
                                         void *workspace, int64_t workspace_bytes, void *stream)
{
    if (n_seeds < 0 || n_seeds > 4 || (n_seeds > 0 && !seeds)) return fail(NGCF_ERR_ARG, "spmm_dropout: 0..4 seeds expected");
    if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(NGCF_ERR_ARG, "spmm_dropout: drop_p=%f not in [0,1)", drop_p);
    EdgeDrop dr{drop_p > 0.f ? n_seeds : 0, (uint32_t)((double)drop_p * 4294967296.0), {0, 0, 0, 0}, eid};
    for (int q = 0; q < n_seeds; ++q) dr.seed[q] = seeds[q];
    return spmm_dispatch(c, E, ldE, d, LE, ldLE, workspace, workspace_bytes, (hipStream_t)stream, dr);
}

