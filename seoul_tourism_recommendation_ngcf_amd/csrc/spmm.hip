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
                                                   float *__restrict__ partial, int dp, EdgeDrop dr_in)
{
    using V = typename VecT<VEC>::type;
    const EdgeDropR dr = resolve_drop(dr_in);
    const int wave = threadIdx.x >> 6;
    int64_t begin, end, row;
    float *dst;
    if ((int64_t)blockIdx.x < seg_blocks) {
        const int64_t s = (int64_t)blockIdx.x * 4 + wave;
        if (s >= n_seg) return;
        begin = seg_begin[s];
        row = seg_row[s];
        const int64_t row_end = rowptr[row + 1];
        end = begin + seg_len < row_end ? begin + seg_len : row_end;
        dst = partial + s * (int64_t)dp;
    } else {
        row = row_begin + ((int64_t)blockIdx.x - seg_blocks) * 4 + wave;
        if (row >= n_rows) return;
        begin = rowptr[row];
        end = rowptr[row + 1];
        if (end - begin > seg_len) return;   // cut row: produced from its segments
        dst = out + row * ldo;
    }
    V acc[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) acc[ch] = vzero<VEC>();
    spmm_accumulate<VEC, LPR, CH, U>(colidx, vals, begin, end, E, ldE, d, acc, dr, 0, row);
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
                                                          float *__restrict__ out, int64_t ldo, EdgeDrop dr_in)
{
    const EdgeDropR dr = resolve_drop(dr_in);
    const int64_t slice = blockIdx.x / row_blocks;
    const int64_t row = row_begin + ((int64_t)blockIdx.x % row_blocks) * 4 + (threadIdx.x >> 6);
    if (row >= row_end) return;
    const int64_t begin = rowptr[row], end = rowptr[row + 1];
    if (end - begin > seg_len) return;   // cut row: produced from its segments
    float4 acc[1];
    acc[0] = vzero4();
    spmm_accumulate<4, 8, 1, U>(colidx, vals, begin, end, E + slice * 32, ldE, 32, acc, dr, 0, row);
    spmm_store<4, 8, 1>(acc, out + row * ldo + slice * 32, 32);
}


// Rows that gather from a table of at most kLdsTableRows rows (a graph with a few hundred items: the user rows of the Seoul
// data gather from ~100 item rows): the 64-float slice of the whole table is staged in LDS once per workgroup and every
// gather is a ds_read_b128 - 128 B/clk per CU instead of the ~33 B/clk a CU gets from L2 for gathers that miss its L1.
// Grid (row blocks, slices); 8 waves share one copy of the table slice (three workgroups per CU at ~70 registers), a wave takes
// every 8th row of its workgroup's block.
constexpr int kLdsTabWaves = 8;
template <int U>
__global__ __launch_bounds__(kLdsTabWaves * 64) void spmm_ldstab_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                          const float *__restrict__ vals, int64_t row_begin, int64_t row_end,
                                                          int rows_per_wg, int seg_len, const float *__restrict__ E, int64_t ldE,
                                                          int d, int col_lo, int n_tab, float *__restrict__ out, int64_t ldo,
                                                          EdgeDrop dr_in)
{
    const EdgeDropR dr = resolve_drop(dr_in);
    extern __shared__ float4 tab4[];               // [n_tab + 1][16]: one 64-float slice of table rows col_lo .. col_lo + n_tab - 1, then a row of zeros
    const int slice = blockIdx.y;
    const int w = d - slice * 64 < 64 ? d - slice * 64 : 64;      // width of this slice (a multiple of 4)
    for (int i = threadIdx.x; i < (n_tab + 1) * 16; i += kLdsTabWaves * 64) {
        const int r = i >> 4, q = i & 15;
        float4 v = vzero4();
        if (q * 4 < w && r < n_tab) v = *reinterpret_cast<const float4 *>(E + (int64_t)(col_lo + r) * ldE + slice * 64 + q * 4);
        tab4[i] = v;
    }
    __syncthreads();
    const float *tab = reinterpret_cast<const float *>(tab4) - (int64_t)col_lo * 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r0 = row_begin + (int64_t)blockIdx.x * rows_per_wg;
    const int64_t r1 = r0 + rows_per_wg < row_end ? r0 + rows_per_wg : row_end;
    // this wave's rows r0 + wave + 8 k, k < K <= 64: lane k reads the bounds of row k (one round trip for all of them)
    const int K = r0 + wave < r1 ? (int)((r1 - r0 - wave + kLdsTabWaves - 1) / kLdsTabWaves) : 0;
    long long rb = 0, re = 0;
    if (lane < K) {
        const int64_t row = r0 + wave + (int64_t)lane * kLdsTabWaves;
        rb = rowptr[row];
        re = rowptr[row + 1];
    }
    auto bcast64 = [](long long x, int k) {       // lane k's 64-bit value as two 32-bit shuffles (the library's 64-bit __shfl goes through the stack)
        const unsigned lo = (unsigned)__shfl((int)(unsigned)(x & 0xffffffffll), k), hi = (unsigned)__shfl((int)(x >> 32), k);
        return (long long)(((unsigned long long)hi << 32) | lo);
    };
    auto store = [&](int k, const float4 &res) {
        if (lane < 16 && lane * 4 < w)
            *reinterpret_cast<float4 *>(out + (r0 + wave + (int64_t)k * kLdsTabWaves) * ldo + slice * 64 + lane * 4) = res;
    };
    if (dr.n > 0) {        // edge dropout: the general walk (hash + in-wave compaction per batch of 64 entries)
        for (int k = 0; k < K; ++k) {
            const int64_t begin = bcast64(rb, k), end = bcast64(re, k);
            if (end - begin > seg_len) continue;       // cut row: produced from its segments
            float4 acc[1];
            acc[0] = vzero4();
            spmm_accumulate<4, 16, 1, U>(colidx, vals, begin, end, tab, 64, 64, acc, dr, col_lo, r0 + wave + (int64_t)k * kLdsTabWaves);
            store(k, acc[0]);
        }
        return;
    }
    // Without dropout the walk is pipelined: the first 128 entries of row k+1 are requested before row k is multiplied, in two
    // fixed register sets (a copy of a register that is still being loaded would be a wait), so a wave pays the memory round
    // trip once, not once per row and batch.  Inside a batch of 64 entries lane (g, p) holds entry 4 p + g, so the four entries
    // of round u sit in lane u of the four 16-lane rows and reach their row with DPP broadcasts (the shuffles of the general
    // walk are LDS instructions and would compete with the table reads); slots past the end point at the zero row.
    const int g = lane >> 4, pl = lane & 15;
    const int held = pl * 4 + g;
    const char *tab_lane = reinterpret_cast<const char *>(tab4) + pl * 16;
    const int zero_off = n_tab * 256;
    struct Batch {
        int c0, c1, n;
        float v0, v1;
        int64_t begin;
    };
    auto fetch = [&](int k, Batch &t) {
        t.begin = bcast64(rb, k);
        const int64_t len = bcast64(re, k) - t.begin;
        t.n = __builtin_amdgcn_readfirstlane(len > seg_len ? -1 : (int)len);   // -1: cut row, produced from its segments
        t.c0 = t.c1 = col_lo;
        t.v0 = t.v1 = 0.f;
        if (held < t.n) {
            t.c0 = colidx[t.begin + held];
            t.v0 = vals[t.begin + held];
        }
        if (held + 64 < t.n) {
            t.c1 = colidx[t.begin + 64 + held];
            t.v1 = vals[t.begin + 64 + held];
        }
    };
    // one batch: `cnt` entries (1..64), this lane's at index `held`
    auto consume = [&](int c, float v, int cnt, float4 &acc) {
        cnt = __builtin_amdgcn_readfirstlane(cnt);         // the same in every lane: let the compiler know
        const int off = held < cnt ? (c - col_lo) * 256 : zero_off;
        v = held < cnt ? v : 0.f;
        // four rounds (16 entries) per step: the four table reads are issued together; the step condition is wave-uniform
#define NGCF_TAB_QUAD(u0, u1, u2, u3)                                                                   \
    if (4 * u0 < cnt) {                                                                                 \
        const float4 x0 = *reinterpret_cast<const float4 *>(tab_lane + row_bcast<u0>(off));            \
        const float4 x1 = *reinterpret_cast<const float4 *>(tab_lane + row_bcast<u1>(off));            \
        const float4 x2 = *reinterpret_cast<const float4 *>(tab_lane + row_bcast<u2>(off));            \
        const float4 x3 = *reinterpret_cast<const float4 *>(tab_lane + row_bcast<u3>(off));            \
        acc = vfma(row_bcast<u0>(v), x0, acc);                                                          \
        acc = vfma(row_bcast<u1>(v), x1, acc);                                                          \
        acc = vfma(row_bcast<u2>(v), x2, acc);                                                          \
        acc = vfma(row_bcast<u3>(v), x3, acc);                                                          \
    }
        NGCF_TAB_QUAD(0, 1, 2, 3) NGCF_TAB_QUAD(4, 5, 6, 7) NGCF_TAB_QUAD(8, 9, 10, 11) NGCF_TAB_QUAD(12, 13, 14, 15)
#undef NGCF_TAB_QUAD
    };
    auto multiply = [&](int k, const Batch &t) {
        if (t.n < 0) return;
        float4 acc[1];
        acc[0] = vzero4();
        if (t.n > 0) consume(t.c0, t.v0, t.n < 64 ? t.n : 64, acc[0]);
        if (t.n > 64) consume(t.c1, t.v1, t.n < 128 ? t.n - 64 : 64, acc[0]);
        for (int done = 128; done < t.n; done += 64) {     // rows of more than 128 entries: the rest batch by batch
            const int cnt = t.n - done < 64 ? t.n - done : 64;
            int c = col_lo;
            float v = 0.f;
            if (held < cnt) {
                c = colidx[t.begin + done + held];
                v = vals[t.begin + done + held];
            }
            consume(c, v, cnt, acc[0]);
        }
        spmm_combine<4, 16, 1>(acc);
        store(k, acc[0]);
    };
    Batch A, B;
    if (K > 0) fetch(0, A);
    for (int k = 0; k < K; k += 2) {
        if (k + 1 < K) fetch(k + 1, B);
        multiply(k, A);
        if (k + 1 < K) {
            if (k + 2 < K) fetch(k + 2, A);
            multiply(k + 1, B);
        }
    }
}

// Compact side table of a 1..4-column tail panel: T[c] = E[c, 0..tail) as ONE small row per table row - TW = 1, 2 or 4 floats
// (4, 8 or 16 bytes: 65 -> 1 column beyond 64, 130 -> 2 beyond 128, 515 -> 3 beyond 512).  The widths the reference forces on the
// first layer (NGCF.py:39-43) leave 1..3 columns beyond the wide panel; gathered from the strided table each of them costs a
// 64-byte sector per stored entry (100 M isolated sectors on C3, ~2 ms), from this table one TW*4-byte load per entry, 64 entries
// per wave instruction.  r03: the table is as narrow as the tail allows (r02: always 16-byte rows) - at 130 the user-side table
// is 8.8 MB instead of 17.6 MB, 16 table rows per 128-byte line instead of 8, and more of it stays in the 4 MB L2 of an XCD.
template <int TW> __device__ inline typename VecT<TW>::type tail_load(const float *e, int tail);
template <> __device__ inline float tail_load<1>(const float *e, int) { return e[0]; }
template <> __device__ inline float2 tail_load<2>(const float *e, int tail) { return make_float2(e[0], tail > 1 ? e[1] : 0.f); }
template <> __device__ inline float4 tail_load<4>(const float *e, int tail)
{
    return make_float4(e[0], tail > 1 ? e[1] : 0.f, tail > 2 ? e[2] : 0.f, tail > 3 ? e[3] : 0.f);
}
__device__ inline void tail_store(float *o, float t, int) { o[0] = t; }
__device__ inline void tail_store(float *o, float2 t, int tail)
{
    o[0] = t.x;
    if (tail > 1) o[1] = t.y;
}
__device__ inline void tail_store(float *o, float4 t, int tail)
{
    o[0] = t.x;
    if (tail > 1) o[1] = t.y;
    if (tail > 2) o[2] = t.z;
    if (tail > 3) o[3] = t.w;
}

template <int TW>
__global__ void tail_pack_kernel(const float *__restrict__ E, int64_t ldE, int64_t n, int tail, typename VecT<TW>::type *__restrict__ T)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        T[r] = tail_load<TW>(E + r * ldE, tail);
}

// the tail product lands in a compact [n_rows, TW] block and is copied into its columns of `out` (which may be a column
// slice of a wider matrix: nothing beyond the `tail` columns is written)
template <int TW>
__global__ void tail_unpack_kernel(const typename VecT<TW>::type *__restrict__ T, int64_t n, int tail, float *__restrict__ out, int64_t ldo)
{
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x)
        tail_store(out + r * ldo, T[r], tail);
}

// The tail product itself: [n_rows, TW] = L . T[n_cols, TW].  16 lanes per row (or per <= seg_len-entry segment of a cut row),
// four of them per wave, four independent (col, val, gather) triples in flight per lane: a 50-entry user row is one pass of
// its group.  One wave per row (spmm_kernel<4,1,1,1>) is latency-bound at ~0.5 us per row: 1.0 ms per product on C3, this
// form 0.3 ms.
// DROP (r04): device-side node dropout - an entry counts iff edge_keep(row, column) says so, the same test the row-wise and the
// swept kernels apply (the training step's 130-wide products ran their two tail columns through the row-wise VEC = 1 kernel:
// 1.55 ms per product at C3 against 0.8 here).
template <int TW, bool DROP>
__global__ __launch_bounds__(256) void spmm_tail_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                        const float *__restrict__ vals, int64_t n_rows,
                                                        const int32_t *__restrict__ seg_row, const int64_t *__restrict__ seg_begin,
                                                        int64_t n_seg, int64_t seg_blocks, int seg_len,
                                                        const typename VecT<TW>::type *__restrict__ T,
                                                        typename VecT<TW>::type *__restrict__ out, typename VecT<TW>::type *__restrict__ partial,
                                                        EdgeDrop dr_in)
{
    using V = typename VecT<TW>::type;
    const int unit = threadIdx.x >> 4, l = threadIdx.x & 15;
    int64_t begin = 0, end = 0, my_row = 0;
    V *dst = nullptr;
    EdgeDropR dr;
    if (DROP) dr = resolve_drop(dr_in);
    if ((int64_t)blockIdx.x < seg_blocks) {
        const int64_t s = (int64_t)blockIdx.x * 16 + unit;
        if (s < n_seg) {
            begin = seg_begin[s];
            my_row = seg_row[s];
            const int64_t row_end = rowptr[seg_row[s] + 1];
            end = begin + seg_len < row_end ? begin + seg_len : row_end;
            dst = partial + s;
        }
    } else {
        const int64_t row = ((int64_t)blockIdx.x - seg_blocks) * 16 + unit;
        if (row < n_rows) {
            begin = rowptr[row];
            end = rowptr[row + 1];
            my_row = row;
            dst = out + row;
            if (end - begin > seg_len) dst = nullptr, end = begin;   // cut row: produced from its segments
        }
    }
    V a0 = vzero<TW>(), a1 = vzero<TW>(), a2 = vzero<TW>(), a3 = vzero<TW>();
    for (int64_t e = begin + l; e < end; e += 64) {
        const int64_t e1 = e + 16, e2 = e + 32, e3 = e + 48;
        const bool k1 = e1 < end, k2 = e2 < end, k3 = e3 < end;
        const int c0 = colidx[e], c1 = k1 ? colidx[e1] : 0, c2 = k2 ? colidx[e2] : 0, c3 = k3 ? colidx[e3] : 0;
        float v0 = vals[e], v1 = k1 ? vals[e1] : 0.f, v2 = k2 ? vals[e2] : 0.f, v3 = k3 ? vals[e3] : 0.f;
        if (DROP) {
            v0 = edge_keep(dr, my_row, c0) ? v0 : 0.f;
            v1 = edge_keep(dr, my_row, c1) ? v1 : 0.f;
            v2 = edge_keep(dr, my_row, c2) ? v2 : 0.f;
            v3 = edge_keep(dr, my_row, c3) ? v3 : 0.f;
        }
        const V t0 = T[c0], t1 = T[c1], t2 = T[c2], t3 = T[c3];
        a0 = vfma(v0, t0, a0);
        a1 = vfma(v1, t1, a1);
        a2 = vfma(v2, t2, a2);
        a3 = vfma(v3, t3, a3);
    }
    V acc = vadd(vadd(a0, a1), vadd(a2, a3));
#pragma unroll
    for (int m = 8; m >= 1; m >>= 1) acc = vadd(acc, vshfl_xor(acc, m));
    if (l == 0 && dst) *dst = acc;
}

// one tail product through its compact tables (workspace: [partial sums ... | T_in [n_cols] | T_out [n_rows]] as 16-byte slots)
template <int TW>
static int launch_tail(const ngcf_csr *c, const float *E, int64_t ldE, int tail, float *out, int64_t ldo, void *workspace,
                       int64_t workspace_bytes, hipStream_t stream, const EdgeDrop &dr)
{
    using V = typename VecT<TW>::type;
    uintptr_t end = reinterpret_cast<uintptr_t>(workspace) + (uintptr_t)workspace_bytes;
    V *Tout = reinterpret_cast<V *>((end - (uintptr_t)(c->n_rows * (int64_t)sizeof(float4))) & ~(uintptr_t)255);
    V *Tin = reinterpret_cast<V *>((reinterpret_cast<uintptr_t>(Tout) - (uintptr_t)(c->n_cols * (int64_t)sizeof(float4))) & ~(uintptr_t)255);
    tail_pack_kernel<TW><<<grid_for(c->n_cols, 256), 256, 0, stream>>>(E, ldE, c->n_cols, tail, Tin);
    LAUNCH_CHECK();
    // cut rows (the CSR's own segment plan): one partial V per segment at the start of the workspace, then the fix-up
    V *tpart = reinterpret_cast<V *>(align_up((int64_t)(uintptr_t)workspace, 256));
    const int64_t seg_blocks = (c->n_seg + 15) / 16, row_blocks = (c->n_rows + 15) / 16;
    if (seg_blocks + row_blocks >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "spmm: too many rows for one launch");
    prof_mark(stream, 0);
    if (dr.n > 0)
        spmm_tail_kernel<TW, true><<<dim3((unsigned)(seg_blocks + row_blocks)), 256, 0, stream>>>(
            c->rowptr, c->colidx, c->vals, c->n_rows, c->seg_row, c->seg_begin, c->n_seg, seg_blocks, c->seg_len, Tin, Tout, tpart, dr);
    else
        spmm_tail_kernel<TW, false><<<dim3((unsigned)(seg_blocks + row_blocks)), 256, 0, stream>>>(
            c->rowptr, c->colidx, c->vals, c->n_rows, c->seg_row, c->seg_begin, c->n_seg, seg_blocks, c->seg_len, Tin, Tout, tpart, dr);
    LAUNCH_CHECK();
    if (c->n_heavy > 0) {
        spmm_fixup_kernel<TW><<<dim3((unsigned)((c->n_heavy + 3) / 4)), 256, 0, stream>>>(
            c->heavy_row, c->heavy_seg_ptr, c->n_heavy, reinterpret_cast<const float *>(tpart), TW, TW, reinterpret_cast<float *>(Tout), TW);
        LAUNCH_CHECK();
    }
    prof_mark(stream, 1);
    tail_unpack_kernel<TW><<<grid_for(c->n_rows, 256), 256, 0, stream>>>(Tout, c->n_rows, tail, out, ldo);
    LAUNCH_CHECK();
    return NGCF_OK;
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

// While the caller's stream is being CAPTURED into a hipGraph (GraphedForward on the launch-bound Seoul-sized graphs), the two
// halves of a row-wise product - the segments of the long rows with their fix-up, and the kernels of the short rows - are
// recorded as parallel branches (fork / join through a second stream): they write disjoint rows and are bound by different
// things (L2 gathers / LDS).  Outside a capture nothing changes: in an eager step the extra event calls would cost more host
// time than the overlap saves.  The stream and events are created on an eager call (GraphedForward warms up before capturing).
struct ForkState {
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int device = -1;
};
static ForkState g_forks[kMaxDevices];          // one side stream + event pair per device

static bool fork_ready(hipStream_t stream)
{
    if (ngcf_opts().no_fork) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    ForkState &g_fork = g_forks[dev % kMaxDevices];
    if (st != hipStreamCaptureStatusActive) {
        if (!g_fork.side) {
            ForkState f;
            if (hipStreamCreateWithFlags(&f.side, hipStreamNonBlocking) == hipSuccess &&
                hipEventCreateWithFlags(&f.ev_fork, hipEventDisableTiming) == hipSuccess &&
                hipEventCreateWithFlags(&f.ev_join, hipEventDisableTiming) == hipSuccess) {
                f.device = dev;
                g_fork = f;
            } else {
                (void)hipGetLastError();
            }
        }
        return false;
    }
    return g_fork.side != nullptr && g_fork.device == dev;
}

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
    const bool can_slice = VEC == 4 && a.d % 32 == 0 && a.d >= 64 && c->mode != 1 && !ngcf_opts().no_slicing;
    // table-in-LDS kernel for the groups that gather from a few hundred rows: 16-byte pieces of 64-float slices
    const bool can_ldstab = VEC == 4 && a.d % 4 == 0 && c->mode != 1 && !ngcf_opts().no_ldstab;
    bool seg_done = seg_blocks == 0;
    // under capture: segments (+ fix-up at the end) on the caller's stream, every group kernel on the side stream
    // (only where the halves are long enough to be worth a dependency edge - tools/fork_lab.py, Seoul-shaped forward as a hipGraph,
    // fork / one branch: d = 64 0.127 / 0.112 ms, 128 0.175 / 0.164, 256 0.247 / 0.312, 384 0.319 / 0.336, 512 0.370 / 0.395)
    const int64_t fork_min = ngcf_opts().fork_min;   // lab knob (tools/fork_lab.py)
    const bool fork = seg_blocks > 0 && !a.with_swept && !c->groups.empty() && c->nnz * (int64_t)a.d >= fork_min &&
                      fork_ready(a.stream);
    hipStream_t gs = a.stream;
    ForkState &g_fork = g_forks[current_device_slot()];
    if (fork) {
        HIP_TRY(hipEventRecord(g_fork.ev_fork, a.stream));
        HIP_TRY(hipStreamWaitEvent(g_fork.side, g_fork.ev_fork, 0));
        gs = g_fork.side;
        spmm_kernel<VEC, LPR, CH, U><<<dim3((unsigned)seg_blocks), 256, 0, a.stream>>>(
            c->rowptr, c->colidx, c->vals, 0, 0, seg_row, seg_begin, n_seg, seg_blocks, c->seg_len, a.E, a.ldE, a.d, a.out, a.ldo,
            a.partial, a.dp, a.dr);
        LAUNCH_CHECK();
        seg_done = true;
    }
    for (size_t g = 0; g <= c->groups.size(); ++g) {
        const bool last = g == c->groups.size();
        if (last && seg_done) break;
        if (!last && a.with_swept && c->swept.group_swept[g]) continue;
        if (!last && can_ldstab && c->groups[g].lds_table && c->groups[g].col_hi >= c->groups[g].col_lo) {
            const auto &grp = c->groups[g];
            const int64_t n_rows_g = grp.end - grp.begin;
            const int n_tab = grp.col_hi - grp.col_lo + 1, n_slices = (a.d + 63) / 64;
            // rows per workgroup: three workgroups per CU when the group is large enough, at least one row per wave
            int64_t rpw = (n_rows_g * n_slices + 767) / 768;
            rpw = std::min<int64_t>(std::max<int64_t>((rpw + kLdsTabWaves - 1) / kLdsTabWaves * kLdsTabWaves, kLdsTabWaves), 64 * kLdsTabWaves);
            const int64_t rb = (n_rows_g + rpw - 1) / rpw;
            const size_t lds = ((size_t)n_tab + 1) * 256;
            static bool attr_set[kMaxDevices] = {};      // the attribute is per device
            const int dev_i = current_device_slot();
            if (!attr_set[dev_i]) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(spmm_ldstab_kernel<4>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (kLdsTableRows + 1) * 256));
                attr_set[dev_i] = true;
            }
            spmm_ldstab_kernel<4><<<dim3((unsigned)rb, (unsigned)n_slices), kLdsTabWaves * 64, lds, gs>>>(
                c->rowptr, c->colidx, c->vals, grp.begin, grp.end, (int)rpw, c->seg_len, a.E, a.ldE, a.d, grp.col_lo, n_tab, a.out,
                a.ldo, a.dr);
            LAUNCH_CHECK();
            continue;
        }
        // (a group of a few hundred rows is not worth a launch of its own: it rides with the segments below)
        if (!last && can_slice && c->groups[g].sliceable && c->groups[g].end - c->groups[g].begin >= 1024) {
            const int64_t rb = (c->groups[g].end - c->groups[g].begin + 3) / 4;
            const int64_t blocks = rb * (a.d / 32);
            if (blocks >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "spmm: too many rows for one launch");
            spmm_sliced_kernel<8><<<dim3((unsigned)blocks), 256, 0, gs>>>(c->rowptr, c->colidx, c->vals, c->groups[g].begin,
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
            spmm_kernel<VEC, LPR, CH, U><<<dim3((unsigned)blocks), 256, 0, gs>>>(
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
    if (fork) {     // join: whatever follows on the caller's stream also follows the group kernels
        HIP_TRY(hipEventRecord(g_fork.ev_join, g_fork.side));
        HIP_TRY(hipStreamWaitEvent(a.stream, g_fork.ev_join, 0));
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
    // wider than one wave covers: column panels of 512.  Up to 768 aligned columns still run in one piece (three float4 per
    // lane): Seoul's 515-wide first layer (BASELINE configs[1]) is 516 padded columns, and a 4-column panel of its own is four
    // more launches on a launch-bound graph
    const bool one_piece = d <= 768 && d % 4 == 0 && ldE % 4 == 0 && ldo % 4 == 0 && aligned16(E) && aligned16(out);
    if (d > 512 && !one_piece) {
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
    // more than the scalar loads on the row-wise kernels, so there those products stay in one piece - they are split
    // only when the main panel then runs on the swept kernel)
    if (rows_aligned && d > 4 && !ngcf_opts().no_panel_split) {
        int main = 0;
        if (d % 64 != 0 && d > 64 && swept_usable(c, ldE, d & ~63)) main = d & ~63;                // swept kernel + tail
        else if (d % 4 != 0 && dr.n == 0) main = d & ~3;                                           // float4 kernel + 1..3 columns
        if (main > 0) {
            const int rc = spmm_dispatch(c, E, ldE, main, out, ldo, workspace, workspace_bytes, stream, dr);
            if (rc != NGCF_OK) return rc;
            const int tail = d - main;
            if (tail <= 4 && workspace && workspace_bytes >= ngcf_spmm_workspace_bytes(c, d) && !ngcf_opts().no_tail_table) {
                if (tail == 1) return launch_tail<1>(c, E + main, ldE, tail, out + main, ldo, workspace, workspace_bytes, stream, dr);
                if (tail == 2) return launch_tail<2>(c, E + main, ldE, tail, out + main, ldo, workspace, workspace_bytes, stream, dr);
                return launch_tail<4>(c, E + main, ldE, tail, out + main, ldo, workspace, workspace_bytes, stream, dr);
            }
            return spmm_dispatch(c, E + main, ldE, tail, out + main, ldo, workspace, workspace_bytes, stream, dr);
        }
    }
    const int dp = (int)align_up(d, 4);
    const bool vec = (d % 4 == 0) && rows_aligned;
    const bool with_swept = vec && swept_usable(c, ldE, d);               // (under edge dropout: the DROP instantiation)
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
    if (with_swept) rc = launch_swept(c, E, ldE, d, out, ldo, partial, dp, stream, dr);
    if (rc != NGCF_OK) return rc;
    if (vec) {
        const int nq = d / 4;
        if (nq <= 8) rc = launch_spmm<4, 8, 1, 4>(a);
        else if (nq <= 16) rc = launch_spmm<4, 16, 1, 8>(a);
        else if (nq <= 32) rc = launch_spmm<4, 32, 1, 8>(a);
        else if (nq <= 64) rc = launch_spmm<4, 64, 1, 8>(a);
        else if (nq <= 128) rc = launch_spmm<4, 64, 2, 4>(a);
        else rc = launch_spmm<4, 64, 3, 2>(a);
    } else if (d <= 8) rc = launch_spmm<1, 8, 1, 8>(a);       // narrow tail panels: 8 gathered rows per wave instruction
    else if (d <= 64) rc = launch_spmm<1, 64, 1, 8>(a);
    else if (d <= 128) rc = launch_spmm<1, 64, 2, 4>(a);
    else if (d <= 256) rc = launch_spmm<1, 64, 4, 2>(a);
    else rc = launch_spmm<1, 64, 8, 1>(a);
    prof_mark(stream, 1);
    return rc;
}

// the width a product of this CSR should be run at when its output rows are padded (see ngcf_layer_fused_f32)
extern "C" int ngcf_spmm_product_width(const ngcf_csr_t *c, const float *E, int64_t ldE, int d)
{
    if (c && d % 4 != 0 && c->nnz < ((int64_t)1 << 22) && ldE % 4 == 0 && ldE >= align_up(d, 4) && aligned16(E) &&
        !ngcf_opts().no_pad_product)
        return (int)align_up(d, 4);
    return d;
}

extern "C" int ngcf_spmm_csr_f32(const ngcf_csr_t *c, const float *E, int64_t ldE, int d, float *LE, int64_t ldLE,
                                 void *workspace, int64_t workspace_bytes, void *stream)
{
    return spmm_dispatch(c, E, ldE, d, LE, ldLE, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int ngcf_spmm_csr_dropout_f32(const ngcf_csr_t *c, const float *E, int64_t ldE, int d, float *LE, int64_t ldLE,
                                         float drop_p, const uint64_t *seeds, int n_seeds, int transposed,
                                         void *workspace, int64_t workspace_bytes, void *stream)
{
    if (n_seeds < 0 || n_seeds > 4 || (n_seeds > 0 && !seeds)) return fail(NGCF_ERR_ARG, "spmm_dropout: 0..4 seeds expected");
    if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(NGCF_ERR_ARG, "spmm_dropout: drop_p=%f not in [0,1)", drop_p);
    EdgeDrop dr{drop_p > 0.f ? n_seeds : 0, (uint32_t)((double)drop_p * 4294967296.0), {0, 0, 0, 0}, transposed ? 1 : 0};
    for (int q = 0; q < n_seeds; ++q) dr.seed[q] = seeds[q];
    return spmm_dispatch(c, E, ldE, d, LE, ldLE, workspace, workspace_bytes, (hipStream_t)stream, dr);
}

