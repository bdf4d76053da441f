// libngcf_hip.so - NGCF embedding propagation for MI355X (gfx950 / CDNA4).  C ABI: include/ngcf_hip.h
//
// Kernels (all fp32, wave = 64 lanes):
//   spmm / spmm_fixup                    : row-segmented CSR SpMM  LE = L.E          (NGCF.py:130)
//   pack_weights                         : [W1^T ; W2^T] chunk-interleaved + 2*b1+b2  (NGCF.py:131-138)
//   layer_dense                          : fp32-MFMA GEMM + bias + LeakyReLU + (dropout) +
//                                          row L2-normalise, writes carry and all_E block (NGCF.py:131-146)
//   feature_inject (3 small kernels)     : NGCF.py:103-115
//   gather_rows                          : NGCF.py:151-155
//   bpr_rows / bpr_finish                : bprloss.py:15-22
// gfx950 only: no other architecture, no compatibility paths.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/ngcf_hip.h"

// ---------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(NGCF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                                   \
    } while (0)

#define LAUNCH_CHECK()                                                                         \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess)                                                                  \
            return fail(NGCF_ERR_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                   \
    } while (0)

extern "C" const char *ngcf_last_error(void) { return g_err; }
extern "C" const char *ngcf_target_arch(void) { return "gfx950"; }
extern "C" int ngcf_version(void) { return 1; }

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---------------------------------------------------------------------------------------------
// CSR object
// ---------------------------------------------------------------------------------------------
struct ngcf_csr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int64_t *rowptr = nullptr;   // device [n_rows+1]
    int32_t *colidx = nullptr;   // device [nnz]
    float *vals = nullptr;       // device [nnz]
    bool owns = false;
    // row segmentation: rows with > seg_len entries are cut into segments
    int32_t seg_len = 0;
    int64_t n_seg = 0, n_heavy = 0;
    int32_t *seg_row = nullptr;        // device [n_seg]   row of each segment
    int64_t *seg_begin = nullptr;      // device [n_seg]   first entry of each segment
    int32_t *heavy_row = nullptr;      // device [n_heavy] rows that were cut
    int64_t *heavy_seg_ptr = nullptr;  // device [n_heavy+1] their segment ranges
    // row groups: maximal runs of rows whose gathered column range is small enough that d-slicing pays
    struct RowGroup { int64_t begin, end; bool sliceable; };
    std::vector<RowGroup> groups;
    // L2-swept plan (see "swept SpMM" below); experimental, only on request
    int mode = 0;                      // 0/1 row-wise kernels, 2 swept kernel whenever the width allows
    struct Swept {
        int64_t n_owners = 0, n_entries = 0, n_partial = 0, n_heavy = 0;
        int32_t block_cols = 0, n_blocks = 0, n_rounds = 0, col_lo = 0;
        int64_t *own_ptr = nullptr;        // device [n_owners+1]   entry range of each owner
        int32_t *own_blk = nullptr;        // device [n_owners][n_blocks] end offset of each column block in that range
        uint32_t *barrier = nullptr;       // device [8*32] per-XCD arrival counters (zeroed before each launch)
        int32_t *e_col = nullptr;          // device [n_entries]    column, sorted by (column block, row) per owner
        float *e_val = nullptr;            // device [n_entries]
        uint8_t *e_row = nullptr;          // device [n_entries]    owner-local row id (0..15)
        int64_t *own_dst = nullptr;        // device [n_owners*16]  >=0 output row, <0 partial -1-p, INT64_MIN unused
        int32_t *heavy_row = nullptr;      // device [n_heavy]
        int64_t *heavy_seg_ptr = nullptr;  // device [n_heavy+1]
    } swept;
};

static const int32_t kDefaultSegLen = 512;

static void free_swept(ngcf_csr *c)
{
    ngcf_csr::Swept &w = c->swept;
    if (w.own_ptr) (void)hipFree(w.own_ptr);
    if (w.own_blk) (void)hipFree(w.own_blk);
    if (w.barrier) (void)hipFree(w.barrier);
    if (w.e_col) (void)hipFree(w.e_col);
    if (w.e_val) (void)hipFree(w.e_val);
    if (w.e_row) (void)hipFree(w.e_row);
    if (w.own_dst) (void)hipFree(w.own_dst);
    if (w.heavy_row) (void)hipFree(w.heavy_row);
    if (w.heavy_seg_ptr) (void)hipFree(w.heavy_seg_ptr);
    w = ngcf_csr::Swept();
}

static void free_plan(ngcf_csr *c)
{
    if (c->seg_row) (void)hipFree(c->seg_row);
    if (c->seg_begin) (void)hipFree(c->seg_begin);
    if (c->heavy_row) (void)hipFree(c->heavy_row);
    if (c->heavy_seg_ptr) (void)hipFree(c->heavy_seg_ptr);
    c->seg_row = nullptr;
    c->seg_begin = nullptr;
    c->heavy_row = nullptr;
    c->heavy_seg_ptr = nullptr;
    c->n_seg = c->n_heavy = 0;
}

extern "C" void ngcf_csr_free(ngcf_csr_t *c)
{
    if (!c) return;
    free_plan(c);
    free_swept(c);
    if (c->owns) {
        if (c->rowptr) (void)hipFree(c->rowptr);
        if (c->colidx) (void)hipFree(c->colidx);
        if (c->vals) (void)hipFree(c->vals);
    }
    delete c;
}

extern "C" int64_t ngcf_csr_nnz(const ngcf_csr_t *c) { return c ? c->nnz : -1; }
extern "C" int64_t ngcf_csr_n_rows(const ngcf_csr_t *c) { return c ? c->n_rows : -1; }
extern "C" int64_t ngcf_csr_n_cols(const ngcf_csr_t *c) { return c ? c->n_cols : -1; }
extern "C" int64_t ngcf_csr_n_segments(const ngcf_csr_t *c) { return c ? c->n_seg : -1; }
extern "C" const int64_t *ngcf_csr_rowptr(const ngcf_csr_t *c) { return c ? c->rowptr : nullptr; }
extern "C" const int32_t *ngcf_csr_colidx(const ngcf_csr_t *c) { return c ? c->colidx : nullptr; }
extern "C" const float *ngcf_csr_vals(const ngcf_csr_t *c) { return c ? c->vals : nullptr; }

// flags[0] |= 1 when rows are not non-decreasing; flags[1] |= 1 when an id is out of range
__global__ void coo_check_kernel(const int64_t *__restrict__ rows, const int64_t *__restrict__ cols,
                                 int64_t nnz, int64_t n_rows, int64_t n_cols, int32_t *flags)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool unsorted = false, bad = false;
    for (; i < nnz; i += stride) {
        const int64_t r = rows[i], c = cols[i];
        bad |= (r < 0) | (r >= n_rows) | (c < 0) | (c >= n_cols);
        if (i + 1 < nnz) unsorted |= rows[i + 1] < r;
    }
    if (unsorted) atomicOr(&flags[0], 1);
    if (bad) atomicOr(&flags[1], 1);
}

// rowptr[r] = first entry whose row id is >= r (rows sorted); one thread per r in [0, n_rows]
__global__ void coo_rowptr_kernel(const int64_t *__restrict__ rows, int64_t nnz, int64_t n_rows,
                                  int64_t *__restrict__ rowptr)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (rows[mid] < r)
            lo = mid + 1;
        else
            hi = mid;
    }
    rowptr[r] = lo;
}

__global__ void coo_copy_kernel(const int64_t *__restrict__ cols, const float *__restrict__ vals, int64_t nnz,
                                int32_t *__restrict__ colidx, float *__restrict__ out_vals)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) {
        colidx[i] = (int32_t)cols[i];
        out_vals[i] = vals[i];
    }
}

static int grid_for(int64_t n, int block)
{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}


// per 1024-row block: smallest and largest column any of its rows gathers (one-time, plan only)
#define NGCF_GROUP_ROWS 1024
__global__ void row_colrange_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                    int64_t n_rows, int32_t *__restrict__ blk_min, int32_t *__restrict__ blk_max)
{
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    int32_t lo = INT32_MAX, hi = -1;
    for (int64_t e = rowptr[row]; e < rowptr[row + 1]; ++e) {
        const int32_t c = colidx[e];
        lo = c < lo ? c : lo;
        hi = c > hi ? c : hi;
    }
    if (hi >= 0) {
        atomicMin(&blk_min[row / NGCF_GROUP_ROWS], lo);
        atomicMax(&blk_max[row / NGCF_GROUP_ROWS], hi);
    }
}

// A row block is "sliceable" when one 32-float slice (128 B) of every row it gathers is at most 48 MiB: then the
// hot part of the table slice lives in the XCD L2s while a slice-major launch walks it (measured on the user half
// of C3: 2.33 ms sliced vs 2.98 ms unsliced; the 1 M-row user table gets slower sliced: 3.8 vs 3.46 ms).
static const int64_t kSliceFootprintRows = (48ll << 20) / 128;

static int build_row_groups(ngcf_csr *c, hipStream_t stream)
{
    c->groups.clear();
    if (c->n_rows == 0) return NGCF_OK;
    const int64_t nb = (c->n_rows + NGCF_GROUP_ROWS - 1) / NGCF_GROUP_ROWS;
    int32_t *d_min = nullptr, *d_max = nullptr;
    std::vector<int32_t> h_min((size_t)nb), h_max((size_t)nb);
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&d_min, sizeof(int32_t) * (size_t)nb));
        HIP_TRY(hipMalloc(&d_max, sizeof(int32_t) * (size_t)nb));
        HIP_TRY(hipMemsetAsync(d_min, 0x7f, sizeof(int32_t) * (size_t)nb, stream));   // 0x7f7f7f7f: large
        HIP_TRY(hipMemsetAsync(d_max, 0xff, sizeof(int32_t) * (size_t)nb, stream));   // -1
        row_colrange_kernel<<<(int)((c->n_rows + 255) / 256), 256, 0, stream>>>(c->rowptr, c->colidx, c->n_rows, d_min, d_max);
        LAUNCH_CHECK();
        HIP_TRY(hipMemcpyAsync(h_min.data(), d_min, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(h_max.data(), d_max, sizeof(int32_t) * (size_t)nb, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        return NGCF_OK;
    };
    const int rc = body();
    if (d_min) (void)hipFree(d_min);
    if (d_max) (void)hipFree(d_max);
    if (rc != NGCF_OK) return rc;
    for (int64_t b = 0; b < nb; ++b) {
        const bool s = h_max[(size_t)b] < 0 || (int64_t)h_max[(size_t)b] - h_min[(size_t)b] + 1 <= kSliceFootprintRows;
        const int64_t lo = b * NGCF_GROUP_ROWS, hi = std::min<int64_t>(c->n_rows, lo + NGCF_GROUP_ROWS);
        if (!c->groups.empty() && c->groups.back().sliceable == s)
            c->groups.back().end = hi;
        else
            c->groups.push_back({lo, hi, s});
    }
    // a group of a few blocks is not worth its own launch: give it its neighbour's class, then fuse equal neighbours
    for (size_t i = 0; i < c->groups.size(); ++i)
        if (c->groups.size() > 1 && c->groups[i].end - c->groups[i].begin < 16 * NGCF_GROUP_ROWS)
            c->groups[i].sliceable = c->groups[i > 0 ? i - 1 : i + 1].sliceable;
    for (size_t k = 1; k < c->groups.size();) {
        if (c->groups[k].sliceable == c->groups[k - 1].sliceable) {
            c->groups[k - 1].end = c->groups[k].end;
            c->groups.erase(c->groups.begin() + (long)k);
        } else {
            ++k;
        }
    }
    return NGCF_OK;
}

static int build_swept_plan(ngcf_csr *c, hipStream_t stream);

extern "C" int ngcf_csr_plan(ngcf_csr_t *c, int32_t seg_len, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!c) return fail(NGCF_ERR_ARG, "ngcf_csr_plan: null csr");
    if (seg_len < 64) return fail(NGCF_ERR_ARG, "ngcf_csr_plan: seg_len must be >= 64 (got %d)", seg_len);
    free_plan(c);
    c->seg_len = seg_len;
    std::vector<int64_t> rp((size_t)c->n_rows + 1);
    HIP_TRY(hipMemcpyAsync(rp.data(), c->rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<int32_t> seg_row, heavy_row;
    std::vector<int64_t> seg_begin, heavy_ptr;
    heavy_ptr.push_back(0);
    for (int64_t r = 0; r < c->n_rows; ++r) {
        const int64_t len = rp[r + 1] - rp[r];
        if (len > seg_len) {
            heavy_row.push_back((int32_t)r);
            for (int64_t b = rp[r]; b < rp[r + 1]; b += seg_len) {
                seg_row.push_back((int32_t)r);
                seg_begin.push_back(b);
            }
            heavy_ptr.push_back((int64_t)seg_row.size());
        }
    }
    c->n_seg = (int64_t)seg_row.size();
    c->n_heavy = (int64_t)heavy_row.size();
    if (c->n_seg > 0) {
        HIP_TRY(hipMalloc(&c->seg_row, sizeof(int32_t) * seg_row.size()));
        HIP_TRY(hipMalloc(&c->seg_begin, sizeof(int64_t) * seg_begin.size()));
        HIP_TRY(hipMalloc(&c->heavy_row, sizeof(int32_t) * heavy_row.size()));
        HIP_TRY(hipMalloc(&c->heavy_seg_ptr, sizeof(int64_t) * heavy_ptr.size()));
        HIP_TRY(hipMemcpyAsync(c->seg_row, seg_row.data(), sizeof(int32_t) * seg_row.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c->seg_begin, seg_begin.data(), sizeof(int64_t) * seg_begin.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c->heavy_row, heavy_row.data(), sizeof(int32_t) * heavy_row.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(c->heavy_seg_ptr, heavy_ptr.data(), sizeof(int64_t) * heavy_ptr.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));
    }
    {
        const int rc = build_row_groups(c, stream);
        if (rc != NGCF_OK) return rc;
    }
    if (c->swept.n_owners == 0 && c->mode == 2) return build_swept_plan(c, stream);
    return NGCF_OK;
}

extern "C" int ngcf_csr_set_mode(ngcf_csr_t *c, int mode, void *stream)
{
    if (!c || mode < 0 || mode > 2) return fail(NGCF_ERR_ARG, "ngcf_csr_set_mode: bad argument");
    c->mode = mode;
    if (mode != 2) {
        free_swept(c);
        return NGCF_OK;
    }
    if (c->swept.n_owners == 0) return build_swept_plan(c, (hipStream_t)stream);
    return NGCF_OK;
}

extern "C" int ngcf_csr_from_arrays(const int64_t *rowptr, const int32_t *colidx, const float *vals,
                                    int64_t n_rows, int64_t n_cols, int64_t nnz, ngcf_csr_t **out, void *stream)
{
    if (!out) return fail(NGCF_ERR_ARG, "ngcf_csr_from_arrays: null out");
    *out = nullptr;
    if (!rowptr || (nnz > 0 && (!colidx || !vals)))
        return fail(NGCF_ERR_ARG, "ngcf_csr_from_arrays: null array");
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || n_cols >= (int64_t)1 << 31 || n_rows >= (int64_t)1 << 31)
        return fail(NGCF_ERR_ARG, "ngcf_csr_from_arrays: bad shape %lld x %lld nnz %lld", (long long)n_rows,
                    (long long)n_cols, (long long)nnz);
    ngcf_csr *c = new ngcf_csr();
    c->n_rows = n_rows;
    c->n_cols = n_cols;
    c->nnz = nnz;
    c->rowptr = const_cast<int64_t *>(rowptr);
    c->colidx = const_cast<int32_t *>(colidx);
    c->vals = const_cast<float *>(vals);
    c->owns = false;
    const int rc = ngcf_csr_plan(c, kDefaultSegLen, stream);
    if (rc != NGCF_OK) {
        ngcf_csr_free(c);
        return rc;
    }
    *out = c;
    return NGCF_OK;
}

extern "C" int ngcf_csr_from_coo(const int64_t *rows, const int64_t *cols, const float *vals, int64_t nnz,
                                 int64_t n_rows, int64_t n_cols, ngcf_csr_t **out, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) return fail(NGCF_ERR_ARG, "ngcf_csr_from_coo: null out");
    *out = nullptr;
    if (nnz < 0 || n_rows < 0 || n_cols < 0 || n_cols >= (int64_t)1 << 31 || n_rows >= (int64_t)1 << 31)
        return fail(NGCF_ERR_ARG, "ngcf_csr_from_coo: bad shape %lld x %lld nnz %lld", (long long)n_rows,
                    (long long)n_cols, (long long)nnz);
    if (nnz > 0 && (!rows || !cols || !vals)) return fail(NGCF_ERR_ARG, "ngcf_csr_from_coo: null array");

    ngcf_csr *c = new ngcf_csr();
    c->n_rows = n_rows;
    c->n_cols = n_cols;
    c->nnz = nnz;
    c->owns = true;
    int32_t *flags = nullptr;
    int rc = NGCF_OK;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&c->rowptr, sizeof(int64_t) * (size_t)(n_rows + 1)));
        HIP_TRY(hipMalloc(&c->colidx, sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1)));
        HIP_TRY(hipMalloc(&c->vals, sizeof(float) * (size_t)std::max<int64_t>(nnz, 1)));
        HIP_TRY(hipMalloc(&flags, 2 * sizeof(int32_t)));
        HIP_TRY(hipMemsetAsync(flags, 0, 2 * sizeof(int32_t), stream));
        int32_t h_flags[2] = {0, 0};
        if (nnz > 0) {
            coo_check_kernel<<<grid_for(nnz, 256), 256, 0, stream>>>(rows, cols, nnz, n_rows, n_cols, flags);
            LAUNCH_CHECK();
        }
        HIP_TRY(hipMemcpyAsync(h_flags, flags, sizeof(h_flags), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (h_flags[1])
            return fail(NGCF_ERR_INDEX, "ngcf_csr_from_coo: index out of range for a %lld x %lld matrix",
                        (long long)n_rows, (long long)n_cols);
        if (!h_flags[0]) {
            // rows already sorted (what matrix.py:79-83 emits): convert on the device
            coo_rowptr_kernel<<<(int)((n_rows + 1 + 255) / 256), 256, 0, stream>>>(rows, nnz, n_rows, c->rowptr);
            LAUNCH_CHECK();
            if (nnz > 0) {
                coo_copy_kernel<<<grid_for(nnz, 256), 256, 0, stream>>>(cols, vals, nnz, c->colidx, c->vals);
                LAUNCH_CHECK();
            }
        } else {
            // unsorted input: stable sort by row on the host (one-time set-up path)
            std::vector<int64_t> hr((size_t)nnz), hc((size_t)nnz);
            std::vector<float> hv((size_t)nnz);
            HIP_TRY(hipMemcpyAsync(hr.data(), rows, sizeof(int64_t) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(hc.data(), cols, sizeof(int64_t) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipMemcpyAsync(hv.data(), vals, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            std::vector<int64_t> rp((size_t)n_rows + 1, 0);
            for (int64_t i = 0; i < nnz; ++i) rp[(size_t)hr[i] + 1]++;
            for (int64_t r = 0; r < n_rows; ++r) rp[(size_t)r + 1] += rp[(size_t)r];
            std::vector<int64_t> cur(rp.begin(), rp.end() - 1);
            std::vector<int32_t> oc((size_t)nnz);
            std::vector<float> ov((size_t)nnz);
            for (int64_t i = 0; i < nnz; ++i) {   // counting sort = stable
                const int64_t dst = cur[(size_t)hr[i]]++;
                oc[(size_t)dst] = (int32_t)hc[i];
                ov[(size_t)dst] = hv[i];
            }
            HIP_TRY(hipMemcpyAsync(c->rowptr, rp.data(), sizeof(int64_t) * rp.size(), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(c->colidx, oc.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(c->vals, ov.data(), sizeof(float) * (size_t)nnz, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));
        }
        return ngcf_csr_plan(c, kDefaultSegLen, stream);
    };
    rc = body();
    if (flags) (void)hipFree(flags);
    if (rc != NGCF_OK) {
        ngcf_csr_free(c);
        return rc;
    }
    *out = c;
    return NGCF_OK;
}

// ---------------------------------------------------------------------------------------------
// SpMM  LE = L.E   (NGCF.py:130)
//
// One wave owns one row (or one <= seg_len-entry segment of a long row).  The wave first reads up
// to 64 (col, val) pairs with one coalesced load per lane, then walks them: LPR lanes cover one
// gathered row of E with one 16-byte load each (VEC = 4), so G = 64/LPR neighbour rows are fetched
// per wave-instruction and U such instructions are kept in flight.  The G lane groups keep
// private partial sums that are combined with DPP/bpermute shuffles at the end.
// VEC = 1 is the any-width / any-alignment form (Seoul's d = 65).
// ---------------------------------------------------------------------------------------------
template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<1> { using type = float; };

__device__ inline float4 vfma(float s, float4 x, float4 a)
{
    a.x = fmaf(s, x.x, a.x);
    a.y = fmaf(s, x.y, a.y);
    a.z = fmaf(s, x.z, a.z);
    a.w = fmaf(s, x.w, a.w);
    return a;
}
__device__ inline float vfma(float s, float x, float a) { return fmaf(s, x, a); }
__device__ inline float4 vsel(bool p, float4 a, float4 b) { return p ? a : b; }
__device__ inline float vsel(bool p, float a, float b) { return p ? a : b; }
__device__ inline float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <int VEC> __device__ inline typename VecT<VEC>::type vzero();
template <> __device__ inline float4 vzero<4>() { return vzero4(); }
template <> __device__ inline float vzero<1>() { return 0.f; }
__device__ inline float4 vshfl_xor(float4 a, int m)
{
    a.x = __shfl_xor(a.x, m);
    a.y = __shfl_xor(a.y, m);
    a.z = __shfl_xor(a.z, m);
    a.w = __shfl_xor(a.w, m);
    return a;
}
__device__ inline float vshfl_xor(float a, int m) { return __shfl_xor(a, m); }
__device__ inline float4 vadd(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ inline float vadd(float a, float b) { return a + b; }

// counter-based hash: the keep masks of node and message dropout are pure functions of (seed, index)
__device__ inline uint32_t mix32(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return (uint32_t)x;
}

// Device-side node dropout (NGCF.py:93-100 semantics: keep each stored entry w.p. 1-p, values NOT rescaled,
// cumulative over layers): entry e survives layer k iff mix32(seed_j ^ e*K) >= thr for every j <= k.
// `eid` maps the entries of a transposed CSR back to the entry numbers of L (NULL: the entry position itself).
struct EdgeDrop {
    int n;                  // number of seeds (0 = no dropout)
    uint32_t thr;           // p * 2^32
    uint64_t seed[4];
    const int64_t *eid;
};

template <int VEC, int LPR, int CH, int U>
__device__ inline void spmm_accumulate(const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                       int64_t begin, int64_t end, const float *__restrict__ E, int64_t ldE,
                                       int d, typename VecT<VEC>::type (&acc)[CH], const EdgeDrop &dr = EdgeDrop{0, 0, {0, 0, 0, 0}, nullptr})
{
    using V = typename VecT<VEC>::type;
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int g = lane / LPR;
    const int l = lane % LPR;
    // column offset of each chunk this lane covers; lanes past the row width read column 0 and
    // are never written back
    int coff[CH];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const int o = (l + ch * LPR) * VEC;
        coff[ch] = o < d ? o : 0;
    }
    for (int64_t base = begin; base < end; base += 64) {
        int cnt = (int)((end - base) < 64 ? (end - base) : 64);
        int c = 0;       // column 0 is always a valid row of E: padding slots read it, masked below
        float v = 0.f;
        if (lane < cnt) {
            c = colidx[base + lane];
            v = vals[base + lane];
        }
        if (dr.n > 0) {
            // drop entries, then compact the survivors to the low lanes (dropped ones go to the top, unused)
            bool keep = lane < cnt;
            if (keep) {
                const uint64_t e = (uint64_t)(dr.eid ? dr.eid[base + lane] : base + lane) * 0x9E3779B97F4A7C15ULL;
                for (int q = 0; q < dr.n; ++q) keep = keep && mix32(dr.seed[q] ^ e) >= dr.thr;
            }
            const unsigned long long m = __ballot(keep);
            const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            const int dst = keep ? __popcll(m & lt) : 63 - __popcll(~m & lt);
            c = __builtin_amdgcn_ds_permute(dst << 2, c);
            v = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v)));
            cnt = __popcll(m);
        }
        int j = 0;
        for (; j + G * U <= cnt; j += G * U) {          // full batches: every slot is a real entry
            V x[U][CH];
            float vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * G + g;
                const int cc = __shfl(c, idx);
                vv[u] = __shfl(v, idx);
                const float *src = E + (int64_t)cc * ldE;
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) x[u][ch] = *reinterpret_cast<const V *>(src + coff[ch]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) acc[ch] = vfma(vv[u], x[u][ch], acc[ch]);
        }
        if (j < cnt) {                                   // tail batch: slots past cnt are masked
            V x[U][CH];
            float vv[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = j + u * G + g;
                ok[u] = idx < cnt;
                const int cc = __shfl(c, idx & 63);
                vv[u] = __shfl(v, idx & 63);
                const float *src = E + (int64_t)cc * ldE;
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) x[u][ch] = *reinterpret_cast<const V *>(src + coff[ch]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int ch = 0; ch < CH; ++ch) acc[ch] = vsel(ok[u], vfma(vv[u], x[u][ch], acc[ch]), acc[ch]);
        }
    }
    // combine the G lane groups
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) acc[ch] = vadd(acc[ch], vshfl_xor(acc[ch], off));
}

template <int VEC, int LPR, int CH>
__device__ inline void spmm_store(typename VecT<VEC>::type (&acc)[CH], float *__restrict__ dst, int d)
{
    using V = typename VecT<VEC>::type;
    const int lane = threadIdx.x & 63;
    if (lane >= LPR) return;
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
        const int o = (lane + ch * LPR) * VEC;
        if (o < d) *reinterpret_cast<V *>(dst + o) = acc[ch];
    }
}

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

// cut rows: add their segments' partial sums in segment order (fixed order, no atomics)
template <int VEC>
__global__ __launch_bounds__(256) void spmm_fixup_kernel(const int32_t *__restrict__ heavy_row,
                                                         const int64_t *__restrict__ heavy_seg_ptr,
                                                         int64_t n_heavy, const float *__restrict__ partial,
                                                         int dp, int d, float *__restrict__ out, int64_t ldo)
{
    using V = typename VecT<VEC>::type;
    const int64_t h = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (h >= n_heavy) return;
    const int lane = threadIdx.x & 63;
    const int64_t s0 = heavy_seg_ptr[h], s1 = heavy_seg_ptr[h + 1];
    float *dst = out + (int64_t)heavy_row[h] * ldo;
    for (int o = lane * VEC; o < d; o += 64 * VEC) {
        V acc = vzero<VEC>();
        for (int64_t s = s0; s < s1; ++s) acc = vadd(acc, *reinterpret_cast<const V *>(partial + s * (int64_t)dp + o));
        *reinterpret_cast<V *>(dst + o) = acc;
    }
}


// ---------------------------------------------------------------------------------------------
// Swept SpMM: the L2-blocked form for large matrices (same product LE = L.E, NGCF.py:130).
//
// Why: on a graph without locality the row-wise kernel above misses L2 on ~9 of 10 gathered rows and runs at
// the chip's L2-miss rate (~8 TB/s of gathered bytes); rows served from L2 arrive 2-3x faster (measured with a
// synchronised sliding window, profiles/r01_window_lab.txt).  Here the gathered table is swept in column blocks
// that fit an XCD's 4 MiB L2 while every CU works on the same block, so a table row is fetched from memory
// once per XCD and re-used from L2 by the other output rows of that XCD that need it.
//
// How: a persistent grid of 256 workgroups (one per CU, 512 threads, 128 KiB of LDS).  Output rows are handed to
// "owners"; an owner is a quarter-wave (16 lanes x 16 B = one 64-float slice of a row) that keeps up to 16
// accumulator rows in LDS and walks its own edge list, which the host plan has sorted by (column block, row)
// and balanced so that every owner has about the same work in every block.  The walk is software-pipelined:
// list entries are fetched two 16-entry chunks ahead, the 16 gathers of the next chunk are issued before the
// current chunk is accumulated.  After each column block the workgroups of one XCD (HW_REG_XCC_ID) meet at a
// counter barrier whose spin is bounded: the barrier only keeps the sweep together for speed, correctness never
// depends on it (an owner touches nothing but its own LDS rows and adds in list order: deterministic result).
// Rows longer than the per-owner budget are dealt round-robin to several pieces whose partial sums are combined
// by spmm_fixup_kernel in a fixed order.  A slice is 64 floats, so d must be a multiple of 64 (other widths use
// the row-wise kernel); slices and owner rounds are walked one after the other inside the kernel.
// Status: opt-in (ngcf_csr_set_mode(csr, 2)).  On the C3 item rows it reaches 3.2 ms against 3.5-3.7 ms for the
// row-wise kernel (69 % L2 hits), still far from the 1.3 ms of a perfectly synchronised sweep: with 8 waves per
// CU the per-entry run/flush logic and the barrier imbalance dominate.
// ---------------------------------------------------------------------------------------------
static const int kSweptRPO = 16;                 // accumulator rows per owner
static const int kSweptOwnersPerWG = 32;         // 8 waves x 4 quarter-waves
static const int kSweptWGs = 256;                // one 512-thread workgroup per CU (128 KiB of LDS)
static const int kSweptGroups = 8;               // XCDs
static const int64_t kSweptUnused = INT64_MIN;

static int32_t swept_block_cols()
{
    // columns per block: block bytes / (64 floats * 4 B); default 2 MiB of table slice per block
    const char *e = getenv("NGCF_SWEPT_BLOCK_KB");
    int64_t kb = e ? atoll(e) : 2048;
    if (kb < 16) kb = 16;
    return (int32_t)std::max<int64_t>(kb * 1024 / 256, 64);
}

template <typename F>
static void parallel_for(int64_t n, F &&fn)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt > 16) nt = 16;
    if (nt < 1) nt = 1;
    if (n < 64 || nt == 1) {
        fn(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t lo = t * chunk, hi = std::min<int64_t>(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([&fn, lo, hi]() { fn(lo, hi); });
    }
    for (auto &t : th) t.join();
}

static int build_swept_plan(ngcf_csr *c, hipStream_t stream)
{
    free_swept(c);
    ngcf_csr::Swept &w = c->swept;
    const int64_t n_rows = c->n_rows, nnz = c->nnz;
    if (n_rows == 0) return NGCF_OK;
    std::vector<int64_t> rp((size_t)n_rows + 1);
    std::vector<int32_t> col((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> val((size_t)std::max<int64_t>(nnz, 1));
    HIP_TRY(hipMemcpyAsync(rp.data(), c->rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost, stream));
    if (nnz > 0) {
        HIP_TRY(hipMemcpyAsync(col.data(), c->colidx, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(val.data(), c->vals, sizeof(float) * (size_t)nnz, hipMemcpyDeviceToHost, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));

    const int64_t per_round = (int64_t)kSweptOwnersPerWG * kSweptWGs;       // 8192 owners are resident at a time
    const int64_t rounds = std::max<int64_t>(1, (n_rows + per_round * 13 - 1) / (per_round * 13));
    const int64_t target = per_round * rounds;
    int64_t T = std::max<int64_t>(64, (nnz + target - 1) / target);
    w.block_cols = swept_block_cols();

    // entries begin+off, begin+off+step, ... < end.  A long row is dealt out to its pieces round-robin so that every
    // piece covers the whole column range evenly (a contiguous cut would pile one piece's work into a few blocks)
    struct Piece { int64_t begin, end, dst, off, step; int64_t count() const { return end - begin <= off ? 0 : (end - begin - off + step - 1) / step; } };
    std::vector<Piece> pieces;
    std::vector<int32_t> heavy_row;
    std::vector<int64_t> heavy_ptr, own_first;
    int64_t n_partial = 0;
    for (int attempt = 0; attempt < 40; ++attempt) {
        // 1) pieces: a row, or a <=T-entry cut of a long row (partial sums, combined in piece order)
        pieces.clear();
        heavy_row.clear();
        heavy_ptr.assign(1, 0);
        n_partial = 0;
        for (int64_t r = 0; r < n_rows; ++r) {
            const int64_t b = rp[r], e = rp[r + 1], len = e - b;
            if (len <= T) {
                pieces.push_back({b, e, r, 0, 1});
            } else {
                const int64_t k = (len + T - 1) / T;
                heavy_row.push_back((int32_t)r);
                for (int64_t j = 0; j < k; ++j) pieces.push_back({b, e, -1 - n_partial++, j, k});
                heavy_ptr.push_back(n_partial);
            }
        }
        // 2) owners: consecutive pieces until the entry budget or kSweptRPO rows are reached
        own_first.assign(1, 0);
        int64_t edges = 0;
        int rows_in = 0;
        for (size_t i = 0; i < pieces.size(); ++i) {
            const int64_t len = pieces[i].count();
            if (rows_in == kSweptRPO || (rows_in > 0 && edges + len > T)) {
                own_first.push_back((int64_t)i);
                edges = 0;
                rows_in = 0;
            }
            edges += len;
            ++rows_in;
        }
        own_first.push_back((int64_t)pieces.size());
        if ((int64_t)own_first.size() - 1 <= target) break;
        T += std::max<int64_t>(1, T / 16);          // too many owners for the resident grid: raise the budget
    }
    const int64_t n_owners = (int64_t)own_first.size() - 1;
    const int64_t n_owners_pad = align_up(n_owners, per_round);
    int32_t col_lo = INT32_MAX, col_hi = 0;
    for (int64_t x = 0; x < nnz; ++x) {
        col_lo = std::min(col_lo, col[x]);
        col_hi = std::max(col_hi, col[x]);
    }
    if (nnz == 0) col_lo = 0;
    w.col_lo = col_lo;
    const int64_t n_blocks = std::max<int64_t>(1, ((int64_t)col_hi - col_lo + w.block_cols) / w.block_cols);
    std::vector<int64_t> own_ptr((size_t)n_owners_pad + 1, 0);
    std::vector<int64_t> own_dst((size_t)n_owners_pad * kSweptRPO, kSweptUnused);
    for (int64_t o = 0; o < n_owners; ++o) {
        int64_t cnt = 0;
        for (int64_t i = own_first[o]; i < own_first[o + 1]; ++i) {
            cnt += pieces[i].count();
            own_dst[(size_t)(o * kSweptRPO + (i - own_first[o]))] = pieces[i].dst;
        }
        if (cnt >= (int64_t)1 << 31) return fail(NGCF_ERR_ARG, "swept plan: owner list too long");
        own_ptr[(size_t)o + 1] = own_ptr[(size_t)o] + cnt;
    }
    for (int64_t o = n_owners; o < n_owners_pad; ++o) own_ptr[(size_t)o + 1] = own_ptr[(size_t)o];
    // 3) per owner: stable counting sort of its entries by (column block, local row); block offsets kept
    std::vector<int32_t> e_col((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> e_val((size_t)std::max<int64_t>(nnz, 1));
    std::vector<uint8_t> e_row((size_t)std::max<int64_t>(nnz, 1));
    const int32_t bc = w.block_cols;
    std::vector<int32_t> own_blk((size_t)n_owners_pad * (size_t)n_blocks, 0);
    parallel_for(n_owners, [&](int64_t lo, int64_t hi) {
        std::vector<int64_t> hist((size_t)n_blocks * kSweptRPO + 1);
        for (int64_t o = lo; o < hi; ++o) {
            std::fill(hist.begin(), hist.end(), 0);
            for (int64_t i = own_first[o]; i < own_first[o + 1]; ++i) {
                const int lr = (int)(i - own_first[o]);
                for (int64_t x = pieces[i].begin + pieces[i].off; x < pieces[i].end; x += pieces[i].step)
                    hist[(size_t)((col[x] - col_lo) / bc) * kSweptRPO + lr + 1]++;
            }
            for (size_t k = 1; k < hist.size(); ++k) hist[k] += hist[k - 1];
            for (int64_t bq = 0; bq < n_blocks; ++bq)      // where block bq ends inside this owner's list
                own_blk[(size_t)o * (size_t)n_blocks + (size_t)bq] = (int32_t)hist[(size_t)(bq + 1) * kSweptRPO];
            const int64_t base = own_ptr[(size_t)o];
            for (int64_t i = own_first[o]; i < own_first[o + 1]; ++i) {
                const int lr = (int)(i - own_first[o]);
                for (int64_t x = pieces[i].begin + pieces[i].off; x < pieces[i].end; x += pieces[i].step) {
                    const int64_t pos = base + hist[(size_t)((col[x] - col_lo) / bc) * kSweptRPO + lr]++;
                    e_col[(size_t)pos] = col[x];
                    e_val[(size_t)pos] = val[x];
                    e_row[(size_t)pos] = (uint8_t)lr;
                }
            }
        }
    });
    if (getenv("NGCF_SWEPT_DEBUG")) {
        int64_t mx = 0, nz = 0;
        for (int64_t o = 0; o < n_owners; ++o) {
            mx = std::max(mx, own_ptr[(size_t)o + 1] - own_ptr[(size_t)o]);
            nz += own_ptr[(size_t)o + 1] > own_ptr[(size_t)o];
        }
        fprintf(stderr, "[swept plan] rows %lld nnz %lld owners %lld (pad %lld, non-empty %lld) rounds %lld T %lld max/owner %lld "
                        "pieces %zu partial %lld blocks %lld x %d cols\n", (long long)n_rows, (long long)nnz, (long long)n_owners,
                (long long)n_owners_pad, (long long)nz, (long long)(n_owners_pad / per_round), (long long)T, (long long)mx,
                pieces.size(), (long long)n_partial, (long long)n_blocks, (int)bc);
    }
    // 4) upload
    w.n_owners = n_owners_pad;
    w.n_rounds = (int32_t)(n_owners_pad / per_round);
    w.n_blocks = (int32_t)n_blocks;
    w.n_entries = nnz;
    w.n_partial = n_partial;
    w.n_heavy = (int64_t)heavy_row.size();
    HIP_TRY(hipMalloc(&w.own_ptr, sizeof(int64_t) * own_ptr.size()));
    HIP_TRY(hipMalloc(&w.own_dst, sizeof(int64_t) * own_dst.size()));
    HIP_TRY(hipMalloc(&w.own_blk, sizeof(int32_t) * own_blk.size()));
    HIP_TRY(hipMalloc(&w.barrier, sizeof(uint32_t) * 32 * kSweptGroups));
    HIP_TRY(hipMemcpyAsync(w.own_blk, own_blk.data(), sizeof(int32_t) * own_blk.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMalloc(&w.e_col, sizeof(int32_t) * e_col.size()));
    HIP_TRY(hipMalloc(&w.e_val, sizeof(float) * e_val.size()));
    HIP_TRY(hipMalloc(&w.e_row, e_row.size()));
    HIP_TRY(hipMemcpyAsync(w.own_ptr, own_ptr.data(), sizeof(int64_t) * own_ptr.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.own_dst, own_dst.data(), sizeof(int64_t) * own_dst.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.e_col, e_col.data(), sizeof(int32_t) * e_col.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.e_val, e_val.data(), sizeof(float) * e_val.size(), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(w.e_row, e_row.data(), e_row.size(), hipMemcpyHostToDevice, stream));
    if (w.n_heavy > 0) {
        HIP_TRY(hipMalloc(&w.heavy_row, sizeof(int32_t) * heavy_row.size()));
        HIP_TRY(hipMalloc(&w.heavy_seg_ptr, sizeof(int64_t) * heavy_ptr.size()));
        HIP_TRY(hipMemcpyAsync(w.heavy_row, heavy_row.data(), sizeof(int32_t) * heavy_row.size(), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(w.heavy_seg_ptr, heavy_ptr.data(), sizeof(int64_t) * heavy_ptr.size(), hipMemcpyHostToDevice, stream));
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return NGCF_OK;
}

// value of lane U of this lane's 16-lane row (DPP row_newbcast: one VALU op, no LDS round trip)
template <int U> __device__ inline int row_bcast(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + U, 0xf, 0xf, false);
}
template <int U> __device__ inline float row_bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + U, 0xf, 0xf, false));
}

// add a 16-lane x float4 accumulator into the owner's LDS row (plain read-modify-write: only this quarter-wave
// ever touches the row; LDS float atomics were measured slower and erratic here)
__device__ inline void swept_flush(float *__restrict__ rowp, float4 a)
{
    float4 *p = reinterpret_cast<float4 *>(rowp);
    float4 t = *p;
    t.x += a.x;
    t.y += a.y;
    t.z += a.z;
    t.w += a.w;
    *p = t;
}

#ifdef NGCF_SWEPT_STAMPS
// diagnostic build only (tools/): cycle stamps of one quarter-wave per workgroup, summed per phase
#define NGCF_STAMP(t) unsigned long long t = __builtin_readcyclecounter()
#define NGCF_STAMP_ADD(i, t) stamp_sum[i] += __builtin_readcyclecounter() - t
#define NGCF_STAMP_USE(x) asm volatile("" ::"v"(x))
#else
#define NGCF_STAMP(t)
#define NGCF_STAMP_ADD(i, t)
#define NGCF_STAMP_USE(x)
#endif

#define NGCF_SWEPT_THREADS (kSweptOwnersPerWG * 16)

// entries k*16 .. k*16+15 of an owner's list: lane l keeps entry l (column, value, local row)
struct SweptEntries {
    int c, r, cnt;
    float v;
};

__device__ inline SweptEntries swept_load_entries(const int32_t *__restrict__ e_col, const float *__restrict__ e_val,
                                                  const uint8_t *__restrict__ e_row, int64_t pos, int64_t end, int l, int idle_col)
{
    SweptEntries e;
    const int64_t left = end - pos;
    e.cnt = left >= 16 ? 16 : (left > 0 ? (int)left : 0);
    e.c = idle_col;
    e.r = 0;
    e.v = 0.f;
    if (l < e.cnt) {
        e.c = e_col[pos + l];
        e.v = e_val[pos + l];
        e.r = e_row[pos + l];
    }
    return e;
}

// issue the 16 gathers of a chunk; idle slots re-read the chunk's first row (an L2 hit), never column 0
__device__ inline void swept_issue(float4 (&x)[16], const SweptEntries &e, int l, const float *__restrict__ Es, int64_t ldE)
{
    const int c0 = row_bcast<0>(e.c);
    const int c = l < e.cnt ? e.c : c0;
#define NGCF_GATHER(u) x[u] = *reinterpret_cast<const float4 *>(Es + (int64_t)row_bcast<u>(c) * ldE);
    NGCF_GATHER(0) NGCF_GATHER(1) NGCF_GATHER(2) NGCF_GATHER(3) NGCF_GATHER(4) NGCF_GATHER(5) NGCF_GATHER(6) NGCF_GATHER(7)
    NGCF_GATHER(8) NGCF_GATHER(9) NGCF_GATHER(10) NGCF_GATHER(11) NGCF_GATHER(12) NGCF_GATHER(13) NGCF_GATHER(14) NGCF_GATHER(15)
#undef NGCF_GATHER
}

// consecutive entries of one row are summed in registers and added to the owner's LDS row when the row changes
__device__ inline void swept_accumulate(const float4 (&x)[16], const SweptEntries &e, float *__restrict__ myacc, int &cur, float4 &a)
{
#define NGCF_ACCUM(u)                                   \
    if (u < e.cnt) {                                    \
        const int rr = row_bcast<u>(e.r);               \
        if (rr != cur) {                                \
            swept_flush(myacc + cur * 64, a);           \
            a = vzero4();                               \
            cur = rr;                                   \
        }                                               \
        a = vfma(row_bcast<u>(e.v), x[u], a);           \
    }
    NGCF_ACCUM(0) NGCF_ACCUM(1) NGCF_ACCUM(2) NGCF_ACCUM(3) NGCF_ACCUM(4) NGCF_ACCUM(5) NGCF_ACCUM(6) NGCF_ACCUM(7)
    NGCF_ACCUM(8) NGCF_ACCUM(9) NGCF_ACCUM(10) NGCF_ACCUM(11) NGCF_ACCUM(12) NGCF_ACCUM(13) NGCF_ACCUM(14) NGCF_ACCUM(15)
#undef NGCF_ACCUM
}

// Meeting point of the workgroups of one XCD after a column block.  Bounded spin: a group that is not resident
// together only loses the L2 re-use; it never hangs and never changes the result.
__device__ inline void swept_group_sync(unsigned *ctr, unsigned target, int max_spin)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < max_spin)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(NGCF_SWEPT_THREADS) void spmm_swept_kernel(
    const int64_t *__restrict__ own_ptr, const int32_t *__restrict__ own_blk, const int32_t *__restrict__ e_col,
    const float *__restrict__ e_val, const uint8_t *__restrict__ e_row, const int64_t *__restrict__ own_dst, int n_rounds,
    int n_blocks, int n_slices, const float *__restrict__ E, int64_t ldE, float *__restrict__ out, int64_t ldo,
    float *__restrict__ partial, int dp, unsigned *bar, int max_spin)
{
    __shared__ float acc_lds[kSweptOwnersPerWG * kSweptRPO * 64 + 4];   // 128 KiB of accumulators (+ the XCD id)
    const int q = threadIdx.x >> 4;          // owner slot in the workgroup
    const int l = threadIdx.x & 15;          // lane in the quarter-wave
    float *myacc = acc_lds + q * (kSweptRPO * 64) + l * 4;
    if (threadIdx.x == 0)
        reinterpret_cast<unsigned *>(acc_lds)[kSweptOwnersPerWG * kSweptRPO * 64] = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
    __syncthreads();
    unsigned *ctr = bar + reinterpret_cast<unsigned *>(acc_lds)[kSweptOwnersPerWG * kSweptRPO * 64] * 32;   // HW_REG_XCC_ID
    const unsigned members = gridDim.x / kSweptGroups;
    unsigned seq = 0;
    for (int slice = 0; slice < n_slices; ++slice) {
        const float *Es = E + slice * 64 + l * 4;
        for (int round = 0; round < n_rounds; ++round) {
            const int64_t owner = ((int64_t)round * gridDim.x + blockIdx.x) * kSweptOwnersPerWG + q;
#pragma unroll
            for (int r = 0; r < kSweptRPO; ++r) *reinterpret_cast<float4 *>(myacc + r * 64) = vzero4();
            const int64_t beg = own_ptr[owner], end = own_ptr[owner + 1];
            const int32_t *blk = own_blk + owner * (int64_t)n_blocks;
            // chunks this WAVE walks: the longest of its four owners
            int my_chunks = (int)((end - beg + 15) >> 4);
            my_chunks = max(my_chunks, __shfl_xor(my_chunks, 16));
            my_chunks = max(my_chunks, __shfl_xor(my_chunks, 32));
            int cur = 0, b = 0;
            int bend = n_blocks > 0 ? blk[0] : 0x7fffffff;     // end of block b in this owner's list (kept one block ahead)
            float4 a = vzero4();
            int idle_col = end > beg ? e_col[beg] : 0;
            // software pipeline: entries two chunks ahead, gathers one chunk ahead of the accumulation
            float4 xA[16], xB[16];
            SweptEntries eA = swept_load_entries(e_col, e_val, e_row, beg, end, l, idle_col);
            SweptEntries eB = swept_load_entries(e_col, e_val, e_row, beg + 16, end, l, idle_col);
            swept_issue(xA, eA, l, Es, ldE);
            for (int k = 0; k < my_chunks; k += 2) {
                // ---- chunk k (set A): prefetch entries k+2, issue gathers k+1, accumulate k
                SweptEntries eC = swept_load_entries(e_col, e_val, e_row, beg + (int64_t)(k + 2) * 16, end, l, idle_col);
                swept_issue(xB, eB, l, Es, ldE);
                swept_accumulate(xA, eA, myacc, cur, a);
                // a column block is finished once every owner of the wave has walked past its end
                while (b < n_blocks) {
                    int done = bend <= (k + 1) * 16 ? 1 : 0;
                    done &= __shfl_xor(done, 16);
                    done &= __shfl_xor(done, 32);
                    if (!done) break;
                    swept_group_sync(ctr, members * (++seq), max_spin);
                    ++b;
                    bend = b < n_blocks ? blk[b] : 0x7fffffff;
                }
                // ---- chunk k+1 (set B)
                eA = swept_load_entries(e_col, e_val, e_row, beg + (int64_t)(k + 3) * 16, end, l, idle_col);
                swept_issue(xA, eC, l, Es, ldE);
                swept_accumulate(xB, eB, myacc, cur, a);
                while (b < n_blocks) {
                    int done = bend <= (k + 2) * 16 ? 1 : 0;
                    done &= __shfl_xor(done, 16);
                    done &= __shfl_xor(done, 32);
                    if (!done) break;
                    swept_group_sync(ctr, members * (++seq), max_spin);
                    ++b;
                    bend = b < n_blocks ? blk[b] : 0x7fffffff;
                }
                eB = eA;
                eA = eC;
                // rotate: next iteration accumulates chunk k+2 from xA (issued above from eC) with entries eA = eC,
                // and needs eB = entries k+3
            }
            for (; b < n_blocks; ++b) swept_group_sync(ctr, members * (++seq), max_spin);   // every wave meets n_blocks times
            swept_flush(myacc + cur * 64, a);
            // write the owner's rows (its own LDS rows only: no barrier needed)
#pragma unroll 1
            for (int r = 0; r < kSweptRPO; ++r) {
                const int64_t dst = own_dst[owner * kSweptRPO + r];
                if (dst == kSweptUnused) continue;
                float *p = dst >= 0 ? out + dst * ldo : partial + (-1 - dst) * (int64_t)dp;
                *reinterpret_cast<float4 *>(p + slice * 64 + l * 4) = *reinterpret_cast<const float4 *>(myacc + r * 64);
            }
        }
    }
}

extern "C" int64_t ngcf_spmm_workspace_bytes(const ngcf_csr_t *c, int d)
{
    if (!c || d <= 0) return -1;
    const int64_t n_part = std::max(c->n_seg, c->swept.n_partial);
    return align_up(n_part * align_up(d, 4) * (int64_t)sizeof(float), 256) + 256;
}

// ---------------------------------------------------------------------------------------------
// optional in-library timing of the dominant kernel (bench.py's roofline figure): when enabled,
// a hipEvent pair is recorded on the launch stream around every spmm_kernel launch.
// ---------------------------------------------------------------------------------------------
static bool g_prof_on = false;
static std::vector<hipEvent_t> g_prof_events;   // pairs: begin, end
static size_t g_prof_used = 0;

static void prof_mark(hipStream_t stream, int which)
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
    const int64_t seg_blocks = (c->n_seg + 3) / 4;
    // d-slicing of the sliceable row groups needs 16-byte slices of 32 floats
    const bool can_slice = VEC == 4 && a.d % 32 == 0 && a.d >= 64 && c->mode != 1 && !getenv("NGCF_NO_SLICING");
    prof_mark(a.stream, 0);
    bool seg_done = seg_blocks == 0;
    for (size_t g = 0; g <= c->groups.size(); ++g) {
        const bool last = g == c->groups.size();
        if (last && seg_done) break;
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
                c->rowptr, c->colidx, c->vals, rbeg, rend, c->seg_row, c->seg_begin, seg_done ? 0 : c->n_seg, sb, c->seg_len,
                a.E, a.ldE, a.d, a.out, a.ldo, a.partial, a.dp, a.dr);
            LAUNCH_CHECK();
        }
        seg_done = true;
    }
    prof_mark(a.stream, 1);
    if (c->n_heavy > 0) {
        const int64_t fb = (c->n_heavy + 3) / 4;
        spmm_fixup_kernel<VEC><<<dim3((unsigned)fb), 256, 0, a.stream>>>(c->heavy_row, c->heavy_seg_ptr, c->n_heavy,
                                                                          a.partial, a.dp, a.d, a.out, a.ldo);
        LAUNCH_CHECK();
    }
    return NGCF_OK;
}
}  // namespace

static int spmm_dispatch(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo,
                         void *workspace, int64_t workspace_bytes, hipStream_t stream,
                         const EdgeDrop &dr = EdgeDrop{0, 0, {0, 0, 0, 0}, nullptr})
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
    const int dp = (int)align_up(d, 4);
    float *partial = nullptr;
    if (c->n_seg > 0) {
        const int64_t need = ngcf_spmm_workspace_bytes(c, d);
        if (!workspace || workspace_bytes < need)
            return fail(NGCF_ERR_WORKSPACE, "spmm: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
        partial = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
    }
    SpmmArgs a{c, E, ldE, d, out, ldo, partial, dp, stream, dr};
    const bool vec = (d % 4 == 0) && (ldE % 4 == 0) && (ldo % 4 == 0) && aligned16(E) && aligned16(out);
    const ngcf_csr::Swept &w = c->swept;
    if (vec && d % 64 == 0 && w.n_owners > 0 && c->mode == 2 && dr.n == 0) {
        if (w.n_partial > 0 && !partial) {
            const int64_t need = ngcf_spmm_workspace_bytes(c, d);
            if (!workspace || workspace_bytes < need)
                return fail(NGCF_ERR_WORKSPACE, "spmm: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
            partial = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
        }
        static const int max_spin = getenv("NGCF_SWEPT_SPIN") ? atoi(getenv("NGCF_SWEPT_SPIN")) : 400;
        HIP_TRY(hipMemsetAsync(w.barrier, 0, sizeof(uint32_t) * 32 * kSweptGroups, stream));
        prof_mark(stream, 0);
        spmm_swept_kernel<<<dim3(kSweptWGs), NGCF_SWEPT_THREADS, 0, stream>>>(
            w.own_ptr, w.own_blk, w.e_col, w.e_val, w.e_row, w.own_dst, w.n_rounds, w.n_blocks, d / 64, E, ldE, out, ldo,
            partial, dp, w.barrier, max_spin);
        LAUNCH_CHECK();
        prof_mark(stream, 1);
        if (w.n_heavy > 0) {
            spmm_fixup_kernel<4><<<dim3((unsigned)((w.n_heavy + 3) / 4)), 256, 0, stream>>>(
                w.heavy_row, w.heavy_seg_ptr, w.n_heavy, partial, dp, d, out, ldo);
            LAUNCH_CHECK();
        }
        return NGCF_OK;
    }
    if (vec) {
        const int nq = d / 4;
        if (nq <= 8) return launch_spmm<4, 8, 1, 4>(a);
        if (nq <= 16) return launch_spmm<4, 16, 1, 8>(a);
        if (nq <= 32) return launch_spmm<4, 32, 1, 8>(a);
        if (nq <= 64) return launch_spmm<4, 64, 1, 8>(a);
        return launch_spmm<4, 64, 2, 4>(a);
    }
    if (d <= 64) return launch_spmm<1, 64, 1, 8>(a);
    if (d <= 128) return launch_spmm<1, 64, 2, 4>(a);
    if (d <= 256) return launch_spmm<1, 64, 4, 2>(a);
    return launch_spmm<1, 64, 8, 1>(a);
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

// ---------------------------------------------------------------------------------------------
// Dense half of a layer (NGCF.py:131-146) on the fp32 matrix cores.
//
//   M = [LE+E | LE*E] . [W1^T ; W2^T] + (2*b1 + b2)
// The K dimension is walked in chunks of DC = 16 input columns: a chunk contributes 16 "sum" values
// and 16 "product" values per row (KC = 32 k-steps), so LE and E are read exactly once.  Weights are
// packed per call into that chunk order ([n_chunks*32][DOP], zero padded) by pack_weights_kernel.
// v_mfma_f32_32x32x2_f32: exact fp32 FMA chain per output element.
// A workgroup of 4 waves owns BM = 32*RW full rows; waves are arranged RW x CW, each wave NT 32x32 tiles,
// so a whole output row (<= 32*NT*CW columns) lives in one workgroup and the L2 row-normalisation is
// done in registers + one LDS exchange.
// ---------------------------------------------------------------------------------------------
#define NGCF_KC 32
#define NGCF_DC 16

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers inside lambdas

__global__ void pack_weights_kernel(const float *__restrict__ W1, const float *__restrict__ b1,
                                    const float *__restrict__ W2, const float *__restrict__ b2, int d_in, int d_out,
                                    int n_chunks, int DOP, int NT, float *__restrict__ Wt, float *__restrict__ bias2)
{
    // Wt[chunk][kl][cw][j][t] = weight of output column (cw*NT + t)*32 + j: a lane reads its NT tile values at once
    const int total = n_chunks * NGCF_KC * DOP;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int within = i % DOP;
        const int t = within % NT, j = (within / NT) % 32, cw = within / (NT * 32);
        const int oc = (cw * NT + t) * 32 + j;
        const int k = i / DOP;
        const int chunk = k / NGCF_KC, kl = k % NGCF_KC;
        const int col = chunk * NGCF_DC + (kl % NGCF_DC);
        float w = 0.f;
        if (oc < d_out && col < d_in) w = (kl < NGCF_DC ? W1 : W2)[(int64_t)oc * d_in + col];
        Wt[i] = w;
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < DOP; j += gridDim.x * blockDim.x)
        bias2[j] = j < d_out ? (b1[j] + b1[j]) + b2[j] : 0.f;   // b1 is added twice, NGCF.py:131,133
}

template <int RW, int CW, int NT, bool ALIGNED>
__global__ __launch_bounds__(256) void layer_dense_kernel(const float *__restrict__ LE, int64_t ldLE,
                                                          const float *__restrict__ Es, int64_t ldE, int64_t n_rows,
                                                          int d_in, int d_out, const float *__restrict__ Wt,
                                                          const float *__restrict__ bias2, int n_chunks,
                                                          float leaky, float drop_p, uint64_t drop_seed,
                                                          float *__restrict__ carry, int64_t ldc,
                                                          float *__restrict__ norm, int64_t ldn)
{
    constexpr int BM = 32 * RW;
    constexpr int WCOLS = 32 * NT * CW;      // == DOP
    constexpr int XLD = NGCF_KC + 4;         // 36: rows stay 16-B aligned and b128 column reads are conflict-free
    constexpr bool DB = WCOLS <= 128;        // double-buffered LDS (one barrier per chunk) where two blocks still fit a CU
    constexpr int NBUF = DB ? 2 : 1;
    constexpr int RR = (BM + 63) / 64;       // X rows staged per thread
    constexpr int WN = (NGCF_KC * WCOLS / 4) / 256;   // W float4s staged per thread
    __shared__ float Xs[NBUF * BM * XLD];
    __shared__ float Ws[NBUF * NGCF_KC * WCOLS];
    __shared__ float ssq[BM * CW];

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int rw = wave / CW, cw = wave % CW;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * BM;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // staging roles: 4 lanes x 4 columns cover the 16 input columns of a chunk for one row
    const int sq = tid & 3;
    const int sr = tid >> 2;   // 0..63
    f32x4 xle[RR], xe[RR], wreg[WN];

    auto load_chunk = [&](int chunk) {       // global -> registers
        const int c0 = chunk * NGCF_DC + sq * 4;
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
            const int r = sr + rr * 64;
            float le[4] = {0.f, 0.f, 0.f, 0.f}, e[4] = {0.f, 0.f, 0.f, 0.f};
            const int64_t grow = row0 + r;
            if (r < BM && grow < n_rows) {
                if (ALIGNED && c0 + 4 <= d_in) {
                    const float4 a = *reinterpret_cast<const float4 *>(LE + grow * ldLE + c0);
                    const float4 b = *reinterpret_cast<const float4 *>(Es + grow * ldE + c0);
                    le[0] = a.x; le[1] = a.y; le[2] = a.z; le[3] = a.w;
                    e[0] = b.x; e[1] = b.y; e[2] = b.z; e[3] = b.w;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < d_in) {
                            le[q] = LE[grow * ldLE + c0 + q];
                            e[q] = Es[grow * ldE + c0 + q];
                        }
                }
            }
            xle[rr] = f32x4{le[0], le[1], le[2], le[3]};
            xe[rr] = f32x4{e[0], e[1], e[2], e[3]};
        }
        const f32x4 *src = reinterpret_cast<const f32x4 *>(Wt + (int64_t)chunk * NGCF_KC * WCOLS);
#pragma unroll
        for (int i = 0; i < WN; ++i) wreg[i] = src[tid + i * 256];
    };
    auto store_chunk = [&](int buf) {        // registers -> LDS: (LE + E) feeds W1, (LE * E) feeds W2
        float *X = Xs + buf * (BM * XLD);
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
            const int r = sr + rr * 64;
            if (r < BM) {
                const f32x4 a = xle[rr], b = xe[rr];
                *reinterpret_cast<f32x4 *>(X + r * XLD + sq * 4) = a + b;
                *reinterpret_cast<f32x4 *>(X + r * XLD + NGCF_DC + sq * 4) = a * b;
            }
        }
        f32x4 *dst = reinterpret_cast<f32x4 *>(Ws + buf * (NGCF_KC * WCOLS));
#pragma unroll
        for (int i = 0; i < WN; ++i) dst[tid + i * 256] = wreg[i];
    };
    auto compute_chunk = [&](int buf) {      // 32 k-values: 4 blocks of (one b128 A read, 4 x NT-wide B reads, 4*NT MFMAs)
        const float *X = Xs + buf * (BM * XLD) + (rw * 32 + li) * XLD + lh * 4;
        const float *W = Ws + buf * (NGCF_KC * WCOLS) + cw * (32 * NT) + li * NT;
#pragma unroll
        for (int kb = 0; kb < NGCF_KC / 8; ++kb) {
            const f32x4 a4 = *reinterpret_cast<const f32x4 *>(X + kb * 8);
#define NGCF_KSTEP(sx, aval)                                                                                   \
    {                                                                                                          \
        const float *wk = W + (kb * 8 + lh * 4 + sx) * WCOLS;                                                  \
        float bv[NT];                                                                                          \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) bv[t] = wk[t];                                          \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                         \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bv[t], acc[t], 0, 0, 0);                       \
    }
            NGCF_KSTEP(0, a4.x) NGCF_KSTEP(1, a4.y) NGCF_KSTEP(2, a4.z) NGCF_KSTEP(3, a4.w)
#undef NGCF_KSTEP
        }
    };

    if (DB) {
        load_chunk(0);
        store_chunk(0);
        __syncthreads();
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const bool more = chunk + 1 < n_chunks;
            if (more) load_chunk(chunk + 1);        // global loads fly under the MFMAs
            compute_chunk(chunk & 1);
            if (more) store_chunk((chunk + 1) & 1);
            __syncthreads();
        }
    } else {
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            load_chunk(chunk);
            store_chunk(0);
            __syncthreads();
            compute_chunk(0);
            __syncthreads();
        }
    }

    // ---- epilogue: bias, LeakyReLU, dropout, row sum of squares
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t drop_thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    float rowss[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rowss[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = (cw * NT + t) * 32 + li;
        const float bz = bias2[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[t][r] + bz;
            v = v >= 0.f ? v : leaky * v;
            if (drop_p > 0.f) {
                const int64_t grow = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const uint32_t h = mix32(drop_seed ^ ((uint64_t)grow * 0x9E3779B97F4A7C15ULL + (uint64_t)col));
                v = h < drop_thr ? 0.f : v * keep_scale;
            }
            acc[t][r] = v;
            rowss[r] = fmaf(v, v, rowss[r]);
        }
    }
    // reduce over the 32 lanes that share a row (lanes li = 0..31 within each half)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float s = rowss[r];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        rowss[r] = s;
    }
    if (CW > 1) {
        if (li == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                ssq[lr * CW + cw] = rowss[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < CW; ++q) s += ssq[lr * CW + q];
            rowss[r] = s;
        }
    }
    // ---- stores: carry (un-normalised, feeds the next layer) and the normalised all_E block
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t grow = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (grow >= n_rows) continue;
        const float nrm = fmaxf(sqrtf(rowss[r]), 1e-12f);   // F.normalize eps, NGCF.py:144
        const float inv = 1.f / nrm;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = (cw * NT + t) * 32 + li;
            if (col < d_out) {
                const float v = acc[t][r];
                if (carry) carry[grow * ldc + col] = v;
                norm[grow * ldn + col] = v * inv;
            }
        }
    }
}

static int dense_dop(int d_out)
{
    if (d_out <= 32) return 32;
    if (d_out <= 64) return 64;
    if (d_out <= 96) return 96;
    if (d_out <= 128) return 128;
    if (d_out <= 256) return 256;
    if (d_out <= 512) return 512;
    return -1;
}

extern "C" int64_t ngcf_dense_workspace_bytes(int d_in, int d_out)
{
    const int dop = dense_dop(d_out);
    if (dop < 0 || d_in <= 0) return -1;
    const int64_t n_chunks = (d_in + NGCF_DC - 1) / NGCF_DC;
    return align_up((n_chunks * NGCF_KC * dop + dop) * (int64_t)sizeof(float), 256) + 256;
}

template <int RW, int CW, int NT>
static int launch_dense(bool al, int64_t n_rows, const float *LE, int64_t ldLE, const float *Es, int64_t ldE, int d_in,
                        int d_out, const float *Wt, const float *bias2, int n_chunks, float leaky, float drop_p,
                        uint64_t seed, float *carry, int64_t ldc, float *norm, int64_t ldn, hipStream_t stream)
{
    const int64_t blocks = (n_rows + 32 * RW - 1) / (32 * RW);
    if (blocks == 0) return NGCF_OK;
    if (al)
        layer_dense_kernel<RW, CW, NT, true><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            LE, ldLE, Es, ldE, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, seed, carry, ldc, norm, ldn);
    else
        layer_dense_kernel<RW, CW, NT, false><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            LE, ldLE, Es, ldE, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, seed, carry, ldc, norm, ldn);
    LAUNCH_CHECK();
    return NGCF_OK;
}

extern "C" int ngcf_layer_dense_f32(const float *LE, int64_t ldLE, const float *Es, int64_t ldEs, int64_t n_rows,
                                    int d_in, const float *W1, const float *b1, const float *W2, const float *b2,
                                    int d_out, float leaky, float drop_p, uint64_t drop_seed, float *carry,
                                    int64_t ldc, float *norm, int64_t ldn, void *workspace, int64_t workspace_bytes,
                                    void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!LE || !Es || !W1 || !b1 || !W2 || !b2 || !norm) return fail(NGCF_ERR_ARG, "layer_dense: null argument");
    if (n_rows < 0 || d_in <= 0 || d_out <= 0) return fail(NGCF_ERR_ARG, "layer_dense: bad sizes");
    const int dop = dense_dop(d_out);
    if (dop < 0) return fail(NGCF_ERR_ARG, "layer_dense: d_out=%d > 512 is not supported", d_out);
    if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(NGCF_ERR_ARG, "layer_dense: drop_p=%f not in [0,1)", drop_p);
    if (ldLE < d_in || ldEs < d_in || ldn < d_out || (carry && ldc < d_out))
        return fail(NGCF_ERR_ARG, "layer_dense: leading dimension too small");
    const int64_t need = ngcf_dense_workspace_bytes(d_in, d_out);
    if (!workspace || workspace_bytes < need)
        return fail(NGCF_ERR_WORKSPACE, "layer_dense: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    const int n_chunks = (d_in + NGCF_DC - 1) / NGCF_DC;
    float *Wt = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
    float *bias2 = Wt + (int64_t)n_chunks * NGCF_KC * dop;
    pack_weights_kernel<<<64, 256, 0, stream>>>(W1, b1, W2, b2, d_in, d_out, n_chunks, dop, dop <= 128 ? dop / 32 : 4, Wt, bias2);
    LAUNCH_CHECK();
    const bool al = (ldLE % 4 == 0) && (ldEs % 4 == 0) && aligned16(LE) && aligned16(Es);
#define NGCF_DENSE(RW, CW, NT) \
    return launch_dense<RW, CW, NT>(al, n_rows, LE, ldLE, Es, ldEs, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, \
                                    drop_seed, carry, ldc, norm, ldn, stream)
    switch (dop) {
    case 32: NGCF_DENSE(4, 1, 1);
    case 64: NGCF_DENSE(4, 1, 2);
    case 96: NGCF_DENSE(4, 1, 3);
    case 128: NGCF_DENSE(4, 1, 4);
    case 256: NGCF_DENSE(2, 2, 4);
    default: NGCF_DENSE(1, 4, 4);
    }
#undef NGCF_DENSE
}

extern "C" int64_t ngcf_layer_workspace_bytes(const ngcf_csr_t *c, int d_in, int d_out)
{
    if (!c) return -1;
    const int64_t a = ngcf_spmm_workspace_bytes(c, d_in);
    const int64_t b = ngcf_dense_workspace_bytes(d_in, d_out);
    if (a < 0 || b < 0) return -1;
    const int64_t le = align_up(c->n_rows * align_up(d_in, 4) * (int64_t)sizeof(float), 256);
    return a + b + le + 256;
}

extern "C" int ngcf_layer_fused_f32(const ngcf_csr_t *c, const float *Eg, int64_t ldEg, const float *Es, int64_t ldEs,
                                    int d_in, const float *W1, const float *b1, const float *W2, const float *b2,
                                    int d_out, float leaky, float drop_p, uint64_t drop_seed, float *carry,
                                    int64_t ldc, float *norm, int64_t ldn, void *workspace, int64_t workspace_bytes,
                                    void *stream)
{
    if (!c) return fail(NGCF_ERR_ARG, "layer_fused: null csr");
    const int64_t need = ngcf_layer_workspace_bytes(c, d_in, d_out);
    if (need < 0) return fail(NGCF_ERR_ARG, "layer_fused: unsupported widths d_in=%d d_out=%d", d_in, d_out);
    if (!workspace || workspace_bytes < need)
        return fail(NGCF_ERR_WORKSPACE, "layer_fused: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    char *ws = reinterpret_cast<char *>(align_up((int64_t)(uintptr_t)workspace, 256));
    const int64_t ldLE = align_up(d_in, 4);
    float *LE = reinterpret_cast<float *>(ws);
    ws += align_up(c->n_rows * ldLE * (int64_t)sizeof(float), 256);
    const int64_t spmm_ws = ngcf_spmm_workspace_bytes(c, d_in);
    void *ws_spmm = ws;
    ws += spmm_ws;
    const int64_t dense_ws = ngcf_dense_workspace_bytes(d_in, d_out);
    void *ws_dense = ws;
    int rc = ngcf_spmm_csr_f32(c, Eg, ldEg, d_in, LE, ldLE, ws_spmm, spmm_ws, stream);
    if (rc != NGCF_OK) return rc;
    return ngcf_layer_dense_f32(LE, ldLE, Es, ldEs, c->n_rows, d_in, W1, b1, W2, b2, d_out, leaky, drop_p, drop_seed,
                                carry, ldc, norm, ldn, ws_dense, dense_ws, stream);
}

// ---------------------------------------------------------------------------------------------
// strided row copy (E0 -> its block of all_E), NGCF.py:120-121,147
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ void copy_rows_kernel(const float *__restrict__ src, int64_t lds, float *__restrict__ dst, int64_t ldd,
                                 int64_t n_rows, int d)
{
    using V = typename VecT<VEC>::type;
    const int per_row = d / VEC;
    const int64_t total = n_rows * per_row;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t r = i / per_row;
        const int q = (int)(i % per_row) * VEC;
        *reinterpret_cast<V *>(dst + r * ldd + q) = *reinterpret_cast<const V *>(src + r * lds + q);
    }
}

extern "C" int ngcf_copy_rows_f32(const float *src, int64_t lds, float *dst, int64_t ldd, int64_t n_rows, int d,
                                  void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst || d <= 0 || n_rows < 0 || lds < d || ldd < d) return fail(NGCF_ERR_ARG, "copy_rows: bad argument");
    if (n_rows == 0) return NGCF_OK;
    const bool vec = d % 4 == 0 && lds % 4 == 0 && ldd % 4 == 0 && aligned16(src) && aligned16(dst);
    if (vec)
        copy_rows_kernel<4><<<grid_for(n_rows * (d / 4), 256), 256, 0, stream>>>(src, lds, dst, ldd, n_rows, d);
    else
        copy_rows_kernel<1><<<grid_for(n_rows * d, 256), 256, 0, stream>>>(src, lds, dst, ldd, n_rows, d);
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

// ---------------------------------------------------------------------------------------------
// BPR, bprloss.py:15-22
// ---------------------------------------------------------------------------------------------
__device__ inline float wave_sum(float x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}

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
// Backward pass (`loss.backward()` in experiment.py:57).
// The two plain GEMMs per layer (dM.[W1|W2] and dM^T.[S|P]) are library GEMMs issued by the host; everything
// around them is fused here: BPR gradient, gather scatter-add, normalise/dropout/LeakyReLU backward,
// the [LE+E | LE*E] operand, and the combination of the GEMM result into dLE and the direct part of dE.
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
        if (Bu == R) du[ru * D + j] = gu; else atomicAdd(&du[j], gu);
        if (Bp == R) dp[rp * D + j] = gp; else atomicAdd(&dp[j], gp);
        if (Bn == R) dn[rn * D + j] = gn; else atomicAdd(&dn[j], gn);
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
    if (Bu != R) HIP_TRY(hipMemsetAsync(du, 0, sizeof(float) * (size_t)D, stream));
    if (Bp != R) HIP_TRY(hipMemsetAsync(dp, 0, sizeof(float) * (size_t)D, stream));
    if (Bn != R) HIP_TRY(hipMemsetAsync(dn, 0, sizeof(float) * (size_t)D, stream));
    bpr_backward_kernel<<<dim3((unsigned)((R + 3) / 4)), 256, 0, stream>>>(u, Bu, p, Bp, n, Bn, R, D, wd, batch_size, grad_out,
                                                                            du, dp, dn);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---- gather backward: G[row_off + idx[b], :] += g[b, :] (duplicates add up) -------------------
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float *__restrict__ G, int64_t ld, int d,
                                                               const int64_t *__restrict__ idx, int64_t B, int64_t row_off,
                                                               int64_t n_idx_rows, const float *__restrict__ g, int64_t ldg)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t i = idx[b];
    if (i < 0 || i >= n_idx_rows) return;
    float *dst = G + (row_off + i) * ld;
    for (int j = threadIdx.x & 63; j < d; j += 64) atomicAdd(&dst[j], g[b * ldg + j]);
}

extern "C" int ngcf_scatter_add_rows_f32(float *G, int64_t ld, int d, const int64_t *idx, int64_t B, int64_t row_off,
                                         int64_t n_idx_rows, const float *g, int64_t ldg, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (B == 0) return NGCF_OK;
    if (!G || !idx || !g || d <= 0 || ld < d || ldg < d) return fail(NGCF_ERR_ARG, "scatter_add_rows: bad argument");
    scatter_add_rows_kernel<<<dim3((unsigned)((B + 3) / 4)), 256, 0, stream>>>(G, ld, d, idx, B, row_off, n_idx_rows, g, ldg);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---- normalise + dropout + LeakyReLU backward: (dN, dC, C) -> dM, one wave per row ------------
// forward: A = leaky(M); C = keep ? A/(1-p) : 0; N = C / max(|C|, eps)    (NGCF.py:140-144)
__global__ __launch_bounds__(256) void layer_bwd_pre_kernel(const float *__restrict__ dN, int64_t ldn,
                                                            const float *__restrict__ dC, int64_t ldc,
                                                            const float *__restrict__ C, int64_t ldC, int64_t n_rows,
                                                            int d, float leaky, float drop_p, uint64_t seed,
                                                            float *__restrict__ dM, int64_t ldm)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const float *c = C + r * ldC, *g = dN + r * ldn;
    float ss = 0.f, dot = 0.f;
    for (int j = lane; j < d; j += 64) {
        ss = fmaf(c[j], c[j], ss);
        dot = fmaf(c[j], g[j], dot);
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float nrm = sqrtf(ss);
    const bool clamped = nrm < 1e-12f;                       // F.normalize's clamp_min: N = C / eps there
    const float den = clamped ? 1e-12f : nrm;
    const float ydot = clamped ? 0.f : dot / (den * den);    // (y.dy)/|x| with y = x/|x|
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    for (int j = lane; j < d; j += 64) {
        float t = g[j] / den - c[j] * (ydot / den);
        if (dC) t += dC[r * ldc + j];
        if (drop_p > 0.f) {
            const uint32_t h = mix32(seed ^ ((uint64_t)r * 0x9E3779B97F4A7C15ULL + (uint64_t)j));
            t = h < thr ? 0.f : t * keep_scale;
        }
        dM[r * ldm + j] = t * (c[j] > 0.f ? 1.f : leaky);     // sign(C) == sign(M) wherever C was kept
    }
}

extern "C" int ngcf_layer_bwd_pre_f32(const float *dN, int64_t ldn, const float *dC, int64_t ldc, const float *C, int64_t ldC,
                                      int64_t n_rows, int d, float leaky, float drop_p, uint64_t seed, float *dM, int64_t ldm,
                                      void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (n_rows == 0) return NGCF_OK;
    if (!dN || !C || !dM || d <= 0) return fail(NGCF_ERR_ARG, "layer_bwd_pre: bad argument");
    layer_bwd_pre_kernel<<<dim3((unsigned)((n_rows + 3) / 4)), 256, 0, stream>>>(dN, ldn, dC, ldc, C, ldC, n_rows, d, leaky, drop_p,
                                                                                 seed, dM, ldm);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---- SP = [LE + E | LE * E]  (the GEMM operand of the forward, needed for dW1/dW2) ------------
__global__ void sp_concat_kernel(const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ E, int64_t ldE,
                                 int64_t n_rows, int d, float *__restrict__ SP)
{
    const int64_t total = n_rows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const int j = (int)(i % d);
        const float a = LE[r * ldLE + j], b = E[r * ldE + j];
        SP[r * 2 * d + j] = a + b;
        SP[r * 2 * d + d + j] = a * b;
    }
}

extern "C" int ngcf_sp_concat_f32(const float *LE, int64_t ldLE, const float *E, int64_t ldE, int64_t n_rows, int d, float *SP,
                                  void *stream_)
{
    if (n_rows == 0) return NGCF_OK;
    if (!LE || !E || !SP || d <= 0) return fail(NGCF_ERR_ARG, "sp_concat: bad argument");
    sp_concat_kernel<<<grid_for(n_rows * d, 256), 256, 0, (hipStream_t)stream_>>>(LE, ldLE, E, ldE, n_rows, d, SP);
    LAUNCH_CHECK();
    return NGCF_OK;
}

// ---- dSP = dM.[W1 | W2] -> dLE = dS + dP*E ; dE_direct = dS + dP*LE ----------------------------
__global__ void layer_bwd_combine_kernel(const float *__restrict__ dSP, const float *__restrict__ LE, int64_t ldLE,
                                         const float *__restrict__ E, int64_t ldE, int64_t n_rows, int d,
                                         float *__restrict__ dLE, float *__restrict__ dE)
{
    const int64_t total = n_rows * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const int j = (int)(i % d);
        const float ds = dSP[r * 2 * d + j], dp = dSP[r * 2 * d + d + j];
        dLE[i] = fmaf(dp, E[r * ldE + j], ds);
        dE[i] = fmaf(dp, LE[r * ldLE + j], ds);
    }
}

extern "C" int ngcf_layer_bwd_combine_f32(const float *dSP, const float *LE, int64_t ldLE, const float *E, int64_t ldE,
                                          int64_t n_rows, int d, float *dLE, float *dE, void *stream_)
{
    if (n_rows == 0) return NGCF_OK;
    if (!dSP || !LE || !E || !dLE || !dE || d <= 0) return fail(NGCF_ERR_ARG, "layer_bwd_combine: bad argument");
    layer_bwd_combine_kernel<<<grid_for(n_rows * d, 256), 256, 0, (hipStream_t)stream_>>>(dSP, LE, ldLE, E, ldE, n_rows, d, dLE, dE);
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

extern "C" int ngcf_add_rows_f32(float *out, int64_t ldo, const float *add, int64_t lda, int64_t n_rows, int d, void *stream_)
{
    if (n_rows == 0) return NGCF_OK;
    if (!out || !add || d <= 0) return fail(NGCF_ERR_ARG, "add_rows: bad argument");
    add_rows_kernel<<<grid_for(n_rows * d, 256), 256, 0, (hipStream_t)stream_>>>(out, ldo, add, lda, n_rows, d);
    LAUNCH_CHECK();
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

__global__ __launch_bounds__(256) void topk_rows_kernel(const float *__restrict__ scores, int64_t ld, int64_t n_cols, int k,
                                                        int kp2, float *__restrict__ out_val, int64_t *__restrict__ out_idx)
{
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sh_prefix, sh_need, sh_cnt_gt, sh_cnt_eq;
    __shared__ uint32_t skey[NGCF_TOPK_MAX];
    __shared__ int32_t sidx[NGCF_TOPK_MAX];
    const float *row = scores + (int64_t)blockIdx.x * ld;
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
        out_idx[(int64_t)blockIdx.x * k + j] = sidx[j];
        out_val[(int64_t)blockIdx.x * k + j] = row[sidx[j]];
    }
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
