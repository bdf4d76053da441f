// csr.hip - library identity, the Laplacian CSR object, its row-segment plan and row groups.
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
#include "common.h"

extern "C" const char *ngcf_last_error(void) { return g_err; }
extern "C" const char *ngcf_target_arch(void) { return "gfx950"; }
extern "C" int ngcf_version(void) { return NGCF_ABI_VERSION; }   // the mirror refuses a library built from another header

// ---------------------------------------------------------------------------------------------
// options (common.h, NgcfOptions): the ONLY place of the library that reads the environment
// ---------------------------------------------------------------------------------------------
NgcfOptions g_opts;

namespace {
struct OptField {
    const char *name;       // option name = environment variable without the NGCF_ prefix, lower case
    int NgcfOptions::*i;
    int64_t NgcfOptions::*l;
};
const OptField kOptFields[] = {
    {"no_fork", &NgcfOptions::no_fork, nullptr},
    {"no_slicing", &NgcfOptions::no_slicing, nullptr},
    {"no_ldstab", &NgcfOptions::no_ldstab, nullptr},
    {"no_panel_split", &NgcfOptions::no_panel_split, nullptr},
    {"no_tail_table", &NgcfOptions::no_tail_table, nullptr},
    {"no_pad_product", &NgcfOptions::no_pad_product, nullptr},
    {"fork_min", nullptr, &NgcfOptions::fork_min},
    {"dense_direct", &NgcfOptions::dense_direct, nullptr},
    {"dense_resident", &NgcfOptions::dense_resident, nullptr},
    {"dense_resident_min_rows", &NgcfOptions::dense_resident_min_rows, nullptr},
    {"dense_small_tiles", &NgcfOptions::dense_small_tiles, nullptr},
    {"dense_il_lab", &NgcfOptions::dense_il_lab, nullptr},
    {"dense_tall", &NgcfOptions::dense_tall, nullptr},
    {"bwd_input_resident", &NgcfOptions::bwd_input_resident, nullptr},
    {"t_rows_bitmap", &NgcfOptions::t_rows_bitmap, nullptr},
    {"slice_max_mb", &NgcfOptions::slice_max_mb, nullptr},
    {"swept_lpe", &NgcfOptions::swept_lpe, nullptr},
    {"swept_waves", &NgcfOptions::swept_waves, nullptr},
    {"swept_cut", &NgcfOptions::swept_cut, nullptr},
    {"swept_no_moments", &NgcfOptions::swept_no_moments, nullptr},
    {"swept_order_rows", &NgcfOptions::swept_order_rows, nullptr},
    {"swept_debug", &NgcfOptions::swept_debug, nullptr},
    {"swept_window_kb", &NgcfOptions::swept_window_kb, nullptr},
    {"swept_spin", &NgcfOptions::swept_spin, nullptr},
    {"swept_lead", &NgcfOptions::swept_lead, nullptr},
    {"swept_sync_every", &NgcfOptions::swept_sync_every, nullptr},
    {"swept_prio_kb", &NgcfOptions::swept_prio_kb, nullptr},
    {"swept_prio_graded", &NgcfOptions::swept_prio_graded, nullptr},
    {"swept_nt", &NgcfOptions::swept_nt, nullptr},
    {"swept_merge", &NgcfOptions::swept_merge, nullptr},
};
bool g_opts_read = false;
}  // namespace

// Defaults, then every NGCF_<NAME> variable that is set (a flag variable that is set but empty counts as 1).  Called once on
// first use; tools call it again after changing os.environ.
extern "C" int ngcf_options_from_env(void)
{
    NgcfOptions o;
    for (const OptField &f : kOptFields) {
        char var[64] = "NGCF_";
        size_t n = strlen(var);
        for (const char *p = f.name; *p && n + 1 < sizeof(var); ++p) var[n++] = (char)((*p >= 'a' && *p <= 'z') ? *p - 32 : *p);
        var[n] = 0;
        const char *e = getenv(var);
        if (!e) continue;
        const long long v = *e ? atoll(e) : 1;
        if (f.i) o.*(f.i) = (int)(*e && (*e == '-' || (*e >= '0' && *e <= '9')) ? v : 1);
        else o.*(f.l) = v;
    }
    if (const char *e = getenv("NGCF_SWEPT_ORDER")) o.swept_order_rows = !strcmp(e, "rows");
    if (const char *e = getenv("NGCF_SWEPT_TRACE")) snprintf(o.swept_trace, sizeof(o.swept_trace), "%s", e);
    g_opts = o;
    g_opts_read = true;
    return NGCF_OK;
}

const NgcfOptions &ngcf_opts()
{
    if (!g_opts_read) ngcf_options_from_env();
    return g_opts;
}

extern "C" int ngcf_set_option(const char *name, int64_t value)
{
    if (!name) return fail(NGCF_ERR_ARG, "ngcf_set_option: null name");
    (void)ngcf_opts();
    for (const OptField &f : kOptFields)
        if (!strcmp(f.name, name)) {
            if (f.i) g_opts.*(f.i) = (int)value;
            else g_opts.*(f.l) = value;
            return NGCF_OK;
        }
    return fail(NGCF_ERR_ARG, "ngcf_set_option: unknown option '%s'", name);
}

extern "C" int ngcf_set_option_str(const char *name, const char *value)
{
    if (!name) return fail(NGCF_ERR_ARG, "ngcf_set_option_str: null name");
    (void)ngcf_opts();
    if (!strcmp(name, "swept_trace")) {
        snprintf(g_opts.swept_trace, sizeof(g_opts.swept_trace), "%s", value ? value : "");
        return NGCF_OK;
    }
    return fail(NGCF_ERR_ARG, "ngcf_set_option_str: unknown option '%s'", name);
}


static void free_plan(ngcf_csr *c)
{
    if (c->borrows_plan) {             // a filtered copy shares its source's segment lists (it owns seg_begin only)
        c->seg_row = nullptr;
        c->heavy_row = nullptr;
        c->heavy_seg_ptr = nullptr;
        c->borrows_plan = false;
    }
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
    if (c->pos) (void)hipFree(c->pos);
    if (c->scan_blk) (void)hipFree(c->scan_blk);
    delete c;
}

extern "C" int64_t ngcf_csr_nnz(const ngcf_csr_t *c) { return c ? c->nnz : -1; }
extern "C" int64_t ngcf_csr_n_rows(const ngcf_csr_t *c) { return c ? c->n_rows : -1; }
extern "C" int64_t ngcf_csr_n_cols(const ngcf_csr_t *c) { return c ? c->n_cols : -1; }
extern "C" int64_t ngcf_csr_n_segments(const ngcf_csr_t *c) { return c ? c->n_seg : -1; }
extern "C" int64_t ngcf_csr_max_row_len(const ngcf_csr_t *c) { return c ? c->max_row_len : -1; }
extern "C" int64_t ngcf_csr_swept_rows(const ngcf_csr_t *c)
{
    if (!c) return -1;
    int64_t n = 0;
    for (const auto &p : c->swept.parts) n += p.row_hi - p.row_lo;
    return n;
}
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


// per block of `group_rows` rows: smallest and largest column any of its rows gathers (one-time, plan only)
__global__ void row_colrange_kernel(const int64_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                    int64_t n_rows, int group_rows, int32_t *__restrict__ blk_min, int32_t *__restrict__ blk_max)
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
        atomicMin(&blk_min[row / group_rows], lo);
        atomicMax(&blk_max[row / group_rows], hi);
    }
}

// A row block is "sliceable" when one 32-float slice (128 B) of every row it gathers is at most 48 MiB: then the
// hot part of the table slice lives in the XCD L2s while a slice-major launch walks it (measured on the user half
// of C3: 2.33 ms sliced vs 2.98 ms unsliced; the 1 M-row user table gets slower sliced: 3.8 vs 3.46 ms).
static int64_t slice_footprint_rows()   // slice_max_mb: lab knob (C5-sized tables: slices that fit the 256 MiB Infinity Cache)
{
    return ((int64_t)ngcf_opts().slice_max_mb << 20) / 128;
}
#define kSliceFootprintRows slice_footprint_rows()

static int build_row_groups(ngcf_csr *c, hipStream_t stream)
{
    c->groups.clear();
    if (c->n_rows == 0) return NGCF_OK;
    // blocks of 1024 rows; 64 on a small matrix, where the border between the user rows and the item rows of a Seoul-sized
    // graph (5 840 + 100 rows) would otherwise put an eighth of the user rows into a mixed block
    const int64_t NGCF_GROUP_ROWS = c->n_rows <= 65536 ? 64 : 1024;
    const int64_t nb = (c->n_rows + NGCF_GROUP_ROWS - 1) / NGCF_GROUP_ROWS;
    int32_t *d_min = nullptr, *d_max = nullptr;
    std::vector<int32_t> h_min((size_t)nb), h_max((size_t)nb);
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&d_min, sizeof(int32_t) * (size_t)nb));
        HIP_TRY(hipMalloc(&d_max, sizeof(int32_t) * (size_t)nb));
        HIP_TRY(hipMemsetAsync(d_min, 0x7f, sizeof(int32_t) * (size_t)nb, stream));   // 0x7f7f7f7f: large
        HIP_TRY(hipMemsetAsync(d_max, 0xff, sizeof(int32_t) * (size_t)nb, stream));   // -1
        row_colrange_kernel<<<(int)((c->n_rows + 255) / 256), 256, 0, stream>>>(c->rowptr, c->colidx, c->n_rows, (int)NGCF_GROUP_ROWS, d_min, d_max);
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
    // class of a row block: 2 = its rows gather from at most kLdsTableRows table rows (the user rows of a graph with a
    // few hundred items - the Seoul data has ~100: the table slice is staged in LDS), 1 = sliceable, 0 = neither
    auto block_class = [&](int64_t b) {
        if (h_max[(size_t)b] < 0) return 1;
        const int64_t range = (int64_t)h_max[(size_t)b] - h_min[(size_t)b] + 1;
        return range <= kLdsTableRows ? 2 : range <= kSliceFootprintRows ? 1 : 0;
    };
    std::vector<int> cls;
    for (int64_t b = 0; b < nb; ++b) {
        const int k = block_class(b);
        const int64_t lo = b * NGCF_GROUP_ROWS, hi = std::min<int64_t>(c->n_rows, lo + NGCF_GROUP_ROWS);
        if (!c->groups.empty() && cls.back() == k) {
            c->groups.back().end = hi;
        } else {
            c->groups.push_back({lo, hi, k >= 1, INT32_MAX, -1, k == 2});
            cls.push_back(k);
        }
    }
    // an LDS-table group must gather from ONE small range as a whole, not only block by block
    for (size_t i = 0; i < c->groups.size(); ++i) {
        if (!c->groups[i].lds_table) continue;
        int32_t lo = INT32_MAX, hi = -1;
        for (int64_t b = c->groups[i].begin / NGCF_GROUP_ROWS; b * NGCF_GROUP_ROWS < c->groups[i].end; ++b)
            if (h_max[(size_t)b] >= 0) lo = std::min(lo, h_min[(size_t)b]), hi = std::max(hi, h_max[(size_t)b]);
        if (hi >= 0 && (int64_t)hi - lo + 1 > kLdsTableRows) c->groups[i].lds_table = false;
    }
    // a group of a few blocks is not worth its own launch: give it its neighbour's class, then fuse equal neighbours
    // (LDS-table groups keep to themselves: on a Seoul-shaped graph they are most of the rows of a six-block matrix)
    auto same = [&](size_t a, size_t b) {
        return c->groups[a].sliceable == c->groups[b].sliceable && !c->groups[a].lds_table && !c->groups[b].lds_table;
    };
    for (size_t i = 0; i < c->groups.size(); ++i) {
        if (c->groups[i].lds_table || c->groups.size() == 1 || c->groups[i].end - c->groups[i].begin >= 16 * NGCF_GROUP_ROWS) continue;
        const size_t nb_i = i > 0 ? i - 1 : i + 1;
        if (!c->groups[nb_i].lds_table) c->groups[i].sliceable = c->groups[nb_i].sliceable;
    }
    for (size_t k = 1; k < c->groups.size();) {
        if (same(k, k - 1)) {
            c->groups[k - 1].end = c->groups[k].end;
            c->groups.erase(c->groups.begin() + (long)k);
        } else {
            ++k;
        }
    }
    for (auto &g : c->groups)                  // column range every group gathers from (the swept plan's first question)
        for (int64_t b = g.begin / NGCF_GROUP_ROWS; b * NGCF_GROUP_ROWS < g.end; ++b)
            if (h_max[(size_t)b] >= 0) {
                g.col_lo = std::min(g.col_lo, h_min[(size_t)b]);
                g.col_hi = std::max(g.col_hi, h_max[(size_t)b]);
            }
    return NGCF_OK;
}


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
    c->max_row_len = 0;
    for (int64_t r = 0; r < c->n_rows; ++r) {
        const int64_t len = rp[r + 1] - rp[r];
        c->max_row_len = std::max(c->max_row_len, len);
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
    if (c->mode >= 2) return build_swept_plan(c, stream);          // the parts follow the row groups and seg_len
    free_swept(c);
    return NGCF_OK;
}

extern "C" int ngcf_csr_set_mode(ngcf_csr_t *c, int mode, void *stream)
{
    if (!c || mode < 0 || mode > 3) return fail(NGCF_ERR_ARG, "ngcf_csr_set_mode: bad argument");
    c->mode = mode;
    if (mode < 2) {
        free_swept(c);
        return NGCF_OK;
    }
    if (c->swept.built_mode != mode) return build_swept_plan(c, (hipStream_t)stream);
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
    const int rc = ngcf_csr_plan(c, default_seg_len(c->nnz), stream);
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
        return ngcf_csr_plan(c, default_seg_len(c->nnz), stream);
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



// =============================================================================================
// ngcf_csr_filter - thinned copy of a CSR on the device (reference-mode node dropout, NGCF.py:93-100,124-126)
//
// The reference rebuilds a COO tensor from `indices[:, mask]` per layer and step.  Entries stay row-sorted under a mask, so
// the thinned CSR is a stream compaction: an exclusive scan of the keep flags over the source's entries gives every kept
// entry its new position and every row its new start (rowptr'[r] = pos[rowptr[r]]).  Three passes over the entries
// (count per 2048-entry block, scan of the block counts in one workgroup, scan inside the blocks + scatter), no host round
// trip, no allocation after the first call: the destination object is re-used step after step.
// The segment structure is the source's: a row cut into k segments there keeps k segments here (seg_begin' = rowptr'[row]
// + j*seg_len; trailing ones may be empty), so seg_row / heavy_row / heavy_seg_ptr are shared and no plan is built.  A row
// that fell to <= seg_len entries is then produced twice - as a row by the row kernels and from its segments by the fix-up,
// which runs last and writes the same sum.
// =============================================================================================
constexpr int kScanBlock = 2048;        // entries per workgroup (256 threads x 8)

__device__ inline int filter_keep(const uint8_t *__restrict__ keep, const int32_t *__restrict__ map, int64_t e)
{
    return keep[map ? (int64_t)map[e] : e] != 0;
}

__global__ __launch_bounds__(256) void filter_count_kernel(const uint8_t *__restrict__ keep, const int32_t *__restrict__ map,
                                                           int64_t nnz, int32_t *__restrict__ blk)
{
    __shared__ int wsum[4];
    const int64_t base = (int64_t)blockIdx.x * kScanBlock;
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t e = base + q * 256 + threadIdx.x;
        if (e < nnz) cnt += filter_keep(keep, map, e);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) cnt += __shfl_xor(cnt, m);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of blk[0..nb) in place, blk[nb] = total; one workgroup of 1024 threads walks the array in chunks
__global__ __launch_bounds__(1024) void filter_scan_blocks_kernel(int32_t *__restrict__ blk, int64_t nb)
{
    __shared__ int wtot[16];
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int x = i < nb ? blk[i] : 0;
        int incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        if (i < nb) blk[i] = before + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) blk[nb] = carry_s;
}

// pos[e] = kept entries before e (all e), kept entries copied to their place
__global__ __launch_bounds__(256) void filter_scatter_kernel(const uint8_t *__restrict__ keep, const int32_t *__restrict__ map,
                                                             const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                                             int64_t nnz, const int32_t *__restrict__ blk, int32_t *__restrict__ pos,
                                                             int32_t *__restrict__ out_col, float *__restrict__ out_val)
{
    __shared__ int wsum[4];
    const int64_t base = (int64_t)blockIdx.x * kScanBlock;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // thread t holds entries base + 8 t .. base + 8 t + 7 (consecutive: one scan over the thread totals orders them)
    int k[8], mine = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t e = base + (int64_t)threadIdx.x * 8 + q;
        k[q] = e < nnz ? filter_keep(keep, map, e) : 0;
        mine += k[q];
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int y = __shfl_up(incl, o);
        if (lane >= o) incl += y;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int p = blk[blockIdx.x] + incl - mine;
    for (int w = 0; w < wave; ++w) p += wsum[w];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int64_t e = base + (int64_t)threadIdx.x * 8 + q;
        if (e < nnz) {
            pos[e] = p;
            if (k[q]) {
                out_col[p] = colidx[e];
                out_val[p] = vals[e];
                ++p;
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) pos[nnz] = blk[gridDim.x];
}

__global__ void filter_rowptr_kernel(const int64_t *__restrict__ rowptr, int64_t n_rows, const int32_t *__restrict__ pos,
                                     int64_t *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_rows) out[r] = pos[rowptr[r]];
}

__global__ void filter_segbegin_kernel(const int64_t *__restrict__ src_rowptr, const int64_t *__restrict__ src_seg_begin,
                                       const int32_t *__restrict__ seg_row, int64_t n_seg, const int64_t *__restrict__ rowptr,
                                       int64_t *__restrict__ seg_begin)
{
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_seg) seg_begin[s] = rowptr[seg_row[s]] + (src_seg_begin[s] - src_rowptr[seg_row[s]]);
}

extern "C" int ngcf_csr_filter(const ngcf_csr_t *src, const uint8_t *keep, const int32_t *map, int64_t nnz_kept,
                               ngcf_csr_t **dst_inout, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!src || !dst_inout) return fail(NGCF_ERR_ARG, "ngcf_csr_filter: null argument");
    if (src->nnz > 0 && !keep) return fail(NGCF_ERR_ARG, "ngcf_csr_filter: null keep flags");
    if (src->nnz >= ((int64_t)1 << 31) - 1) return fail(NGCF_ERR_ARG, "ngcf_csr_filter: more than 2^31 stored entries");
    if (nnz_kept > src->nnz) return fail(NGCF_ERR_ARG, "ngcf_csr_filter: nnz_kept exceeds the source's entries");
    ngcf_csr *d = *dst_inout;
    const int64_t nb = (src->nnz + kScanBlock - 1) / kScanBlock;
    if (d && (d->n_rows != src->n_rows || d->n_cols != src->n_cols || d->cap < src->nnz || !d->owns || d->pos_len < src->nnz + 1 ||
              d->n_seg != src->n_seg)) {
        ngcf_csr_free(d);
        d = nullptr;
        *dst_inout = nullptr;
    }
    if (!d) {                                        // first call for this destination: the only allocations
        d = new ngcf_csr();
        d->n_rows = src->n_rows;
        d->n_cols = src->n_cols;
        d->owns = true;
        d->cap = std::max<int64_t>(src->nnz, 1);
        d->pos_len = src->nnz + 1;
        auto body = [&]() -> int {
            HIP_TRY(hipMalloc(&d->rowptr, sizeof(int64_t) * (size_t)(d->n_rows + 1)));
            HIP_TRY(hipMalloc(&d->colidx, sizeof(int32_t) * (size_t)d->cap));
            HIP_TRY(hipMalloc(&d->vals, sizeof(float) * (size_t)d->cap));
            HIP_TRY(hipMalloc(&d->pos, sizeof(int32_t) * (size_t)d->pos_len));
            HIP_TRY(hipMalloc(&d->scan_blk, sizeof(int32_t) * (size_t)(nb + 2)));
            if (src->n_seg > 0) HIP_TRY(hipMalloc(&d->seg_begin, sizeof(int64_t) * (size_t)src->n_seg));
            return NGCF_OK;
        };
        const int rc = body();
        if (rc != NGCF_OK) {
            ngcf_csr_free(d);
            return rc;
        }
    }
    // plan: the source's segment structure and row groups (a subset of the entries gathers from a subset of the columns)
    d->seg_len = src->seg_len;
    d->n_seg = src->n_seg;
    d->n_heavy = src->n_heavy;
    d->seg_row = src->seg_row;
    d->heavy_row = src->heavy_row;
    d->heavy_seg_ptr = src->heavy_seg_ptr;
    d->borrows_plan = true;
    d->filter_src = src;
    d->groups = src->groups;
    d->max_row_len = src->max_row_len;               // upper bound
    d->mode = 0;
    d->nnz = nnz_kept >= 0 ? nnz_kept : src->nnz;    // exact when the caller knows it (it drew the mask), else an upper bound
    if (src->nnz == 0) {
        HIP_TRY(hipMemsetAsync(d->rowptr, 0, sizeof(int64_t) * (size_t)(d->n_rows + 1), stream));
        *dst_inout = d;
        return NGCF_OK;
    }
    filter_count_kernel<<<dim3((unsigned)nb), 256, 0, stream>>>(keep, map, src->nnz, d->scan_blk);
    LAUNCH_CHECK();
    filter_scan_blocks_kernel<<<1, 1024, 0, stream>>>(d->scan_blk, nb);
    LAUNCH_CHECK();
    filter_scatter_kernel<<<dim3((unsigned)nb), 256, 0, stream>>>(keep, map, src->colidx, src->vals, src->nnz, d->scan_blk, d->pos,
                                                                 d->colidx, d->vals);
    LAUNCH_CHECK();
    filter_rowptr_kernel<<<(unsigned)((d->n_rows + 1 + 255) / 256), 256, 0, stream>>>(src->rowptr, d->n_rows, d->pos, d->rowptr);
    LAUNCH_CHECK();
    if (d->n_seg > 0) {
        filter_segbegin_kernel<<<(unsigned)((d->n_seg + 255) / 256), 256, 0, stream>>>(src->rowptr, src->seg_begin, d->seg_row, d->n_seg,
                                                                                     d->rowptr, d->seg_begin);
        LAUNCH_CHECK();
    }
    *dst_inout = d;
    return NGCF_OK;
}

extern "C" const int32_t *ngcf_csr_filter_pos(const ngcf_csr_t *c) { return c ? c->pos : nullptr; }

// map_out[pos_t[j]] = pos_l[map[j]] for every kept entry j of the transposed source: where the entries of the thinned transpose
// sit in the thinned matrix (the map the NEXT layer's thinning of the transpose needs)
__global__ void filter_remap_kernel(const uint8_t *__restrict__ keep, const int32_t *__restrict__ map, int64_t nnz,
                                    const int32_t *__restrict__ pos_t, const int32_t *__restrict__ pos_l, int32_t *__restrict__ map_out)
{
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; j < nnz; j += stride) {
        const int32_t e = map[j];
        if (keep[e]) map_out[pos_t[j]] = pos_l[e];
    }
}

extern "C" int ngcf_csr_filter_remap(const ngcf_csr_t *dst_t, const uint8_t *keep, const int32_t *map, int64_t n_src_entries,
                                     const int32_t *pos_l, int32_t *map_out, void *stream)
{
    if (!dst_t || !dst_t->pos) return fail(NGCF_ERR_ARG, "ngcf_csr_filter_remap: not a filtered CSR");
    if (n_src_entries == 0) return NGCF_OK;
    if (!keep || !map || !pos_l || !map_out || n_src_entries + 1 > dst_t->pos_len)
        return fail(NGCF_ERR_ARG, "ngcf_csr_filter_remap: bad argument");
    filter_remap_kernel<<<grid_for(n_src_entries, 256), 256, 0, (hipStream_t)stream>>>(keep, map, n_src_entries, dst_t->pos, pos_l, map_out);
    LAUNCH_CHECK();
    return NGCF_OK;
}
