// Shared host/device declarations of libngcf_hip.so (internal; the public C ABI is include/ngcf_hip.h).
#ifndef NGCF_COMMON_H
#define NGCF_COMMON_H

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
inline thread_local char g_err[512] = "";

inline int fail(int code, const char *fmt, ...)
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


// ---------------------------------------------------------------------------------------------
// options: every tunable of the dispatch code in ONE struct.  csr.hip fills it once per process (the only getenv() of the
// library is in ngcf_options_from_env there: launches never touch the environment) and ngcf_set_option changes single
// fields at run time (tests, tools/).  Fields marked LAB select code that is compiled only with -DNGCF_LAB.
// ---------------------------------------------------------------------------------------------
struct NgcfOptions {
    // spmm.hip
    int no_fork = 0;               // NGCF_NO_FORK: never record the two halves of a row-wise product as parallel graph branches
    int no_slicing = 0;            // NGCF_NO_SLICING: no d-sliced launches of the sliceable row groups
    int no_ldstab = 0;             // NGCF_NO_LDSTAB: no table-in-LDS kernel
    int no_panel_split = 0;        // NGCF_NO_PANEL_SPLIT: odd widths in one piece on the scalar kernels
    int no_tail_table = 0;         // NGCF_NO_TAIL_TABLE: tail columns gathered out of the strided table
    int no_pad_product = 0;        // NGCF_NO_PAD_PRODUCT: small matrices: do not multiply the padding columns along
    int64_t fork_min = 200000000;  // NGCF_FORK_MIN: entry-columns from which the halves are forked under capture
    // dense.hip
    int dense_direct = 1;          // NGCF_DENSE_DIRECT: 0 never, 1 up to 8 192 rows, 2 at any row count
    int dense_resident = 1;        // NGCF_DENSE_RESIDENT: 0 keeps the staged kernel; (LAB) 2: layer_dense_resident_il_kernel where it applies
    int dense_resident_min_rows = 106496;  // NGCF_DENSE_RESIDENT_MIN_ROWS: rows from which the weights-resident kernel runs (65 536 = one round of its 2 048 waves; measured cross-over, profiles/r04_dense_rows_lab.txt)
    int dense_small_tiles = 1;     // NGCF_DENSE_SMALL_TILES: 32-row tiles for <= 128 output columns on <= 16 384 rows
    int dense_tall = 1;            // NGCF_DENSE_TALL: 256 / 512 output columns as 96-row x 128-column workgroups: 0 never, 1 where measured faster, 2 always
    int dense_il_lab = 0;          // NGCF_DENSE_IL_LAB (LAB): the interleaved kernel taken apart: 1 no stores, 2 no loads, 3 neither
    // backward.hip
    int t_rows_bitmap = 1;         // NGCF_T_ROWS_BITMAP: 0 keeps the row-sparse transposed product on the slot-table kernel at every size
    int bwd_input_resident = 1;    // NGCF_BWD_INPUT_RESIDENT: 0 keeps the staged input-gradient kernel at every size
    // csr.hip
    int slice_max_mb = 48;         // NGCF_SLICE_MAX_MB: largest table slice a d-sliced group may gather from
    // spmm_swept.hip (plan)
    int swept_lpe = 16;            // NGCF_SWEPT_LPE: lanes per entry of the swept plan: 16 = 64-float slices, 32 = 128-float slices
    int swept_waves = 0;           // NGCF_SWEPT_WAVES (LAB: 8)
    int swept_cut = 4;             // NGCF_SWEPT_CUT: rows are cut at 1/cut of a wave task's share
    int swept_no_moments = 0;      // NGCF_SWEPT_NO_MOMENTS
    int swept_order_rows = 0;      // NGCF_SWEPT_ORDER=rows
    int swept_debug = 0;           // NGCF_SWEPT_DEBUG: print the plan
    int swept_window_kb = 0;       // NGCF_SWEPT_WINDOW_KB: 0 = by table size
    // spmm_swept.hip (launch)
    int swept_spin = 500;          // NGCF_SWEPT_SPIN
    int swept_lead = -2;           // NGCF_SWEPT_LEAD: -2 = by table size, -1 = no synchronisation
    int swept_sync_every = 1;      // NGCF_SWEPT_SYNC_EVERY
    int swept_prio_kb = 256;       // NGCF_SWEPT_PRIO_KB
    int swept_prio_graded = 1;     // NGCF_SWEPT_PRIO_GRADED
    int swept_nt = 0;              // NGCF_SWEPT_NT (LAB)
    int swept_merge = 0;           // NGCF_SWEPT_MERGE (LAB)
    char swept_trace[480] = "";    // NGCF_SWEPT_TRACE (LAB): file prefix of the sweep-spread trace
};
extern NgcfOptions g_opts;                      // csr.hip
const NgcfOptions &ngcf_opts();                 // csr.hip: reads the environment on first use


// per-device state (function attributes, side streams) lives in small arrays indexed by the current device
constexpr int kMaxDevices = 64;
inline int current_device_slot()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev % kMaxDevices;
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }


inline int grid_for(int64_t n, int block)

{
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 16) g = 256 * 16;
    return (int)g;
}



// ---------------------------------------------------------------------------------------------
// CSR object (csr.hip owns its lifetime; spmm*.hip read it)
// ---------------------------------------------------------------------------------------------
struct ngcf_csr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    int64_t max_row_len = 0;     // stored entries of the longest row (set by the segmentation plan)
    int64_t *rowptr = nullptr;   // device [n_rows+1]
    int32_t *colidx = nullptr;   // device [nnz]
    float *vals = nullptr;       // device [nnz]
    bool owns = false;
    // a thinned copy made by ngcf_csr_filter (csr.hip): capacity of colidx / vals, the exclusive scan of the keep flags over the
    // source's entries (pos[e] = kept entries before e, pos[src nnz] = kept in all), scan scratch; seg_row / heavy_row /
    // heavy_seg_ptr are then BORROWED from the source (same segment structure, own seg_begin) and nnz may be an upper bound
    int64_t cap = 0, pos_len = 0;
    int32_t *pos = nullptr;
    int32_t *scan_blk = nullptr;
    bool borrows_plan = false;
    const struct ngcf_csr *filter_src = nullptr;
    // row segmentation: rows with > seg_len entries are cut into segments
    int32_t seg_len = 0;
    int64_t n_seg = 0, n_heavy = 0;
    int32_t *seg_row = nullptr;        // device [n_seg]   row of each segment
    int64_t *seg_begin = nullptr;      // device [n_seg]   first entry of each segment
    int32_t *heavy_row = nullptr;      // device [n_heavy] rows that were cut
    int64_t *heavy_seg_ptr = nullptr;  // device [n_heavy+1] their segment ranges
    // row groups: maximal runs of rows whose gathered column range is small enough that d-slicing pays
    struct RowGroup {                   // col_hi < col_lo: no entries
        int64_t begin, end;
        bool sliceable;
        int32_t col_lo, col_hi;
        bool lds_table = false;        // the rows gather from <= kLdsTableRows table rows: spmm_ldstab_kernel
    };
    std::vector<RowGroup> groups;
    // L2-swept plan (spmm_swept.hip): built on request (mode 2/3) for long-lived matrices
    int mode = 0;                      // 0 row-wise (+ d-sliced groups), 1 row-wise only, 2 swept wherever the shape
                                       // allows, 3 swept on the row groups where the plan expects L2 re-use to pay
    struct SegSet {                    // rows cut into segments (device arrays, same meaning as seg_row.. above)
        int64_t n_seg = 0, n_heavy = 0;
        int32_t *seg_row = nullptr;
        int64_t *seg_begin = nullptr;
        int32_t *heavy_row = nullptr;
        int64_t *heavy_seg_ptr = nullptr;
    };
    struct Swept {
        struct Part {                      // one row group handled by the swept kernel
            int64_t row_lo = 0, row_hi = 0;
            int waves = 0, rows_per_wave = 0;          // workgroup shape the entry lists were laid out for
            int lpe = 16;                              // lanes per entry (16: 64-float slices, 32: 128-float slices)
            int32_t n_rowpass = 0, n_win = 0, win_cols = 0, col_lo = 0, col_hi = 0;
            int64_t n_slots = 0, n_partial = 0, n_heavy = 0, partial_base = 0;   // partial rows: workspace rows base..base+n
            int64_t *tptr = nullptr;           // device [n_tasks*n_win+1] slot range of every (wave task, column window)
            int32_t *e_pack = nullptr;         // device [n_slots] (local row << 25) | column; negative: empty slot
            float *e_val = nullptr;            // device [n_slots]
            int32_t *dst = nullptr;            // device [n_tasks*rows_per_wave] >=0 output row, <=-2 the part's partial row -2-p, -1 unused
            int32_t *prow = nullptr;           // device [n_tasks*rows_per_wave] the matrix row behind every accumulator row (cut rows too)
            int32_t *heavy_row = nullptr;      // device [n_heavy]   rows cut into pieces
            int64_t *heavy_seg_ptr = nullptr;  // device [n_heavy+1] their ranges of the part's partial rows
        };
        std::vector<Part> parts;
        std::vector<char> group_swept;     // per entry of `groups`: handled by a part
        SegSet out;                        // segments of the cut rows that no part covers
        int64_t n_partial = 0;             // partial rows of all parts (they follow out.n_seg in the workspace)
        // device [kSweptStreams][8*32] per-XCD sweep counters (zeroed before each launch): one block per stream that has
        // launched products of this CSR, so two streams multiplying the same matrix at once do not pace each other's sweep
        uint32_t *barrier = nullptr;
        mutable hipStream_t barrier_owner[4] = {nullptr, nullptr, nullptr, nullptr};
        mutable int barrier_used = 0;
        int built_mode = 0;
        int lpe = 16;                      // geometry of the parts (spmm_swept.hip): lanes per entry
    } swept;
};


static const int32_t kLdsTableRows = 512;      // x 256 B (one 64-float slice) = 128 KiB of LDS
static const int32_t kDefaultSegLen = 2048;   // measured on C3: 512 -> 20.9 ms/step, 2048 -> 20.6, 4096 -> 20.5
// Default segment length of a matrix with `nnz` stored entries.  A wave walks its segment 8 gathers at a time, so a small matrix
// with a few very long rows (the Seoul graph: 100 item rows of ~4 400 entries) is latency-bound on a few hundred waves at 2048;
// aim at >= 4 096 segments instead.  Seoul-shaped C2 / C1 (876 K entries, whole forward as a hipGraph): 2048 -> 0.967 / 0.232 ms,
// 512 -> 0.607 / 0.145, 256 -> 0.577 / 0.139, 128 -> 0.621 / 0.149, 64 -> 0.652 / 0.169.
static inline int32_t default_seg_len(int64_t nnz)
{
    int32_t s = 64;
    while (s < kDefaultSegLen && (int64_t)s * 4096 < nnz) s *= 2;
    return s;
}

void free_swept(ngcf_csr *c);                                  // spmm_swept.hip
int build_swept_plan(ngcf_csr *c, hipStream_t stream);         // spmm_swept.hip (reads c->mode)
bool swept_usable(const ngcf_csr *c, int64_t ldE, int d);       // spmm_swept.hip: can this call use the parts?
void prof_mark(hipStream_t stream, int which);                 // spmm.hip: hipEvent around the SpMM launches

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
// cumulative over layers): the entry (i, j) of L survives layer k iff mix32(seed_q ^ key(i, j)) >= thr for every q <= k.
// The key is the entry's (row, column) in L - not its position in some storage order - so every kernel that walks L in
// whatever layout (row-wise CSR, the L2-swept plan, the rows of L^T, a scatter over selected rows) thins it the same way
// without a map between layouts; `transposed` says that the CSR being walked is L^T (its rows are L's columns).
// A 64-bit dropout seed is either a value or - top 16 bits == kSeedPtrTag - the device address (low 48 bits) of a uint64_t that
// holds the value: a captured hipGraph bakes its kernel arguments in, and with the seeds behind a pointer every replay still draws
// new masks (the mirror advances the words on the device).  Values handed over by the mirror are below 2^62, so they never
// carry the tag.  Kernels resolve a seed ONCE, at their start.
constexpr uint64_t kSeedPtrTag = 0xD5EDull;
__device__ inline uint64_t resolve_seed(uint64_t s)
{
    if ((s >> 48) == kSeedPtrTag) s = *reinterpret_cast<const uint64_t *>(s & 0xFFFFFFFFFFFFull);
    // the same in every lane by construction: say so, and the value stays in scalar registers (as a by-value kernel argument would)
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)s), hi = __builtin_amdgcn_readfirstlane((uint32_t)(s >> 32));
    return ((uint64_t)hi << 32) | lo;
}

struct EdgeDrop {
    int n;                  // number of seeds (0 = no dropout)
    uint32_t thr;           // p * 2^32
    uint64_t seed[4];
    int transposed;
};

// The kernel-side form: the seeds resolved (tagged addresses read once, at kernel start) into NAMED scalars - an array indexed by
// a loop counter in a local copy of the argument would live in scratch memory (measured: spmm_kernel 12.6 -> 28.7 us on the Seoul
// graph); the argument struct itself stays in the kernarg segment.
struct EdgeDropR {
    int n;
    uint32_t thr;
    uint64_t s0, s1, s2, s3;
    int transposed;
};

__device__ inline EdgeDropR resolve_drop(const EdgeDrop &dr)
{
    EdgeDropR r;
    r.n = dr.n;
    r.thr = dr.thr;
    r.transposed = dr.transposed;
    r.s0 = dr.n > 0 ? resolve_seed(dr.seed[0]) : 0;
    r.s1 = dr.n > 1 ? resolve_seed(dr.seed[1]) : 0;
    r.s2 = dr.n > 2 ? resolve_seed(dr.seed[2]) : 0;
    r.s3 = dr.n > 3 ? resolve_seed(dr.seed[3]) : 0;
    return r;
}

__device__ inline bool edge_keep(const EdgeDropR &dr, int64_t row, int64_t col)   // row, col of the CSR being walked
{
    const uint64_t i = (uint64_t)(dr.transposed ? col : row), j = (uint64_t)(dr.transposed ? row : col);
    const uint64_t e = ((i << 32) | (j & 0xffffffffull)) * 0x9E3779B97F4A7C15ULL;
    bool keep = true;
    if (dr.n > 0) keep = mix32(dr.s0 ^ e) >= dr.thr;
    if (dr.n > 1) keep = keep && mix32(dr.s1 ^ e) >= dr.thr;
    if (dr.n > 2) keep = keep && mix32(dr.s2 ^ e) >= dr.thr;
    if (dr.n > 3) keep = keep && mix32(dr.s3 ^ e) >= dr.thr;
    return keep;
}


int spmm_dispatch(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo, void *workspace,
                  int64_t workspace_bytes, hipStream_t stream, const EdgeDrop &dr = EdgeDrop{0, 0, {0, 0, 0, 0}, 0});
// swept parts (spmm_swept.hip): kernels + fix-ups of every part; `partial` is the workspace base (rows of dp floats)
int launch_swept(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo, float *partial, int dp,
                 hipStream_t stream, const EdgeDrop &dr);

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// value of lane U of this lane's 16-lane row (DPP row_newbcast: one VALU op, no LDS round trip)
template <int U> __device__ inline int row_bcast(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + U, 0xf, 0xf, false);
}
template <int U> __device__ inline float row_bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + U, 0xf, 0xf, false));
}

template <int VEC> struct VecT;
template <> struct VecT<4> { using type = float4; };
template <> struct VecT<2> { using type = float2; };
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
__device__ inline float2 vfma(float s, float2 x, float2 a)
{
    a.x = fmaf(s, x.x, a.x);
    a.y = fmaf(s, x.y, a.y);
    return a;
}
__device__ inline float2 vadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ inline float2 vshfl_xor(float2 a, int m)
{
    a.x = __shfl_xor(a.x, m);
    a.y = __shfl_xor(a.y, m);
    return a;
}
__device__ inline float4 vsel(bool p, float4 a, float4 b) { return p ? a : b; }
__device__ inline float vsel(bool p, float a, float b) { return p ? a : b; }
__device__ inline float4 vzero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <int VEC> __device__ inline typename VecT<VEC>::type vzero();
template <> __device__ inline float4 vzero<4>() { return vzero4(); }
template <> __device__ inline float vzero<1>() { return 0.f; }
template <> __device__ inline float2 vzero<2>() { return make_float2(0.f, 0.f); }
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

__device__ inline float wave_sum(float x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}


#endif  // NGCF_COMMON_H
