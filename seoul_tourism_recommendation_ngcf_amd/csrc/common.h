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


static const int32_t kDefaultSegLen = 2048;   // measured on C3: 512 -> 20.9 ms/step, 2048 -> 20.6, 4096 -> 20.5

void free_swept(ngcf_csr *c);                                  // spmm_swept.hip
int build_swept_plan(ngcf_csr *c, hipStream_t stream);         // spmm_swept.hip
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
// cumulative over layers): entry e survives layer k iff mix32(seed_j ^ e*K) >= thr for every j <= k.
// `eid` maps the entries of a transposed CSR back to the entry numbers of L (NULL: the entry position itself).
struct EdgeDrop {
    int n;                  // number of seeds (0 = no dropout)
    uint32_t thr;           // p * 2^32
    uint64_t seed[4];
    const int64_t *eid;
};


int spmm_dispatch(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo, void *workspace,
                  int64_t workspace_bytes, hipStream_t stream, const EdgeDrop &dr = EdgeDrop{0, 0, {0, 0, 0, 0}, nullptr});
// swept kernel launch (spmm_swept.hip); returns NGCF_OK or an error code
int launch_swept(const ngcf_csr *c, const float *E, int64_t ldE, int d, float *out, int64_t ldo, float *partial, int dp,
                 hipStream_t stream);

// ---------------------------------------------------------------------------------------------
// small device helpers
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

__device__ inline float wave_sum(float x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
    return x;
}


#endif  // NGCF_COMMON_H
