// Device-side building blocks of the row-wise SpMM (shared by spmm.hip and tools/spmm_lab.hip).
#ifndef NGCF_SPMM_DEVICE_H
#define NGCF_SPMM_DEVICE_H

#include "common.h"

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
// The entries a wave holds in registers - lane i has the i-th (column, value) pair, `cnt` of them - are walked G = 64/LPR at a
// time, U wave instructions in flight; `coff[ch]` is the column offset of the lane's ch-th piece of a gathered row.
template <int VEC, int LPR, int CH, int U>
__device__ inline void spmm_consume(int c, float v, int cnt, const float *__restrict__ E, int64_t ldE, const int (&coff)[CH],
                                    typename VecT<VEC>::type (&acc)[CH])
{
    using V = typename VecT<VEC>::type;
    constexpr int G = 64 / LPR;
    const int g = (threadIdx.x & 63) / LPR;
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

// the G lane groups of a wave hold private partial sums: combine them (every lane ends up with the total of its piece)
template <int VEC, int LPR, int CH>
__device__ inline void spmm_combine(typename VecT<VEC>::type (&acc)[CH])
{
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) acc[ch] = vadd(acc[ch], vshfl_xor(acc[ch], off));
}

template <int VEC, int LPR, int CH, int U>
__device__ inline void spmm_accumulate(const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                       int64_t begin, int64_t end, const float *__restrict__ E, int64_t ldE,
                                       int d, typename VecT<VEC>::type (&acc)[CH], const EdgeDropR &dr = EdgeDropR{0, 0, 0, 0, 0, 0, 0},
                                       int pad_col = 0, int64_t row = 0)     // row: the CSR row being walked (edge dropout key)
{
    const int lane = threadIdx.x & 63;
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
        int c = pad_col;   // a valid row of E (column 0 unless the caller's table starts elsewhere): padding slots read it, masked below
        float v = 0.f;
        if (lane < cnt) {
            c = colidx[base + lane];
            v = vals[base + lane];
        }
        if (dr.n > 0) {
            // drop entries, then compact the survivors to the low lanes (dropped ones go to the top, unused)
            const bool keep = lane < cnt && edge_keep(dr, row, c);
            const unsigned long long m = __ballot(keep);
            const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
            const int dst = keep ? __popcll(m & lt) : 63 - __popcll(~m & lt);
            c = __builtin_amdgcn_ds_permute(dst << 2, c);
            v = __int_as_float(__builtin_amdgcn_ds_permute(dst << 2, __float_as_int(v)));
            cnt = __popcll(m);
        }
        spmm_consume<VEC, LPR, CH, U>(c, v, cnt, E, ldE, coff, acc);
    }
    spmm_combine<VEC, LPR, CH>(acc);
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
        for (int64_t s = s0; s < s1; s += 8) {                  // eight partial sums in flight, added in segment order
            V t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const V *>(partial + (s + k < s1 ? s + k : s) * (int64_t)dp + o);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (s + k < s1) acc = vadd(acc, t[k]);
        }
        *reinterpret_cast<V *>(dst + o) = acc;
    }
}



#endif  // NGCF_SPMM_DEVICE_H
