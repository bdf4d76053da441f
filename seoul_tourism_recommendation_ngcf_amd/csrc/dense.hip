// dense.hip - the dense half of a layer on the fp32 matrix cores, and the fused layer entry point.
#include "common.h"
#include <type_traits>

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
#ifndef NGCF_DC
#define NGCF_DC 16
#endif
#define NGCF_KC (2 * NGCF_DC)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));   // native vector: stays in registers inside lambdas

__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ W1, const float *__restrict__ b1,
                                                           const float *__restrict__ W2, const float *__restrict__ b2, int d_in,
                                                           int d_out, int n_chunks, int DOP, int NT, float *__restrict__ Wt,
                                                           float *__restrict__ bias2)
{
    // Wt[chunk][kl][cw][j][t] = weight of output column (cw*NT + t)*32 + j: a lane reads its NT tile values at once.
    // One workgroup per (chunk, 32 packed columns): the 16 input columns of the chunk are read along the rows of W1 / W2
    // (64-byte runs), turned in LDS, and leave as 128-byte runs of the packed row.
    __shared__ float tile[NGCF_KC][33];
    const int groups = DOP / 32;
    const int chunk = blockIdx.x / groups, g = blockIdx.x % groups;
    for (int idx = threadIdx.x; idx < 32 * NGCF_KC; idx += 256) {
        const int w = idx / NGCF_KC, kl = idx % NGCF_KC;
        const int within = g * 32 + w;
        const int t = within % NT, j = (within / NT) % 32, cw = within / (NT * 32);
        const int oc = (cw * NT + t) * 32 + j;
        const int col = chunk * NGCF_DC + (kl % NGCF_DC);
        float v = 0.f;
        if (oc < d_out && col < d_in) v = (kl < NGCF_DC ? W1 : W2)[(int64_t)oc * d_in + col];
        tile[kl][w] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 32 * NGCF_KC; idx += 256) {
        const int kl = idx / 32, w = idx % 32;
        Wt[((int64_t)chunk * NGCF_KC + kl) * DOP + g * 32 + w] = tile[kl][w];
    }
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < DOP; j += 256)
            bias2[j] = j < d_out ? (b1[j] + b1[j]) + b2[j] : 0.f;   // b1 is added twice, NGCF.py:131,133
}

#ifndef NGCF_DENSE_WAVES_PER_EU
#define NGCF_DENSE_WAVES_PER_EU 1
#endif
template <int RW, int CW, int NT, bool ALIGNED, bool FAST>
__global__ __launch_bounds__(256, NGCF_DENSE_WAVES_PER_EU) void layer_dense_kernel(const float *__restrict__ LE, int64_t ldLE,
                                                          const float *__restrict__ Es, int64_t ldE, int64_t n_rows,
                                                          int d_in, int d_out, const float *__restrict__ Wt,
                                                          const float *__restrict__ bias2, int n_chunks,
                                                          float leaky, float drop_p, uint64_t drop_seed_in,
                                                          const float *__restrict__ drop_mask, int64_t ldm,
                                                          float *__restrict__ carry, int64_t ldc,
                                                          float *__restrict__ norm, int64_t ldn)
{
    const uint64_t drop_seed = drop_p > 0.f ? resolve_seed(drop_seed_in) : drop_seed_in;
    constexpr int BM = 32 * RW;
    constexpr int WCOLS = 32 * NT * CW;      // == DOP
    constexpr int XLD = NGCF_KC + 4;         // 36: rows stay 16-B aligned and b128 column reads are conflict-free
    constexpr bool DB = WCOLS <= 128;        // double-buffered LDS (one barrier per chunk) where two blocks still fit a CU
    constexpr int NBUF = DB ? 2 : 1;
    constexpr int SQ = NGCF_DC / 4;          // lanes that cover the input columns of a chunk for one row
    constexpr int RPP = 256 / SQ;            // rows staged per pass of the workgroup
    constexpr int RR = (BM + RPP - 1) / RPP; // X rows staged per thread
    constexpr int WF4 = NGCF_KC * WCOLS / 4;           // float4s of a W chunk
    constexpr int WN = (WF4 + 255) / 256;              // W float4s staged per thread
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
    const int sq = tid % SQ;
    const int sr = tid / SQ;   // 0..RPP-1
    f32x4 xle[RR], xe[RR], wreg[WN];

    auto load_chunk = [&](int chunk) {       // global -> registers
        const int c0 = chunk * NGCF_DC + sq * 4;
        if constexpr (FAST) {   // 16-byte aligned rows padded to a multiple of 4 floats: one float4 per lane and operand
            // branch-free: rows past the end re-read the last row (never stored) and columns past d_in re-read the last
            // float4 of the row and are zeroed (the padding columns of LE hold no defined values), so the loads stay in
            // flight under the MFMAs of the current chunk instead of being waited for inside a conditional
            const int d4 = (d_in + 3) & ~3;
            const int cc = c0 < d4 ? c0 : d4 - 4;
            const bool k0 = c0 < d_in, k1 = c0 + 1 < d_in, k2 = c0 + 2 < d_in, k3 = c0 + 3 < d_in;
#pragma unroll
            for (int rr = 0; rr < RR; ++rr) {
                int64_t grow = row0 + (sr + rr * RPP) % BM;
                grow = grow < n_rows ? grow : n_rows - 1;
                f32x4 a = *reinterpret_cast<const f32x4 *>(LE + grow * ldLE + cc);
                f32x4 b = *reinterpret_cast<const f32x4 *>(Es + grow * ldE + cc);
                a.x = k0 ? a.x : 0.f; a.y = k1 ? a.y : 0.f; a.z = k2 ? a.z : 0.f; a.w = k3 ? a.w : 0.f;
                b.x = k0 ? b.x : 0.f; b.y = k1 ? b.y : 0.f; b.z = k2 ? b.z : 0.f; b.w = k3 ? b.w : 0.f;
                xle[rr] = a;
                xe[rr] = b;
            }
            const f32x4 *srcw = reinterpret_cast<const f32x4 *>(Wt + (int64_t)chunk * NGCF_KC * WCOLS);
#pragma unroll
            for (int i = 0; i < WN; ++i)
                if (WF4 % 256 == 0 || tid + i * 256 < WF4) wreg[i] = srcw[tid + i * 256];
            return;
        }
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
            const int r = sr + rr * RPP;
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
        for (int i = 0; i < WN; ++i)
            if (WF4 % 256 == 0 || tid + i * 256 < WF4) wreg[i] = src[tid + i * 256];
    };
    auto store_chunk = [&](int buf) {        // registers -> LDS: (LE + E) feeds W1, (LE * E) feeds W2
        float *X = Xs + buf * (BM * XLD);
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) {
            const int r = sr + rr * RPP;
            if (r < BM) {
                const f32x4 a = xle[rr], b = xe[rr];
                *reinterpret_cast<f32x4 *>(X + r * XLD + sq * 4) = a + b;
                *reinterpret_cast<f32x4 *>(X + r * XLD + NGCF_DC + sq * 4) = a * b;
            }
        }
        f32x4 *dst = reinterpret_cast<f32x4 *>(Ws + buf * (NGCF_KC * WCOLS));
#pragma unroll
        for (int i = 0; i < WN; ++i)
            if (WF4 % 256 == 0 || tid + i * 256 < WF4) dst[tid + i * 256] = wreg[i];
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
        // one LDS buffer (256 / 512 output columns: a chunk of W is 32 / 64 KB): the next chunk waits in registers while this
        // one is multiplied, so the global loads are hidden and only the LDS stores sit between the two barriers
        load_chunk(0);
        store_chunk(0);
        __syncthreads();
        for (int chunk = 0; chunk < n_chunks; ++chunk) {
            const bool more = chunk + 1 < n_chunks;
            if (more) load_chunk(chunk + 1);
            compute_chunk(0);
            __syncthreads();
            if (more) {
                store_chunk(0);
                __syncthreads();
            }
        }
    }

    // ---- epilogue: bias, LeakyReLU, dropout, row sum of squares
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t drop_thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    float rowss[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rowss[r] = 0.f;
    if (!drop_mask && !(drop_p > 0.f)) {
        // eval mode / no message dropout: the same loop without the two per-element tests of the dropout form
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = (cw * NT + t) * 32 + li;
            const float bz = bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r] + bz;
                v = v >= 0.f ? v : leaky * v;
                acc[t][r] = v;
                rowss[r] = fmaf(v, v, rowss[r]);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = (cw * NT + t) * 32 + li;
            const float bz = bias2[col];
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[t][r] + bz;
                v = v >= 0.f ? v : leaky * v;
                if (drop_mask) {       // "reference" mode: the noise tensor nn.Dropout drew on the host (0 or 1/(1-p)), NGCF.py:142
                    const int64_t grow = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    v *= (grow < n_rows && col < d_out) ? drop_mask[grow * ldm + col] : 0.f;
                } else if (drop_p > 0.f) {
                    const int64_t grow = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const uint32_t h = mix32(drop_seed ^ ((uint64_t)grow * 0x9E3779B97F4A7C15ULL + (uint64_t)col));
                    v = h < drop_thr ? 0.f : v * keep_scale;
                }
                acc[t][r] = v;
                rowss[r] = fmaf(v, v, rowss[r]);
            }
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
    // ---- stores: carry (un-normalised, feeds the next layer) and the normalised all_E block.  A full tile (every row and
    // column inside the matrix: all but the last workgroup) stores without the per-element tests.
    if (row0 + BM <= n_rows && d_out == WCOLS) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t grow = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const float inv = 1.f / fmaxf(sqrtf(rowss[r]), 1e-12f);   // F.normalize eps, NGCF.py:144
            float *nrow = norm + grow * ldn + cw * NT * 32 + li;
#pragma unroll
            for (int t = 0; t < NT; ++t) nrow[t * 32] = acc[t][r] * inv;
        }
        if (carry) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t grow = row0 + rw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float *crow = carry + grow * ldc + cw * NT * 32 + li;
#pragma unroll
                for (int t = 0; t < NT; ++t) crow[t * 32] = acc[t][r];
            }
        }
        return;
    }
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

// ---------------------------------------------------------------------------------------------
// The same layer with the packed weights RESIDENT in LDS and no barrier in the main loop (r02).
// layer_dense_kernel stages the rows of LE / E through LDS, chunk by chunk, for the four waves of a workgroup: one barrier per
// chunk, and the matrix pipe is busy only 51 % of the time (profiles/r02_dense_lab.txt).  Here a wave owns its 32 rows outright:
// each lane reads the two 16-byte pieces of LE and of E it needs for a chunk straight from global memory into the MFMA A layout
// (lane (li, lh) holds input columns c*16 + lh*4 + {0..3} and c*16 + 8 + lh*4 + {0..3} of row li), one chunk ahead; the only shared
// operand is the weight matrix, and all of it (2 d_in x 128 floats <= 147 KB) sits in LDS for the lifetime of a persistent
// workgroup of 8 waves (one per CU).  No __syncthreads after the prologue: a wave in its epilogue or waiting for memory leaves
// the matrix pipe to the other wave of its SIMD.  For d_out in (96, 128], d_in <= 144, 16-byte aligned padded rows.
// ---------------------------------------------------------------------------------------------
constexpr int kResWaves = 8, kResWGs = 256;

__global__ __launch_bounds__(kResWaves * 64) void layer_dense_resident_kernel(
    const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ Es, int64_t ldE, int64_t n_rows, int d_in, int d_out,
    const float *__restrict__ Wt, const float *__restrict__ bias2, int n_chunks, float leaky, float drop_p, uint64_t drop_seed_in,
    const float *__restrict__ drop_mask, int64_t ldm, float *__restrict__ carry, int64_t ldc, float *__restrict__ norm, int64_t ldn)
{
    const uint64_t drop_seed = drop_p > 0.f ? resolve_seed(drop_seed_in) : drop_seed_in;
    constexpr int NT = 4, WCOLS = 128;
    extern __shared__ float Wres[];                 // [n_chunks * 32][128], the layout of pack_weights_kernel
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    {   // prologue: the whole packed weight matrix, once per workgroup
        const f32x4 *src = reinterpret_cast<const f32x4 *>(Wt);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wres);
        const int n4 = n_chunks * NGCF_KC * WCOLS / 4;
        for (int i = tid; i < n4; i += kResWaves * 64) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t n_tiles = (n_rows + 31) / 32;
    const int d4 = (d_in + 3) & ~3;
    const float *W = Wres + li * NT;
    float bz[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bz[t] = bias2[t * 32 + li];
    // Stores and loads share one in-order counter (vmcnt), so a wave that stores its finished tile and THEN asks for the first
    // chunks of its next tile waits for all 256 stores to be acknowledged before its first MFMA: the per-tile cost that no
    // staggering of the waves could hide.  The first two chunks of the next tile are therefore requested BEFORE the epilogue of
    // the current one; by the time anything younger than the stores is waited for, two chunks of MFMAs have passed.
    const int last = n_chunks - 1;
    const int64_t tile_step = (int64_t)gridDim.x * kResWaves;
    auto row_of = [&](int64_t t) {                 // the lane's row of tile t (rows past the end re-read the last row, never stored)
        int64_t g = t * 32 + li;
        return g < n_rows ? g : n_rows - 1;
    };
    // the lane's four 16-byte pieces of a chunk: LE and E at columns c*16 + lh*4 (a) and c*16 + 8 + lh*4 (b); columns past
    // d_in are re-read from the row's last float4 and zeroed
    auto fetch = [&](const float *le_row, const float *e_row, int c, f32x4 &la, f32x4 &lb, f32x4 &ea, f32x4 &eb) {
        const int ca = c * NGCF_DC + lh * 4, cb = ca + 8;
        const int cca = ca < d4 ? ca : d4 - 4, ccb = cb < d4 ? cb : d4 - 4;
        la = *reinterpret_cast<const f32x4 *>(le_row + cca);
        ea = *reinterpret_cast<const f32x4 *>(e_row + cca);
        lb = *reinterpret_cast<const f32x4 *>(le_row + ccb);
        eb = *reinterpret_cast<const f32x4 *>(e_row + ccb);
    };
    // Two chunks of look-ahead in two fixed register sets (no rotation copies - a copy of a register that is still being
    // loaded is a wait): the sums and products of a chunk are formed first, which frees its set for the chunk after next.
    // Every prefetch is UNCONDITIONAL (past the end the last chunk is read again and never used): behind a branch the
    // compiler cannot count the loads in flight and waits for all of them (s_waitcnt vmcnt(0)) at the next use - the
    // look-ahead then exists in the source only.  The odd last chunk is peeled off the loop for the same reason.
    f32x4 la0, lb0, ea0, eb0, la1, lb1, ea1, eb1;
    int64_t tile = (int64_t)blockIdx.x * kResWaves + wave;
    {
        const int64_t g0 = row_of(tile < n_tiles ? tile : 0);
        fetch(LE + g0 * ldLE, Es + g0 * ldE, 0, la0, lb0, ea0, eb0);
        fetch(LE + g0 * ldLE, Es + g0 * ldE, last < 1 ? last : 1, la1, lb1, ea1, eb1);
    }
    for (; tile < n_tiles; tile += tile_step) {
        const int64_t row0 = tile * 32;
        const int64_t grow_l = row_of(tile);
        const float *le_row = LE + grow_l * ldLE, *e_row = Es + grow_l * ldE;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // sums and products of a chunk from its raw pieces; the zeroing of the columns past d_in happens HERE, at use - in
        // fetch() it would make the loads wait right where they are issued
        auto form = [&](int c, f32x4 la, f32x4 lb, f32x4 ea, f32x4 eb, f32x4 (&a4)[4]) {
            const int ca = c * NGCF_DC + lh * 4, cb = ca + 8;
            if (cb + 4 > d_in) {                      // only the last chunk of an odd width
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (ca + q >= d_in) la[q] = 0.f, ea[q] = 0.f;
                    if (cb + q >= d_in) lb[q] = 0.f, eb[q] = 0.f;
                }
            }
            a4[0] = la + ea, a4[1] = lb + eb, a4[2] = la * ea, a4[3] = lb * eb;   // k-blocks: sum 0-7, sum 8-15, product 0-7, product 8-15
        };
        auto chunk_mfma = [&](int c, const f32x4 (&a4)[4]) {
            const float *wc = W + (int64_t)c * NGCF_KC * WCOLS;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(wc + (kb * 8 + lh * 4 + sx) * WCOLS);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kb][sx], bv.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kb][sx], bv.y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kb][sx], bv.z, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kb][sx], bv.w, acc[3], 0, 0, 0);
                }
            }
        };
        int c = 0;
        for (; c + 1 < n_chunks; c += 2) {
            {
                f32x4 a4[4];
                form(c, la0, lb0, ea0, eb0, a4);
                fetch(le_row, e_row, c + 2 < last ? c + 2 : last, la0, lb0, ea0, eb0);   // in flight under two chunks of MFMAs
                __builtin_amdgcn_sched_barrier(0);      // (left alone the compiler sinks these loads to just before their use)
                chunk_mfma(c, a4);
            }
            {
                f32x4 a4[4];
                form(c + 1, la1, lb1, ea1, eb1, a4);
                fetch(le_row, e_row, c + 3 < last ? c + 3 : last, la1, lb1, ea1, eb1);
                __builtin_amdgcn_sched_barrier(0);
                chunk_mfma(c + 1, a4);
            }
        }
        if (c < n_chunks) {
            f32x4 a4[4];
            form(c, la0, lb0, ea0, eb0, a4);
            chunk_mfma(c, a4);
        }
        {   // the next tile's first two chunks, ahead of this tile's stores (the last tile of a wave re-reads its own)
            const int64_t gn = row_of(tile + tile_step < n_tiles ? tile + tile_step : tile);
            fetch(LE + gn * ldLE, Es + gn * ldE, 0, la0, lb0, ea0, eb0);
            fetch(LE + gn * ldLE, Es + gn * ldE, last < 1 ? last : 1, la1, lb1, ea1, eb1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue (wave-local): bias, LeakyReLU, dropout, row norm, stores
        const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        const uint32_t drop_thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
        float rowss[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rowss[r] = 0.f;
        const bool any_drop = drop_mask || drop_p > 0.f;
        if (!any_drop) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[t][r] + bz[t];
                    v = v >= 0.f ? v : leaky * v;
                    acc[t][r] = v;
                    rowss[r] = fmaf(v, v, rowss[r]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[t][r] + bz[t];
                    v = v >= 0.f ? v : leaky * v;
                    const int64_t grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (drop_mask) v *= (grow < n_rows && col < d_out) ? drop_mask[grow * ldm + col] : 0.f;
                    else {
                        const uint32_t h = mix32(drop_seed ^ ((uint64_t)grow * 0x9E3779B97F4A7C15ULL + (uint64_t)col));
                        v = h < drop_thr ? 0.f : v * keep_scale;
                    }
                    acc[t][r] = v;
                    rowss[r] = fmaf(v, v, rowss[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s2 = rowss[r];
            s2 += __shfl_xor(s2, 1);
            s2 += __shfl_xor(s2, 2);
            s2 += __shfl_xor(s2, 4);
            s2 += __shfl_xor(s2, 8);
            s2 += __shfl_xor(s2, 16);
            rowss[r] = s2;
        }
        if (row0 + 32 <= n_rows && d_out == WCOLS) {          // full tile: no per-element tests
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float inv = 1.f / fmaxf(sqrtf(rowss[r]), 1e-12f);   // F.normalize eps, NGCF.py:144
                float *nrow = norm + grow * ldn + li;
#pragma unroll
                for (int t = 0; t < NT; ++t) nrow[t * 32] = acc[t][r] * inv;
            }
            if (carry) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *crow = carry + (row0 + (r & 3) + 8 * (r >> 2) + 4 * lh) * ldc + li;
#pragma unroll
                    for (int t = 0; t < NT; ++t) crow[t * 32] = acc[t][r];
                }
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (grow >= n_rows) continue;
            const float inv = 1.f / fmaxf(sqrtf(rowss[r]), 1e-12f);   // F.normalize eps, NGCF.py:144
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 32 + li;
                if (col < d_out) {
                    const float v = acc[t][r];
                    if (carry) carry[grow * ldc + col] = v;
                    norm[grow * ldn + col] = v * inv;
                }
            }
        }
    }
}

#ifdef NGCF_LAB
// ---------------------------------------------------------------------------------------------
// LAB ONLY: layer_dense_resident_kernel with TWO row tiles per wave (64 rows x 128 columns, 128 accumulator registers): a B operand
// read from LDS feeds eight MFMAs instead of four (half the LDS read bytes per MFMA - MI355X_MICROARCH.md "DVFS give-back": what
// raises the clock is less energy per MFMA), raw operands one chunk (128 MFMAs) ahead in one register set per tile.  Same k order
// and epilogue arithmetic: bit-identical.  dense_resident = 3.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kResWaves * 64) void layer_dense_resident2_kernel(
    const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ Es, int64_t ldE, int64_t n_rows, int d_in, int d_out,
    const float *__restrict__ Wt, const float *__restrict__ bias2, int n_chunks, float leaky, float drop_p, uint64_t drop_seed_in,
    const float *__restrict__ drop_mask, int64_t ldm, float *__restrict__ carry, int64_t ldc, float *__restrict__ norm, int64_t ldn)
{
    const uint64_t drop_seed = drop_p > 0.f ? resolve_seed(drop_seed_in) : drop_seed_in;
    constexpr int NT = 4, WCOLS = 128, TP = 2;
    extern __shared__ float Wres[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(Wt);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wres);
        const int n4 = n_chunks * NGCF_KC * WCOLS / 4;
        for (int i = tid; i < n4; i += kResWaves * 64) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t n_pairs = (n_rows + 32 * TP - 1) / (32 * TP);
    const int d4 = (d_in + 3) & ~3;
    const float *W = Wres + li * NT + lh * 4 * WCOLS;
    float bz[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bz[t] = bias2[t * 32 + li];
    const int last = n_chunks - 1;
    const int64_t pair_step = (int64_t)gridDim.x * kResWaves;
    auto row_of = [&](int64_t pr, int u) {
        int64_t g = pr * (32 * TP) + u * 32 + li;
        return g < n_rows ? g : n_rows - 1;
    };
    auto fetch = [&](const float *le_row, const float *e_row, int c, f32x4 &la, f32x4 &lb, f32x4 &ea, f32x4 &eb) {
        const int ca = c * NGCF_DC + lh * 4, cb = ca + 8;
        const int cca = ca < d4 ? ca : d4 - 4, ccb = cb < d4 ? cb : d4 - 4;
        la = *reinterpret_cast<const f32x4 *>(le_row + cca);
        ea = *reinterpret_cast<const f32x4 *>(e_row + cca);
        lb = *reinterpret_cast<const f32x4 *>(le_row + ccb);
        eb = *reinterpret_cast<const f32x4 *>(e_row + ccb);
    };
    f32x4 la[TP], lb[TP], ea[TP], eb[TP];
    int64_t pair = (int64_t)blockIdx.x * kResWaves + wave;
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        const int64_t g0 = row_of(pair < n_pairs ? pair : 0, u);
        fetch(LE + g0 * ldLE, Es + g0 * ldE, 0, la[u], lb[u], ea[u], eb[u]);
    }
    for (; pair < n_pairs; pair += pair_step) {
        const int64_t row0 = pair * (32 * TP);
        const float *le_row[TP], *e_row[TP];
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            const int64_t g = row_of(pair, u);
            le_row[u] = LE + g * ldLE, e_row[u] = Es + g * ldE;
        }
        f32x16 acc[TP][NT];
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][t][r] = 0.f;
        for (int c = 0; c < n_chunks; ++c) {
            f32x4 a4[TP][4];
            const int ca = c * NGCF_DC + lh * 4, cb = ca + 8;
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                if (cb + 4 > d_in) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (ca + q >= d_in) la[u][q] = 0.f, ea[u][q] = 0.f;
                        if (cb + q >= d_in) lb[u][q] = 0.f, eb[u][q] = 0.f;
                    }
                }
                a4[u][0] = la[u] + ea[u], a4[u][1] = lb[u] + eb[u], a4[u][2] = la[u] * ea[u], a4[u][3] = lb[u] * eb[u];
            }
            {   // the next chunk - of this pair, or chunk 0 of the wave's next pair (the last pair re-reads its own): unconditional
                const bool nxt = c == last;
                const int64_t np = pair + pair_step < n_pairs ? pair + pair_step : pair;
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    const int64_t gn = row_of(np, u);
                    const float *lr = nxt ? LE + gn * ldLE : le_row[u], *er = nxt ? Es + gn * ldE : e_row[u];
                    fetch(lr, er, nxt ? 0 : c + 1, la[u], lb[u], ea[u], eb[u]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const float *wc = W + (int64_t)c * NGCF_KC * WCOLS;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) {
                    const f32x4 bv = *reinterpret_cast<const f32x4 *>(wc + (kb * 8 + sx) * WCOLS);
#pragma unroll
                    for (int u = 0; u < TP; ++u) {
                        acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.x, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.y, acc[u][1], 0, 0, 0);
                        acc[u][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.z, acc[u][2], 0, 0, 0);
                        acc[u][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.w, acc[u][3], 0, 0, 0);
                    }
                }
        }
        // ---- epilogue per tile (wave-local): bias, LeakyReLU, dropout, row norm, stores
        const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
        const uint32_t drop_thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            const int64_t t0 = row0 + u * 32;
            if (t0 >= n_rows) break;
            float rowss[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rowss[r] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int col = t * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[u][t][r] + bz[t];
                    v = v >= 0.f ? v : leaky * v;
                    if (drop_mask || drop_p > 0.f) {
                        const int64_t grow = t0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (drop_mask) v *= (grow < n_rows && col < d_out) ? drop_mask[grow * ldm + col] : 0.f;
                        else {
                            const uint32_t h = mix32(drop_seed ^ ((uint64_t)grow * 0x9E3779B97F4A7C15ULL + (uint64_t)col));
                            v = h < drop_thr ? 0.f : v * keep_scale;
                        }
                    }
                    acc[u][t][r] = v;
                    rowss[r] = fmaf(v, v, rowss[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s2 = rowss[r];
                s2 += __shfl_xor(s2, 1);
                s2 += __shfl_xor(s2, 2);
                s2 += __shfl_xor(s2, 4);
                s2 += __shfl_xor(s2, 8);
                s2 += __shfl_xor(s2, 16);
                rowss[r] = s2;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t grow = t0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (grow >= n_rows) continue;
                const float inv = 1.f / fmaxf(sqrtf(rowss[r]), 1e-12f);   // F.normalize eps, NGCF.py:144
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int col = t * 32 + li;
                    if (col < d_out) {
                        const float v = acc[u][t][r];
                        if (carry) carry[grow * ldc + col] = v;
                        norm[grow * ldn + col] = v * inv;
                    }
                }
            }
        }
    }
}
#endif  // NGCF_LAB

#ifdef NGCF_LAB
// ---------------------------------------------------------------------------------------------
// LAB ONLY (-DNGCF_LAB; measured slower than layer_dense_resident_kernel: profiles/r03_dense_il_lab.txt).
// The resident kernel with the finished tiles leaving UNDER the next tiles' K loop (r03).
// layer_dense_resident_kernel ends a tile with 128 (256 with a carry) store instructions per wave in one burst and relies on the
// second wave of the SIMD to keep the matrix pipe busy meanwhile.  Here a SIMD runs ONE wave with the whole register file (512
// per lane: __launch_bounds__(256), one workgroup per CU) that owns TWO row tiles at a time (64 rows x 128 columns, 128 accumulator
// registers; a B operand read from LDS feeds eight MFMAs).  The activated values of a finished pair stay in 128 registers (+ 32 row
// scales) and the K loop of the wave's NEXT pair issues two stores after every group of eight MFMAs (16 groups per chunk, 8
// chunks: 256 stores), so that a wave's memory traffic is spread evenly over its matrix work instead of alternating with it.  The
// K loop is unrolled completely (NCH chunks, a template parameter: 8 for d_in 113..128, 9 for 129..144) - the element of the
// previous pair a group stores is then a fixed register - and there is no branch in it: a store behind a branch makes the
// compiler wait for every load in flight at the join.  Raw operands are requested one chunk (128 MFMAs = 3.4 us) ahead.  The first
// pair of a wave runs the loop without stores, the last one leaves in a burst after the loop (with the bounds tests of a partial
// tile).  Same k order and the same epilogue arithmetic as the other two kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------
constexpr int kIlWaves = 4;

template <int NCH, bool CARRY, int LAB = 0>     // LAB (-DNGCF_LAB builds): 1 no stores, 2 no loads, +4 stores to L2-resident rows, +8 staged but not stored - wrong results, timing only
__global__ __launch_bounds__(kIlWaves * 64) void layer_dense_resident_il_kernel(
    const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ Es, int64_t ldE, int64_t n_rows, int d_in, int d_out,
    const float *__restrict__ Wt, const float *__restrict__ bias2, float leaky, float drop_p, uint64_t drop_seed_in,
    const float *__restrict__ drop_mask, int64_t ldm, float *__restrict__ carry, int64_t ldc, float *__restrict__ norm, int64_t ldn)
{
    const uint64_t drop_seed = drop_p > 0.f ? resolve_seed(drop_seed_in) : drop_seed_in;
    constexpr int NT = 4, WCOLS = 128, TP = 1;   // TP: row tiles a wave works on at a time (2: 204 registers spilled)
    extern __shared__ float Wres[];                 // [NCH * 32][128], the layout of pack_weights_kernel
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(Wt);
        f32x4 *dst = reinterpret_cast<f32x4 *>(Wres);
        constexpr int n4 = NCH * NGCF_KC * WCOLS / 4;
        for (int i = tid; i < n4; i += kIlWaves * 64) dst[i] = src[i];
    }
    __syncthreads();
    const int64_t n_pairs = (n_rows + 32 * TP - 1) / (32 * TP);
    const int d4 = (d_in + 3) & ~3;
    const float *W = Wres + li * NT + lh * 4 * WCOLS;     // the lane's k rows of a group: 4 lh + sx
    float bz[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bz[t] = bias2[t * 32 + li];
    const int64_t pair_step = (int64_t)gridDim.x * kIlWaves;
    const int ldn_i = (int)ldn, ldc_i = (int)ldc;          // (host: 64 rows of either fit 31 bits)
    auto row_of = [&](int64_t pr, int u) {                 // the lane's row of tile u of pair pr (past the end: the last row, never stored)
        int64_t g = pr * (32 * TP) + u * 32 + li;
        return g < n_rows ? g : n_rows - 1;
    };
    // the lane's four 16-byte pieces of chunk C (le_l / e_l: the lane's row + 4 lh): only the LAST chunk can reach past the padded
    // width (d_in > 16 (NCH - 1)), every other one is the row pointer + a constant
    auto fetch = [&](auto Cc, const float *le_l, const float *e_l, f32x4 &la, f32x4 &lb, f32x4 &ea, f32x4 &eb) {
        constexpr int C = decltype(Cc)::value;
        if constexpr (LAB & 2) return;
        if constexpr (C < NCH - 1) {
            la = *reinterpret_cast<const f32x4 *>(le_l + C * NGCF_DC);
            ea = *reinterpret_cast<const f32x4 *>(e_l + C * NGCF_DC);
            lb = *reinterpret_cast<const f32x4 *>(le_l + C * NGCF_DC + 8);
            eb = *reinterpret_cast<const f32x4 *>(e_l + C * NGCF_DC + 8);
        } else {
            const int ca = C * NGCF_DC + lh * 4, cb = ca + 8;
            const int cca = (ca < d4 ? ca : d4 - 4) - lh * 4, ccb = (cb < d4 ? cb : d4 - 4) - lh * 4;
            la = *reinterpret_cast<const f32x4 *>(le_l + cca);
            ea = *reinterpret_cast<const f32x4 *>(e_l + cca);
            lb = *reinterpret_cast<const f32x4 *>(le_l + ccb);
            eb = *reinterpret_cast<const f32x4 *>(e_l + ccb);
        }
    };
    f32x4 la[2][TP], lb[2][TP], ea[2][TP], eb[2][TP];   // the raw pieces of the next two chunks (set = chunk & 1)
    if constexpr (LAB & 2) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int u = 0; u < TP; ++u) la[q][u] = lb[q][u] = ea[q][u] = eb[q][u] = f32x4{0.01f * lane, 0.02f, 0.03f, 0.04f};
    }
    int64_t pair = (int64_t)blockIdx.x * kIlWaves + wave;
    if (pair >= n_pairs) return;
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        const int64_t g0 = row_of(pair, u);
        fetch(std::integral_constant<int, 0>{}, LE + g0 * ldLE + lh * 4, Es + g0 * ldE + lh * 4, la[0][u], lb[0][u], ea[0][u], eb[0][u]);
        fetch(std::integral_constant<int, 1>{}, LE + g0 * ldLE + lh * 4, Es + g0 * ldE + lh * 4, la[1][u], lb[1][u], ea[1][u], eb[1][u]);
    }
    f32x16 acc[TP][NT];
    f32x16 pv[TP][NT];             // the previous pair: activated values, un-normalised (what `carry` receives)
    float pinv[TP][16];            // and its row scales
    float *pn = norm, *pc = carry; // the lane's 16 bytes (columns 4 li ..) of row 4 lh of the previous pair's destination
    float *stg = Wres + NCH * NGCF_KC * WCOLS + wave * 512;   // the wave's two 1 KB slabs (a row pair each) for turning a row
    f32x4 qn[TP][16], qc[TP][16];  // the previous tile as it leaves: lane = 16 bytes (columns 4 li ..) of row r (+ 4 lh), normalised / carry
    f32x4 lab_sink = {0.f, 0.f, 0.f, 0.f};
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t drop_thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const bool any_drop = drop_mask || drop_p > 0.f;

    // one chunk: sums / products from the raw pieces, the next chunk requested (NXT: of the next pair), 16 groups of (B operand of
    // the next group from LDS, eight MFMAs, two stores of the previous pair)
    auto chunk = [&](auto Cc, auto STc, const float *const (&le_row)[TP], const float *const (&e_row)[TP],
                     const float *const (&le_nxt)[TP], const float *const (&e_nxt)[TP]) {
        constexpr int C = decltype(Cc)::value;
        constexpr bool ST = decltype(STc)::value;
        f32x4 a4[TP][4];
        constexpr int S = C & 1;
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            if constexpr (C == NCH - 1) {
                const int ca = C * NGCF_DC + lh * 4, cb = ca + 8;
                if (cb + 4 > d_in) {                      // only the last chunk of an odd width
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (ca + q >= d_in) la[S][u][q] = 0.f, ea[S][u][q] = 0.f;
                        if (cb + q >= d_in) lb[S][u][q] = 0.f, eb[S][u][q] = 0.f;
                    }
                }
            }
            a4[u][0] = la[S][u] + ea[S][u], a4[u][1] = lb[S][u] + eb[S][u], a4[u][2] = la[S][u] * ea[S][u], a4[u][3] = lb[S][u] * eb[S][u];
            // the chunk after next - of this tile, or chunk 0 / 1 of the wave's next one (NCH odd: the sets swap roles from tile to
            // tile, which a fixed register assignment cannot follow; then chunk NCH - 1 leaves its set to chunk 1 of the next tile
            // and chunk 0 is requested by chunk NCH - 2 into the other one ... only for even NCH; odd NCH: see k_loop)
            if constexpr (C + 2 < NCH) fetch(std::integral_constant<int, C + 2>{}, le_row[u], e_row[u], la[S][u], lb[S][u], ea[S][u], eb[S][u]);
            else if constexpr ((NCH & 1) == 0)
                fetch(std::integral_constant<int, C + 2 - NCH>{}, le_nxt[u], e_nxt[u], la[S][u], lb[S][u], ea[S][u], eb[S][u]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const float *wc = W + C * NGCF_KC * WCOLS;
        f32x4 bv = *reinterpret_cast<const f32x4 *>(wc);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int kb = g >> 2, sx = g & 3;
            f32x4 bn = bv;
            if (g < 15) bn = *reinterpret_cast<const f32x4 *>(wc + (((g + 1) >> 2) * 8 + ((g + 1) & 3)) * WCOLS);
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.x, acc[u][0], 0, 0, 0);
                acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.y, acc[u][1], 0, 0, 0);
                acc[u][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.z, acc[u][2], 0, 0, 0);
                acc[u][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[u][kb][sx], bv.w, acc[u][3], 0, 0, 0);
                if constexpr (ST && C < 8 && !(LAB & 1)) {
                    // chunk C stores rows r = 2 C and 2 C + 1 (of both halves lh) of the previous tile: four 16-byte stores, four
                    // groups apart (the values were turned and scaled by activate(): instructions between the MFMAs of this loop
                    // cost several times what they cost there - tools/dense_il_parts_lab.py)
                    if ((g & 3) == 2) {
                        const int js = g >> 3, rs = 2 * C + js, ros = u * 32 + (rs & 3) + 8 * (rs >> 2);
                        if constexpr (LAB & 8) lab_sink += qn[u][rs];
                        else if ((g >> 2 & 1) == 0) *reinterpret_cast<f32x4 *>(pn + ros * ldn_i) = qn[u][rs];
                        else if (CARRY) *reinterpret_cast<f32x4 *>(pc + ros * ldc_i) = qc[u][rs];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            bv = bn;
        }
    };
    auto k_loop = [&](auto STc, int64_t pr) {
        const float *le_row[TP], *e_row[TP], *le_nxt[TP], *e_nxt[TP];
        const int64_t nx = pr + pair_step < n_pairs ? pr + pair_step : pr;   // (the last pair of a wave re-reads its own first chunk)
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            const int64_t g = row_of(pr, u), gn = row_of(nx, u);
            le_row[u] = LE + g * ldLE + lh * 4, e_row[u] = Es + g * ldE + lh * 4;
            le_nxt[u] = LE + gn * ldLE + lh * 4, e_nxt[u] = Es + gn * ldE + lh * 4;
#pragma unroll
            for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][tt][r] = 0.f;
        }
        chunk(std::integral_constant<int, 0>{}, STc, le_row, e_row, le_nxt, e_nxt);
        chunk(std::integral_constant<int, 1>{}, STc, le_row, e_row, le_nxt, e_nxt);
        chunk(std::integral_constant<int, 2>{}, STc, le_row, e_row, le_nxt, e_nxt);
        chunk(std::integral_constant<int, 3>{}, STc, le_row, e_row, le_nxt, e_nxt);
        chunk(std::integral_constant<int, 4>{}, STc, le_row, e_row, le_nxt, e_nxt);
        chunk(std::integral_constant<int, 5>{}, STc, le_row, e_row, le_nxt, e_nxt);
        chunk(std::integral_constant<int, 6>{}, STc, le_row, e_row, le_nxt, e_nxt);
        chunk(std::integral_constant<int, 7>{}, STc, le_row, e_row, le_nxt, e_nxt);
        if constexpr (NCH > 8) chunk(std::integral_constant<int, 8>{}, STc, le_row, e_row, le_nxt, e_nxt);
        if constexpr (NCH & 1) {     // odd: both sets are free only now
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                fetch(std::integral_constant<int, 0>{}, le_nxt[u], e_nxt[u], la[0][u], lb[0][u], ea[0][u], eb[0][u]);
                fetch(std::integral_constant<int, 1>{}, le_nxt[u], e_nxt[u], la[1][u], lb[1][u], ea[1][u], eb[1][u]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // bias, LeakyReLU, dropout, row scale: acc -> pv, pinv, pn, pc
    auto activate = [&](int64_t pr) {
        const int64_t row0 = pr * (32 * TP);
#pragma unroll
        for (int u = 0; u < TP; ++u) {
            float rowss[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) rowss[r] = 0.f;
            if (!any_drop) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[u][tt][r] + bz[tt];
                        v = v >= 0.f ? v : leaky * v;
                        pv[u][tt][r] = v;
                        rowss[r] = fmaf(v, v, rowss[r]);
                    }
            } else {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int col = tt * 32 + li;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = acc[u][tt][r] + bz[tt];
                        v = v >= 0.f ? v : leaky * v;
                        const int64_t grow = row0 + u * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (drop_mask) v *= (grow < n_rows && col < d_out) ? drop_mask[grow * ldm + col] : 0.f;
                        else {
                            const uint32_t h = mix32(drop_seed ^ ((uint64_t)grow * 0x9E3779B97F4A7C15ULL + (uint64_t)col));
                            v = h < drop_thr ? 0.f : v * keep_scale;
                        }
                        pv[u][tt][r] = v;
                        rowss[r] = fmaf(v, v, rowss[r]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s2 = rowss[r];
                s2 += __shfl_xor(s2, 1);
                s2 += __shfl_xor(s2, 2);
                s2 += __shfl_xor(s2, 4);
                s2 += __shfl_xor(s2, 8);
                s2 += __shfl_xor(s2, 16);
                pinv[u][r] = 1.f / fmaxf(sqrtf(s2), 1e-12f);   // F.normalize eps, NGCF.py:144
            }
            // a lane holds 4 values of a row 32 columns apart; through a wave-private 1 KB slab of LDS (two, alternating) they
            // become 16 contiguous bytes per lane: a store instruction then writes two complete rows
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) stg[(r & 1) * 256 + lh * 128 + tt * 32 + li] = pv[u][tt][r];
                const f32x4 q = *reinterpret_cast<const f32x4 *>(stg + (r & 1) * 256 + lane * 4);
                qn[u][r] = q * pinv[u][r];
                if (CARRY) qc[u][r] = q;
            }
        }
        const int64_t dst0 = (LAB & 4) ? ((int64_t)blockIdx.x * kIlWaves + wave) * (32 * TP) : row0;   // lab: every tile to the wave's first rows (L2 hits)
        pn = norm + (dst0 + 4 * lh) * ldn + 4 * li;
        if (CARRY) pc = carry + (dst0 + 4 * lh) * ldc + 4 * li;
    };

    k_loop(std::false_type{}, pair);
    activate(pair);
    int64_t prev = pair;
    for (pair += pair_step; pair < n_pairs; pair += pair_step) {
        k_loop(std::true_type{}, pair);               // stores pair `prev` (full: it is not the last one) on the way
        activate(pair);
        prev = pair;
    }
    // the wave's last pair leaves in a burst
    if constexpr (LAB & 1) {           // (one store keeps the arithmetic alive)
        f32x4 x = lab_sink;
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) x += qn[u][r] + (CARRY ? qc[u][r] : qn[u][r]);
        if (x.x + x.y + x.z + x.w == 1.2345f) norm[0] = x.x;
        return;
    }
    const int64_t row0 = prev * (32 * TP);
#pragma unroll
    for (int u = 0; u < TP; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = u * 32 + (r & 3) + 8 * (r >> 2);
            if (row0 + ro + 4 * lh >= n_rows) continue;
            *reinterpret_cast<f32x4 *>(pn + ro * ldn_i) = qn[u][r];
            if (CARRY) *reinterpret_cast<f32x4 *>(pc + ro * ldc_i) = qc[u][r];
        }
    if constexpr (LAB & 8)
        if (lab_sink.x + lab_sink.y + lab_sink.z + lab_sink.w == 1.2345f) norm[1] = lab_sink.x;
}

#endif  // NGCF_LAB

// ---------------------------------------------------------------------------------------------
// 256 / 512 output columns with NO operand in LDS (r02).  At these widths a workgroup of layer_dense_kernel owns 32 (or 64) rows
// and each of its waves its own 128 output columns: the weights a wave multiplies by are shared with nobody, yet a 32 / 64 KB
// chunk of them goes through the one LDS buffer per chunk between two barriers, and with a few thousand rows (the Seoul graph:
// 186 workgroups, one wave per SIMD) nothing hides that.  Here a wave reads its B operands - the lane's four tile values of a
// k-pair are 16 contiguous bytes of the packed row - straight from the L2-resident packed matrix into registers, one chunk ahead
// in two fixed register sets, and its A operands like layer_dense_resident_kernel; the only LDS traffic is the exchange of the
// row sums of squares at the end.  Same k order per output element as the other two kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NT> struct TileVec;
template <> struct TileVec<4> { using type = f32x4; };
template <> struct TileVec<2> { using type = f32x2; };

template <int CW, int NT>
__global__ __launch_bounds__(CW * 64) void layer_dense_direct_kernel(
    const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ Es, int64_t ldE, int64_t n_rows, int d_in, int d_out,
    const float *__restrict__ Wt, const float *__restrict__ bias2, int n_chunks, float leaky, float drop_p, uint64_t drop_seed_in,
    const float *__restrict__ drop_mask, int64_t ldm, float *__restrict__ carry, int64_t ldc, float *__restrict__ norm, int64_t ldn)
{
    const uint64_t drop_seed = drop_p > 0.f ? resolve_seed(drop_seed_in) : drop_seed_in;
    constexpr int WCOLS = 32 * NT * CW;            // weights packed with this NT: a lane's NT tile values are contiguous
    using BV = typename TileVec<NT>::type;
    __shared__ float ssq[32 * CW];
    const int tid = threadIdx.x, cw = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int64_t row0 = (int64_t)blockIdx.x * 32;
    int64_t grow_l = row0 + li;
    grow_l = grow_l < n_rows ? grow_l : n_rows - 1;              // rows past the end re-read the last row, never stored
    const float *le_row = LE + grow_l * ldLE, *e_row = Es + grow_l * ldE;
    const int d4 = (d_in + 3) & ~3;
    const float *Wl = Wt + (int64_t)lh * 4 * WCOLS + cw * (32 * NT) + li * NT;   // k-pair (kb, sx) of chunk c: + ((c*32 + kb*8 + sx) * WCOLS)
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    auto fetch_a = [&](int c, f32x4 &la, f32x4 &lb, f32x4 &ea, f32x4 &eb) {
        const int ca = c * NGCF_DC + lh * 4, cb = ca + 8;
        const int cca = ca < d4 ? ca : d4 - 4, ccb = cb < d4 ? cb : d4 - 4;
        la = *reinterpret_cast<const f32x4 *>(le_row + cca);
        ea = *reinterpret_cast<const f32x4 *>(e_row + cca);
        lb = *reinterpret_cast<const f32x4 *>(le_row + ccb);
        eb = *reinterpret_cast<const f32x4 *>(e_row + ccb);
    };
    auto fetch_b = [&](int c, BV (&b)[16]) {
        const float *wc = Wl + (int64_t)c * NGCF_KC * WCOLS;
#pragma unroll
        for (int q = 0; q < 16; ++q) b[q] = *reinterpret_cast<const BV *>(wc + ((q >> 2) * 8 + (q & 3)) * WCOLS);
    };
    auto form = [&](int c, f32x4 la, f32x4 lb, f32x4 ea, f32x4 eb, f32x4 (&a4)[4]) {   // zeroing of the columns past d_in at use
        const int ca = c * NGCF_DC + lh * 4, cb = ca + 8;
        if (cb + 4 > d_in) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (ca + q >= d_in) la[q] = 0.f, ea[q] = 0.f;
                if (cb + q >= d_in) lb[q] = 0.f, eb[q] = 0.f;
            }
        }
        a4[0] = la + ea, a4[1] = lb + eb, a4[2] = la * ea, a4[3] = lb * eb;
    };
    auto chunk_mfma = [&](const f32x4 (&a4)[4], const BV (&b)[16]) {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int sx = 0; sx < 4; ++sx) {
                const BV bv = b[kb * 4 + sx];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[kb][sx], bv[t], acc[t], 0, 0, 0);
            }
    };
    // every prefetch unconditional, the odd last chunk peeled off (see layer_dense_resident_kernel)
    const int last = n_chunks - 1;
    f32x4 la0, lb0, ea0, eb0, la1, lb1, ea1, eb1;
    BV b0[16], b1[16];
    fetch_a(0, la0, lb0, ea0, eb0);
    fetch_b(0, b0);
    fetch_a(last < 1 ? last : 1, la1, lb1, ea1, eb1);
    fetch_b(last < 1 ? last : 1, b1);
    int c = 0;
    for (; c + 1 < n_chunks; c += 2) {
        {
            f32x4 a4[4];
            form(c, la0, lb0, ea0, eb0, a4);
            fetch_a(c + 2 < last ? c + 2 : last, la0, lb0, ea0, eb0);
            __builtin_amdgcn_sched_barrier(0);
            chunk_mfma(a4, b0);
            fetch_b(c + 2 < last ? c + 2 : last, b0);          // in flight under the next chunk's MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            f32x4 a4[4];
            form(c + 1, la1, lb1, ea1, eb1, a4);
            fetch_a(c + 3 < last ? c + 3 : last, la1, lb1, ea1, eb1);
            __builtin_amdgcn_sched_barrier(0);
            chunk_mfma(a4, b1);
            fetch_b(c + 3 < last ? c + 3 : last, b1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (c < n_chunks) {
        f32x4 a4[4];
        form(c, la0, lb0, ea0, eb0, a4);
        chunk_mfma(a4, b0);
    }
    // ---- epilogue: bias, LeakyReLU, dropout, row sum of squares (as layer_dense_kernel with RW = 1)
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
            if (drop_mask) {           // "reference" mode: the noise tensor nn.Dropout drew on the host, NGCF.py:142
                const int64_t grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                v *= (grow < n_rows && col < d_out) ? drop_mask[grow * ldm + col] : 0.f;
            } else if (drop_p > 0.f) {
                const int64_t grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const uint32_t h = mix32(drop_seed ^ ((uint64_t)grow * 0x9E3779B97F4A7C15ULL + (uint64_t)col));
                v = h < drop_thr ? 0.f : v * keep_scale;
            }
            acc[t][r] = v;
            rowss[r] = fmaf(v, v, rowss[r]);
        }
    }
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
    if (li == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ssq[((r & 3) + 8 * (r >> 2) + 4 * lh) * CW + cw] = rowss[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < CW; ++q) s += ssq[lr * CW + q];
        rowss[r] = s;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t grow = row0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (grow >= n_rows) continue;
        const float inv = 1.f / fmaxf(sqrtf(rowss[r]), 1e-12f);   // F.normalize eps, NGCF.py:144
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

// ---------------------------------------------------------------------------------------------
// 256 / 512 output columns on a FEW THOUSAND rows, balanced over the whole chip (r03; VERDICT r2 #7: the Seoul shape).
// layer_dense_direct_kernel gives a workgroup 32 complete rows: 5 940 rows are 186 workgroups on 256 CUs and every wave multiplies
// 32 rows x 128 columns x K - 2 060 MFMAs at 515 -> 512, 55 us before anything else.  What balances is three 32 x 32 tiles per
// SIMD: 186 x 16 tiles / 1 024 SIMDs = 2.9.  Here a workgroup owns 96 rows x 128 columns and each of its four waves THREE row
// tiles of ONE 32-column tile (62 x 4 = 248 workgroups, 1 545 MFMAs per wave = 41 us): the rows of LE / E go through LDS once per
// workgroup (sums and products formed on the way in, double-buffered, one barrier per chunk) and feed all four waves; a wave's
// weights come straight from the packed matrix (a lane's four k-values of a k-block are 16 contiguous bytes: pack_weights_tall_kernel),
// one chunk ahead.  A row is no longer complete inside a workgroup, so the kernel writes the activated values into BOTH outputs
// and row_scale_kernel turns the `norm` rows into unit rows in place - adding the squares in exactly the order the other dense
// kernels use (per 128-column panel: four columns 32 apart per lane, butterfly over 32 lanes; panels in order), so the results
// are bit-identical to theirs.
// ---------------------------------------------------------------------------------------------
constexpr int kTallRows = 96;

// Wp[chunk][ct][kb][lh][li][sx] = packed k-row kb*8 + lh*4 + sx (0..15: W1 column chunk*16 + kl, 16..31: W2 column chunk*16 + kl - 16)
// of output column ct*32 + li; zero outside the matrices
__global__ __launch_bounds__(256) void pack_weights_tall_kernel(const float *__restrict__ W1, const float *__restrict__ b1,
                                                                const float *__restrict__ W2, const float *__restrict__ b2, int d_in,
                                                                int d_out, int DOP, float *__restrict__ Wp, float *__restrict__ bias2)
{
    __shared__ float tile[NGCF_KC][33];
    const int n_ct = DOP / 32;
    const int chunk = blockIdx.x / n_ct, ct = blockIdx.x % n_ct;
    for (int idx = threadIdx.x; idx < 32 * NGCF_KC; idx += 256) {
        const int j = idx / NGCF_KC, kl = idx % NGCF_KC;          // consecutive threads: consecutive input columns of one output row
        const int oc = ct * 32 + j, col = chunk * NGCF_DC + (kl % NGCF_DC);
        float v = 0.f;
        if (oc < d_out && col < d_in) v = (kl < NGCF_DC ? W1 : W2)[(int64_t)oc * d_in + col];
        tile[kl][j] = v;
    }
    __syncthreads();
    float *dst = Wp + ((int64_t)chunk * n_ct + ct) * (32 * NGCF_KC);
    for (int idx = threadIdx.x; idx < 32 * NGCF_KC; idx += 256) {
        const int sx = idx & 3, li = (idx >> 2) & 31, lh = (idx >> 7) & 1, kb = idx >> 8;
        dst[idx] = tile[kb * 8 + lh * 4 + sx][li];
    }
    if (blockIdx.x == 0)
        for (int j = threadIdx.x; j < DOP; j += 256)
            bias2[j] = j < d_out ? (b1[j] + b1[j]) + b2[j] : 0.f;   // b1 is added twice, NGCF.py:131,133
}

template <int LAB>      // LAB (-DNGCF_LAB builds, timing only): 1 no barriers, 2 no row staging, 4 no weight loads in the loop
__global__ __launch_bounds__(256) void layer_dense_tall_kernel(
    const float *__restrict__ LE, int64_t ldLE, const float *__restrict__ Es, int64_t ldE, int64_t n_rows, int d_in, int d_out,
    const float *__restrict__ Wp, const float *__restrict__ bias2, int n_chunks, int n_ct, float leaky, float drop_p,
    uint64_t drop_seed_in, const float *__restrict__ drop_mask, int64_t ldm, float *__restrict__ carry, int64_t ldc,
    float *__restrict__ norm, int64_t ldn)
{
    const uint64_t drop_seed = drop_p > 0.f ? resolve_seed(drop_seed_in) : drop_seed_in;
    constexpr int XLD = NGCF_KC + 4;               // 36: [16 sums | 16 products | pad], b128 reads of a column block conflict-free
    constexpr int MT = kTallRows / 32;             // row tiles per wave
    constexpr int XBUF = kTallRows * XLD;          // floats of one LDS buffer
    __shared__ float Xs[3 * XBUF];                 // THREE buffers: see the step comment below
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int n_cp = n_ct / 4;
    const int64_t row0 = (int64_t)(blockIdx.x / n_cp) * kTallRows;
    const int ct = (blockIdx.x % n_cp) * 4 + wave;                    // the wave's 32-column tile
    const int d4 = (d_in + 3) & ~3;
    // staging: 96 rows x 4 pieces of 16 bytes per operand and chunk = 384 pieces: thread t takes pieces t and t + 256 (< 384)
    const bool two = tid < 384 - 256;
    int64_t grow_a = row0 + (tid >> 2), grow_b = row0 + 64 + (tid >> 2);
    grow_a = grow_a < n_rows ? grow_a : n_rows - 1;                   // rows past the end re-read the last row, never stored
    grow_b = two ? (grow_b < n_rows ? grow_b : n_rows - 1) : grow_a;
    const int sq4 = (tid & 3) * 4;
    const float *le_a = LE + grow_a * ldLE + sq4, *e_a = Es + grow_a * ldE + sq4, *le_b = LE + grow_b * ldLE + sq4, *e_b = Es + grow_b * ldE + sq4;
    const int last = n_chunks - 1;
    const int cc_last = (last * NGCF_DC + sq4 < d4 ? last * NGCF_DC + sq4 : d4 - 4) - sq4;   // the last chunk may reach past the padded width
    struct XRegs { f32x4 al, ae, bl, be; };     // a thread's pieces of one chunk: rows tid / 4 and 64 + tid / 4
    XRegs x0, x1;                               // TWO chunks in flight: the rows come from beyond the L2 (12 MB each on the Seoul graph),
                                                // one chunk of MFMAs (1.3 us) does not cover that latency
    auto load_x = [&](int c, XRegs &x) {        // c <= last (callers clamp); every load unconditional (a load behind a branch cannot be counted)
        const int off = c < last ? c * NGCF_DC : cc_last;
        x.al = *reinterpret_cast<const f32x4 *>(le_a + off);
        x.ae = *reinterpret_cast<const f32x4 *>(e_a + off);
        x.bl = *reinterpret_cast<const f32x4 *>(le_b + off);
        x.be = *reinterpret_cast<const f32x4 *>(e_b + off);
    };
    float *xs_w = Xs + (tid >> 2) * XLD + sq4;
    auto store_x = [&](int c, int buf, XRegs x) {
        if (c == last) {                        // only the last chunk can hold columns past d_in: they contribute zeros
            const int c0 = c * NGCF_DC + sq4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c0 + q >= d_in) x.al[q] = 0.f, x.ae[q] = 0.f, x.bl[q] = 0.f, x.be[q] = 0.f;
        }
        float *xr = xs_w + buf * XBUF;
        *reinterpret_cast<f32x4 *>(xr) = x.al + x.ae;
        *reinterpret_cast<f32x4 *>(xr + NGCF_DC) = x.al * x.ae;
        if (two) {
            *reinterpret_cast<f32x4 *>(xr + 64 * XLD) = x.bl + x.be;
            *reinterpret_cast<f32x4 *>(xr + 64 * XLD + NGCF_DC) = x.bl * x.be;
        }
    };
    const float *wl = Wp + (int64_t)ct * (32 * NGCF_KC) + lane * 4;    // chunk c, k-block kb: + c * n_ct * 1024 + kb * 256
    const int64_t wstep = (int64_t)n_ct * (32 * NGCF_KC);
    f32x4 b0[4], b1[4];
    auto load_b = [&](int c, f32x4 (&b)[4]) {
        const float *w = wl + (int64_t)c * wstep;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) b[kb] = *reinterpret_cast<const f32x4 *>(w + kb * 256);
    };
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    const float *xs_r = Xs + li * XLD + lh * 4;
    auto read_a = [&](int buf, int kb, f32x4 (&a)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
            a[m] = *reinterpret_cast<const f32x4 *>(xs_r + buf * XBUF + m * 32 * XLD + (kb >> 1) * NGCF_DC + (kb & 1) * 8);
    };
    // One wave per SIMD: nobody else covers a latency, so every operand is requested a step ahead.
    // At the top of step c: LDS buffers c % 3 and (c + 1) % 3 hold chunks c and c + 1; `an` holds the first k-block of chunk c
    // (read from LDS at the end of step c - 1); register set c & 1 holds the rows of chunk c + 2 and set (c + 1) & 1 those of
    // chunk c + 3 (in flight).  Step c: the MFMAs of chunk c (k-block kb + 1 read from LDS before the MFMAs of k-block kb);
    // chunk c + 2 goes into buffer (c + 2) % 3 (last read in step c - 1, a barrier ago); chunk c + 4 is requested; the first
    // k-block of chunk c + 1 is read; ONE barrier.  With three buffers the first LDS reads of a chunk are issued BEFORE the
    // barrier that ends the previous step, not after it.  Chunk indices past the end are clamped (re-read, never multiplied).
    auto cl = [&](int c) { return c < last ? c : last; };
    f32x4 an[MT];
    auto step = [&](int c, int buf, const f32x4 (&b)[4], XRegs &x) {
        f32x4 a0[MT], a1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) a0[m] = an[m];
#pragma unroll
        for (int kb = 0; kb < 4; kb += 2) {
            read_a(buf, kb + 1, a1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sx = 0; sx < 4; ++sx)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[m][sx], b[kb][sx], acc[m], 0, 0, 0);
            if (kb + 2 < 4) read_a(buf, kb + 2, a0);
            else {
                const int b2 = buf >= 1 ? buf - 1 : 2;                 // (c + 2) % 3
                if constexpr (!(LAB & 2)) {
                    store_x(cl(c + 2), b2, x);
                    load_x(cl(c + 4), x);
                }
                read_a(buf == 2 ? 0 : buf + 1, 0, an);                 // chunk c + 1, written a step ago
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sx = 0; sx < 4; ++sx)
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[m][sx], b[kb + 1][sx], acc[m], 0, 0, 0);
        }
    };
    load_x(0, x0);
    load_b(0, b0);
    load_x(cl(1), x1);
    load_b(cl(1), b1);
    store_x(0, 0, x0);
    load_x(cl(2), x0);
    store_x(cl(1), 1, x1);
    load_x(cl(3), x1);
    __syncthreads();
    read_a(0, 0, an);
    int buf = 0;
    int c = 0;
    for (; c + 1 < n_chunks; c += 2) {
        step(c, buf, b0, x0);
        if constexpr (!(LAB & 4)) load_b(cl(c + 2), b0);
        if constexpr (!(LAB & 1)) __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
        step(c + 1, buf, b1, x1);
        if constexpr (!(LAB & 4)) load_b(cl(c + 3), b1);
        if constexpr (!(LAB & 1)) __syncthreads();
        buf = buf == 2 ? 0 : buf + 1;
    }
    if (c < n_chunks) step(c, buf, b0, x0);
    // ---- epilogue: bias, LeakyReLU, dropout; the activated value goes to both outputs (row_scale_kernel finishes `norm`)
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const uint32_t drop_thr = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
    const int col = ct * 32 + li;
    const float bz = bias2[col];
    if (col >= d_out) return;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t grow = row0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (grow >= n_rows) continue;
            float v = acc[m][r] + bz;
            v = v >= 0.f ? v : leaky * v;
            if (drop_mask) v *= drop_mask[grow * ldm + col];          // "reference" mode: the noise tensor nn.Dropout drew, NGCF.py:142
            else if (drop_p > 0.f) {
                const uint32_t h = mix32(drop_seed ^ ((uint64_t)grow * 0x9E3779B97F4A7C15ULL + (uint64_t)col));
                v = h < drop_thr ? 0.f : v * keep_scale;
            }
            if (carry) carry[grow * ldc + col] = v;
            norm[grow * ldn + col] = v;
        }
}

// norm[row, :] /= max(||norm[row, :]||, 1e-12) (F.normalize, NGCF.py:144), one wave per row.  The squares are added exactly as
// the dense kernels add them: per panel of 128 columns lane li takes columns li, 32 + li, 64 + li, 96 + li in that order (fma
// chain from 0), the 32 lanes are combined by the xor butterfly 1, 2, 4, 8, 16, and the panel sums are added in panel order.
template <int CW>      // panels: 2 (<= 256 columns) or 4
__global__ __launch_bounds__(256) void row_scale_kernel(float *__restrict__ norm, int64_t ldn, int64_t n_rows, int d_out)
{
    const int lane = threadIdx.x & 63, li = lane & 31, half = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    float *x = norm + row * ldn;
    constexpr int PP = CW / 2;                     // panels per lane half: half h takes panels h, h + 2
    float v[PP][4], part[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const int cw = half + 2 * p;
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = cw * 128 + t * 32 + li;
            v[p][t] = col < d_out ? x[col] : 0.f;
            s = fmaf(v[p][t], v[p][t], s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 16);
        part[p] = s;
    }
    // panel order 0, 1, 2, 3: this half holds panels half, half + 2, the other one the rest
    float other[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) other[p] = __shfl_xor(part[p], 32);
    float ss = 0.f;
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const float even = half == 0 ? part[p] : other[p], odd = half == 0 ? other[p] : part[p];
        ss += even;
        ss += odd;
    }
    const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
    for (int p = 0; p < PP; ++p)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = (half + 2 * p) * 128 + t * 32 + li;
            if (col < d_out) x[col] = v[p][t] * inv;
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
    int dop = dense_dop(d_out);
    if (dop < 0 || d_in <= 0) return -1;
    dop = std::max(dop, 128);                  // small matrices run narrow layers on 32-row x 128-column tiles
    const int64_t n_chunks = (d_in + NGCF_DC - 1) / NGCF_DC;
    return align_up((n_chunks * NGCF_KC * dop + dop) * (int64_t)sizeof(float), 256) + 256;
}

template <int RW, int CW, int NT>
static int launch_dense(bool al, int64_t n_rows, const float *LE, int64_t ldLE, const float *Es, int64_t ldE, int d_in,
                        int d_out, const float *Wt, const float *bias2, int n_chunks, float leaky, float drop_p,
                        uint64_t seed, const float *drop_mask, int64_t ldm, float *carry, int64_t ldc, float *norm, int64_t ldn,
                        hipStream_t stream)
{
    const int64_t blocks = (n_rows + 32 * RW - 1) / (32 * RW);
    if (blocks == 0) return NGCF_OK;
    if (al && ldLE >= align_up(d_in, 4) && ldE >= align_up(d_in, 4) && d_in >= 4)   // padded, aligned rows (any d_in)
        layer_dense_kernel<RW, CW, NT, true, true><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            LE, ldLE, Es, ldE, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, seed, drop_mask, ldm, carry, ldc, norm, ldn);
    else if (al)
        layer_dense_kernel<RW, CW, NT, true, false><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            LE, ldLE, Es, ldE, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, seed, drop_mask, ldm, carry, ldc, norm, ldn);
    else
        layer_dense_kernel<RW, CW, NT, false, false><<<dim3((unsigned)blocks), 256, 0, stream>>>(
            LE, ldLE, Es, ldE, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, seed, drop_mask, ldm, carry, ldc, norm, ldn);
    LAUNCH_CHECK();
    return NGCF_OK;
}

extern "C" int ngcf_layer_dense_f32(const float *LE, int64_t ldLE, const float *Es, int64_t ldEs, int64_t n_rows,
                                    int d_in, const float *W1, const float *b1, const float *W2, const float *b2,
                                    int d_out, float leaky, float drop_p, uint64_t drop_seed, const float *drop_mask,
                                    int64_t ld_mask, float *carry, int64_t ldc, float *norm, int64_t ldn, void *workspace,
                                    int64_t workspace_bytes, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!LE || !Es || !W1 || !b1 || !W2 || !b2 || !norm) return fail(NGCF_ERR_ARG, "layer_dense: null argument");
    if (n_rows < 0 || d_in <= 0 || d_out <= 0) return fail(NGCF_ERR_ARG, "layer_dense: bad sizes");
    int dop = dense_dop(d_out);
    if (dop < 0) return fail(NGCF_ERR_ARG, "layer_dense: d_out=%d > 512 is not supported", d_out);
    // Up to 128 output columns on a SMALL matrix (the Seoul graph: 5 940 rows): 128-row tiles are 47 workgroups on 256 CUs and the
    // 65 -> 65 layer of the reference's own configuration took 28-30 us; 32-row x 128-column tiles (four waves side by side, one
    // tile each) are 186 workgroups.  From 16 384 rows on (128 tiles of 128 rows) the tall tiles stay.
    const bool small_rows = dop <= 128 && n_rows <= 16384 && ngcf_opts().dense_small_tiles;
    if (small_rows) dop = 128;
    if (!(drop_p >= 0.f && drop_p < 1.f)) return fail(NGCF_ERR_ARG, "layer_dense: drop_p=%f not in [0,1)", drop_p);
    if (ldLE < d_in || ldEs < d_in || ldn < d_out || (carry && ldc < d_out) || (drop_mask && ld_mask < d_out))
        return fail(NGCF_ERR_ARG, "layer_dense: leading dimension too small");
    const int64_t need = ngcf_dense_workspace_bytes(d_in, d_out);
    if (!workspace || workspace_bytes < need)
        return fail(NGCF_ERR_WORKSPACE, "layer_dense: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    const int n_chunks = (d_in + NGCF_DC - 1) / NGCF_DC;
    float *Wt = reinterpret_cast<float *>(align_up((int64_t)(uintptr_t)workspace, 256));
    float *bias2 = Wt + (int64_t)n_chunks * NGCF_KC * dop;
    const bool al = (ldLE % 4 == 0) && (ldEs % 4 == 0) && aligned16(LE) && aligned16(Es);
    // 256 / 512 output columns and at most ONE workgroup per CU (<= 8 192 rows - the Seoul graph has 5 940): operands straight
    // from global memory / L2, no staging (layer_dense_direct_kernel).  tools/dense_wide_lab.py: 94 vs 114 us at 5 940 x 515 -> 512,
    // 50 vs 57 us at 256 -> 256, 97 vs 117 us at 8 192 x 512 -> 512; as soon as a CU gets a second workgroup the staged kernel
    // (two workgroups share a CU's LDS and matrix pipe; the direct kernel runs one wave per SIMD) wins clearly: 170 vs 277 us at
    // 12 288 rows, 1.08 vs 1.40 ms at 100 K.  NGCF_DENSE_DIRECT=0 / 2: never / at any row count.
    const int direct_env = ngcf_opts().dense_direct;
    const bool direct = direct_env && dop >= 256 && al && ldLE >= align_up(d_in, 4) && ldEs >= align_up(d_in, 4) && d_in >= 4 &&
                        n_rows > 0 && (n_rows <= 8192 || direct_env == 2);
    // 256 / 512 output columns as 96-row x 128-column workgroups, three row tiles per wave, the row norm in a second kernel
    // (layer_dense_tall_kernel): 248 workgroups for the Seoul graph's 5 940 rows where the direct kernel has 186.  Measured
    // (tools/dense_wide_lab.py, profiles/r03_dense_wide_lab.txt; tall / direct / staged, us incl. pack and row scale):
    //   512 columns: 5 940 rows 84 / 92 / 109; 6 144: 84 / 93 / 109; 7 000: 127 / 89 / 107 (a second round of workgroups; the direct
    //   kernel still has one per CU up to 8 192 rows); 9 000: 132 / 267 / 165; 100 K: 956 / 1 384 / 1 078; 1.1 M: 10 129 / 12 114 / 10 858
    //   256 columns: 5 940 rows 47 / 47 / 58 (124 workgroups: no gain); 9 000: 48 / 78 / 59; 100 K: 290 / 418 / 310; 200 K: 569 / 735 / 552
    // dense_tall = 0: never; 2: wherever the shape allows; 1: by these numbers.
    const int tall_env = ngcf_opts().dense_tall;
    const bool tall_pays = dop == 512 ? (n_rows <= 6144 || n_rows > 8192) : (n_rows > 8192 && n_rows <= 131072);
    if (tall_env && dop >= 256 && al && ldLE >= align_up(d_in, 4) && ldEs >= align_up(d_in, 4) && d_in >= 4 && n_rows > 0 &&
        (tall_pays || tall_env == 2)) {
        const int n_ct = dop / 32;
        pack_weights_tall_kernel<<<dim3((unsigned)(n_chunks * n_ct)), 256, 0, stream>>>(W1, b1, W2, b2, d_in, d_out, dop, Wt, bias2);
        LAUNCH_CHECK();
        const int64_t groups = (n_rows + kTallRows - 1) / kTallRows;
#define NGCF_TALL(L) \
    layer_dense_tall_kernel<L><<<dim3((unsigned)(groups * (n_ct / 4))), 256, 0, stream>>>( \
        LE, ldLE, Es, ldEs, n_rows, d_in, d_out, Wt, bias2, n_chunks, n_ct, leaky, drop_p, drop_seed, drop_mask, ld_mask, carry, ldc, norm, ldn)
#ifdef NGCF_LAB
        switch (ngcf_opts().dense_il_lab) {
        case 1: NGCF_TALL(1); break;
        case 2: NGCF_TALL(2); break;
        case 3: NGCF_TALL(3); break;
        case 4: NGCF_TALL(4); break;
        case 7: NGCF_TALL(7); break;
        default: NGCF_TALL(0); break;
        }
#else
        NGCF_TALL(0);
#endif
#undef NGCF_TALL
        LAUNCH_CHECK();
        if (dop == 256) row_scale_kernel<2><<<dim3((unsigned)((n_rows + 3) / 4)), 256, 0, stream>>>(norm, ldn, n_rows, d_out);
        else row_scale_kernel<4><<<dim3((unsigned)((n_rows + 3) / 4)), 256, 0, stream>>>(norm, ldn, n_rows, d_out);
        LAUNCH_CHECK();
        return NGCF_OK;
    }
    pack_weights_kernel<<<dim3((unsigned)(n_chunks * (dop / 32))), 256, 0, stream>>>(W1, b1, W2, b2, d_in, d_out, n_chunks, dop,
                                                                                    small_rows ? 1 : dop <= 128 ? dop / 32 : 4, Wt, bias2);
    LAUNCH_CHECK();
#define NGCF_DENSE(RW, CW, NT) \
    return launch_dense<RW, CW, NT>(al, n_rows, LE, ldLE, Es, ldEs, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, \
                                    drop_seed, drop_mask, ld_mask, carry, ldc, norm, ldn, stream)
    {
        // weights resident in LDS, no barriers (layer_dense_resident_kernel): large row counts at the 128-wide shapes, from two
        // tiles per wave on (128 -> 128, resident / staged us: 65 536 rows 59 / 59, 98 304 rows 95 / 85, 131 072 rows 96 / 104,
        // 262 144 rows 184 / 212, C3's 1.1 M rows 633 / 780; NGCF_DENSE_RESIDENT=0 keeps the staged kernel)
        const int resident = ngcf_opts().dense_resident;
        const int64_t lds_bytes = (int64_t)n_chunks * NGCF_KC * 128 * (int64_t)sizeof(float);
        if (resident && dop == 128 && al && ldLE >= align_up(d_in, 4) && ldEs >= align_up(d_in, 4) && d_in >= 4 &&
            lds_bytes <= 150 * 1024 && n_rows >= (int64_t)ngcf_opts().dense_resident_min_rows) {
            static bool attr_set[kMaxDevices] = {};      // the attribute is per device
            const int dev_i = current_device_slot();
            if (!attr_set[dev_i]) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                attr_set[dev_i] = true;
            }
#ifdef NGCF_LAB
            if (resident == 3) {
                static bool r2_set = false;
                if (!r2_set) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    r2_set = true;
                }
                layer_dense_resident2_kernel<<<dim3(kResWGs), kResWaves * 64, (size_t)lds_bytes, stream>>>(
                    LE, ldLE, Es, ldEs, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, drop_seed, drop_mask, ld_mask, carry, ldc,
                    norm, ldn);
                LAUNCH_CHECK();
                return NGCF_OK;
            }
#endif
#ifdef NGCF_LAB
            // resident == 2: the finished tile leaves under the next tile's K loop (layer_dense_resident_il_kernel; full 128
            // output columns, 8 or 9 chunks)
            if (resident == 2 && d_out == 128 && (n_chunks == 8 || n_chunks == 9) && ldn < (1 << 24) && ldc < (1 << 24) && ldn % 4 == 0 &&
                aligned16(norm) && (!carry || (ldc % 4 == 0 && aligned16(carry)))) {
                const int64_t il_lds = lds_bytes + kIlWaves * 2048;      // + two 1 KB slabs per wave
#define NGCF_IL(NCH, CARRY) \
    do { \
        static bool il_set[kMaxDevices] = {}; \
        if (!il_set[dev_i]) { \
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident_il_kernel<NCH, CARRY>), \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
            il_set[dev_i] = true; \
        } \
        layer_dense_resident_il_kernel<NCH, CARRY><<<dim3(kResWGs), kIlWaves * 64, (size_t)il_lds, stream>>>( \
            LE, ldLE, Es, ldEs, n_rows, d_in, d_out, Wt, bias2, leaky, drop_p, drop_seed, drop_mask, ld_mask, carry, ldc, norm, ldn); \
    } while (0)
#define NGCF_IL_LAB(L) \
    layer_dense_resident_il_kernel<8, true, L><<<dim3(kResWGs), kIlWaves * 64, (size_t)il_lds, stream>>>( \
        LE, ldLE, Es, ldEs, n_rows, d_in, d_out, Wt, bias2, leaky, drop_p, drop_seed, drop_mask, ld_mask, carry, ldc, norm, ldn)
                if (const int lab = ngcf_opts().dense_il_lab; lab && n_chunks == 8 && carry) {
                    static bool lab_set = false;
                    if (!lab_set) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident_il_kernel<8, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident_il_kernel<8, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident_il_kernel<8, true, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident_il_kernel<8, true, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(layer_dense_resident_il_kernel<8, true, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                        lab_set = true;
                    }
                    switch (lab) {
                    case 1: NGCF_IL_LAB(1); break;
                    case 2: NGCF_IL_LAB(2); break;
                    case 3: NGCF_IL_LAB(3); break;
                    case 6: NGCF_IL_LAB(6); break;       // no loads, every store to the wave's first tile
                    default: NGCF_IL_LAB(10); break;     // no loads, the stored values formed (LDS round trip, scaling) but not stored
                    }
                    LAUNCH_CHECK();
                    return NGCF_OK;
                }
#undef NGCF_IL_LAB
                if (n_chunks == 8) { if (carry) NGCF_IL(8, true); else NGCF_IL(8, false); }
                else { if (carry) NGCF_IL(9, true); else NGCF_IL(9, false); }
#undef NGCF_IL
                LAUNCH_CHECK();
                return NGCF_OK;
            }
#endif
            layer_dense_resident_kernel<<<dim3(kResWGs), kResWaves * 64, (size_t)lds_bytes, stream>>>(
                LE, ldLE, Es, ldEs, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, drop_seed, drop_mask, ld_mask, carry, ldc,
                norm, ldn);
            LAUNCH_CHECK();
            return NGCF_OK;
        }
    }
    if (direct) {
        const int64_t blocks = (n_rows + 31) / 32;
        if (dop == 256)
            layer_dense_direct_kernel<2, 4><<<dim3((unsigned)blocks), 128, 0, stream>>>(
                LE, ldLE, Es, ldEs, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, drop_seed, drop_mask, ld_mask, carry, ldc,
                norm, ldn);
        else
            layer_dense_direct_kernel<4, 4><<<dim3((unsigned)blocks), 256, 0, stream>>>(
                LE, ldLE, Es, ldEs, n_rows, d_in, d_out, Wt, bias2, n_chunks, leaky, drop_p, drop_seed, drop_mask, ld_mask, carry, ldc,
                norm, ldn);
        LAUNCH_CHECK();
        return NGCF_OK;
    }
    if (small_rows) NGCF_DENSE(1, 4, 1);
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
    const int64_t le = align_up(c->n_rows * align_up(d_in, 32) * (int64_t)sizeof(float), 256);   // LE rows start on 128-byte lines
    return a + b + le + 256;
}

extern "C" int ngcf_layer_fused_f32(const ngcf_csr_t *c, const float *Eg, int64_t ldEg, const float *Es, int64_t ldEs,
                                    int d_in, const float *W1, const float *b1, const float *W2, const float *b2,
                                    int d_out, float leaky, float drop_p, uint64_t drop_seed, const float *drop_mask,
                                    int64_t ld_mask, float *carry, int64_t ldc, float *norm, int64_t ldn, void *workspace,
                                    int64_t workspace_bytes, void *stream)
{
    if (!c) return fail(NGCF_ERR_ARG, "layer_fused: null csr");
    const int64_t need = ngcf_layer_workspace_bytes(c, d_in, d_out);
    if (need < 0) return fail(NGCF_ERR_ARG, "layer_fused: unsupported widths d_in=%d d_out=%d", d_in, d_out);
    if (!workspace || workspace_bytes < need)
        return fail(NGCF_ERR_WORKSPACE, "layer_fused: workspace %lld B < %lld B", (long long)workspace_bytes, (long long)need);
    char *ws = reinterpret_cast<char *>(align_up((int64_t)(uintptr_t)workspace, 256));
    const int64_t ldLE = align_up(d_in, 32);
    float *LE = reinterpret_cast<float *>(ws);
    ws += align_up(c->n_rows * ldLE * (int64_t)sizeof(float), 256);
    const int64_t spmm_ws = ngcf_spmm_workspace_bytes(c, d_in);
    void *ws_spmm = ws;
    ws += spmm_ws;
    const int64_t dense_ws = ngcf_dense_workspace_bytes(d_in, d_out);
    void *ws_dense = ws;
    // Widths that are not a multiple of 4 (65, 130, 515: NGCF.py:39-43) on a SMALL matrix: the product is launch-bound, and the
    // main + tail panel split costs four extra launches (pack, tail product, fix-up, unpack: 23 of 141 us on the Seoul-shaped C1).
    // When the gathered rows are 16-byte aligned and padded, the columns up to the next multiple of 4 are simply multiplied along:
    // whatever they hold ends up in the padding columns of LE, which the dense half never reads (it selects zeros there).
    // (Reading them cannot fault: the 16-byte piece that holds the last column of a 16-byte aligned row is read whole.  The
    // mirror's engine.spmm asks the same function for the products of the training path, so both paths give the same bits.)
    const int d_sp = ngcf_spmm_product_width(c, Eg, ldEg, d_in);
    int rc = ngcf_spmm_csr_f32(c, Eg, ldEg, d_sp, LE, ldLE, ws_spmm, spmm_ws, stream);
    if (rc != NGCF_OK) return rc;
    return ngcf_layer_dense_f32(LE, ldLE, Es, ldEs, c->n_rows, d_in, W1, b1, W2, b2, d_out, leaky, drop_p, drop_seed,
                                drop_mask, ld_mask, carry, ldc, norm, ldn, ws_dense, dense_ws, stream);
}

