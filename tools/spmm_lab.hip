// Experimental SpMM variants (NOT product code): d-slicing and unroll depth on top of the product's
// spmm_accumulate.  One wave per work unit (row or row segment); units come from tools/spmm_lab.py.
#include "../seoul_tourism_recommendation_ngcf_amd/csrc/spmm_device.h"

void prof_mark(hipStream_t, int) {}   // the lab does not time through the library

// LPR lanes cover one slice of S = 4*LPR floats of a gathered row; G = 64/LPR rows per wave-instruction.
template <int LPR, int U>
__global__ __launch_bounds__(256) void lab_spmm(const int64_t *__restrict__ ubeg, const int64_t *__restrict__ uend,
                                                const int64_t *__restrict__ udst, int64_t n_units, int64_t unit_blocks,
                                                const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                                const float *__restrict__ E, int64_t ldE, float *__restrict__ out,
                                                int64_t ldo)
{
    const int64_t slice = blockIdx.x / unit_blocks;
    const int64_t ub = blockIdx.x % unit_blocks;
    const int64_t unit = ub * 4 + (threadIdx.x >> 6);
    if (unit >= n_units) return;
    constexpr int S = LPR * 4;
    float4 acc[1];
    acc[0] = vzero4();
    spmm_accumulate<4, LPR, 1, U>(colidx, vals, ubeg[unit], uend[unit], E + slice * S, ldE, S, acc);
    spmm_store<4, LPR, 1>(acc, out + udst[unit] * ldo + slice * S, S);
}

extern "C" int lab_launch(int variant, const int64_t *ubeg, const int64_t *uend, const int64_t *udst, int64_t n_units,
                          const int32_t *colidx, const float *vals, const float *E, int64_t ldE, int d, float *out,
                          int64_t ldo, void *stream_)
{
    hipStream_t s = (hipStream_t)stream_;
    const int64_t ubk = (n_units + 3) / 4;
#define GO(LPR, U) \
    lab_spmm<LPR, U><<<dim3((unsigned)(ubk * (d / (4 * LPR)))), 256, 0, s>>>(ubeg, uend, udst, n_units, ubk, colidx, vals, E, ldE, out, ldo)
    switch (variant) {
    case 0: GO(32, 8); break;      // full width (d=128), the product kernel's shape
    case 1: GO(32, 4); break;
    case 2: GO(32, 16); break;
    case 3: GO(16, 8); break;      // 2 slices of 64 floats
    case 4: GO(16, 16); break;
    case 5: GO(8, 8); break;       // 4 slices of 32 floats (one 128-B line per gathered row)
    case 6: GO(8, 16); break;
    case 7: GO(8, 4); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// accumulate variant: out[dst] (+)= sum over the unit; used by the column-blocked experiment
template <int LPR, int U>
__global__ __launch_bounds__(256) void lab_spmm_acc(const int64_t *__restrict__ ubeg, const int64_t *__restrict__ uend,
                                                    int64_t n_units, int64_t unit_blocks,
                                                    const int32_t *__restrict__ colidx, const float *__restrict__ vals,
                                                    const float *__restrict__ E, int64_t ldE, float *__restrict__ out,
                                                    int64_t ldo, int accumulate)
{
    const int64_t slice = blockIdx.x / unit_blocks;
    const int64_t unit = (blockIdx.x % unit_blocks) * 4 + (threadIdx.x >> 6);
    if (unit >= n_units) return;
    constexpr int S = LPR * 4;
    const int64_t b = ubeg[unit], e = uend[unit];
    if (b == e && accumulate) return;
    float4 acc[1];
    acc[0] = vzero4();
    spmm_accumulate<4, LPR, 1, U>(colidx, vals, b, e, E + slice * S, ldE, S, acc);
    const int lane = threadIdx.x & 63;
    if (lane < LPR) {
        float4 *p = reinterpret_cast<float4 *>(out + unit * ldo + slice * S + lane * 4);
        if (accumulate) {
            const float4 t = *p;
            acc[0] = vadd(acc[0], t);
        }
        *p = acc[0];
    }
}

extern "C" int lab_launch_acc(int variant, const int64_t *ubeg, const int64_t *uend, int64_t n_units, const int32_t *colidx,
                              const float *vals, const float *E, int64_t ldE, int d, float *out, int64_t ldo, int accumulate,
                              void *stream_)
{
    hipStream_t s = (hipStream_t)stream_;
    const int64_t ubk = (n_units + 3) / 4;
#define GO2(LPR, U) \
    lab_spmm_acc<LPR, U><<<dim3((unsigned)(ubk * (d / (4 * LPR)))), 256, 0, s>>>(ubeg, uend, n_units, ubk, colidx, vals, E, ldE, out, ldo, accumulate)
    switch (variant) {
    case 0: GO2(8, 2); break;
    case 1: GO2(8, 4); break;
    case 2: GO2(8, 8); break;
    case 3: GO2(16, 2); break;
    case 4: GO2(16, 4); break;
    case 5: GO2(32, 2); break;
    case 6: GO2(32, 4); break;
    default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// mixed launch: every workgroup is either a d-sliced block of 4 user rows or an unsliced block of 4 item units;
// `sched[blockIdx]` >= 0: user block (slice-major index), < 0: item block -1-index.  Does interleaving the
// L2-friendly half with the fabric-bound half overlap them?
__global__ __launch_bounds__(256) void lab_mixed(const int32_t *__restrict__ sched, const int64_t *__restrict__ rowptr,
                                                 int64_t n_user, int64_t user_row_blocks, const int64_t *__restrict__ ubeg,
                                                 const int64_t *__restrict__ uend, const int64_t *__restrict__ udst,
                                                 int64_t n_units, const int32_t *__restrict__ colidx,
                                                 const float *__restrict__ vals, const float *__restrict__ E, int64_t ldE,
                                                 float *__restrict__ out, int64_t ldo)
{
    const int32_t sc = sched[blockIdx.x];
    const int wave = threadIdx.x >> 6;
    if (sc >= 0) {
        const int64_t slice = sc / user_row_blocks;
        const int64_t row = ((int64_t)sc % user_row_blocks) * 4 + wave;
        if (row >= n_user) return;
        float4 acc[1];
        acc[0] = vzero4();
        spmm_accumulate<4, 8, 1, 8>(colidx, vals, rowptr[row], rowptr[row + 1], E + slice * 32, ldE, 32, acc);
        spmm_store<4, 8, 1>(acc, out + row * ldo + slice * 32, 32);
    } else {
        const int64_t unit = (int64_t)(-1 - sc) * 4 + wave;
        if (unit >= n_units) return;
        float4 acc[1];
        acc[0] = vzero4();
        spmm_accumulate<4, 32, 1, 8>(colidx, vals, ubeg[unit], uend[unit], E, ldE, 128, acc);
        spmm_store<4, 32, 1>(acc, out + udst[unit] * ldo, 128);
    }
}

extern "C" int lab_launch_mixed(const int32_t *sched, int64_t n_blocks, const int64_t *rowptr, int64_t n_user,
                                int64_t user_row_blocks, const int64_t *ubeg, const int64_t *uend, const int64_t *udst,
                                int64_t n_units, const int32_t *colidx, const float *vals, const float *E, int64_t ldE,
                                float *out, int64_t ldo, void *stream_)
{
    lab_mixed<<<dim3((unsigned)n_blocks), 256, 0, (hipStream_t)stream_>>>(sched, rowptr, n_user, user_row_blocks, ubeg, uend, udst,
                                                                          n_units, colidx, vals, E, ldE, out, ldo);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
