// Experiment (NOT product code): the L2-swept SpMM of csrc/spmm_swept.hip with narrower slices - 8 lanes per entry
// (32 floats = one 128-B line per gathered row, 8 entries per round) against 16 lanes per entry, both with
// ds_bpermute broadcasts.  Narrower slices double the output rows that fit in LDS and the table rows per window.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
static constexpr int kRing = 8;
static constexpr int kRowBits = 24;
static constexpr int kColMask = (1 << kRowBits) - 1;

template <int LPE, int RW, int NW>
__global__ __launch_bounds__(NW * 64) void ldsacc8_kernel(const int64_t *__restrict__ tptr, const int32_t *__restrict__ e_pack,
                                                          const float *__restrict__ e_val, const int32_t *__restrict__ dst,
                                                          int n_rowpass, int n_win, int n_slices, const float *__restrict__ E,
                                                          int64_t ldE, float *__restrict__ out, int64_t ldo, unsigned *bar,
                                                          int max_spin, int lead)
{
    constexpr int SW = LPE * 4, EPR = 64 / LPE, CH = 64, NR = CH / EPR;   // NR rounds per 64-entry chunk
    __shared__ float acc_lds[NW * (RW + 1) * SW];
    __shared__ unsigned wg_cnt[kRing];
    __shared__ int perm_lds;
    __shared__ unsigned xcc_id;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane / LPE, p = lane % LPE;
    float *wacc = acc_lds + wave * ((RW + 1) * SW);
    if (threadIdx.x < kRing) wg_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
        perm_lds = lead;
    }
    __syncthreads();
    unsigned *ctr = bar + xcc_id * 32;
    const unsigned members = gridDim.x / 8;
    const bool sync = lead >= 0;
    int perm = lead, step0 = 0;
    const unsigned ld_bytes = (unsigned)ldE * 4u, lane_off = p * 16;
    for (int rp = 0; rp < n_rowpass; ++rp) {
        const int64_t task = ((int64_t)rp * gridDim.x + blockIdx.x) * NW + wave;
        const int64_t *tp = tptr + task * n_win;
        const int64_t beg = tp[0], end = tp[n_win];
        for (int slice = 0; slice < n_slices; ++slice, step0 += n_win) {
            const char *Eb = reinterpret_cast<const char *>(E + slice * SW);
            for (int i = lane; i < (RW + 1) * SW; i += 64) wacc[i] = 0.f;
            int b = 0;
            int64_t wend = tp[1], wend_next = n_win > 1 ? tp[2] : end;
            auto arrive = [&](int s) {
                if (sync && lane == 0) {
                    const unsigned old = atomicAdd(&wg_cnt[s % kRing], 1u);
                    if ((old + 1) % NW == 0) __hip_atomic_fetch_add(ctr + (s % kRing), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            auto cross = [&](int64_t pos, bool crossed) {
                while (b < n_win - 1 && pos >= wend) {
                    arrive(step0 + b);
                    ++b;
                    wend = wend_next;
                    wend_next = b + 2 <= n_win ? tp[b + 2] : end;
                    crossed = true;
                }
                const int s = step0 + b;
                if (crossed && sync && max_spin > 0 && perm < s) {
                    perm = __hip_atomic_load(&perm_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    int spins = 0;
                    while (perm < s) {
                        const unsigned mine = lane < kRing ? __hip_atomic_load(ctr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                        int t = perm - lead;
                        for (int k = 0; k < kRing - 2 - lead; ++k, ++t) {
                            const unsigned have = __builtin_amdgcn_readlane(mine, t % kRing);
                            if (have < members * (unsigned)(t / kRing + 1)) break;
                        }
                        perm = t + lead;
                        if (perm >= s) break;
                        if (++spins >= max_spin) {
                            max_spin = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (lane == 0) atomicMax(&perm_lds, perm);
                }
            };
            const int idle_pk = end > beg ? (RW << kRowBits) | (e_pack[beg] & kColMask) : 0;
            auto load_entries = [&](int64_t pos, int &pk, float &v) {
                const int64_t idx = pos + lane;
                pk = idle_pk;
                v = 0.f;
                if (idx < end) {
                    pk = e_pack[idx];
                    v = e_val[idx];
                }
            };
            auto issue = [&](f32x4(&x)[NR], int pk) {
                const unsigned off = (unsigned)(pk & kColMask) * ld_bytes;
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const unsigned o = (unsigned)__builtin_amdgcn_ds_bpermute((u * EPR + g) << 2, (int)off) + lane_off;
                    x[u] = *reinterpret_cast<const f32x4 *>(Eb + o);
                }
            };
            auto accumulate = [&](const f32x4(&x)[NR], int pk, float v) {
#pragma unroll
                for (int u = 0; u < NR; ++u) {
                    const unsigned pku = (unsigned)__builtin_amdgcn_ds_bpermute((u * EPR + g) << 2, pk);
                    const float vu = __int_as_float(__builtin_amdgcn_ds_bpermute((u * EPR + g) << 2, __float_as_int(v)));
                    f32x4 *a = reinterpret_cast<f32x4 *>(wacc + (pku >> kRowBits) * SW + p * 4);
                    f32x4 t = *a;
                    t.x = fmaf(vu, x[u].x, t.x);
                    t.y = fmaf(vu, x[u].y, t.y);
                    t.z = fmaf(vu, x[u].z, t.z);
                    t.w = fmaf(vu, x[u].w, t.w);
                    *a = t;
                }
            };
            if (end > beg) {
                int pkA, pkB, pkC;
                float vA, vB, vC;
                f32x4 xa[NR], xb[NR];
                load_entries(beg, pkA, vA);
                load_entries(beg + CH, pkB, vB);
                cross(beg, true);
                issue(xa, pkA);
                for (int64_t pos = beg; pos < end; pos += 2 * CH) {
                    // chunk at pos (set A): entries two chunks ahead, gathers one chunk ahead
                    load_entries(pos + 2 * CH, pkC, vC);
                    if (pos + CH < end) {
                        cross(pos + CH, false);
                        issue(xb, pkB);
                    }
                    accumulate(xa, pkA, vA);
                    if (pos + CH >= end) break;
                    // chunk at pos + CH (set B)
                    load_entries(pos + 3 * CH, pkA, vA);
                    if (pos + 2 * CH < end) {
                        cross(pos + 2 * CH, false);
                        issue(xa, pkC);
                    }
                    accumulate(xb, pkB, vB);
                    pkB = pkA;           // entries of chunk pos + 3 CH
                    vB = vA;
                    pkA = pkC;           // entries of chunk pos + 2 CH (their gathers are in xa)
                    vA = vC;
                }
            }
            cross(INT64_MAX - 1, false);
            arrive(step0 + n_win - 1);
            for (int r0 = 0; r0 < RW; r0 += EPR) {
                const int r = r0 + g;
                if (r < RW) {
                    const int drow = dst[task * RW + r];
                    if (drow >= 0)
                        *reinterpret_cast<f32x4 *>(out + (int64_t)drow * ldo + slice * SW + p * 4) =
                            *reinterpret_cast<const f32x4 *>(wacc + r * SW + p * 4);
                }
            }
        }
    }
}

#define LAB8(L, R, W)                                                                                                            \
    ldsacc8_kernel<L, R, W><<<dim3(256), W * 64, 0, stream>>>(tptr, e_pack, e_val, dst, n_rowpass, n_win, d / (L * 4), E, ldE, out, \
                                                              ldo, bar, max_spin, lead)

extern "C" int ldsacc8_launch(int lpe, int waves, const int64_t *tptr, const int32_t *e_pack, const float *e_val, const int32_t *dst,
                              int n_rowpass, int n_win, int d, const float *E, int64_t ldE, float *out, int64_t ldo, unsigned *bar,
                              int max_spin, int lead, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (lead > kRing - 4) return 4;
    if (hipMemsetAsync(bar, 0, sizeof(unsigned) * 32 * 8, stream) != hipSuccess) return 1;
    if (lpe == 8 && waves == 16) LAB8(8, 72, 16);
    else if (lpe == 8 && waves == 8) LAB8(8, 144, 8);
    else if (lpe == 16 && waves == 16) LAB8(16, 36, 16);
    else if (lpe == 16 && waves == 8) LAB8(16, 72, 8);
    else return 2;
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
