// Experiment (NOT product code): swept SpMM with the output rows resident in LDS.
// A persistent grid (one 1024-thread workgroup per CU) sweeps the gathered table in column windows; every wave owns
// a few output rows whose accumulators stay in LDS for the whole sweep.  Each entry's contribution is added with a
// plain LDS read-modify-write (ds_add_f32 was measured 15x slower), so the plan must keep the entries that share a
// round (one wave instruction) on distinct rows.  d is walked in slices of 4*LPE floats so that more output rows
// fit the chip's LDS at once (more re-use of a fetched table row per XCD).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U> __device__ inline int row_bcast(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, 0x150 + U, 0xf, 0xf, false);
}
template <int U> __device__ inline float row_bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + U, 0xf, 0xf, false));
}

static constexpr int kRing = 8;

// LPE lanes share one entry (LPE*4 floats of the gathered row); RW accumulator rows per wave
template <int LPE, int RW, int ACC, int kWaves>
__global__ __launch_bounds__(kWaves * 64) void ldsacc_kernel(const int64_t *__restrict__ tptr, const int32_t *__restrict__ e_pack,
                                                      const float *__restrict__ e_val, const int32_t *__restrict__ dst,
                                                      int n_rowpass, int n_win, int n_slices, const float *__restrict__ E,
                                                      int64_t ldE, float *__restrict__ out, int64_t ldo, unsigned *bar,
                                                      int max_spin, int lead)
{
    constexpr int SW = LPE * 4;                 // floats per slice
    constexpr int EPR = 64 / LPE;               // entries per round (one wave instruction)
    constexpr int CH = 16 * EPR;                // entries per chunk (16 rounds)
    __shared__ float acc_lds[kWaves * RW * SW];
    __shared__ unsigned wg_cnt[kRing];
    __shared__ int perm_lds;                    // highest sweep step this workgroup knows to be permitted
    __shared__ unsigned xcc_id;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, k16 = lane & 15;
    const int g = lane / LPE, p = lane % LPE;   // group in the wave, position in the slice
    const int held = LPE == 16 ? k16 * 4 + q : k16 * 2 + (q >> 1);   // chunk entry this lane keeps for the DPP broadcasts
    float *wacc = acc_lds + wave * (RW * SW);
    if (threadIdx.x < kRing) wg_cnt[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        xcc_id = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;
        perm_lds = lead;                        // steps 0..lead need nobody
    }
    __syncthreads();
    unsigned *ctr = bar + xcc_id * 32;          // kRing rotating monotonic counters per XCD
    const unsigned members = gridDim.x / 8;
    const bool sync = lead >= 0;
    int perm = lead;                            // this wave's copy of perm_lds
    int step0 = 0;                              // sweep steps finished before this pass
    f32x4 sink = {0.f, 0.f, 0.f, 0.f};
    for (int rp = 0; rp < n_rowpass; ++rp) {
        const int64_t task = ((int64_t)rp * gridDim.x + blockIdx.x) * kWaves + wave;
        const int64_t *tp = tptr + task * n_win;
        const int64_t beg = tp[0], end = tp[n_win];
        for (int slice = 0; slice < n_slices; ++slice, step0 += n_win) {
            const char *Eb = reinterpret_cast<const char *>(E + slice * SW + p * 4);
            const unsigned ld_bytes = (unsigned)ldE * 4u;
            for (int i = lane; i < RW * SW; i += 64) wacc[i] = 0.f;
            int b = 0;
            int64_t wend = tp[1], wend_next = n_win > 1 ? tp[2] : end;
            auto arrive = [&](int s) {           // this wave has left sweep step s behind
                if (sync && lane == 0) {
                    const unsigned old = atomicAdd(&wg_cnt[s % kRing], 1u);
                    if ((old + 1) % kWaves == 0)
                        __hip_atomic_fetch_add(ctr + (s % kRing), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            };
            auto cross = [&](int64_t pos) {      // windows left behind by a wave whose next entry is `pos`, then permission
                bool crossed = false;
                while (b < n_win - 1 && pos >= wend) {
                    arrive(step0 + b);
                    ++b;
                    wend = wend_next;
                    wend_next = b + 2 <= n_win ? tp[b + 2] : end;
                    crossed = true;
                }
                const int s = step0 + b;         // the step being entered: needs step s-1-lead finished by the whole XCD
                if (crossed && sync && max_spin > 0 && perm < s) {
                    perm = __hip_atomic_load(&perm_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    int spins = 0;
                    while (perm < s) {
                        // lanes 0..kRing-1 read the ring; step t is finished when its slot reached members*(t/kRing+1)
                        unsigned mine = lane < kRing ? __hip_atomic_load(ctr + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                        int t = perm - lead;     // first step not yet known finished
                        for (int k = 0; k < kRing - 2 - lead; ++k, ++t) {
                            const unsigned have = __builtin_amdgcn_readlane(mine, t % kRing);
                            if (have < members * (unsigned)(t / kRing + 1)) break;
                        }
                        perm = t + lead;         // steps <= perm may start
                        if (perm >= s) break;
                        if (++spins >= max_spin) {
                            max_spin = 0;        // the XCD's workgroups are not resident together: stop waiting for good
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (lane == 0) atomicMax(&perm_lds, perm);
                }
            };
            auto load_entries = [&](int64_t pos, int &pk, float &v) {
                const int64_t idx = pos + held;
                pk = (int)0x80000000;
                v = 0.f;
                if (idx < end) {
                    pk = e_pack[idx];
                    v = e_val[idx];
                }
            };
            if (end > beg) {
                const unsigned idle_off = (unsigned)(e_pack[beg] & 0xffffff) * ld_bytes;
                int pkA, pkB;
                float vA, vB;
                f32x4 xa[8], xb[8];
                unsigned offA;
#define LAB_GATHER(x, u, j) x[j] = *reinterpret_cast<const f32x4 *>(Eb + row_bcast<u>((int)offA));
#define LAB_GATHER_LO(x) LAB_GATHER(x, 0, 0) LAB_GATHER(x, 1, 1) LAB_GATHER(x, 2, 2) LAB_GATHER(x, 3, 3) LAB_GATHER(x, 4, 4) LAB_GATHER(x, 5, 5) LAB_GATHER(x, 6, 6) LAB_GATHER(x, 7, 7)
#define LAB_GATHER_HI(x) LAB_GATHER(x, 8, 0) LAB_GATHER(x, 9, 1) LAB_GATHER(x, 10, 2) LAB_GATHER(x, 11, 3) LAB_GATHER(x, 12, 4) LAB_GATHER(x, 13, 5) LAB_GATHER(x, 14, 6) LAB_GATHER(x, 15, 7)
#define LAB_ACC(x, u, j)                                                                  \
    {                                                                                     \
        const int pku = row_bcast<u>(pkA);                                                \
        const float vu = row_bcast<u>(vA);                                                \
        if (pku >= 0) {                                                                   \
            if (ACC == 3) {                                                               \
                f32x4 *a = reinterpret_cast<f32x4 *>(wacc + (pku >> 24) * SW + p * 4);    \
                f32x4 t = *a;                                                             \
                t += vu * x[j];                                                           \
                *a = t;                                                                   \
            } else {                                                                      \
                sink += vu * x[j];                                                        \
            }                                                                             \
        }                                                                                 \
    }
#define LAB_ACC_LO(x) LAB_ACC(x, 0, 0) LAB_ACC(x, 1, 1) LAB_ACC(x, 2, 2) LAB_ACC(x, 3, 3) LAB_ACC(x, 4, 4) LAB_ACC(x, 5, 5) LAB_ACC(x, 6, 6) LAB_ACC(x, 7, 7)
#define LAB_ACC_HI(x) LAB_ACC(x, 8, 0) LAB_ACC(x, 9, 1) LAB_ACC(x, 10, 2) LAB_ACC(x, 11, 3) LAB_ACC(x, 12, 4) LAB_ACC(x, 13, 5) LAB_ACC(x, 14, 6) LAB_ACC(x, 15, 7)
                load_entries(beg, pkA, vA);
                cross(beg);
                offA = pkA < 0 ? idle_off : (unsigned)(pkA & 0xffffff) * ld_bytes;
                LAB_GATHER_LO(xa)
                for (int64_t pos = beg; pos < end; pos += CH) {
                    load_entries(pos + CH, pkB, vB);
                    const bool hi = pos + 8 * EPR < end;
                    if (hi) { LAB_GATHER_HI(xb) }
                    LAB_ACC_LO(xa)
                    const unsigned offB = pkB < 0 ? idle_off : (unsigned)(pkB & 0xffffff) * ld_bytes;
                    if (pos + CH < end) {
                        cross(pos + CH);
                        offA = offB;
                        LAB_GATHER_LO(xa)        // first half of the next chunk, in flight during the second half's adds
                    }
                    if (hi) { LAB_ACC_HI(xb) }
                    pkA = pkB;
                    vA = vB;
                }
            }
            cross(INT64_MAX - 1);                // leave the remaining windows (b ends at n_win-1) ...
            arrive(step0 + n_win - 1);           // ... and the last one
            // write this wave's rows (its own LDS rows: the wave's LDS operations complete in order)
            for (int r0 = 0; r0 < RW; r0 += EPR) {
                const int r = r0 + g;
                if (r < RW) {
                    const int drow = dst[task * RW + r];
                    if (drow >= 0) {
                        f32x4 o = *reinterpret_cast<const f32x4 *>(wacc + r * SW + p * 4);
                        if (ACC == 2) o += sink;
                        *reinterpret_cast<f32x4 *>(out + (int64_t)drow * ldo + slice * SW + p * 4) = o;
                    }
                }
            }
        }
    }
}

#define LAB_LAUNCH(L, R, A, W)                                                                                          \
    ldsacc_kernel<L, R, A, W><<<dim3(256), W * 64, 0, stream>>>(tptr, e_pack, e_val, dst, n_rowpass, n_win, d / (L * 4), E, ldE, \
                                                                out, ldo, bar, max_spin, lead)

extern "C" int ldsacc_launch(int lpe, int acc, int waves, const int64_t *tptr, const int32_t *e_pack, const float *e_val,
                             const int32_t *dst, int n_rowpass, int n_win, int d, const float *E, int64_t ldE, float *out,
                             int64_t ldo, unsigned *bar, int max_spin, int lead, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (lead > kRing - 4) return 4;
    if (hipMemsetAsync(bar, 0, sizeof(unsigned) * 32 * 8, stream) != hipSuccess) return 1;
    if (lpe == 16 && acc == 2 && waves == 16) LAB_LAUNCH(16, 26, 2, 16);
    else if (lpe == 16 && acc == 3 && waves == 16) LAB_LAUNCH(16, 26, 3, 16);
    else if (lpe == 32 && acc == 3 && waves == 16) LAB_LAUNCH(32, 13, 3, 16);
    else if (lpe == 16 && acc == 2 && waves == 8) LAB_LAUNCH(16, 52, 2, 8);
    else if (lpe == 16 && acc == 3 && waves == 8) LAB_LAUNCH(16, 52, 3, 8);
    else if (lpe == 32 && acc == 3 && waves == 8) LAB_LAUNCH(32, 26, 3, 8);
    else return 2;
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
