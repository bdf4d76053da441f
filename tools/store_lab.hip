// r04 lab: how fast do 512-byte rows leave a CU?  (The row-sparse transposed product writes every row of a [1.1 M, 128] fp32
// matrix; its row writes took 0.6 ms of 0.9 - this isolates the store pattern from everything else in that kernel.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_lab tools/store_lab.hip && /tmp/store_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int64_t N = 1100000;
constexpr int D = 128;

// A: persistent, 16 waves a CU, a wave writes 16 consecutive rows, a dword a lane (two stores a row) - the pattern of the kernel
template <int WAVES, int ROWS, int LDS_KB>
__global__ __launch_bounds__(WAVES * 64) void rows_dword(float *out, int64_t n_rows)
{
    extern __shared__ uint32_t lds[];
    if (LDS_KB) lds[threadIdx.x] = 1;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n_units = (n_rows + ROWS - 1) / ROWS;
    for (int64_t u = (int64_t)blockIdx.x * WAVES + wave; u < n_units; u += (int64_t)gridDim.x * WAVES) {
        const int64_t r0 = (n_units - 1 - u) * ROWS;
        for (int r = 0; r < ROWS && r0 + r < n_rows; ++r) {
            float *dst = out + (r0 + r) * D;
            dst[lane] = 0.f;
            dst[lane + 64] = 0.f;
        }
    }
}
// B: the same rows, 16 bytes a lane: a store instruction covers two rows
template <int WAVES, int ROWS>
__global__ __launch_bounds__(WAVES * 64) void rows_x4(float *out, int64_t n_rows)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n_units = (n_rows + ROWS - 1) / ROWS;
    for (int64_t u = (int64_t)blockIdx.x * WAVES + wave; u < n_units; u += (int64_t)gridDim.x * WAVES) {
        const int64_t r0 = (n_units - 1 - u) * ROWS;
        for (int r = 0; r < ROWS && r0 + r + 1 < n_rows; r += 2)
            reinterpret_cast<float4 *>(out + (r0 + r) * D)[lane] = make_float4(0, 0, 0, 0);
    }
}
// C: a plain fill, one float4 a thread
__global__ void fill_x4(float4 *out, int64_t n4)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) out[i] = make_float4(0, 0, 0, 0);
}
// D: one wave per row, four rows a workgroup (the shape of spmm_t_rows_kernel)
__global__ __launch_bounds__(256) void row_per_wave(float *out, int64_t n_rows)
{
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    out[row * D + lane] = 0.f;
    out[row * D + lane + 64] = 0.f;
}

template <class F>
static void timeit(const char *name, F f)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-70s %8.1f us  %6.2f TB/s\n", name, ms * 100, N * D * 4.0 / (ms * 1e-4) / 1e12);
}

int main()
{
    float *out;
    CHECK(hipMalloc(&out, N * D * sizeof(float)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(rows_dword<16, 16, 137>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    timeit("A  persistent 256 x 16 waves, 16-row units, dword stores, 137 KB LDS", [&] { rows_dword<16, 16, 137><<<256, 1024, 137 * 1024>>>(out, N); });
    timeit("A' the same without the LDS (still one workgroup a CU)", [&] { rows_dword<16, 16, 0><<<256, 1024, 0>>>(out, N); });
    timeit("A2 persistent 512 x 16 waves (two workgroups a CU)", [&] { rows_dword<16, 16, 0><<<512, 1024, 0>>>(out, N); });
    timeit("A4 persistent 1024 x 8 waves", [&] { rows_dword<8, 16, 0><<<1024, 512, 0>>>(out, N); });
    timeit("A8 persistent 2048 x 4 waves, 4-row units", [&] { rows_dword<4, 4, 0><<<2048, 256, 0>>>(out, N); });
    timeit("B  persistent 256 x 16 waves, 16-row units, dwordx4 stores", [&] { rows_x4<16, 16><<<256, 1024>>>(out, N); });
    timeit("B2 persistent 512 x 16 waves, dwordx4 stores", [&] { rows_x4<16, 16><<<512, 1024>>>(out, N); });
    timeit("C  plain fill, a float4 a thread", [&] { fill_x4<<<(unsigned)((N * D / 4 + 255) / 256), 256>>>(reinterpret_cast<float4 *>(out), N * D / 4); });
    timeit("D  a wave a row, 4 rows a workgroup (275 000 workgroups)", [&] { row_per_wave<<<(unsigned)((N + 3) / 4), 256>>>(out, N); });
    CHECK(hipDeviceSynchronize());
    return 0;
}
