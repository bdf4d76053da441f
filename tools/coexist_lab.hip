// Stand-in for a collective's kernel: a few long-lived workgroups that stream memory, to see what a persistent
// one-workgroup-per-CU launch (the L2-swept SpMM) does when it is not alone on the GPU.  Lab only.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(512) void hog_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, long n, int iters)
{
    for (int it = 0; it < iters; ++it)
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
extern "C" int hog_launch(const void *src, void *dst, long n_float4, int iters, int wgs, void *stream)
{
    hog_kernel<<<dim3(wgs), 512, 0, (hipStream_t)stream>>>((const float4 *)src, (float4 *)dst, n_float4, iters);
    return (int)hipGetLastError();
}
