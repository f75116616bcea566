// dev: how fast can gfx950 read a 64.5 MB array once (the size of the SMPL-X blendshape matrix)?  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void rd(const f4 *__restrict__ p, size_t n4, float *out)
{
    size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (; i + 256 * (U - 1) < n4; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = p[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1;
}
template <int U>
float run(const f4 *d, size_t n4, float *o, int blocks, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(rd<U>, dim3(blocks), dim3(256), 0, 0, d, n4, o);
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(rd<U>, dim3(blocks), dim3(256), 0, 0, d, n4, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main()
{
    size_t bytes = 64ull * 1024 * 1024 + 512 * 1024, n4 = bytes / 16;
    f4 *d; float *o; hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 0, bytes);
    int blks[] = {256, 512, 1024, 2048, 4096, 16384};
    for (int b : blks) {
        float t4 = run<4>(d, n4, o, b, 50), t8 = run<8>(d, n4, o, b, 50), t16 = run<16>(d, n4, o, b, 50);
        printf("blocks %5d: U4 %.1f us (%.0f GB/s)  U8 %.1f us (%.0f GB/s)  U16 %.1f us (%.0f GB/s)\n", b, t4, bytes / t4 * 1e-3, t8, bytes / t8 * 1e-3,
               t16, bytes / t16 * 1e-3);
    }
    // big array (beyond the 256 MB Infinity Cache) for the HBM number
    size_t big = 2048ull * 1024 * 1024; f4 *D; hipMalloc(&D, big); hipMemset(D, 0, big);
    float tb = run<8>(D, big / 16, o, 16384, 10);
    printf("2 GB: %.1f us (%.0f GB/s)\n", tb, big / tb * 1e-3);
    return 0;
}
