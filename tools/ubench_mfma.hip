// Development tool: issue rate of v_mfma_f32_16x16x4_f32 on gfx950 (N independent accumulators, W waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
#define ITERS 2048
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f4){0, 0, 0, 0};
    float av = a + threadIdx.x, bv = b - threadIdx.x;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456f) out[threadIdx.x] = s;
}
template <int NACC>
void run(int w)
{
    float *out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(256 * w), dim3(256), 0, 0, out, 1.0f, 2.0f);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float t; (void)hipEventElapsedTime(&t, e0, e1); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double t = ms[2] * 1e-3;
    double n = (double)ITERS * NACC * w;     // MFMAs per SIMD
    double flops = (double)256 * w * 4 * ITERS * NACC * 2048.0;
    printf("NACC=%d waves/SIMD=%d: %.3f ms, %.1f cycles/MFMA/SIMD @2.4GHz, %.1f TFLOP/s\n", NACC, w, t * 1e3, t * 2.4e9 / n, flops / t * 1e-12);
    (void)hipFree(out);
}
int main() { run<1>(1); run<2>(1); run<4>(1); run<8>(1); run<8>(2); run<4>(2); run<8>(4); return 0; }
