// Micro-benchmark (development tool, not part of the product): fp32 VALU issue rates on gfx950 that
// decide the Chamfer inner-loop design — plain v_mul/v_add/v_min/v_fma vs packed v_pk_mul/v_pk_add/v_pk_fma.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o gpurun_out/ubench_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

#define ITERS 4096
#define REP8(X) X X X X X X X X

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, p4 = {r1, r0}, p5 = {r3, r2}, p6 = {r5, r4}, p7 = {r7, r6};
    f2 pa = {a, b};
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) {  // v_fma_f32 x8 independent
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (MODE == 1) {  // v_mul_f32 x8
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (MODE == 2) {  // v_min_f32 x8
            asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n"
                         "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (MODE == 3) {  // v_pk_mul_f32 x8 (16 lane-results)
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
        } else if (MODE == 4) {  // v_pk_add_f32 x8
            asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                         "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
        } else if (MODE == 5) {  // v_pk_fma_f32 x8
            asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                         "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
        } else if (MODE == 6) {  // v_min3_f32 x8
            asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n"
                         "v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b));
        } else if (MODE == 7) {  // v_sub_f32 with SGPR operand x8
            asm volatile("v_sub_f32 %0, %8, %0\n v_sub_f32 %1, %8, %1\n v_sub_f32 %2, %8, %2\n v_sub_f32 %3, %8, %3\n"
                         "v_sub_f32 %4, %8, %4\n v_sub_f32 %5, %8, %5\n v_sub_f32 %6, %8, %6\n v_sub_f32 %7, %8, %7\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(a));
        }
    }
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
double run(const char *name, int lanes_per_inst, int waves_per_simd)
{
    float *out;
    hipMalloc(&out, 4096);
    int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves -> one wave per SIMD per block
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double t = ms[ms.size() / 2] * 1e-3;
    double insts = (double)blocks * 4 * ITERS * 8;           // wave-instructions
    double lane_ops = insts * 64 * lanes_per_inst;
    printf("%-14s waves/SIMD=%d  %.3f ms  %.2f T lane-results/s  %.2f cycles/wave-inst/SIMD @2.4GHz\n", name, waves_per_simd,
           t * 1e3, lane_ops / t * 1e-12, t * 2.4e9 / ((double)ITERS * 8 * waves_per_simd));
    hipFree(out);
    return t;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 1, w);
        run<1>("v_mul_f32", 1, w);
        run<2>("v_min_f32", 1, w);
        run<6>("v_min3_f32", 1, w);
        run<7>("v_sub_f32(sgpr)", 1, w);
        run<3>("v_pk_mul_f32", 2, w);
        run<4>("v_pk_add_f32", 2, w);
        run<5>("v_pk_fma_f32", 2, w);
    }
    return 0;
}
