// Development tool: per-instruction fp32 VALU issue cost on gfx950 (cycles per wave64 instruction per SIMD at
// 8 waves/SIMD), VGPR vs SGPR operands, plain vs packed.  Decides the Chamfer inner-loop instruction mix.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define ITERS 2048
typedef float f2 __attribute__((ext_vector_type(2)));

#define OP8_V(INS)  asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" \
                                 INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
#define OP8_S(INS)  asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" \
                                 INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n" \
      : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(a));
#define OP8_PV(INS, MOD) asm volatile(INS " %0, %0, %8 " MOD "\n" INS " %1, %1, %8 " MOD "\n" INS " %2, %2, %8 " MOD "\n" INS " %3, %3, %8 " MOD "\n" \
                                 INS " %4, %4, %8 " MOD "\n" INS " %5, %5, %8 " MOD "\n" INS " %6, %6, %8 " MOD "\n" INS " %7, %7, %8 " MOD "\n" \
      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pa));
#define OP8_PS(INS, MOD) asm volatile(INS " %0, %8, %0 " MOD "\n" INS " %1, %8, %1 " MOD "\n" INS " %2, %8, %2 " MOD "\n" INS " %3, %8, %3 " MOD "\n" \
                                 INS " %4, %8, %4 " MOD "\n" INS " %5, %8, %5 " MOD "\n" INS " %6, %8, %6 " MOD "\n" INS " %7, %8, %7 " MOD "\n" \
      : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "s"(pa));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b)
{
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    f2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7}, p4 = {r1, r0}, p5 = {r3, r2}, p6 = {r5, r4}, p7 = {r7, r6};
    f2 pa = {a, b};
    for (int i = 0; i < ITERS; i++) {
        if (MODE == 0) { OP8_V("v_add_f32") }
        else if (MODE == 1) { OP8_V("v_sub_f32") }
        else if (MODE == 2) { OP8_V("v_mul_f32") }
        else if (MODE == 3) { OP8_V("v_min_f32") }
        else if (MODE == 4) { OP8_V("v_max_f32") }
        else if (MODE == 5) { OP8_S("v_add_f32") }
        else if (MODE == 6) { OP8_S("v_sub_f32") }
        else if (MODE == 7) { OP8_S("v_mul_f32") }
        else if (MODE == 8) { OP8_PV("v_pk_add_f32", "") }
        else if (MODE == 9) { OP8_PV("v_pk_mul_f32", "") }
        else if (MODE == 10) { OP8_PS("v_pk_add_f32", "op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]") }
        else if (MODE == 11) { OP8_PS("v_pk_mul_f32", "op_sel_hi:[0,1]") }
        else if (MODE == 12) {
            asm volatile("v_fmac_f32 %0, %8, %8\n v_fmac_f32 %1, %8, %8\n v_fmac_f32 %2, %8, %8\n v_fmac_f32 %3, %8, %8\n"
                         "v_fmac_f32 %4, %8, %8\n v_fmac_f32 %5, %8, %8\n v_fmac_f32 %6, %8, %8\n v_fmac_f32 %7, %8, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (MODE == 13) {   // cmp + cndmask pair
            asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a) : "vcc");
        } else if (MODE == 14) {   // v_sub_f32 VOP3 encoding with VGPRs (e64)
            OP8_V("v_sub_f32_e64")
        } else if (MODE == 15) {   // v_mul then v_add alternating (pipe mix)
            asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        } else if (MODE == 16) {   // v_fma as subtraction: fma(q, -1, t) with t in SGPR
            asm volatile("v_fma_f32 %0, %0, -1.0, %8\n v_fma_f32 %1, %1, -1.0, %8\n v_fma_f32 %2, %2, -1.0, %8\n v_fma_f32 %3, %3, -1.0, %8\n"
                         "v_fma_f32 %4, %4, -1.0, %8\n v_fma_f32 %5, %5, -1.0, %8\n v_fma_f32 %6, %6, -1.0, %8\n v_fma_f32 %7, %7, -1.0, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "s"(a));
        } else if (MODE == 17) {   // v_min_f32 with dpp? no: v_med3 as alt
            asm volatile("v_min3_f32 %0, %0, %8, %8\n v_min3_f32 %1, %1, %8, %8\n v_min3_f32 %2, %2, %8, %8\n v_min3_f32 %3, %3, %8, %8\n"
                         "v_min3_f32 %4, %4, %8, %8\n v_min3_f32 %5, %5, %8, %8\n v_min3_f32 %6, %6, %8, %8\n v_min3_f32 %7, %7, %8, %8\n"
                : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a));
        }
    }
    float s = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char *name)
{
    float *out;
    (void)hipMalloc(&out, 4096);
    const int w = 8;
    int blocks = 256 * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::vector<float> ms;
    for (int r = 0; r < 7; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float t;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    double t = ms[ms.size() / 2] * 1e-3;
    printf("%-34s %.3f ms  %.2f cycles/wave-inst/SIMD @2.4GHz\n", name, t * 1e3, t * 2.4e9 / ((double)ITERS * 8 * w));
    (void)hipFree(out);
}

int main()
{
    run<0>("v_add_f32 v,v"); run<1>("v_sub_f32 v,v"); run<2>("v_mul_f32 v,v"); run<3>("v_min_f32 v,v"); run<4>("v_max_f32 v,v");
    run<5>("v_add_f32 s,v"); run<6>("v_sub_f32 s,v"); run<7>("v_mul_f32 s,v");
    run<8>("v_pk_add_f32 v,v"); run<9>("v_pk_mul_f32 v,v");
    run<10>("v_pk_add_f32 s(bcast),-v"); run<11>("v_pk_mul_f32 s(bcast),v");
    run<12>("v_fmac_f32 v,v"); run<13>("v_cmp+v_cndmask (per pair of insts)"); run<14>("v_sub_f32_e64 v,v");
    run<15>("v_mul/v_add alternating"); run<16>("v_fma_f32 v,-1.0,s"); run<17>("v_min3_f32 v,v,v");
    return 0;
}
