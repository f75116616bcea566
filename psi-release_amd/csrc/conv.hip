// 3x3 convolutions of the scene trunk (stride 1, padding 1, no dilation) as a hand-written implicit GEMM on the bf16 matrix cores, gfx950.
//
// Replaces, for the BasicBlocks of the ResNet-18 prefix the reference builds (cvae.py:427-435: torchvision resnet18 layer1 = 4 x
// Conv2d(64,64,3,1,1), layer2 = Conv2d(128,128,3,1,1) x 3 behind the strided first one) and the 128 -> 128 head convolution of
// BodyLocalPoseVAE (net_layers.py:160-164), the library convolution in BOTH directions that are convolutions: the forward pass and the
// input gradient (dX = conv(dY, W rotated by 180 degrees with its channel axes swapped) — the same kernel on a re-laid-out weight).
// The weight gradient stays with the library.
//
// Layout: activations NHWC bf16 (torch channels_last), weights [Cout][kh][kw][Cin] bf16 (a channels_last Conv2d weight), fp32 accumulate,
// bf16 output (+ optional fp32 bias).  GEMM view: D[co][pixel] = sum_{tap, ci} W[co][tap][ci] * X[pixel + tap][ci]; with channels fastest,
// the eight consecutive contraction elements an MFMA lane needs are eight consecutive channels of one pixel / one filter tap: 16 bytes.
//
// Workgroup = 4 waves, wave tile = 64 output pixels x 64 output channels (2 x 2 v_mfma_f32_32x32x16_bf16 accumulators):
//   Cin = 64 : 256 pixels (8 rows x 32 columns of one image) x 64 output channels per workgroup
//   Cin = 128: 128 pixels (8 x 16) x 128 output channels
// The input tile WITH ITS HALO ((TH + 2) x (TW + 2) pixels, all input channels: 48 KB) is staged in LDS once and serves all nine taps —
// every input byte crosses L2 -> LDS once per workgroup instead of nine times; the weights of one tap (Cout_tile x Cin) are staged per
// tap, the next tap's global loads in flight while the current tap is multiplied (register-staged double buffer).  Pixel / filter rows in
// LDS are padded by 8 elements, which makes the 16-byte operand reads of 16 adjacent lanes fall on 16 different bank groups.
// Output: a lane ends up with four consecutive output channels of one pixel per accumulator quarter -> 8-byte stores.
#include "psi_internal.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int CIN, int WPX, int WCO, int TW>
__global__ __launch_bounds__(256) void conv3x3_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ w, const float *__restrict__ bias,
                                                      __bf16 *__restrict__ y, int N, int H, int W, int COUT)
{
    constexpr int PX = WPX * 64, TH = PX / TW, COT = WCO * 64, P = CIN + 8, HW_ = TW + 2, HH_ = TH + 2, CH = CIN / 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 (*Xs)[P] = (__bf16 (*)[P])smem;                                        // [HH_ * HW_][P]
    __bf16 (*Ws)[COT][P] = (__bf16 (*)[COT][P])(smem + (size_t)HH_ * HW_ * P * 2);   // [2][COT][P]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wp = wv % WPX, wc = wv / WPX;
    const int li = lane & 31, kb = (lane >> 5) * 8;
    const int tiles_w = W / TW, tiles_h = H / TH;
    int t = blockIdx.x;
    const int tx0 = (t % tiles_w) * TW;
    t /= tiles_w;
    const int ty0 = (t % tiles_h) * TH, n = t / tiles_h;
    const int co0 = blockIdx.y * COT;

    // ---- weights of one tap: COT rows of CIN channels = COT * CH 16-byte pieces.  A tap's MFMAs take 0.2-0.5 us, an L2 round trip
    // about 1 us, so the weights of tap t + PF are requested while tap t is multiplied (a ring of PF register sets; with a prefetch
    // distance of one tap every tap waited for its weights and the kernel took 18 us instead of ~10)
    constexpr int WLD = (COT * CH + 255) / 256, PF = 4;
    u4 wr[PF][WLD];
    auto load_w = [&](int tap, u4 (&r)[WLD]) {
#pragma unroll
        for (int i = 0; i < WLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            if (COT * CH % 256 == 0 || idx < COT * CH) {
                const int row = idx / CH, c = idx % CH;
                r[i] = *(const u4 *)(w + ((size_t)(co0 + row) * 9 + tap) * CIN + c * 8);
            }
        }
    };
    auto store_w = [&](int buf, const u4 (&r)[WLD]) {
#pragma unroll
        for (int i = 0; i < WLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            if (COT * CH % 256 == 0 || idx < COT * CH) {
                const int row = idx / CH, c = idx % CH;
                *(u4 *)&Ws[buf][row][c * 8] = r[i];
            }
        }
    };
#pragma unroll
    for (int tp = 0; tp < PF; tp++) load_w(tp, wr[tp]);
    // ---- input tile with halo (zero outside the image): every thread requests ALL of its 16-byte pieces before it stores the first one
    // (a load -> store loop is one dependent round trip to L2 / memory per piece: 11 of them made this kernel 4x slower than its MFMAs)
    constexpr int NPC = HH_ * HW_ * CH, XLD = (NPC + 255) / 256;
    {
        u4 xr[XLD];
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            xr[i] = (u4){0u, 0u, 0u, 0u};
            if (idx < NPC) {
                const int pix = idx / CH, c = idx % CH;
                const int iy = ty0 - 1 + pix / HW_, ix = tx0 - 1 + pix % HW_;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) xr[i] = *(const u4 *)(x + (((size_t)n * H + iy) * W + ix) * CIN + c * 8);
            }
        }
        __builtin_amdgcn_sched_barrier(0);                   // keep the loads together: the scheduler must not sink them next to their stores
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            if (idx < NPC) *(u4 *)&Xs[idx / CH][(idx % CH) * 8] = xr[i];
        }
    }
    store_w(0, wr[0]);
    __syncthreads();

    f16v acc[2][2];                                          // [co tile][pixel tile]
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[a][b][i] = 0.0f;
    int pix_base[2];                                         // LDS pixel index of (my pixel, tap (0,0))
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        const int p = wp * 64 + pt * 32 + li;
        pix_base[pt] = (p / TW) * HW_ + (p % TW);
    }
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {                      // fully unrolled: the ring slots are compile-time register sets
        const int buf = tap & 1;
        if (tap + PF < 9) load_w(tap + PF, wr[tap % PF]);    // slot tap % PF held tap `tap`, which is already in LDS
        const int toff = (tap / 3) * HW_ + (tap % 3);
#pragma unroll
        for (int c0 = 0; c0 < CIN; c0 += 16) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int ct = 0; ct < 2; ct++) a[ct] = *(const bf16x8 *)&Ws[buf][wc * 64 + ct * 32 + li][c0 + kb];
#pragma unroll
            for (int pt = 0; pt < 2; pt++) b[pt] = *(const bf16x8 *)&Xs[pix_base[pt] + toff][c0 + kb];
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int pt = 0; pt < 2; pt++) acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[pt], acc[ct][pt], 0, 0, 0);
        }
        if (tap + 1 < 9) store_w(buf ^ 1, wr[(tap + 1) % PF]);
        __syncthreads();
    }
    // ---- epilogue: D[row = co][col = pixel]; lane (li, h) holds rows 8g + 4h + (0..3), g = 0..3, of column li
    const int h = lane >> 5;
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        const int p = wp * 64 + pt * 32 + li;
        const int oy = ty0 + p / TW, ox = tx0 + p % TW;
        __bf16 *yo = y + (((size_t)n * H + oy) * W + ox) * COUT + co0 + wc * 64;
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = ct * 32 + 8 * g + 4 * h;
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float v = acc[ct][pt][4 * g + e];
                    if (bias) v += bias[co0 + wc * 64 + co + e];
                    o[e] = (__bf16)v;
                }
                *(bf16x4 *)(yo + co) = o;
            }
    }
}

template <int CIN, int WPX, int WCO, int TW>
int launch_conv(const void *x, const void *w, const float *bias, void *y, int N, int H, int W, int COUT, hipStream_t st)
{
    constexpr int PX = WPX * 64, TH = PX / TW, COT = WCO * 64, P = CIN + 8;
    const size_t lds = ((size_t)(TH + 2) * (TW + 2) * P + (size_t)2 * COT * P) * 2;
    auto kern = conv3x3_kernel<CIN, WPX, WCO, TW>;
    static bool attr_set = false;
    if (!attr_set) {
        PSI_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    dim3 grid((unsigned)(N * (H / TH) * (W / TW)), (unsigned)(COUT / COT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const __bf16 *)x, (const __bf16 *)w, bias, (__bf16 *)y, N, H, W, COUT);
    PSI_CHECK_LAUNCH("conv3x3_kernel");
    psi_mark("conv3x3_kernel", st);
    return 0;
}

// weight re-layout for the input gradient: W[co][kh][kw][ci] -> Wt[ci][2-kh][2-kw][co]
__global__ void conv_weight_rot_kernel(const __bf16 *__restrict__ w, __bf16 *__restrict__ wt, int COUT, int CIN)
{
    const int i = blockIdx.x * 256 + threadIdx.x, total = COUT * 9 * CIN;
    if (i >= total) return;
    const int co = i % COUT, tap = (i / COUT) % 9, ci = i / (COUT * 9);
    wt[i] = w[((size_t)co * 9 + (8 - tap)) * CIN + ci];
}

}  // namespace

extern "C" int psi_conv3x3_supported(int Cin, int Cout, int H, int W)
{
    if (Cin == 64) return Cout % 64 == 0 && H % 8 == 0 && W % 32 == 0;
    if (Cin == 128) return Cout % 128 == 0 && H % 8 == 0 && W % 16 == 0;
    return 0;
}

extern "C" int psi_conv3x3_forward(const void *x, const void *w, const float *bias, int N, int H, int W, int Cin, int Cout, void *y, void *stream)
{
    PSI_REQUIRE(x && w && y && N > 0, "null pointer");
    PSI_REQUIRE(psi_conv3x3_supported(Cin, Cout, H, W), "shape not covered: Cin 64 (Cout % 64, H % 8, W % 32) or Cin 128 (Cout % 128, H % 8, W % 16)");
    hipStream_t st = (hipStream_t)stream;
    if (Cin == 64) return launch_conv<64, 4, 1, 32>(x, w, bias, y, N, H, W, Cout, st);
    return launch_conv<128, 2, 2, 16>(x, w, bias, y, N, H, W, Cout, st);
}

extern "C" int psi_conv3x3_rotate_weight(const void *w, int Cin, int Cout, void *wt, void *stream)
{
    PSI_REQUIRE(w && wt && Cin > 0 && Cout > 0, "bad arguments");
    const int total = Cout * 9 * Cin;
    hipLaunchKernelGGL(conv_weight_rot_kernel, dim3(psi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16 *)w, (__bf16 *)wt, Cout, Cin);
    PSI_CHECK_LAUNCH("conv_weight_rot_kernel");
    return 0;
}
