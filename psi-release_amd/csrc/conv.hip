// 3x3 convolutions of the scene trunk (stride 1, padding 1, no dilation) as a hand-written implicit GEMM on the bf16 matrix cores, gfx950.
//
// Replaces, for the BasicBlocks of the ResNet-18 prefix the reference builds (cvae.py:427-435: torchvision resnet18 layer1 = 4 x
// Conv2d(64,64,3,1,1), layer2 = Conv2d(128,128,3,1,1) x 3 behind the strided first one) and the 128 -> 128 head convolution of
// BodyLocalPoseVAE (net_layers.py:160-164), the library convolution in BOTH directions that are convolutions: the forward pass and the
// input gradient (dX = conv(dY, W rotated by 180 degrees with its channel axes swapped) — the same kernel on a re-laid-out weight).
// The weight gradient is the third kernel of this file (conv3x3_wrw_kernel: contraction over pixels, split-K with an ordered reduce).
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
#include <atomic>

namespace {

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: set once per device (bit = device id), safely from any thread
static inline hipError_t psi_set_max_lds(const void *kern, size_t lds, std::atomic<unsigned long long> &done)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

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

// ------------------------------------------------------------------------------------------------
// The same convolution at the fp32 model's precision (round 6): fp32 NHWC map in, fp32 map out, three-term split products.  The input
// tile WITH ITS HALO is split into hi = bf16(v) / lo = bf16(v - hi) planes on its way into LDS — once per workgroup, for all nine taps and
// all output channels (the general kernel of conv_gemm.hip re-loads and re-splits its 64-pixel x 64-channel activation chunk for every tap:
// 1423 vector instructions per wave against 108 MFMAs, profiles/r06_pmc_conv_gemm.txt) — the weights arrive PREPARED
// (psi_conv2d_prepare_weight: hi parts [Cout][9][Cin], the lo parts Cout * 9 * Cin elements behind them) and go to LDS as they are, a tap
// at a time, double-buffered.  A product is lo*hi + hi*lo + hi*hi, fp32 accumulate: the arithmetic conv_gemm.hip documents.
// REV: the taps are walked backwards (w row tap 8 - t): with the weight in the input gradient's layout [Cin][9][Cout] this is dX of the
// same layer — the forward kernel on dY.
// Cin = 64: 256 pixels (8 x 32) x 64 output channels per workgroup; two LDS planes of the halo tile (98 KB) + two weight buffers of two
// planes (37 KB): one workgroup per CU.
// ------------------------------------------------------------------------------------------------
template <int CIN, int WPX, int WCO, int TW, int NBUF, bool REV>
__global__ __launch_bounds__(256) void conv3x3s_kernel(const float *__restrict__ x, const __bf16 *__restrict__ wp, const float *__restrict__ bias,
                                                       float *__restrict__ y, int N, int H, int W, int COUT)
{
    // a weight STAGE = the 64 input channels [64 sub, 64 sub + 64) of one tap for the workgroup's COT output channels, both planes; Cin = 64: one
    // stage per tap in two alternating LDS buffers, Cin = 128: two stages per tap through ONE buffer (two barriers per stage: the LDS budget)
    constexpr int PX = WPX * 64, TH = PX / TW, COT = WCO * 64, P = CIN + 8, PW = 64 + 8, HW_ = TW + 2, HH_ = TH + 2, CH = CIN / 8, NSUB = CIN / 64, NST = 9 * NSUB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 (*Xh)[P] = (__bf16 (*)[P])smem;                                           // [HH_ * HW_][P] hi parts
    __bf16 (*Xl)[P] = Xh + HH_ * HW_;                                                // ... lo parts
    __bf16 (*Ws)[2][COT][PW] = (__bf16 (*)[2][COT][PW])(smem + (size_t)2 * HH_ * HW_ * P * 2);   // [buffer][plane][COT][PW]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wp_ = wv % WPX, wc = wv / WPX;
    const int li = lane & 31, kb = (lane >> 5) * 8;
    const int tiles_w = W / TW, tiles_h = H / TH;
    int t = blockIdx.x;
    const int tx0 = (t % tiles_w) * TW;
    t /= tiles_w;
    const int ty0 = (t % tiles_h) * TH, n = t / tiles_h;
    const int co0 = blockIdx.y * COT;
    const size_t wplane = (size_t)COUT * 9 * CIN;                 // elements between the hi and the lo parts of the prepared weight

    constexpr int WLD = (2 * COT * 8 + 255) / 256, PF = 2;
    u4 wr[PF][WLD];
    auto load_w = [&](int st, u4 (&r)[WLD]) {
        const int tap = st / NSUB, sub = st % NSUB, tsrc = REV ? 8 - tap : tap;
#pragma unroll
        for (int i = 0; i < WLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            const int pl = idx / (COT * 8), rem = idx % (COT * 8), row = rem / 8, c = rem % 8;
            r[i] = *(const u4 *)(wp + (size_t)pl * wplane + ((size_t)(co0 + row) * 9 + tsrc) * CIN + sub * 64 + c * 8);
        }
    };
    auto store_w = [&](int buf, const u4 (&r)[WLD]) {
#pragma unroll
        for (int i = 0; i < WLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            const int pl = idx / (COT * 8), rem = idx % (COT * 8), row = rem / 8, c = rem % 8;
            *(u4 *)&Ws[buf][pl][row][c * 8] = r[i];
        }
    };
#pragma unroll
    for (int st = 0; st < PF; st++) load_w(st, wr[st]);
    // ---- input tile with halo (zero outside the image): all of a thread's 32-byte pieces are requested before the first one is split and stored
    constexpr int NPC = HH_ * HW_ * CH, XLD = (NPC + 255) / 256;
    {
        f4 xr[XLD][2];
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            xr[i][0] = xr[i][1] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
            if (idx < NPC) {
                const int pix = idx / CH, c = idx % CH;
                const int iy = ty0 - 1 + pix / HW_, ix = tx0 - 1 + pix % HW_;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const float *src = x + (((size_t)n * H + iy) * W + ix) * CIN + c * 8;
                    xr[i][0] = *(const f4 *)src;
                    xr[i][1] = *(const f4 *)(src + 4);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i;
            if (idx < NPC) {
                bf16x8 hv, lv;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const float v = xr[i][e >> 2][e & 3];
                    const __bf16 hh = (__bf16)v;
                    hv[e] = hh;
                    lv[e] = (__bf16)(v - (float)hh);
                }
                *(bf16x8 *)&Xh[idx / CH][(idx % CH) * 8] = hv;
                *(bf16x8 *)&Xl[idx / CH][(idx % CH) * 8] = lv;
            }
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
    int pix_base[2];
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        const int p = wp_ * 64 + pt * 32 + li;
        pix_base[pt] = (p / TW) * HW_ + (p % TW);
    }
#pragma unroll
    for (int st = 0; st < NST; st++) {
        const int buf = NBUF == 2 ? (st & 1) : 0, tap = st / NSUB, sub = st % NSUB;
        if (st + PF < NST) load_w(st + PF, wr[st % PF]);
        const int toff = (tap / 3) * HW_ + (tap % 3);
#pragma unroll
        for (int c0 = 0; c0 < 64; c0 += 16) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                ah[ct] = *(const bf16x8 *)&Ws[buf][0][wc * 64 + ct * 32 + li][c0 + kb];
                al[ct] = *(const bf16x8 *)&Ws[buf][1][wc * 64 + ct * 32 + li][c0 + kb];
            }
#pragma unroll
            for (int pt = 0; pt < 2; pt++) {
                bh[pt] = *(const bf16x8 *)&Xh[pix_base[pt] + toff][sub * 64 + c0 + kb];
                bl[pt] = *(const bf16x8 *)&Xl[pix_base[pt] + toff][sub * 64 + c0 + kb];
            }
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int pt = 0; pt < 2; pt++) {
                    f16v a = acc[ct][pt];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ct], bh[pt], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ct], bl[pt], a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ct], bh[pt], a, 0, 0, 0);
                    acc[ct][pt] = a;
                }
        }
        if (NBUF == 2) {
            if (st + 1 < NST) store_w(buf ^ 1, wr[(st + 1) % PF]);
            __syncthreads();
        } else if (st + 1 < NST) {
            __syncthreads();                                 // every wave is done with the buffer
            store_w(0, wr[(st + 1) % PF]);
            __syncthreads();
        }
    }
    // ---- epilogue: D[row = co][col = pixel]; lane (li, h) holds rows 8g + 4h + (0..3), g = 0..3, of column li: 16-byte stores
    const int h = lane >> 5;
#pragma unroll
    for (int pt = 0; pt < 2; pt++) {
        const int p = wp_ * 64 + pt * 32 + li;
        const int oy = ty0 + p / TW, ox = tx0 + p % TW;
        float *yo = y + (((size_t)n * H + oy) * W + ox) * COUT + co0 + wc * 64;
#pragma unroll
        for (int ct = 0; ct < 2; ct++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = ct * 32 + 8 * g + 4 * h;
                f4 o;
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    float v = acc[ct][pt][4 * g + e];
                    if (bias) v += bias[co0 + wc * 64 + co + e];
                    o[e] = v;
                }
                *(f4 *)(yo + co) = o;
            }
    }
}

template <int CIN, int WPX, int WCO, int TW>
int launch_conv(const void *x, const void *w, const float *bias, void *y, int N, int H, int W, int COUT, hipStream_t st)
{
    constexpr int PX = WPX * 64, TH = PX / TW, COT = WCO * 64, P = CIN + 8;
    const size_t lds = ((size_t)(TH + 2) * (TW + 2) * P + (size_t)2 * COT * P) * 2;
    auto kern = conv3x3_kernel<CIN, WPX, WCO, TW>;
    static std::atomic<unsigned long long> attr_set{0};           // one bit per device: the attribute is per device, and forward / autograd threads race here
    PSI_CHECK_HIP(psi_set_max_lds((const void *)kern, lds, attr_set));
    dim3 grid((unsigned)(N * (H / TH) * (W / TW)), (unsigned)(COUT / COT));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const __bf16 *)x, (const __bf16 *)w, bias, (__bf16 *)y, N, H, W, COUT);
    PSI_CHECK_LAUNCH("conv3x3_kernel");
    psi_mark("conv3x3_kernel", st);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Weight gradient  dW[co][kh][kw][ci] = sum_{n,oy,ox} dY[n,oy,ox,co] * X[n, oy+kh-1, ox+kw-1, ci]   (fp32 output).
// The contraction runs over PIXELS, along which neither operand is contiguous (channels are fastest), so both tiles are transposed on
// their way into LDS: dYt[co][pixel], Xt[ci][row][8 zeros | TW pixels | 8 zeros].  A thread loads the 16-byte channel pieces of TWO
// adjacent pixels and stores one 32-bit {pixel, pixel + 1} pair per channel (the first version stored single bf16 values into three
// column-shifted copies of the input tile: 176 conflicting 2-byte LDS stores per thread and stage, 38 us per layer).  The MFMA operand of
// tap column kw is the run of eight pixels starting at ox + kw - 1: for kw = 1 an aligned 16-byte LDS read, for kw = 0 / 2 the same run
// shifted by one pixel, assembled in registers from the aligned chunk and its left / right neighbour with five v_alignbit — the zero
// columns left and right of a row are the convolution's padding.  The row shift kh moves by whole rows and keeps the alignment.
// A stage is TR output rows of one image (128 pixels); a workgroup owns a 64 (co) x 64 (ci) tile of dW for all nine taps — wave (ct, it)
// one 32 x 32 quadrant, nine accumulators — and walks the stages s = split, split + S, ...; the S partial results are summed in split
// order by conv_wrw_reduce_kernel (deterministic; the library's split-K kernels use atomics).
// ------------------------------------------------------------------------------------------------
template <int TW>                                            // image width == tile width (16 or 32); TR = 128 / TW output rows per stage
__global__ __launch_bounds__(256) void conv3x3_wrw_kernel(const __bf16 *__restrict__ x, const __bf16 *__restrict__ dy, float *__restrict__ part,
                                                          int N, int H, int CIN, int COUT, int nsplit)
{
    constexpr int TR = 128 / TW, PXP = 128 + 8, RP = TW + 16, XP = (TR + 2) * RP + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 (*dYt)[PXP] = (__bf16 (*)[PXP])smem;                                    // [64 co][128 px]
    __bf16 (*Xt)[XP] = (__bf16 (*)[XP])(smem + (size_t)64 * PXP * 2);               // [64 ci][(TR + 2) rows of RP]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ct = wv & 1, it = wv >> 1;
    const int li = lane & 31, kb = (lane >> 5) * 8, h = lane >> 5;
    const int split = blockIdx.x, co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int stages_per_img = H / TR, nstage = N * stages_per_img;
    for (int i = threadIdx.x; i < 64 * XP / 2; i += 256) ((unsigned *)&Xt[0][0])[i] = 0u;      // the padding columns are never written again
    f16v acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[tp][i] = 0.0f;
    // item = (pixel pair q, channel piece c); consecutive lanes take consecutive PAIRS of one piece: their 32-bit LDS stores fall on
    // consecutive banks, and the eight pieces of a pixel are still requested by one load instruction of the wave
    constexpr int DP = 64, DLD = DP * 8 / 256;                    // dY: 64 pairs x 8 pieces = 512 items, 2 per thread
    constexpr int XPR = (TR + 2) * TW / 2, XIT = XPR * 8, XLD = (XIT + 255) / 256;
    for (int stg = split; stg < nstage; stg += nsplit) {
        const int n = stg / stages_per_img, oy0 = (stg % stages_per_img) * TR;
        __syncthreads();                                     // previous stage's MFMAs are done with LDS (and the zero fill is visible)
        u4 dv[DLD][2], xv[XLD][2];
#pragma unroll
        for (int i = 0; i < DLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % DP, c = idx / DP, px = 2 * q;
            const __bf16 *src = dy + (((size_t)n * H + oy0 + px / TW) * TW + px % TW) * COUT + co0 + c * 8;
            dv[i][0] = *(const u4 *)src;
            dv[i][1] = *(const u4 *)(src + COUT);
        }
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % XPR, c = idx / XPR, px = 2 * q;
            const int iy = oy0 - 1 + px / TW;
            xv[i][0] = xv[i][1] = (u4){0u, 0u, 0u, 0u};
            if (idx < XIT && iy >= 0 && iy < H) {
                const __bf16 *src = x + (((size_t)n * H + iy) * TW + px % TW) * CIN + ci0 + c * 8;
                xv[i][0] = *(const u4 *)src;
                xv[i][1] = *(const u4 *)(src + CIN);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        auto scatter = [&](const u4 &p0, const u4 &p1, __bf16 *row0, int pitch) {     // 8 channels x {pixel, pixel + 1} -> 8 rows, one dword each
#pragma unroll
            for (int e = 0; e < 4; e++) {
                *(unsigned *)(row0 + (size_t)(2 * e) * pitch) = (p0[e] & 0xffffu) | (p1[e] << 16);
                *(unsigned *)(row0 + (size_t)(2 * e + 1) * pitch) = (p0[e] >> 16) | (p1[e] & 0xffff0000u);
            }
        };
#pragma unroll
        for (int i = 0; i < DLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % DP, c = idx / DP;
            scatter(dv[i][0], dv[i][1], &dYt[c * 8][2 * q], PXP);
        }
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % XPR, c = idx / XPR, px = 2 * q;
            if (idx < XIT) scatter(xv[i][0], xv[i][1], &Xt[c * 8][(px / TW) * RP + 8 + px % TW], XP);
        }
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < 128; k0 += 16) {
            const bf16x8 a = *(const bf16x8 *)&dYt[ct * 32 + li][k0 + kb];
            const int r = k0 / TW, c0 = k0 % TW;
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const __bf16 *mid = &Xt[it * 32 + li][(r + kh) * RP + 8 + c0 + kb];
                const u4 L = *(const u4 *)(mid - 8), M = *(const u4 *)mid, R = *(const u4 *)(mid + 8);
                u4 s0, s2;                                    // the run shifted one pixel to the left (kw = 0) / right (kw = 2)
                s0[0] = __builtin_amdgcn_alignbit(M[0], L[3], 16);
                s0[1] = __builtin_amdgcn_alignbit(M[1], M[0], 16);
                s0[2] = __builtin_amdgcn_alignbit(M[2], M[1], 16);
                s0[3] = __builtin_amdgcn_alignbit(M[3], M[2], 16);
                s2[0] = s0[1];
                s2[1] = s0[2];
                s2[2] = s0[3];
                s2[3] = __builtin_amdgcn_alignbit(R[0], M[3], 16);
                acc[kh * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, s0), acc[kh * 3 + 0], 0, 0, 0);
                acc[kh * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, M), acc[kh * 3 + 1], 0, 0, 0);
                acc[kh * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, s2), acc[kh * 3 + 2], 0, 0, 0);
            }
        }
    }
    // ---- partial tile: D[row = co][col = ci] per tap -> part[split][co][tap][ci]
    float *po = part + (size_t)split * COUT * 9 * CIN;
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            po[((size_t)co * 9 + tp) * CIN + ci0 + it * 32 + li] = acc[tp][r];
        }
}

// ------------------------------------------------------------------------------------------------
// The same weight gradient at the fp32 model's precision (round 6): fp32 maps in, every operand split into hi = bf16(v), lo = bf16(v - hi)
// on its way into LDS — ONCE per workgroup and stage, for all nine taps — and a product is lo*hi + hi*lo + hi*hi with fp32 accumulation
// (conv_gemm.hip documents the arithmetic; the general kernel's weight gradient gives a workgroup one tap, so every element of dY and of
// X is loaded, split and transposed nine times: 13 161 vector instructions per wave against 439 MFMAs, profiles/r06_pmc_conv_wgrad.txt).
// Same tile as the bf16 kernel above (64 co x 64 ci, all nine taps, 128-pixel stages, the shifted runs assembled with v_alignbit), two
// LDS planes per operand (110-119 KB: one workgroup per CU, one wave per SIMD), so the next stage's 20 16-byte loads per thread are
// requested before the current stage's 216 MFMAs per wave and wait in registers.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned psi_pack_bf16(__bf16 a, __bf16 b)
{
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

template <int TW>
__global__ __launch_bounds__(256) void conv3x3_wrw3_kernel(const float *__restrict__ x, const float *__restrict__ dy, float *__restrict__ part,
                                                           int N, int H, int CIN, int COUT, int nsplit)
{
    constexpr int TR = 128 / TW, PXP = 128 + 8, RP = TW + 16, XP = (TR + 2) * RP + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 (*dYh)[PXP] = (__bf16 (*)[PXP])smem;                                       // [64 co][128 px], hi parts
    __bf16 (*dYl)[PXP] = dYh + 64;                                                    // ... lo parts
    __bf16 (*Xh)[XP] = (__bf16 (*)[XP])(smem + (size_t)2 * 64 * PXP * 2);              // [64 ci][(TR + 2) rows of RP]
    __bf16 (*Xl)[XP] = Xh + 64;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ct = wv & 1, it = wv >> 1;
    const int li = lane & 31, kb = (lane >> 5) * 8, h = lane >> 5;
    const int split = blockIdx.x, co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int stages_per_img = H / TR, nstage = N * stages_per_img;
    for (int i = threadIdx.x; i < 2 * 64 * XP / 2; i += 256) ((unsigned *)&Xh[0][0])[i] = 0u;     // the padding columns (both planes) are never written again
    f16v acc[9];
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[tp][i] = 0.0f;
    // item = (pixel pair q, channel piece c of 8 channels): two pixels x two 16-byte halves
    constexpr int DP = 64, DLD = DP * 8 / 256;
    constexpr int XPR = (TR + 2) * TW / 2, XIT = XPR * 8, XLD = (XIT + 255) / 256;
    f4 dv[DLD][2][2], xv[XLD][2][2];
    auto load_stage = [&](int stg) {
        const int n = stg / stages_per_img, oy0 = (stg % stages_per_img) * TR;
#pragma unroll
        for (int i = 0; i < DLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % DP, c = idx / DP, px = 2 * q;
            const float *src = dy + (((size_t)n * H + oy0 + px / TW) * TW + px % TW) * COUT + co0 + c * 8;
#pragma unroll
            for (int u = 0; u < 2; u++) {
                dv[i][u][0] = *(const f4 *)(src + (size_t)u * COUT);
                dv[i][u][1] = *(const f4 *)(src + (size_t)u * COUT + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % XPR, c = idx / XPR, px = 2 * q;
            const int iy = oy0 - 1 + px / TW;
#pragma unroll
            for (int u = 0; u < 2; u++) xv[i][u][0] = xv[i][u][1] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
            if (idx < XIT && iy >= 0 && iy < H) {
                const float *src = x + (((size_t)n * H + iy) * TW + px % TW) * CIN + ci0 + c * 8;
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    xv[i][u][0] = *(const f4 *)(src + (size_t)u * CIN);
                    xv[i][u][1] = *(const f4 *)(src + (size_t)u * CIN + 4);
                }
            }
        }
    };
    // 8 channels x {pixel, pixel + 1}, fp32 -> one {hi, hi} dword and one {lo, lo} dword per channel row
    auto scatter = [&](const f4 (&p)[2][2], __bf16 *rowh, __bf16 *rowl, int pitch) {
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v0 = p[0][e >> 2][e & 3], v1 = p[1][e >> 2][e & 3];
            const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
            const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
            *(unsigned *)(rowh + (size_t)e * pitch) = psi_pack_bf16(h0, h1);
            *(unsigned *)(rowl + (size_t)e * pitch) = psi_pack_bf16(l0, l1);
        }
    };
    if (split < nstage) load_stage(split);
    for (int stg = split; stg < nstage; stg += nsplit) {
        __syncthreads();                                     // previous stage's MFMAs are done with LDS (and the zero fill is visible)
#pragma unroll
        for (int i = 0; i < DLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % DP, c = idx / DP;
            scatter(dv[i], &dYh[c * 8][2 * q], &dYl[c * 8][2 * q], PXP);
        }
#pragma unroll
        for (int i = 0; i < XLD; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx % XPR, c = idx / XPR, px = 2 * q;
            if (idx < XIT) scatter(xv[i], &Xh[c * 8][(px / TW) * RP + 8 + px % TW], &Xl[c * 8][(px / TW) * RP + 8 + px % TW], XP);
        }
        __syncthreads();
        if (stg + nsplit < nstage) load_stage(stg + nsplit);     // in flight during this stage's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k0 = 0; k0 < 128; k0 += 16) {
            const bf16x8 ah = *(const bf16x8 *)&dYh[ct * 32 + li][k0 + kb];
            const bf16x8 al = *(const bf16x8 *)&dYl[ct * 32 + li][k0 + kb];
            const int r = k0 / TW, c0 = k0 % TW;
#pragma unroll
            for (int kh = 0; kh < 3; kh++) {
                const int off = (r + kh) * RP + 8 + c0 + kb;
                u4 B[2][3];                                   // [plane][kw]: the run of eight pixels starting at ox + kw - 1
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const __bf16 *mid = (pl ? &Xl[it * 32 + li][0] : &Xh[it * 32 + li][0]) + off;
                    const u4 L = *(const u4 *)(mid - 8), M = *(const u4 *)mid, R = *(const u4 *)(mid + 8);
                    B[pl][0][0] = __builtin_amdgcn_alignbit(M[0], L[3], 16);
                    B[pl][0][1] = __builtin_amdgcn_alignbit(M[1], M[0], 16);
                    B[pl][0][2] = __builtin_amdgcn_alignbit(M[2], M[1], 16);
                    B[pl][0][3] = __builtin_amdgcn_alignbit(M[3], M[2], 16);
                    B[pl][1] = M;
                    B[pl][2][0] = B[pl][0][1];
                    B[pl][2][1] = B[pl][0][2];
                    B[pl][2][2] = B[pl][0][3];
                    B[pl][2][3] = __builtin_amdgcn_alignbit(R[0], M[3], 16);
                }
#pragma unroll
                for (int kw = 0; kw < 3; kw++) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, B[0][kw]), bl = __builtin_bit_cast(bf16x8, B[1][kw]);
                    f16v a = acc[kh * 3 + kw];
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, a, 0, 0, 0);
                    a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, a, 0, 0, 0);
                    acc[kh * 3 + kw] = a;
                }
            }
        }
    }
    // ---- partial tile: D[row = co][col = ci] per tap -> part[split][co][tap][ci]
    float *po = part + (size_t)split * COUT * 9 * CIN;
#pragma unroll
    for (int tp = 0; tp < 9; tp++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            po[((size_t)co * 9 + tp) * CIN + ci0 + it * 32 + li] = acc[tp][r];
        }
}

// FOUR threads per output (lanes 4 i .. 4 i + 3 take the splits s = 0, 1, 2, 3 (mod 4), eight loads in flight each; combined in lane order
// with two shuffles): 36 864 outputs of a 64 -> 64 layer were 144 workgroups reading 38 MB of partials — 12.7 us, a third of the weight
// gradient itself; with four times the threads in flight the same sums take half of that.  Fixed order: deterministic.
__global__ __launch_bounds__(256) void conv_wrw_reduce_kernel(const float *__restrict__ part, int nsplit, long n, float *__restrict__ gw)
{
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = t >> 2;
    const int s0 = (int)(t & 3);
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    if (i < n) {
        int s = s0;
        for (; s + 28 < nsplit; s += 32) {                       // eight loads in flight; fixed order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = part[(size_t)(s + 4 * u) * n + i];
            a0 += v[0] + v[4];
            a1 += v[1] + v[5];
            a2 += v[2] + v[6];
            a3 += v[3] + v[7];
        }
        for (; s < nsplit; s += 4) a0 += part[(size_t)s * n + i];
    }
    float r = (a0 + a1) + (a2 + a3);
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    if (i < n && s0 == 0) gw[i] = r;
}

// weight re-layout for the input gradient: W[co][kh][kw][ci] -> Wt[ci][2-kh][2-kw][co]
__global__ void conv_weight_rot_kernel(const __bf16 *__restrict__ w, __bf16 *__restrict__ wt, int COUT, int CIN)
{
    const int i = blockIdx.x * 256 + threadIdx.x, total = COUT * 9 * CIN;
    if (i >= total) return;
    const int co = i % COUT, tap = (i / COUT) % 9, ci = i / (COUT * 9);
    wt[i] = w[((size_t)co * 9 + (8 - tap)) * CIN + ci];
}

// fp32 master weight -> bf16 in BOTH layouts at once: W[co][kh][kw][ci] for the forward, Wt[ci][2-kh][2-kw][co] for the input gradient
// (one launch per layer and step instead of a cast and a re-layout)
__global__ void conv_weight_prepare_kernel(const float *__restrict__ w32, __bf16 *__restrict__ wb, __bf16 *__restrict__ wt, int COUT, int CIN)
{
    const int i = blockIdx.x * 256 + threadIdx.x, total = COUT * 9 * CIN;
    if (i >= total) return;
    const int ci = i % CIN, tap = (i / CIN) % 9, co = i / (CIN * 9);
    const __bf16 v = (__bf16)w32[i];
    wb[i] = v;
    if (wt) wt[((size_t)ci * 9 + (8 - tap)) * COUT + co] = v;
}

}  // namespace

extern "C" int psi_conv3x3_prepare_weight(const float *w32, int Cin, int Cout, void *wb, void *wt, void *stream)
{
    PSI_REQUIRE(w32 && wb && Cin > 0 && Cout > 0, "bad arguments");
    const int total = Cout * 9 * Cin;
    hipLaunchKernelGGL(conv_weight_prepare_kernel, dim3(psi_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, w32, (__bf16 *)wb, (__bf16 *)wt, Cout,
                       Cin);
    PSI_CHECK_LAUNCH("conv_weight_prepare_kernel");
    return 0;
}

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

// splits of the pixel range for the weight gradient: about one workgroup per compute unit
static int wrw_splits(int N, int H, int W, int Cin, int Cout)
{
    const int TR = 128 / W, nstage = N * (H / TR), tiles = (Cin / 64) * (Cout / 64);
    int S = 256 / tiles;
    if (S > nstage) S = nstage;
    return S < 1 ? 1 : S;
}

extern "C" size_t psi_conv3x3_wrw_workspace_floats(int N, int H, int W, int Cin, int Cout)
{
    if (!(W == 16 || W == 32) || H % (128 / W) || Cin % 64 || Cout % 64) return 0;
    return (size_t)wrw_splits(N, H, W, Cin, Cout) * Cout * 9 * Cin;
}

extern "C" int psi_conv3x3_weight_grad(const void *x, const void *dy, int N, int H, int W, int Cin, int Cout, float *gw, float *ws, void *stream)
{
    PSI_REQUIRE(x && dy && gw && ws && N > 0, "null pointer");
    PSI_REQUIRE((W == 16 || W == 32) && H % (128 / W) == 0 && Cin % 64 == 0 && Cout % 64 == 0, "shape not covered: W 16 or 32, H % (128 / W) == 0, channels % 64 == 0");
    hipStream_t st = (hipStream_t)stream;
    const int S = wrw_splits(N, H, W, Cin, Cout);
    const int TR = 128 / W;
    const size_t lds = ((size_t)64 * (128 + 8) + (size_t)64 * ((TR + 2) * (W + 16) + 8)) * 2;
    dim3 grid(S, Cout / 64, Cin / 64);
    if (W == 32) {
        static std::atomic<unsigned long long> a32{0};
        PSI_CHECK_HIP(psi_set_max_lds((const void *)conv3x3_wrw_kernel<32>, lds, a32));
        hipLaunchKernelGGL(conv3x3_wrw_kernel<32>, grid, dim3(256), lds, st, (const __bf16 *)x, (const __bf16 *)dy, ws, N, H, Cin, Cout, S);
    } else {
        static std::atomic<unsigned long long> a16{0};
        PSI_CHECK_HIP(psi_set_max_lds((const void *)conv3x3_wrw_kernel<16>, lds, a16));
        hipLaunchKernelGGL(conv3x3_wrw_kernel<16>, grid, dim3(256), lds, st, (const __bf16 *)x, (const __bf16 *)dy, ws, N, H, Cin, Cout, S);
    }
    PSI_CHECK_LAUNCH("conv3x3_wrw_kernel");
    psi_mark("conv3x3_wrw_kernel", st);
    const long n = (long)Cout * 9 * Cin;
    hipLaunchKernelGGL(conv_wrw_reduce_kernel, dim3(psi_cdiv(4 * n, 256)), dim3(256), 0, st, ws, S, n, gw);
    PSI_CHECK_LAUNCH("conv_wrw_reduce_kernel");
    return 0;
}

// ---- the fp32 (three-term) weight gradient of the stride-1 3x3 layers: conv_gemm.hip's psi_conv2d_weight_grad routes the shapes this covers here
int psi_conv3x3_wrw3_ok(int N, int H, int W, int Cin, int Cout)
{
    return N > 0 && (W == 16 || W == 32) && H % (128 / W) == 0 && Cin % 64 == 0 && Cout % 64 == 0;
}

size_t psi_conv3x3_wrw3_workspace_floats(int N, int H, int W, int Cin, int Cout)
{
    if (!psi_conv3x3_wrw3_ok(N, H, W, Cin, Cout)) return 0;
    return (size_t)wrw_splits(N, H, W, Cin, Cout) * Cout * 9 * Cin;
}

int psi_conv3x3_weight_grad3(const float *x, const float *dy, int N, int H, int W, int Cin, int Cout, float *gw, float *ws, hipStream_t st)
{
    PSI_REQUIRE(x && dy && gw && ws && psi_conv3x3_wrw3_ok(N, H, W, Cin, Cout), "shape not covered by the three-term 3x3 weight gradient");
    const int S = wrw_splits(N, H, W, Cin, Cout);
    const int TR = 128 / W;
    const size_t lds = ((size_t)2 * 64 * (128 + 8) + (size_t)2 * 64 * ((TR + 2) * (W + 16) + 8)) * 2;
    dim3 grid(S, Cout / 64, Cin / 64);
    if (W == 32) {
        static std::atomic<unsigned long long> a32{0};
        PSI_CHECK_HIP(psi_set_max_lds((const void *)conv3x3_wrw3_kernel<32>, lds, a32));
        hipLaunchKernelGGL(conv3x3_wrw3_kernel<32>, grid, dim3(256), lds, st, x, dy, ws, N, H, Cin, Cout, S);
    } else {
        static std::atomic<unsigned long long> a16{0};
        PSI_CHECK_HIP(psi_set_max_lds((const void *)conv3x3_wrw3_kernel<16>, lds, a16));
        hipLaunchKernelGGL(conv3x3_wrw3_kernel<16>, grid, dim3(256), lds, st, x, dy, ws, N, H, Cin, Cout, S);
    }
    PSI_CHECK_LAUNCH("conv3x3_wrw3_kernel");
    psi_mark("conv3x3_wrw3_kernel", st);
    const long n = (long)Cout * 9 * Cin;
    hipLaunchKernelGGL(conv_wrw_reduce_kernel, dim3(psi_cdiv(4 * n, 256)), dim3(256), 0, st, ws, S, n, gw);
    PSI_CHECK_LAUNCH("conv_wrw_reduce_kernel");
    return 0;
}

// ---- forward / input gradient of the stride-1 3x3 layers at the fp32 model's precision, prepared weights (conv_gemm.hip routes here)
int psi_conv3x3s_ok(int N, int H, int W, int Cin, int Cout)
{
    if (N <= 0 || H % 8) return 0;
    if (Cin == 64) return Cout % 64 == 0 && W % 32 == 0;
    if (Cin == 128) return Cout % 128 == 0 && W % 16 == 0;
    return 0;
}

template <int CIN, int WPX, int WCO, int TW, int NBUF>
static int launch_conv3x3s(const float *x, const void *wp, const float *bias, int N, int H, int W, int Cout, float *y, int reversed_taps, hipStream_t st)
{
    constexpr int PX = WPX * 64, TH = PX / TW, COT = WCO * 64, P = CIN + 8, PW = 64 + 8;
    const size_t lds = ((size_t)2 * (TH + 2) * (TW + 2) * P + (size_t)NBUF * 2 * COT * PW) * 2;
    dim3 grid((unsigned)(N * (H / TH) * (W / TW)), (unsigned)(Cout / COT));
    if (reversed_taps) {
        static std::atomic<unsigned long long> a1{0};
        PSI_CHECK_HIP(psi_set_max_lds((const void *)conv3x3s_kernel<CIN, WPX, WCO, TW, NBUF, true>, lds, a1));
        hipLaunchKernelGGL((conv3x3s_kernel<CIN, WPX, WCO, TW, NBUF, true>), grid, dim3(256), lds, st, x, (const __bf16 *)wp, bias, y, N, H, W, Cout);
    } else {
        static std::atomic<unsigned long long> a0{0};
        PSI_CHECK_HIP(psi_set_max_lds((const void *)conv3x3s_kernel<CIN, WPX, WCO, TW, NBUF, false>, lds, a0));
        hipLaunchKernelGGL((conv3x3s_kernel<CIN, WPX, WCO, TW, NBUF, false>), grid, dim3(256), lds, st, x, (const __bf16 *)wp, bias, y, N, H, W, Cout);
    }
    PSI_CHECK_LAUNCH("conv3x3s_kernel");
    psi_mark("conv3x3s_kernel", st);
    return 0;
}

int psi_conv3x3s_forward(const float *x, const void *wp, const float *bias, int N, int H, int W, int Cin, int Cout, float *y, int reversed_taps, hipStream_t st)
{
    PSI_REQUIRE(x && wp && y && psi_conv3x3s_ok(N, H, W, Cin, Cout), "shape not covered by the three-term 3x3 kernel");
    if (Cin == 64) return launch_conv3x3s<64, 4, 1, 32, 2>(x, wp, bias, N, H, W, Cout, y, reversed_taps, st);
    return launch_conv3x3s<128, 2, 2, 16, 1>(x, wp, bias, N, H, W, Cout, y, reversed_taps, st);
}
