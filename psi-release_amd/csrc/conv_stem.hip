// The stem convolution of the scene encoders — Conv2d(2, 64, kernel 7, stride 2, padding 3) on 128 x 128 maps (cvae.py:427-429: the first
// layer of torchvision's resnet18 re-made for the 2-channel depth / semantics input) — forward and weight gradient, as kernels of their own.
//
// Why not the general implicit GEMM of conv_gemm.hip: with two input channels a filter tap is 8 bytes of an NHWC map, so the general kernel
// gathers its K range element by element (98 values per output pixel, each its own bounds-checked load), and the layer is anything but
// GEMM-bound: 6.6 GFLOP against 67 MB (bf16) / 134 MB (fp32) of output.  Here the INPUT PATCH of an 8 x 16 tile of output pixels
// (21 rows x 37 pixels x 2 channels = 1554 values, read once, coalesced) is staged in LDS as bf16 (hi and lo parts for the fp32 model) and
// the matrix-core operands are read straight out of the patch: one filter ROW of a pixel — 7 taps x 2 channels = 14 consecutive values —
// is 16 operand slots (two of them zero), i.e. one k-step of v_mfma_f32_32x32x16_bf16, and K = 7 filter rows = 7 k-steps.
//   forward   D[co][pixel]: the wave's 32 pixels x 64 channels leave through an LDS transpose, so that the stores are whole 128 / 256-byte
//             pixel rows (16 bytes per lane, consecutive lanes consecutive addresses) instead of 8-byte pieces of 32 different rows;
//   gradient  dW[co][slot] = sum over pixels of dY[pixel][co] * patch[pixel, slot]: the dY tile is transposed into LDS (pixel pairs packed),
//             the patch operand is gathered from the LDS patch; a workgroup walks tiles (persistent, 2 per compute unit) and keeps its
//             64 x 112 partial sum in registers; the workgroups' partials are summed in a fixed order by a second small kernel.
// NTERM = 1 (bf16 products) | 3 (hi*hi + hi*lo + lo*hi: the fp32 model's precision), as in conv_gemm.hip.
#include "psi_internal.h"
#include <atomic>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int CIN = 2, KH = 7, KW = 7, ST = 2, PAD = 3, CO = 64;
constexpr int TY = 8, TX = 16;                                    // output pixels of a tile: 8 rows x 16 columns = 4 waves x 32
constexpr int PR = (TY - 1) * ST + KH;                            // 21 patch rows
constexpr int PE = ((TX - 1) * ST + KW) * CIN;                    // 74 patch elements per row
constexpr int PP = 80;                                            // patch row pitch (elements; operand reads reach element 75)
constexpr int NLD = (PR * PE + 255) / 256;                        // 7 patch elements per thread
constexpr int ROWLEN = KW * CIN;                                  // 14 of the 16 operand slots of a filter row
constexpr int WP = KH * 16 + 8;                                   // filter pitch in LDS: 112 slots + 8 (bank spread)
constexpr int DP = TY * TX + 8;                                   // transposed dY pitch (pixels)
constexpr int KTOT = KH * KW * CIN;                               // 98

__device__ __forceinline__ float ldf(const float *p) { return *p; }
__device__ __forceinline__ float ldf(const __bf16 *p) { return (float)*p; }

template <typename TIN> struct Patch {
    float v[NLD];
    // rows oy0*ST-PAD .. of image n, elements from pixel ox0*ST-PAD; outside the map: zeros (the convolution's padding)
    __device__ __forceinline__ void load(const TIN *x, int n, int oy0, int ox0, int H, int W, int t)
    {
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = t + 256 * i;
            const int r = idx / PE, e = idx - r * PE;
            const int iy = oy0 * ST - PAD + r, ix = ox0 * ST - PAD + e / CIN;
            const bool ok = idx < PR * PE && iy >= 0 && iy < H && ix >= 0 && ix < W;
            v[i] = ok ? ldf(x + (((size_t)n * H + iy) * W + ix) * CIN + (e % CIN)) : 0.0f;
        }
    }
    template <int NTERM> __device__ __forceinline__ void store(__bf16 (*Ph)[PP], __bf16 (*Pl)[PP], int t) const
    {
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = t + 256 * i;
            if (idx < PR * PE) {
                const int r = idx / PE, e = idx - r * PE;
                const __bf16 hi = (__bf16)v[i];
                Ph[r][e] = hi;
                if (NTERM > 1) Pl[r][e] = (__bf16)(v[i] - (float)hi);
            }
        }
    }
};

// eight operand slots of one filter row of a pixel, straight out of the LDS patch (8-byte aligned: two 8-byte reads)
__device__ __forceinline__ bf16x8 patch_row8(const __bf16 *p, int h)
{
    const bf16x4 a = *(const bf16x4 *)p, b = *(const bf16x4 *)(p + 4);
    bf16x8 r = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    if (h) {                                                      // slots 14, 15 are not taps of this pixel (they multiply zero weights; a
        r[6] = (__bf16)0.0f;                                      // non-finite neighbour must not reach the product either)
        r[7] = (__bf16)0.0f;
    }
    return r;
}

template <int NTERM, typename TIN, typename TOUT>
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(const TIN *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                          TOUT *__restrict__ y, int N, int H, int W, int OH, int OW, int tiles_y, int tiles_x)
{
    constexpr int SPITCH = sizeof(TOUT) == 2 ? CO + 8 : CO + 4;   // staging row of a pixel: 144 / 272 bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 (*Wh)[WP] = (__bf16 (*)[WP])smem;
    __bf16 (*Wl)[WP] = (__bf16 (*)[WP])(smem + (size_t)CO * WP * 2);
    unsigned char *sp = smem + (size_t)CO * WP * 2 * (NTERM > 1 ? 2 : 1);
    __bf16 (*Ph)[PP] = (__bf16 (*)[PP])sp;
    __bf16 (*Pl)[PP] = (__bf16 (*)[PP])(sp + (size_t)PR * PP * 2);
    sp += (size_t)PR * PP * 2 * (NTERM > 1 ? 2 : 1);
    TOUT (*S)[SPITCH] = (TOUT (*)[SPITCH])sp;                     // [4 waves * 32 pixels][SPITCH]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / tiles_y, oy0 = (blockIdx.x - n * tiles_y) * TY;
    Patch<TIN> patch;
    patch.load(x, n, oy0, 0, H, W, t);
    // ---- the filters, once per workgroup: [co][kh][16 slots], slots 14 and 15 zero
    for (int it = t; it < CO * KH * 2; it += 256) {
        const int co = it / (KH * 2), r = it - co * (KH * 2), kh = r >> 1, j0 = (r & 1) * 8;
        bf16x8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const float v = j0 + e < ROWLEN ? w[(size_t)co * KTOT + kh * ROWLEN + j0 + e] : 0.0f;
            hi[e] = (__bf16)v;
            if (NTERM > 1) lo[e] = (__bf16)(v - (float)hi[e]);
        }
        *(bf16x8 *)&Wh[co][kh * 16 + j0] = hi;
        if (NTERM > 1) *(bf16x8 *)&Wl[co][kh * 16 + j0] = lo;
    }
    for (int it = t; it < PR * (PP - PE); it += 256) {            // the patch rows' tail: read by the last pixel's upper slots, never written again
        const int r = it / (PP - PE), e = PE + it % (PP - PE);
        Ph[r][e] = (__bf16)0.0f;
        if (NTERM > 1) Pl[r][e] = (__bf16)0.0f;
    }
    float bv[2][4][4];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int g = 0; g < 4; g++)
#pragma unroll
            for (int e = 0; e < 4; e++) bv[c][g][e] = bias ? bias[c * 32 + 8 * g + 4 * h + e] : 0.0f;
    const int ly = 2 * wv + (li >> 4), lx = li & 15;              // my pixel of the tile (MFMA column li of wave wv)
    for (int sx = 0; sx < tiles_x; sx++) {
        __syncthreads();                                          // the previous tile is done with the patch and the staging rows
        patch.template store<NTERM>(Ph, Pl, t);
        __syncthreads();
        if (sx + 1 < tiles_x) patch.load(x, n, oy0, (sx + 1) * TX, H, W, t);      // in flight during this tile's products
        f16v acc[2];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[c][i] = 0.0f;
#pragma unroll
        for (int kh = 0; kh < KH; kh++) {
            const bf16x8 bh = patch_row8(&Ph[ly * ST + kh][lx * ST * CIN + h * 8], h);
            bf16x8 bl;
            if (NTERM > 1) bl = patch_row8(&Pl[ly * ST + kh][lx * ST * CIN + h * 8], h);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const bf16x8 ah = *(const bf16x8 *)&Wh[c * 32 + li][kh * 16 + h * 8];
                if (NTERM > 1) {
                    const bf16x8 al = *(const bf16x8 *)&Wl[c * 32 + li][kh * 16 + h * 8];
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[c], 0, 0, 0);
                }
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[c], 0, 0, 0);
            }
        }
        // ---- D[row = co][col = pixel li]: lane (li, h) holds channels c*32 + 8g + 4h + (0..3) -> the pixel's staging row
        TOUT *srow = &S[wv * 32 + li][0];
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int co = c * 32 + 8 * g + 4 * h;
                if constexpr (sizeof(TOUT) == 2) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] = (__bf16)(acc[c][4 * g + e] + bv[c][g][e]);
                    *(bf16x4 *)(srow + co) = o;
                } else {
                    *(f4 *)(srow + co) = (f4){acc[c][4 * g] + bv[c][g][0], acc[c][4 * g + 1] + bv[c][g][1], acc[c][4 * g + 2] + bv[c][g][2],
                                              acc[c][4 * g + 3] + bv[c][g][3]};
                }
            }
        __syncthreads();
        // ---- whole pixel rows out: 16 bytes per lane, 8 (bf16) / 16 (fp32) lanes per pixel
        constexpr int CPP = CO * sizeof(TOUT) / 16, EPC = 16 / sizeof(TOUT);      // chunks per pixel, elements per chunk
        const int ox0 = sx * TX;
#pragma unroll
        for (int it = 0; it < 32 * CPP / 64; it++) {
            const int q = it * 64 + lane, px = q / CPP, part = q % CPP;
            const int oy = oy0 + 2 * wv + (px >> 4), ox = ox0 + (px & 15);
            if (oy < OH && ox < OW)
                *(u4 *)(y + (((size_t)n * OH + oy) * OW + ox) * CO + part * EPC) = *(const u4 *)&S[wv * 32 + px][part * EPC];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8(const float *p, bool ok, float (&v)[8])
{
    const f4 a = ok ? *(const f4 *)p : (f4){0, 0, 0, 0}, b = ok ? *(const f4 *)(p + 4) : (f4){0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; e++) { v[e] = a[e]; v[4 + e] = b[e]; }
}
__device__ __forceinline__ void load8(const __bf16 *p, bool ok, float (&v)[8])
{
    bf16x8 a;
    if (ok) a = *(const bf16x8 *)p;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = ok ? (float)a[e] : 0.0f;
}

template <int NTERM, typename TIN, typename TDY>
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const TIN *__restrict__ x, const TDY *__restrict__ dy, float *__restrict__ part, int N,
                                                            int H, int W, int OH, int OW, int tiles_y, int tiles_x)
{
    __shared__ __attribute__((aligned(16))) __bf16 Dh[CO][DP], Ph[PR + 1][PP];
    __shared__ __attribute__((aligned(16))) __bf16 Dl[NTERM > 1 ? CO : 1][DP], Pl[NTERM > 1 ? PR + 1 : 1][PP];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 31, h = lane >> 5;
    const long ntiles = (long)N * tiles_y * tiles_x;
    for (int it = t; it < (PR + 1) * PP; it += 256) {             // the pitch tail and the spare row (filter row "7" of the last slot tile reads it)
        (&Ph[0][0])[it] = (__bf16)0.0f;
        if (NTERM > 1) (&Pl[0][0])[it] = (__bf16)0.0f;
    }
    Patch<TIN> patch;
    float dv[2][2][8];                                            // [item][pixel of the pair][channel]
    auto load_tile = [&](long tile) {
        const int n = (int)(tile / (tiles_y * tiles_x)), r = (int)(tile - (long)n * tiles_y * tiles_x);
        const int oy0 = (r / tiles_x) * TY, ox0 = (r % tiles_x) * TX;
        patch.load(x, n, oy0, ox0, H, W, t);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int item = t + 256 * u, pp = item >> 3, q8 = (item & 7) * 8;
#pragma unroll
            for (int s = 0; s < 2; s++) {
                const int px = 2 * pp + s, oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
                const bool ok = oy < OH && ox < OW;
                load8(dy + (((size_t)n * OH + (ok ? oy : 0)) * OW + (ok ? ox : 0)) * CO + q8, ok, dv[u][s]);
            }
        }
    };
    f16v acc[2];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[c][i] = 0.0f;
    const int kh = 2 * wv + (li >> 4), j = li & 15;               // my operand slot: filter row kh (7 = none: the spare zero row region), element j
    long tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    for (; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
        patch.template store<NTERM>(Ph, Pl, t);
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int item = t + 256 * u, pp = item >> 3, q8 = (item & 7) * 8;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const __bf16 h0 = (__bf16)dv[u][0][e], h1 = (__bf16)dv[u][1][e];
                *(unsigned *)&Dh[q8 + e][2 * pp] =
                    (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                if (NTERM > 1) {
                    const __bf16 l0 = (__bf16)(dv[u][0][e] - (float)h0), l1 = (__bf16)(dv[u][1][e] - (float)h1);
                    *(unsigned *)&Dl[q8 + e][2 * pp] =
                        (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                }
            }
        }
        __syncthreads();
        if (tile + gridDim.x < ntiles) load_tile(tile + gridDim.x);       // in flight during this tile's products
#pragma unroll
        for (int s = 0; s < TY; s++) {                            // one k-step = the 16 pixels of output row s of the tile
            bf16x8 bh, bl;
            const __bf16 *ph = &Ph[s * ST + kh][h * 8 * ST * CIN + j], *pl = &Pl[NTERM > 1 ? s * ST + kh : 0][h * 8 * ST * CIN + j];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                bh[e] = ph[e * ST * CIN];
                if (NTERM > 1) bl[e] = pl[e * ST * CIN];
            }
#pragma unroll
            for (int c = 0; c < 2; c++) {
                const bf16x8 ah = *(const bf16x8 *)&Dh[c * 32 + li][s * 16 + h * 8];
                if (NTERM > 1) {
                    const bf16x8 al = *(const bf16x8 *)&Dl[c * 32 + li][s * 16 + h * 8];
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[c], 0, 0, 0);
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[c], 0, 0, 0);
                }
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[c], 0, 0, 0);
            }
        }
    }
    // ---- D[row = co][col = slot (kh, j)] -> part[workgroup][co][kh * 14 + j]
    if (kh < KH && j < ROWLEN) {
        float *po = part + (size_t)blockIdx.x * CO * KTOT + kh * ROWLEN + j;
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int r = 0; r < 16; r++) po[(size_t)(c * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * KTOT] = acc[c][r];
    }
}

// sum of the workgroups' partial tiles in a fixed order: 16 outputs x 16 interleaved groups of slices per workgroup, then the groups in order
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float *__restrict__ part, int nsplit, int n, float *__restrict__ gw)
{
    __shared__ float red[16][17];
    const int t = threadIdx.x, o = blockIdx.x * 16 + (t & 15), sg = t >> 4;
    float a0 = 0.0f, a1 = 0.0f;
    if (o < n) {
        int s = sg;
#pragma unroll 4
        for (; s + 16 < nsplit; s += 32) {
            a0 += part[(size_t)s * n + o];
            a1 += part[(size_t)(s + 16) * n + o];
        }
        if (s < nsplit) a0 += part[(size_t)s * n + o];
    }
    red[sg][t & 15] = a0 + a1;
    __syncthreads();
    if (t < 16 && o < n) {
        float a = 0.0f;
#pragma unroll
        for (int g = 0; g < 16; g++) a += red[g][t];
        gw[o] = a;
    }
}

constexpr int WGRAD_GROUPS = 512;                                 // two persistent workgroups per compute unit

static inline hipError_t set_max_lds(const void *kern, size_t lds, std::atomic<unsigned long long> &done)
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

template <int NTERM, typename TIN, typename TOUT>
int fwd_launch(const void *x, const float *w, const float *bias, void *y, int N, int H, int W, int OH, int OW, hipStream_t st)
{
    const size_t lds = (size_t)CO * WP * 2 * (NTERM > 1 ? 2 : 1) + (size_t)PR * PP * 2 * (NTERM > 1 ? 2 : 1) +
                       (size_t)128 * (sizeof(TOUT) == 2 ? CO + 8 : CO + 4) * sizeof(TOUT);
    auto kern = stem_fwd_kernel<NTERM, TIN, TOUT>;
    static std::atomic<unsigned long long> attr_set{0};
    PSI_CHECK_HIP(set_max_lds((const void *)kern, lds, attr_set));
    const int tiles_y = (OH + TY - 1) / TY, tiles_x = (OW + TX - 1) / TX;
    hipLaunchKernelGGL(kern, dim3((unsigned)(N * tiles_y)), dim3(256), lds, st, (const TIN *)x, w, bias, (TOUT *)y, N, H, W, OH, OW, tiles_y, tiles_x);
    PSI_CHECK_LAUNCH("stem_fwd_kernel");
    psi_mark("stem_fwd_kernel", st);
    return 0;
}

}  // namespace

bool psi_conv_stem_shape(int Cin, int Cout, int kh, int kw, int stride, int pad)
{
    return Cin == CIN && Cout == CO && kh == KH && kw == KW && stride == ST && pad == PAD;
}

int psi_conv_stem_forward(const void *x, int x_bf16, const float *w, const float *bias, int N, int H, int W, void *y, int y_bf16, int nterm,
                          hipStream_t st)
{
    const int OH = (H + 2 * PAD - KH) / ST + 1, OW = (W + 2 * PAD - KW) / ST + 1;
#define PSI_STEM_ARGS x, w, bias, y, N, H, W, OH, OW, st
    if (nterm == 3) {
        if (x_bf16) return y_bf16 ? fwd_launch<3, __bf16, __bf16>(PSI_STEM_ARGS) : fwd_launch<3, __bf16, float>(PSI_STEM_ARGS);
        return y_bf16 ? fwd_launch<3, float, __bf16>(PSI_STEM_ARGS) : fwd_launch<3, float, float>(PSI_STEM_ARGS);
    }
    if (x_bf16) return y_bf16 ? fwd_launch<1, __bf16, __bf16>(PSI_STEM_ARGS) : fwd_launch<1, __bf16, float>(PSI_STEM_ARGS);
    return y_bf16 ? fwd_launch<1, float, __bf16>(PSI_STEM_ARGS) : fwd_launch<1, float, float>(PSI_STEM_ARGS);
#undef PSI_STEM_ARGS
}

static int stem_wgrad_groups(int N, int OH, int OW)
{
    const long ntiles = (long)N * ((OH + TY - 1) / TY) * ((OW + TX - 1) / TX);
    return (int)(ntiles < WGRAD_GROUPS ? ntiles : WGRAD_GROUPS);
}

size_t psi_conv_stem_wgrad_floats(int N, int H, int W)
{
    const int OH = (H + 2 * PAD - KH) / ST + 1, OW = (W + 2 * PAD - KW) / ST + 1;
    return (size_t)stem_wgrad_groups(N, OH, OW) * CO * KTOT;
}

int psi_conv_stem_weight_grad(const void *x, int x_bf16, const void *dy, int dy_bf16, int N, int H, int W, float *gw, float *ws, int nterm,
                              hipStream_t st)
{
    const int OH = (H + 2 * PAD - KH) / ST + 1, OW = (W + 2 * PAD - KW) / ST + 1;
    const int tiles_y = (OH + TY - 1) / TY, tiles_x = (OW + TX - 1) / TX;
    const int G = stem_wgrad_groups(N, OH, OW);
#define PSI_SW_LAUNCH(NT_, TX_, TD_)                                                                                                        \
    hipLaunchKernelGGL((stem_wgrad_kernel<NT_, TX_, TD_>), dim3((unsigned)G), dim3(256), 0, st, (const TX_ *)x, (const TD_ *)dy, ws, N, H, W, OH, OW, \
                       tiles_y, tiles_x)
    if (nterm == 3) {
        if (x_bf16 && dy_bf16) PSI_SW_LAUNCH(3, __bf16, __bf16);
        else if (x_bf16) PSI_SW_LAUNCH(3, __bf16, float);
        else if (dy_bf16) PSI_SW_LAUNCH(3, float, __bf16);
        else PSI_SW_LAUNCH(3, float, float);
    } else {
        if (x_bf16 && dy_bf16) PSI_SW_LAUNCH(1, __bf16, __bf16);
        else if (x_bf16) PSI_SW_LAUNCH(1, __bf16, float);
        else if (dy_bf16) PSI_SW_LAUNCH(1, float, __bf16);
        else PSI_SW_LAUNCH(1, float, float);
    }
#undef PSI_SW_LAUNCH
    PSI_CHECK_LAUNCH("stem_wgrad_kernel");
    psi_mark("stem_wgrad_kernel", st);
    const int n = CO * KTOT;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, ws, G, n, gw);
    PSI_CHECK_LAUNCH("stem_wgrad_reduce_kernel");
    return 0;
}
