// General 2-D convolution forward of the scene encoders (cvae.py:427-438: Conv2d(2,64,7,2,3) stem, torchvision resnet18 layer1 / layer2 with
// their stride-2 3x3 and 1x1 downsample convolutions, and the 3x3 head convolutions net_layers.py:64,162 / cvae.py:436) as ONE hand-written
// implicit GEMM on the gfx950 matrix cores, in two arithmetic modes:
//
//   NTERM = 1   operands rounded to bf16 (RNE), fp32 accumulate — the bf16 mode of the trunk (models.py autocast_bf16=True): the
//               convolutions the stride-1 kernel of conv.hip does not cover (7x7 stem, strided 3x3, 1x1 downsample, 128 -> 32 head);
//   NTERM = 3   the REFERENCE'S PRECISION (fp32 model, cvae.py:427-455) on the bf16 matrix cores: every fp32 operand is split into
//               hi = bf16(v), lo = bf16(v - hi) and a product is hi*hi + hi*lo + lo*hi with fp32 accumulation (the dropped lo*lo term and the
//               split residue are < 2^-16 of the product).  Measured against the reference's recorded fp32 forward passes
//               (tests/golden/cvae.npz): 0.6-3.2e-5 relative — the bound of the parity tests is 2e-4, a one-term bf16 product is at 3e-3..1e-2.
//               v_mfma_f32_32x32x16_bf16 does 8x the multiply-adds of v_mfma_f32_32x32x2_f32 per issue slot, so three of them are 2.7x
//               faster than the exact-fp32 matrix instruction.
//
// GEMM view: D[co][pixel] = sum_k Wt[co][k] * A[pixel][k], k = (kh, kw, ci) with channels fastest = the memory order of an NHWC activation
// and of a channels_last Conv2d weight [Cout][KH][KW][Cin].  The weight arrives either as the fp32 master (split / rounded per tile on load)
// or PREPARED (psi_conv2d_prepare_weight: its bf16 parts written once per layer and step, in this layout and in the input gradient's).
// Workgroup = 4 waves = 64 output pixels x 64 output channels (four workgroups per CU; 128 pixels x 32 channels for the 32-channel head); the
// K range is walked in chunks of 64 staged through LDS (pixel rows / filter rows padded to 72 elements: the 16-byte operand reads of 16
// adjacent lanes fall on different bank groups); the next chunk's global loads are in flight while the current chunk is multiplied.  With
// Cin % 16 == 0 a thread's run of a chunk is 16 consecutive channels of ONE filter tap (64 contiguous bytes per pixel).  The same kernel in its
// transposed-gather form is the input gradient; the weight gradient (pixel contraction) follows below.  The 2-channel 7x7 stem has kernels
// of its own (conv_stem.hip); what bounds this one (vector issue: 2033 vector instructions per wave against 108 MFMAs) is in DESIGN.md section 5.
#include "psi_internal.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>
#ifndef PSI_CONV_BM_DEFAULT
#define PSI_CONV_BM_DEFAULT 64
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

constexpr int KC = 64, PITCH = KC + 8;

__device__ __forceinline__ float bf_round(float v, __bf16 &hi)
{
    hi = (__bf16)v;
    return v - (float)hi;
}

// 32 consecutive input elements of one pixel (or zeros) as floats
__device__ __forceinline__ void load32(const float *p, bool ok, float (&v)[32])
{
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const f4 a = ok ? *(const f4 *)(p + 4 * i) : (f4){0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; e++) v[4 * i + e] = a[e];
    }
}
__device__ __forceinline__ void load32(const __bf16 *p, bool ok, float (&v)[32])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bf16x8 a;
        if (ok) a = *(const bf16x8 *)(p + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; e++) v[8 * i + e] = ok ? (float)a[e] : 0.0f;
    }
}
// 16 consecutive elements (or zeros) as floats
__device__ __forceinline__ void load16(const float *p, bool ok, float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const f4 a = ok ? *(const f4 *)(p + 4 * i) : (f4){0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; e++) v[4 * i + e] = a[e];
    }
}
__device__ __forceinline__ void load16(const __bf16 *p, bool ok, float (&v)[16])
{
#pragma unroll
    for (int i = 0; i < 2; i++) {
        bf16x8 a;
        if (ok) a = *(const bf16x8 *)(p + 8 * i);
#pragma unroll
        for (int e = 0; e < 8; e++) v[8 * i + e] = ok ? (float)a[e] : 0.0f;
    }
}
__device__ __forceinline__ float ldf(const float *p) { return *p; }
__device__ __forceinline__ float ldf(const __bf16 *p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void loadN(const T *p, bool ok, float (&v)[32]) { load32(p, ok, v); }
template <typename T> __device__ __forceinline__ void loadN(const T *p, bool ok, float (&v)[16]) { load16(p, ok, v); }

__device__ __forceinline__ void store_out(float *y, const float (&v)[4]) { *(f4 *)y = (f4){v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void store_out(__bf16 *y, const float (&v)[4])
{
    bf16x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = (__bf16)v[e];
    *(bf16x4 *)y = o;
}

// floats -> LDS row pieces: hi (and lo) bf16, 8 elements per 16-byte store
template <int NTERM, int NV>
__device__ __forceinline__ void split_store(const float (&v)[NV], __bf16 *hi, __bf16 *lo)
{
#pragma unroll
    for (int i = 0; i < NV / 8; i++) {
        bf16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            __bf16 hh;
            const float r = bf_round(v[8 * i + e], hh);
            h[e] = hh;
            if (NTERM > 1) l[e] = (__bf16)r;
        }
        *(bf16x8 *)(hi + 8 * i) = h;
        if (NTERM > 1) *(bf16x8 *)(lo + 8 * i) = l;
    }
}

// PREP: the weights arrive already split (psi_conv2d_prepare_weight: [Cout][K] bf16 hi parts, the lo parts Cout * K elements behind them) and
// go from memory to LDS as they are — the forward and the input gradient of a layer share one preparation per step, and a workgroup no longer
// spends a quarter of its vector instructions re-splitting a 64 x 64 weight tile that 2000 other workgroups split as well.
template <int NTERM, typename TIN, typename TOUT, int BN, int BM, bool RUNS, bool PREP>
__global__ __launch_bounds__(256, BM == 64 ? 4 : 2) void conv_gemm_kernel(const TIN *__restrict__ x, const float *__restrict__ w, const __bf16 *__restrict__ wp,
                                                           const float *__restrict__ bias, TOUT *__restrict__ y, int N, int H, int W, int Cin, int OH,
                                                           int OW, int Cout, int KH, int KW, int stride, int pad, int transposed)
{
    // transposed != 0: the INPUT GRADIENT of a convolution — "x" is dY [N,H,W,Cin] (H x W = the forward's output size, Cin = its Cout), "y" is
    // dX [N,OH,OW,Cout] (the forward's input), w = the forward weight re-laid out as [Cin_fwd][KH][KW][Cout_fwd]; output pixel (iy, ix) takes
    // tap (kh, kw) from dY[(iy + pad - kh) / stride, (ix + pad - kw) / stride] where that division is exact (no weight flip in this form)
    // BM = 128 pixels per workgroup (wave w = pixel tile w, all channel tiles), or 64 (waves = 2 pixel tiles x 2 channel tiles): the small
    // tile has half the LDS and registers, FOUR workgroups fit a compute unit instead of two — the kernel waits for its global loads once per
    // 64-wide K chunk (one chunk's MFMAs are ~0.3 us, a load round trip ~1.5 us), and more resident workgroups are what covers that
    static_assert((BN == 64 || BN == 32) && (BM == 128 || BM == 64) && !(BN == 32 && BM == 64), "tile shape");
    constexpr int NCT = BN / 32;                                   // 32-channel MFMA tiles of the workgroup
    constexpr int WPX = BM / 32, WCO = 4 / WPX, CPW = NCT / WCO;   // waves along pixels / channels; channel tiles per wave
    constexpr int APT = BM * KC / 256;                             // input elements per thread and chunk (32 or 16)
    constexpr int WPT = BN * KC / 256;                             // weight elements per thread and chunk (16 or 8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __bf16 (*Ah)[PITCH] = (__bf16 (*)[PITCH])smem;                                           // [BM][PITCH]   pixels, hi
    __bf16 (*Bh)[PITCH] = (__bf16 (*)[PITCH])(smem + (size_t)BM * PITCH * 2);                 // [BN][PITCH]   filters, hi
    __bf16 (*Al)[PITCH] = (__bf16 (*)[PITCH])(smem + (size_t)(BM + BN) * PITCH * 2);          // lo parts (NTERM = 3)
    __bf16 (*Bl)[PITCH] = (__bf16 (*)[PITCH])(smem + (size_t)(2 * BM + BN) * PITCH * 2);
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int li = lane & 31, kb = (lane >> 5) * 8;
    const long M = (long)N * OH * OW;
    const long m0 = (long)blockIdx.x * BM;
    const int co0 = blockIdx.y * BN;
    const int K = KH * KW * Cin;
    // how a thread's run of APT consecutive k of a chunk is fetched:
    //   RUNS    Cin % APT == 0: the run lies inside ONE filter tap — APT consecutive channels of one pixel, vector loads;
    //   ROWS    small Cin (Cin * KW <= 16: the 7x7 stem, Cin = 2): k is re-indexed as (kh, 16 slots) — the Cin * KW values of one filter ROW are
    //           contiguous in an NHWC map (14 floats for the stem), a thread fetches one such row per 16 slots, two slots stay zero;
    //   GATHER  anything else: element by element.
    const int rowlen = Cin * KW;
    const int mode = RUNS ? 0 : ((rowlen <= 16 && !transposed) ? 1 : 2);      // (RUNS <=> Cin % APT == 0: decided at launch, its own instantiation)
    const int Keff = mode == 1 ? KH * 16 : K;
    const int nck = (Keff + KC - 1) / KC;
    // ---- this thread's share of a chunk: pixel row ar (APT of its 64 elements), filter row br (WPT of its 64)
    const int ar = t / (KC / APT), ah = (t % (KC / APT)) * APT;
    // pixel order.  The input gradient of a stride-2 convolution (transposed != 0) takes a tap only where (i + pad - k) is even: the pixels are
    // numbered CLASS-major (class = parity of (iy, ix)) so that a workgroup's pixels share their class — a tap is then valid for all of them or
    // for none, and the chunks of the other 5-8 of 9 taps are skipped outright (they would multiply zeros)
    const bool classes = transposed && stride == 2 && (OH & 1) == 0 && (OW & 1) == 0 && ((M / 4) % BM) == 0;
    const long Mq = M / 4;
    auto pixel = [&](long m, int &n, int &oy, int &ox) {
        if (classes) {
            const int cls = (int)(m / Mq);
            const long r = m - (long)cls * Mq;
            const int OHq = OH >> 1, OWq = OW >> 1;
            n = (int)(r / ((long)OHq * OWq));
            const int rem = (int)(r - (long)n * OHq * OWq);
            oy = 2 * (rem / OWq) + (cls >> 1);
            ox = 2 * (rem % OWq) + (cls & 1);
        } else {
            n = (int)(m / ((long)OH * OW));
            const int rem = (int)(m - (long)n * OH * OW);
            oy = rem / OW;
            ox = rem - oy * OW;
        }
    };
    const int wg_cls = classes ? (int)(m0 / Mq) : 0;
    const long am = m0 + ar;
    const bool a_live = am < M;
    int an = 0, aoy = 0, aox = 0;
    if (a_live) pixel(am, an, aoy, aox);
    const int br = t / (KC / WPT), bq = (t % (KC / WPT)) * WPT;
    const float *wrow = w + (size_t)(co0 + br) * K;
    float av[APT], bv[WPT];
    bf16x8 pwh[WPT / 8], pwl[WPT / 8];                            // PREP: my piece of the chunk's weight tile, as stored
    auto load_prepared = [&](int ck) {
        const size_t at = (size_t)(co0 + br) * K + (size_t)ck * KC + bq;
#pragma unroll
        for (int i = 0; i < WPT / 8; i++) {
            const bool ok = ck * KC + bq + 8 * i < K;              // (K % 8 == 0)
#pragma unroll
            for (int e = 0; e < 8; e++) pwh[i][e] = pwl[i][e] = (__bf16)0.0f;
            if (ok) {
                pwh[i] = *(const bf16x8 *)(wp + at + 8 * i);
                if (NTERM > 1) pwl[i] = *(const bf16x8 *)(wp + (size_t)Cout * K + at + 8 * i);
            }
        }
    };
    auto src = [&](int kh, int kw, int &iy, int &ix) {            // input pixel of tap (kh, kw) for my output pixel; false: outside / no such tap
        if (transposed) {
            const int ty = aoy + pad - kh, tx = aox + pad - kw;
            if (stride <= 2) {                                     // (no run-time division in the chunk loop; negative ty / tx are rejected below)
                iy = ty >> (stride - 1);
                ix = tx >> (stride - 1);
            } else {
                iy = ty / stride;
                ix = tx / stride;
            }
            return ty >= 0 && tx >= 0 && iy * stride == ty && ix * stride == tx && iy < H && ix < W;
        }
        iy = aoy * stride - pad + kh;
        ix = aox * stride - pad + kw;
        return iy >= 0 && iy < H && ix >= 0 && ix < W;
    };
    auto chunk_live = [&](int ck) {                                // (workgroup-uniform) does any tap of this chunk exist for this pixel class?
        if (!classes) return true;
        const int t0 = (ck * KC) / Cin, t1 = min((ck * KC + KC - 1) / Cin, KH * KW - 1);
        for (int tp = t0; tp <= t1; tp++) {
            const int kh = tp / KW, kw = tp - kh * KW;
            if ((((wg_cls >> 1) + pad - kh) & 1) == 0 && (((wg_cls & 1) + pad - kw) & 1) == 0) return true;
        }
        return false;
    };
    // mode 0: my run of a chunk = one tap, channels rc0 .. rc0 + APT.  (kh, kw, channel) of the run are carried from chunk to chunk: the
    // chunk loop had two integer divisions by run-time values per thread and chunk
    int rck = 0, rkh, rkw, rc0;
    {
        const int tap = ah / Cin;
        rc0 = ah - tap * Cin;
        rkh = tap / KW;
        rkw = tap - rkh * KW;
    }
    auto load_chunk = [&](int ck) {
        if (mode == 0) {
            rc0 += (ck - rck) * KC;                                // (ck only grows)
            rck = ck;
            while (rc0 >= Cin) {
                rc0 -= Cin;
                if (++rkw == KW) { rkw = 0; rkh++; }
            }
            int iy = 0, ix = 0;
            const bool ok = a_live && rkh < KH && src(rkh, rkw, iy, ix);
            loadN(x + (((size_t)an * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cin + (ok ? rc0 : 0), ok, av);
            if (PREP) {
                load_prepared(ck);
            } else {
                const bool wok = ck * KC + bq < K;
#pragma unroll
                for (int i = 0; i < WPT / 4; i++) {
                    const f4 a = wok ? *(const f4 *)(wrow + (size_t)ck * KC + bq + 4 * i) : (f4){0, 0, 0, 0};
#pragma unroll
                    for (int e = 0; e < 4; e++) bv[4 * i + e] = a[e];
                }
            }
        } else if (mode == 1) {
#pragma unroll
            for (int r0 = 0; r0 < APT; r0 += 16) {                 // one filter row per 16 slots
                const int kh = (ck * KC + ah + r0) >> 4;
                const int iy = aoy * stride - pad + kh, ixb = aox * stride - pad;
                const bool rok = a_live && kh < KH && iy >= 0 && iy < H;
                const TIN *row = x + ((size_t)an * H + (rok ? iy : 0)) * W * Cin;
                int kw = 0, ci = 0;
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int ix = ixb + kw;
                    av[r0 + j] = (rok && j < rowlen && ix >= 0 && ix < W) ? ldf(row + (size_t)ix * Cin + ci) : 0.0f;
                    if (++ci == Cin) { ci = 0; kw++; }
                }
            }
#pragma unroll
            for (int e = 0; e < WPT; e++) {
                const int kk = ck * KC + bq + e, kh = kk >> 4, j = kk & 15;
                bv[e] = (kh < KH && j < rowlen) ? wrow[kh * rowlen + j] : 0.0f;
            }
        } else {
            // element k = (tap, ci), gathered one by one; rows beyond K are zeros
#pragma unroll 8
            for (int e = 0; e < APT; e++) {
                const int k = ck * KC + ah + e;
                float v = 0.0f;
                if (a_live && k < K) {
                    const int tap = k / Cin, ci = k - tap * Cin;
                    const int kh = tap / KW, kw = tap - kh * KW;
                    int iy, ix;
                    if (src(kh, kw, iy, ix)) v = ldf(x + (((size_t)an * H + iy) * W + ix) * Cin + ci);
                }
                av[e] = v;
            }
            if (PREP) {
                load_prepared(ck);
            } else {
#pragma unroll
                for (int e = 0; e < WPT; e++) {
                    const int k = ck * KC + bq + e;
                    bv[e] = k < K ? wrow[k] : 0.0f;
                }
            }
        }
    };
    const int wpt = wv % WPX, wc0 = (wv / WPX) * CPW;              // this wave's pixel tile and first channel tile
    f16v acc[CPW];
#pragma unroll
    for (int c = 0; c < CPW; c++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[c][i] = 0.0f;
    int ck = 0;
    while (ck < nck && !chunk_live(ck)) ck++;
    if (ck < nck) load_chunk(ck);
    while (ck < nck) {
        int nx = ck + 1;
        while (nx < nck && !chunk_live(nx)) nx++;
        __syncthreads();                                           // the previous chunk's MFMAs are done with LDS
        split_store<NTERM, APT>(av, &Ah[ar][ah], &Al[ar][ah]);
        if (PREP) {
#pragma unroll
            for (int i = 0; i < WPT / 8; i++) {
                *(bf16x8 *)&Bh[br][bq + 8 * i] = pwh[i];
                if (NTERM > 1) *(bf16x8 *)&Bl[br][bq + 8 * i] = pwl[i];
            }
        } else {
            split_store<NTERM, WPT>(bv, &Bh[br][bq], &Bl[br][bq]);
        }
        __syncthreads();
        if (nx < nck) load_chunk(nx);                              // in flight during this chunk's MFMAs
        ck = nx;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ks++) {
            const bf16x8 bh = *(const bf16x8 *)&Ah[wpt * 32 + li][ks * 16 + kb];     // B operand: my pixel tile
            bf16x8 bl;
            if (NTERM > 1) bl = *(const bf16x8 *)&Al[wpt * 32 + li][ks * 16 + kb];
#pragma unroll
            for (int c = 0; c < CPW; c++) {
                const bf16x8 ahh = *(const bf16x8 *)&Bh[(wc0 + c) * 32 + li][ks * 16 + kb];  // A operand: filter rows
                if (NTERM > 1) {
                    const bf16x8 all = *(const bf16x8 *)&Bl[(wc0 + c) * 32 + li][ks * 16 + kb];
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(all, bh, acc[c], 0, 0, 0);      // the small terms first
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahh, bl, acc[c], 0, 0, 0);
                }
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahh, bh, acc[c], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: D[row = co][col = pixel]; lane (li, h) holds rows 8g + 4h + (0..3), g = 0..3, of column li
    const long om = m0 + wpt * 32 + li;
    if (om >= M) return;
    const int h = lane >> 5;
    long opix = om;
    if (classes) {
        int on, ooy, oox;
        pixel(om, on, ooy, oox);
        opix = ((long)on * OH + ooy) * OW + oox;
    }
    TOUT *yo = y + (size_t)opix * Cout + co0;
#pragma unroll
    for (int c = 0; c < CPW; c++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int co = (wc0 + c) * 32 + 8 * g + 4 * h;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = acc[c][4 * g + e] + (bias ? bias[co0 + co + e] : 0.0f);
            store_out(yo + co, v);
        }
}

// fp32 master weight [Cout][taps][Cin] -> bf16 parts, in the forward layout ([Cout][K]: hi parts, then lo parts for NTERM = 3) and, when the
// layer's input needs a gradient, in the input gradient's layout ([Cin][taps][Cout], the same way): one small launch per layer and step
// instead of a re-layout copy for the backward plus a re-split of the tile in every workgroup of both kernels
template <int NTERM>
__global__ __launch_bounds__(256) void conv_weight_split_kernel(const float *__restrict__ w, int Cout, int taps, int Cin, __bf16 *__restrict__ wf,
                                                                __bf16 *__restrict__ wt)
{
    const int total = Cout * taps * Cin, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    __bf16 hi;
    const float r = bf_round(w[i], hi);
    const __bf16 lo = (__bf16)r;
    if (wf) {
        wf[i] = hi;
        if (NTERM > 1) wf[(size_t)total + i] = lo;
    }
    if (wt) {
        const int ci = i % Cin, tap = (i / Cin) % taps, co = i / (Cin * taps);
        const size_t j = ((size_t)ci * taps + tap) * Cout + co;
        wt[j] = hi;
        if (NTERM > 1) wt[(size_t)total + j] = lo;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the same convolutions:  dW[co][k] = sum_m dY[m][co] * A[m][k],  m = output pixel, k = (kh, kw, ci), A the im2col view
// the forward gathers.  The contraction runs over PIXELS, along which neither operand is contiguous in memory (channels are fastest), so
// both 64-pixel tiles are TRANSPOSED on their way into LDS ([channel][pixel] rows, 2-byte stores; the split into hi / lo parts happens in
// the same pass) and the MFMA operands are 16-byte LDS reads again.  A workgroup owns the 64 x 64 tile (co-tile, one 64-wide K chunk = 64
// channels of one tap) and a slice of the pixel range; the slices' partial tiles are summed in slice order by conv_wgrad_reduce_kernel
// (deterministic: no atomics).  NTERM as in the forward: 3 = both operands split (the fp32 model's precision), 1 = bf16 products.
// ------------------------------------------------------------------------------------------------
constexpr int WG_PX = 64, WG_PITCH = WG_PX + 8;

// eight values of pixel 2p and of pixel 2p + 1 -> eight {pixel 2p, pixel 2p + 1} bf16 pairs (one 32-bit LDS store per channel row and part): a
// thread that stored single 2-byte values did four times as many LDS writes, and those writes — not the MFMAs — set the pace of a stage
template <int NTERM>
__device__ __forceinline__ void split_store_pairs(const float (&v0)[8], const float (&v1)[8], __bf16 (*hi)[WG_PITCH], __bf16 (*lo)[WG_PITCH], int row0,
                                                  int col)
{
#pragma unroll
    for (int e = 0; e < 8; e++) {
        __bf16 h0, h1;
        const float r0 = bf_round(v0[e], h0), r1 = bf_round(v1[e], h1);
        *(unsigned *)&hi[row0 + e][col] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
        if (NTERM > 1) {
            const __bf16 l0 = (__bf16)r0, l1 = (__bf16)r1;
            *(unsigned *)&lo[row0 + e][col] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
    }
}

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
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const TIN *__restrict__ x, const TDY *__restrict__ dy, float *__restrict__ part, int N,
                                                            int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW, int stride, int pad,
                                                            int nsplit, int stages_per_split)
{
    __shared__ __attribute__((aligned(16))) __bf16 Dh[64][WG_PITCH], Xh[64][WG_PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Dl[NTERM > 1 ? 64 : 1][WG_PITCH], Xl[NTERM > 1 ? 64 : 1][WG_PITCH];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int li = lane & 31, kb = (lane >> 5) * 8, h = lane >> 5;
    const int ct = wv & 1, kt = wv >> 1;                          // this wave's 32 x 32 quadrant of the tile
    const int K = KH * KW * Cin, rowlen = Cin * KW;
    const int mode = (Cin % 8) == 0 ? 0 : (rowlen <= 16 ? 1 : 2);  // as in conv_gemm_kernel: tap runs / filter rows of a small Cin / element gather
    const int ck = blockIdx.x, co0 = blockIdx.y * 64, split = blockIdx.z;
    const long M = (long)N * OH * OW;
    // this thread's share of a stage: the pixel PAIR (2 pp, 2 pp + 1) of the 64, eight of the tile's 64 channels / k-columns
    const int pp = t >> 3, q8 = (t & 7) * 8;
    const int k0r = ck * KC + q8;                                  // my eight k-columns (mode 0: one tap; mode 1: eight slots of one filter row)
    int kh = 0, kw = 0, c0 = 0, j0 = 0;
    if (mode == 0) {
        const int tap = k0r / Cin;
        c0 = k0r - tap * Cin;
        kh = tap / KW;
        kw = tap - kh * KW;
    } else if (mode == 1) {
        kh = k0r >> 4;
        j0 = k0r & 15;
    }
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    float dv[2][8], xv[2][8];
    // (n, oy, ox) of my first pixel are carried from stage to stage (consecutive stages are 64 pixels apart): the stage loop had four integer
    // divisions by run-time values per thread
    long pm = -1;
    int pn = 0, poy = 0, pox = 0;
    auto load_stage = [&](long m0) {
        const long ma = m0 + 2 * pp;
        if (pm < 0) {
            pn = (int)(ma / ((long)OH * OW));
            const int rem = (int)(ma - (long)pn * OH * OW);
            poy = rem / OW;
            pox = rem - poy * OW;
        } else {
            pox += (int)(ma - pm);
            while (pox >= OW) {
                pox -= OW;
                if (++poy == OH) { poy = 0; pn++; }
            }
        }
        pm = ma;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const long m = ma + u;
            const bool live = m < M;
            int n = pn, oy = poy, ox = pox + u;
            if (ox >= OW) {
                ox -= OW;
                if (++oy == OH) { oy = 0; n++; }
            }
            const bool dok = live && co0 + q8 + 8 <= Cout;        // (Cout % 8 == 0: whole 8-channel pieces)
            load8(dy + (size_t)(dok ? m : 0) * Cout + (dok ? co0 + q8 : 0), dok, dv[u]);
            if (mode == 0) {
                const int iy = oy * stride - pad + kh, ix = ox * stride - pad + kw;
                const bool ok = live && k0r < K && iy >= 0 && iy < H && ix >= 0 && ix < W;
                load8(x + (((size_t)n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * Cin + (ok ? c0 : 0), ok, xv[u]);
            } else if (mode == 1) {
                const int iy = oy * stride - pad + kh, ixb = ox * stride - pad;
                const bool rok = live && kh < KH && iy >= 0 && iy < H;
                const TIN *row = x + ((size_t)n * H + (rok ? iy : 0)) * W * Cin;
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int j = j0 + e, kwj = j / Cin, ci = j - kwj * Cin, ix = ixb + kwj;
                    xv[u][e] = (rok && j < rowlen && ix >= 0 && ix < W) ? ldf(row + (size_t)ix * Cin + ci) : 0.0f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int k = k0r + e;
                    float v = 0.0f;
                    if (live && k < K) {
                        const int tp = k / Cin, ci = k - tp * Cin;
                        const int kh2 = tp / KW, kw2 = tp - kh2 * KW;
                        const int iy = oy * stride - pad + kh2, ix = ox * stride - pad + kw2;
                        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = ldf(x + (((size_t)n * H + iy) * W + ix) * Cin + ci);
                    }
                    xv[u][e] = v;
                }
            }
        }
    };
    const long stage0 = (long)split * stages_per_split, nstage = (M + WG_PX - 1) / WG_PX;
    const long stage1 = stage0 + stages_per_split < nstage ? stage0 + stages_per_split : nstage;
    if (stage0 < stage1) load_stage(stage0 * WG_PX);
    for (long sg = stage0; sg < stage1; sg++) {
        __syncthreads();                                           // the previous stage's MFMAs are done with LDS
        split_store_pairs<NTERM>(dv[0], dv[1], Dh, Dl, q8, 2 * pp);
        split_store_pairs<NTERM>(xv[0], xv[1], Xh, Xl, q8, 2 * pp);
        __syncthreads();
        if (sg + 1 < stage1) load_stage((sg + 1) * WG_PX);         // in flight during this stage's MFMAs
#pragma unroll
        for (int ks = 0; ks < WG_PX / 16; ks++) {
            const bf16x8 a = *(const bf16x8 *)&Dh[ct * 32 + li][ks * 16 + kb];
            const bf16x8 b = *(const bf16x8 *)&Xh[kt * 32 + li][ks * 16 + kb];
            if (NTERM > 1) {
                const bf16x8 al = *(const bf16x8 *)&Dl[ct * 32 + li][ks * 16 + kb];
                const bf16x8 bl = *(const bf16x8 *)&Xl[kt * 32 + li][ks * 16 + kb];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    // partial tile: D[row = co][col = k-column] -> part[split][co][k]
    float *po = part + (size_t)split * Cout * K;
    int kcol = ck * KC + kt * 32 + li;
    if (mode == 1) {                                              // slot (kh, j) -> k = kh * rowlen + j (the two padding slots of a row: nothing to store)
        const int khc = kcol >> 4, j = kcol & 15;
        kcol = (khc < KH && j < rowlen) ? khc * rowlen + j : K;
    }
    if (kcol < K) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int co = co0 + ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (co < Cout) po[(size_t)co * K + kcol] = acc[r];
        }
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float *__restrict__ part, int nsplit, long n, float *__restrict__ gw)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int s = 0;
    for (; s + 7 < nsplit; s += 8) {                          // eight loads in flight; fixed order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = part[(size_t)(s + u) * n + i];
        a0 += v[0] + v[4];
        a1 += v[1] + v[5];
        a2 += v[2] + v[6];
        a3 += v[3] + v[7];
    }
    for (; s < nsplit; s++) a0 += part[(size_t)s * n + i];
    gw[i] = (a0 + a1) + (a2 + a3);
}

static int wgrad_splits(long M, int K, int Cout)
{
    const long tiles = (long)((K + KC - 1) / KC) * ((Cout + 63) / 64), nstage = (M + WG_PX - 1) / WG_PX;
    long S = 512 / tiles;                                          // about two workgroups per compute unit
    if (S > nstage) S = nstage;
    return (int)(S < 1 ? 1 : S);
}

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

template <int NTERM, typename TIN, typename TOUT, int BN, int BM, bool RUNS, bool PREP>
int launch_p(const void *x, const float *w, const __bf16 *wp, const float *bias, void *y, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH,
             int KW, int stride, int pad, int transposed, hipStream_t st)
{
    const size_t lds = (size_t)(BM + BN) * PITCH * 2 * (NTERM > 1 ? 2 : 1);
    auto kern = conv_gemm_kernel<NTERM, TIN, TOUT, BN, BM, RUNS, PREP>;
    static std::atomic<unsigned long long> attr_set{0};
    PSI_CHECK_HIP(set_max_lds((const void *)kern, lds, attr_set));
    const long M = (long)N * OH * OW;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)(Cout / BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const TIN *)x, w, wp, bias, (TOUT *)y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, transposed);
    PSI_CHECK_LAUNCH("conv_gemm_kernel");
    psi_mark("conv_gemm_kernel", st);
    return 0;
}

// w: fp32 [Cout][K] as the layer holds it, or (prepared != 0) the bf16 parts psi_conv2d_prepare_weight wrote
template <int NTERM, typename TIN, typename TOUT, int BN, int BM, bool RUNS>
int launch_r(const void *x, const float *w, const float *bias, void *y, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
             int stride, int pad, int transposed, hipStream_t st, int prepared)
{
    if (prepared) {
        PSI_REQUIRE((KH * KW * Cin) % 8 == 0 && (RUNS || transposed || Cin * KW > 16), "prepared weights: K % 8 == 0, not the filter-row path");
        return launch_p<NTERM, TIN, TOUT, BN, BM, RUNS, true>(x, nullptr, (const __bf16 *)w, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad,
                                                              transposed, st);
    }
    return launch_p<NTERM, TIN, TOUT, BN, BM, RUNS, false>(x, w, nullptr, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, transposed, st);
}

template <int NTERM, typename TIN, typename TOUT, int BN, int BM>
int launch(const void *x, const float *w, const float *bias, void *y, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
           int stride, int pad, int transposed, hipStream_t st, int prepared)
{
    // a thread's run of BM * 64 / 256 consecutive k lies inside one filter tap: the vector-load instantiation; else filter rows / element gather
    if (Cin % (BM * KC / 256) == 0)
        return launch_r<NTERM, TIN, TOUT, BN, BM, true>(x, w, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, transposed, st, prepared);
    return launch_r<NTERM, TIN, TOUT, BN, BM, false>(x, w, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, transposed, st, prepared);
}

template <int NTERM, typename TIN, typename TOUT>
int launch_bn(const void *x, const float *w, const float *bias, void *y, int N, int H, int W, int Cin, int OH, int OW, int Cout, int KH, int KW,
              int stride, int pad, int transposed, hipStream_t st, int prepared = 0)
{
    // pixel tile: 64 (four workgroups per compute unit) unless PSI_CONV_BM=128 says otherwise; the 32-channel tile keeps 128 pixels
    static const int bm = getenv("PSI_CONV_BM") ? atoi(getenv("PSI_CONV_BM")) : PSI_CONV_BM_DEFAULT;
    if (Cout % 64 == 0) {
        if (bm == 64 || Cin % 32 != 0)
            return launch<NTERM, TIN, TOUT, 64, 64>(x, w, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, transposed, st, prepared);
        return launch<NTERM, TIN, TOUT, 64, 128>(x, w, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, transposed, st, prepared);
    }
    return launch<NTERM, TIN, TOUT, 32, 128>(x, w, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, transposed, st, prepared);
}

// the stem has kernels of its own (conv_stem.hip); PSI_CONV_STEM=0 sends it through the general kernels (dev A/B)
static bool stem_route(int Cin, int Cout, int KH, int KW, int stride, int pad)
{
    static const bool on = !(getenv("PSI_CONV_STEM") && atoi(getenv("PSI_CONV_STEM")) == 0);
    return on && psi_conv_stem_shape(Cin, Cout, KH, KW, stride, pad);
}

}  // namespace

extern "C" int psi_conv2d_supported(int Cin, int Cout, int KH, int KW, int stride, int pad)
{
    return Cin > 0 && Cout > 0 && Cout % 32 == 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0 && (Cin % 16 == 0 || Cin * KH * KW <= 4096);
}

// x [N,H,W,Cin] NHWC (x_bf16: bf16, else fp32); w [Cout,KH,KW,Cin] fp32 (a channels_last Conv2d weight); bias [Cout] fp32 or NULL;
// y [N,OH,OW,Cout] NHWC (y_bf16: bf16, else fp32), OH = (H + 2 pad - KH) / stride + 1.  nterm = 1 | 3 (see the header of this file).
// PSI_CONV3X3_SPLIT=0 (dev A/B): the stride-1 3x3 layers of the fp32 model stay on the general kernels of this file
static bool conv3x3_split_route()
{
    static const bool on = !(getenv("PSI_CONV3X3_SPLIT") && getenv("PSI_CONV3X3_SPLIT")[0] == '0');
    return on;
}

static int conv2d_forward_any(const void *x, int x_bf16, const float *w, int prepared, const float *bias, int N, int H, int W, int Cin, int Cout, int KH,
                              int KW, int stride, int pad, void *y, int y_bf16, int nterm, void *stream)
{
    PSI_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, "bad arguments");
    PSI_REQUIRE(psi_conv2d_supported(Cin, Cout, KH, KW, stride, pad), "shape not covered: Cout % 32 == 0 and (Cin % 64 == 0 or a small gather case)");
    PSI_REQUIRE(nterm == 1 || nterm == 3, "nterm is 1 (bf16 products) or 3 (split products: the fp32 model's precision)");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    PSI_REQUIRE(OH > 0 && OW > 0, "empty output");
    hipStream_t st = (hipStream_t)stream;
    if (stem_route(Cin, Cout, KH, KW, stride, pad)) {
        PSI_REQUIRE(!prepared, "the stem's kernels read the fp32 weight");
        return psi_conv_stem_forward(x, x_bf16, w, bias, N, H, W, y, y_bf16, nterm, st);
    }
    // the stride-1 3x3 layers of the fp32 model with 64 input channels: the halo tile split ONCE per workgroup (conv.hip: conv3x3s_kernel)
    if (conv3x3_split_route() && prepared && nterm == 3 && !x_bf16 && !y_bf16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && psi_conv3x3s_ok(N, H, W, Cin, Cout))
        return psi_conv3x3s_forward((const float *)x, (const void *)w, bias, N, H, W, Cin, Cout, (float *)y, 0, st);
#define PSI_CONV_ARGS x, w, bias, y, N, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad, 0, st, prepared
    if (nterm == 3) {
        if (x_bf16) return y_bf16 ? launch_bn<3, __bf16, __bf16>(PSI_CONV_ARGS) : launch_bn<3, __bf16, float>(PSI_CONV_ARGS);
        return y_bf16 ? launch_bn<3, float, __bf16>(PSI_CONV_ARGS) : launch_bn<3, float, float>(PSI_CONV_ARGS);
    }
    if (x_bf16) return y_bf16 ? launch_bn<1, __bf16, __bf16>(PSI_CONV_ARGS) : launch_bn<1, __bf16, float>(PSI_CONV_ARGS);
    return y_bf16 ? launch_bn<1, float, __bf16>(PSI_CONV_ARGS) : launch_bn<1, float, float>(PSI_CONV_ARGS);
#undef PSI_CONV_ARGS
}

extern "C" int psi_conv2d_forward(const void *x, int x_bf16, const float *w, const float *bias, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                  int stride, int pad, void *y, int y_bf16, int nterm, void *stream)
{
    return conv2d_forward_any(x, x_bf16, w, 0, bias, N, H, W, Cin, Cout, KH, KW, stride, pad, y, y_bf16, nterm, stream);
}

// The same with a PREPARED weight (psi_conv2d_prepare_weight's `wf`): no per-workgroup rounding / splitting of the weight tile.
extern "C" int psi_conv2d_forward_p(const void *x, int x_bf16, const void *wf, const float *bias, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                    int stride, int pad, void *y, int y_bf16, int nterm, void *stream)
{
    return conv2d_forward_any(x, x_bf16, (const float *)wf, 1, bias, N, H, W, Cin, Cout, KH, KW, stride, pad, y, y_bf16, nterm, stream);
}

extern "C" int psi_conv2d_prepared_ok(int Cin, int Cout, int KH, int KW, int stride, int pad)
{
    return psi_conv2d_supported(Cin, Cout, KH, KW, stride, pad) && Cin % 16 == 0 && !psi_conv_stem_shape(Cin, Cout, KH, KW, stride, pad);
}

// w [Cout][KH][KW][Cin] fp32 -> wf (forward layout) and / or wt (input-gradient layout [Cin][KH][KW][Cout]); each Cout*KH*KW*Cin bf16 (nterm = 1)
// or twice that (nterm = 3: hi parts, then lo parts).  Either output may be NULL.
extern "C" int psi_conv2d_prepare_weight(const float *w, int Cout, int KH, int KW, int Cin, int nterm, void *wf, void *wt, void *stream)
{
    PSI_REQUIRE(w && (wf || wt) && Cout > 0 && KH > 0 && KW > 0 && Cin > 0, "bad arguments");
    PSI_REQUIRE(nterm == 1 || nterm == 3, "nterm is 1 or 3");
    const int total = Cout * KH * KW * Cin;
    hipStream_t st = (hipStream_t)stream;
    if (nterm == 3)
        hipLaunchKernelGGL(conv_weight_split_kernel<3>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, Cout, KH * KW, Cin, (__bf16 *)wf, (__bf16 *)wt);
    else
        hipLaunchKernelGGL(conv_weight_split_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, Cout, KH * KW, Cin, (__bf16 *)wf, (__bf16 *)wt);
    PSI_CHECK_LAUNCH("conv_weight_split_kernel");
    return 0;
}

// Input gradient of the convolution above: dy [N,OH,OW,Cout] -> dx [N,H,W,Cin] (OVERWRITTEN), wt = the forward weight re-laid out as
// [Cin][KH][KW][Cout] fp32 (torch: weight.permute(1, 2, 3, 0).contiguous()).  The same kernel in its transposed-gather form; Cin % 32 == 0,
// Cout % 64 == 0 (or Cout * KH * KW <= 4096: the element-gather path, e.g. the 128 -> 32 head).
static int conv2d_input_grad_any(const void *dy, int dy_bf16, const float *wt, int prepared, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                 int stride, int pad, void *dx, int dx_bf16, int nterm, void *stream)
{
    PSI_REQUIRE(dy && wt && dx && N > 0 && H > 0 && W > 0, "bad arguments");
    PSI_REQUIRE(Cin % 32 == 0 && (Cout % 64 == 0 || Cout * KH * KW <= 4096) && KH > 0 && KW > 0 && stride > 0 && pad >= 0,
                "shape not covered: Cin % 32 == 0 and (Cout % 64 == 0 or a small gather case)");
    PSI_REQUIRE(nterm == 1 || nterm == 3, "nterm is 1 or 3");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    // (the forward kernel of conv.hip on dY with the taps walked backwards: its "input channels" are Cout)
    if (conv3x3_split_route() && prepared && nterm == 3 && !dy_bf16 && !dx_bf16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && psi_conv3x3s_ok(N, H, W, Cout, Cin))
        return psi_conv3x3s_forward((const float *)dy, (const void *)wt, nullptr, N, H, W, Cout, Cin, (float *)dx, 1, st);
    // roles swapped: the kernel's "input" is dY (OH x OW x Cout), its "output" dX (H x W x Cin)
#define PSI_DG_ARGS dy, wt, nullptr, dx, N, OH, OW, Cout, H, W, Cin, KH, KW, stride, pad, 1, st, prepared
    if (nterm == 3) {
        if (dy_bf16) return dx_bf16 ? launch_bn<3, __bf16, __bf16>(PSI_DG_ARGS) : launch_bn<3, __bf16, float>(PSI_DG_ARGS);
        return dx_bf16 ? launch_bn<3, float, __bf16>(PSI_DG_ARGS) : launch_bn<3, float, float>(PSI_DG_ARGS);
    }
    if (dy_bf16) return dx_bf16 ? launch_bn<1, __bf16, __bf16>(PSI_DG_ARGS) : launch_bn<1, __bf16, float>(PSI_DG_ARGS);
    return dx_bf16 ? launch_bn<1, float, __bf16>(PSI_DG_ARGS) : launch_bn<1, float, float>(PSI_DG_ARGS);
#undef PSI_DG_ARGS
}

extern "C" int psi_conv2d_input_grad(const void *dy, int dy_bf16, const float *wt, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                     int pad, void *dx, int dx_bf16, int nterm, void *stream)
{
    return conv2d_input_grad_any(dy, dy_bf16, wt, 0, N, H, W, Cin, Cout, KH, KW, stride, pad, dx, dx_bf16, nterm, stream);
}

// The same with psi_conv2d_prepare_weight's `wt`.
extern "C" int psi_conv2d_input_grad_p(const void *dy, int dy_bf16, const void *wt, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                                       int pad, void *dx, int dx_bf16, int nterm, void *stream)
{
    return conv2d_input_grad_any(dy, dy_bf16, (const float *)wt, 1, N, H, W, Cin, Cout, KH, KW, stride, pad, dx, dx_bf16, nterm, stream);
}

extern "C" size_t psi_conv2d_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad)
{
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    if (OH <= 0 || OW <= 0) return 0;
    const int K = KH * KW * Cin;
    const size_t general = (size_t)wgrad_splits((long)N * OH * OW, K, Cout) * Cout * K;
    if (psi_conv_stem_shape(Cin, Cout, KH, KW, stride, pad)) return std::max(general, psi_conv_stem_wgrad_floats(N, H, W));
    if (KH == 3 && KW == 3 && stride == 1 && pad == 1) return std::max(general, psi_conv3x3_wrw3_workspace_floats(N, H, W, Cin, Cout));
    return general;
}

// Weight gradient: x [N,H,W,Cin], dy [N,OH,OW,Cout] (each fp32 or bf16) -> gw [Cout,KH,KW,Cin] fp32 (OVERWRITTEN; the memory of a
// channels_last Conv2d weight gradient).  ws: psi_conv2d_wgrad_workspace_floats floats.  Any shape psi_conv2d_supported accepts (Cout % 8 == 0).
extern "C" int psi_conv2d_weight_grad(const void *x, int x_bf16, const void *dy, int dy_bf16, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                                      int stride, int pad, float *gw, float *ws, int nterm, void *stream)
{
    PSI_REQUIRE(x && dy && gw && ws && N > 0 && H > 0 && W > 0, "bad arguments");
    PSI_REQUIRE(KH > 0 && KW > 0 && stride > 0 && pad >= 0 && Cout % 8 == 0 && (Cin % 8 == 0 || Cin * KH * KW <= 4096), "shape not covered");
    PSI_REQUIRE(nterm == 1 || nterm == 3, "nterm is 1 or 3");
    const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
    PSI_REQUIRE(OH > 0 && OW > 0, "empty output");
    hipStream_t st = (hipStream_t)stream;
    if (stem_route(Cin, Cout, KH, KW, stride, pad)) return psi_conv_stem_weight_grad(x, x_bf16, dy, dy_bf16, N, H, W, gw, ws, nterm, st);
    // the stride-1 3x3 layers of the fp32 model (8 of a trunk's 11 convolutions): both operand tiles split ONCE per workgroup for all nine taps
    // (conv.hip: conv3x3_wrw3_kernel; PSI_CONV3X3_SPLIT=0: the general kernel, dev A/B)
    if (conv3x3_split_route() && nterm == 3 && !x_bf16 && !dy_bf16 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && psi_conv3x3_wrw3_ok(N, H, W, Cin, Cout))
        return psi_conv3x3_weight_grad3((const float *)x, (const float *)dy, N, H, W, Cin, Cout, gw, ws, st);
    const int K = KH * KW * Cin;
    const int Keff = (Cin % 8) != 0 && Cin * KW <= 16 ? KH * 16 : K;      // (filter-row indexing of a small Cin: conv_wgrad_kernel mode 1)
    const long M = (long)N * OH * OW, nstage = (M + WG_PX - 1) / WG_PX;
    const int S = wgrad_splits(M, K, Cout);
    const int sps = (int)((nstage + S - 1) / S);
    dim3 grid((unsigned)((Keff + KC - 1) / KC), (unsigned)((Cout + 63) / 64), (unsigned)S);
#define PSI_WG_LAUNCH(NT_, TX_, TD_)                                                                                                         \
    hipLaunchKernelGGL((conv_wgrad_kernel<NT_, TX_, TD_>), grid, dim3(256), 0, st, (const TX_ *)x, (const TD_ *)dy, ws, N, H, W, Cin, OH, OW, Cout, \
                       KH, KW, stride, pad, S, sps)
    if (nterm == 3) {
        if (x_bf16 && dy_bf16) PSI_WG_LAUNCH(3, __bf16, __bf16);
        else if (x_bf16) PSI_WG_LAUNCH(3, __bf16, float);
        else if (dy_bf16) PSI_WG_LAUNCH(3, float, __bf16);
        else PSI_WG_LAUNCH(3, float, float);
    } else {
        if (x_bf16 && dy_bf16) PSI_WG_LAUNCH(1, __bf16, __bf16);
        else if (x_bf16) PSI_WG_LAUNCH(1, __bf16, float);
        else if (dy_bf16) PSI_WG_LAUNCH(1, float, __bf16);
        else PSI_WG_LAUNCH(1, float, float);
    }
#undef PSI_WG_LAUNCH
    PSI_CHECK_LAUNCH("conv_wgrad_kernel");
    psi_mark("conv_wgrad_kernel", st);
    const long n = (long)Cout * K;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, S, n, gw);
    PSI_CHECK_LAUNCH("conv_wgrad_reduce_kernel");
    return 0;
}
