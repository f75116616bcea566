// Batch normalisation of the scene trunk fused with what follows it — ReLU and the BasicBlock's skip connection — for NHWC bf16
// feature maps, forward (training statistics) and backward, gfx950.
//
// Replaces, inside torchvision's BasicBlock as the reference builds it (cvae.py:427-435 -> resnet18 children[1:6]; train_s1.py /
// train_s2.py run the trunk in .train() mode, so the statistics are the batch's):
//     out = relu(bn1(conv1(x)));  out = bn2(conv2(out));  out = relu(out + identity)          and the stem's relu(bn1(conv(x)))
// i.e.  y = act(gamma * (x - mean_c) / sqrt(var_c + eps) + beta (+ residual)),  act = ReLU or identity, with
// running_mean / running_var updated like nn.BatchNorm2d (momentum, unbiased variance) and num_batches_tracked += 1.
//
// These are HBM-bound passes over [M = N*H*W, C] tensors (C = 64 or 128 channels, channel-fastest), so what matters is how many
// times the feature map crosses HBM and that every access is a full 16-byte lane load: the library path (MIOpen spatial BN: three
// launches forward, three backward, plus separate ReLU / add / clamp launches) moves a 17 MB layer1 map in 25-31 us per pass;
// a plain bandwidth-bound pass is ~6 us.
//   forward : stats (read x) -> finalize (C values) -> apply (read x [+ residual], write y)
//   backward: reduce (read dy, y|x) -> finalize -> apply (read dy, x, y, write dx [, d_residual])
// The reductions are deterministic: fixed per-block partials, summed in order by the finalize kernels (16 channels per block).  The partials
// are fp32 sums of x and x^2; the finalize step combines them in double precision, which keeps the COMBINATION exact but not the partials:
// for a channel whose |mean| is far above its standard deviation (mean 10, variance 0.01) E[x^2] - mean^2 loses the digits the fp32 partials
// no longer hold and the variance can be off by about a percent, where the library's Welford pass is not.  Convolution outputs of this
// trunk have |mean| of the order of their standard deviation (the parity tests against the library path hold to bf16 rounding).
#include "psi_internal.h"
#include <hip/hip_bf16.h>

namespace {

typedef unsigned short bf16raw;
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f2bf(float f)           // round to nearest even (finite inputs; NaN stays NaN)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void unpack8(const u4 &v, float (&f)[8])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ u4 pack8(const float (&f)[8])
{
    u4 v;
#pragma unroll
    for (int i = 0; i < 4; i++) v[i] = (unsigned)f2bf(f[2 * i]) | ((unsigned)f2bf(f[2 * i + 1]) << 16);
    return v;
}

// Eight consecutive channels of a feature map of element type MT: bf16 (the bf16 mode of the trunk: one 16-byte access) or fp32 (the
// fp32 model, whose convolutions run as three-term split products — conv_gemm.hip: two 16-byte accesses).  The arithmetic of every
// kernel below is fp32 either way.
typedef float f4v __attribute__((ext_vector_type(4)));
template <typename MT> struct Map8;
template <> struct Map8<bf16raw> {
    typedef u4 raw;
    static __device__ __forceinline__ raw ld(const void *p, long i) { return ((const u4 *)p)[i]; }
    static __device__ __forceinline__ void st(void *p, long i, const raw &v) { ((u4 *)p)[i] = v; }
    static __device__ __forceinline__ void unpack(const raw &v, float (&f)[8]) { unpack8(v, f); }
    static __device__ __forceinline__ raw pack(const float (&f)[8]) { return pack8(f); }
};
template <> struct Map8<float> {
    struct raw { f4v a, b; };
    static __device__ __forceinline__ raw ld(const void *p, long i) { const f4v *q = (const f4v *)p + 2 * i; return raw{q[0], q[1]}; }
    static __device__ __forceinline__ void st(void *p, long i, const raw &v) { f4v *q = (f4v *)p + 2 * i; q[0] = v.a; q[1] = v.b; }
    static __device__ __forceinline__ void unpack(const raw &v, float (&f)[8])
    {
#pragma unroll
        for (int k = 0; k < 4; k++) { f[k] = v.a[k]; f[4 + k] = v.b[k]; }
    }
    static __device__ __forceinline__ raw pack(const float (&f)[8]) { return raw{f4v{f[0], f[1], f[2], f[3]}, f4v{f[4], f[5], f[6], f[7]}}; }
};

constexpr int BN_BLK = 256;
constexpr int BN_MAXC = 256;

// A block covers RPB = 256 / (C/8) rows per pass; thread (r, g): row r, channels 8g..8g+7.  NACC accumulators per channel per thread
// are reduced over the block's rows through LDS; result for channel c in out[blockIdx.x][q][c].
template <int NQ>
__device__ __forceinline__ void block_reduce_store(float (&acc)[NQ][8], int C, int cg, int rr, int rpb, float *__restrict__ out)
{
    __shared__ float sh[NQ][BN_BLK][8 + 1];
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < NQ; q++)
#pragma unroll
        for (int i = 0; i < 8; i++) sh[q][t][i] = acc[q][i];
    __syncthreads();
    // thread t < NQ * C sums channel c of quantity q over the rpb row-threads, in row order
    const int ngrp = C / 8;
    for (int o = t; o < NQ * C; o += BN_BLK) {
        const int q = o / C, c = o % C, g = c >> 3, i = c & 7;
        float s = 0.0f;
        for (int r = 0; r < rpb; r++) s += sh[q][r * ngrp + g][i];
        out[((size_t)blockIdx.x * NQ + q) * C + c] = s;
    }
}

template <typename MT>
__global__ __launch_bounds__(BN_BLK) void bn_stats_kernel(const void *__restrict__ x, long M, int C, float *__restrict__ part)
{
    typedef Map8<MT> V;
    const int ngrp = C / 8, rpb = BN_BLK / ngrp;
    const int g = threadIdx.x % ngrp, rr = threadIdx.x / ngrp;
    float acc[2][8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[0][i] = acc[1][i] = 0.0f;
    const long stride = (long)gridDim.x * rpb;
    long r = (long)blockIdx.x * rpb + rr;
    // four rows in flight per thread
    for (; r + 3 * stride < M; r += 4 * stride) {
        typename V::raw v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = V::ld(x, (r + u * stride) * ngrp + g);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float f[8];
            V::unpack(v[u], f);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                acc[0][i] += f[i];
                acc[1][i] += f[i] * f[i];
            }
        }
    }
    for (; r < M; r += stride) {
        float fa[8];
        V::unpack(V::ld(x, r * ngrp + g), fa);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            acc[0][i] += fa[i];
            acc[1][i] += fa[i] * fa[i];
        }
    }
    block_reduce_store<2>(acc, C, g, rr, rpb, part);
}

// Sum of the per-block partials.  A block of 1024 threads takes BN_FIN_CH channels (both quantities): output o = q * Cb + c is handled
// by 1024 / (2 Cb) threads, thread slice s adding partials s, s + nsl, ... with four independent accumulators, the slices are then
// combined in slice order: deterministic.  The kernel is pure latency (the partials were written by other compute units a moment ago,
// every dependent round of loads is a trip to L2 / memory): with 16 channels per block a thread's share of 512 partials is ONE round of
// 16 loads (one block for all channels needed four rounds at C = 64: 8.1 us per launch, 80 launches per train_s2 step).
constexpr int BN_FIN_CH = 16;
__host__ __device__ inline int bn_fin_blocks(int C) { return C > BN_FIN_CH ? C / BN_FIN_CH : 1; }

__device__ __forceinline__ void sum_partials(const float *__restrict__ part, int nblk, int C, double *sh /* [1024] */, double &s_out, double &q_out,
                                             int &c_out)
{
    const int Cb = C > BN_FIN_CH ? BN_FIN_CH : C, c_lo = blockIdx.x * Cb;
    const int t = threadIdx.x, no = 2 * Cb, nsl = 1024 / no;
    const int o = t % no, sl = t / no;
    const int col = (o / Cb) * C + c_lo + (o % Cb);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    {
        int b = sl;
        for (; b + 15 * nsl < nblk; b += 16 * nsl) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; u++) v[u] = part[(size_t)(b + u * nsl) * (2 * C) + col];
#pragma unroll
            for (int u = 0; u < 16; u += 4) {
                a0 += (double)v[u];
                a1 += (double)v[u + 1];
                a2 += (double)v[u + 2];
                a3 += (double)v[u + 3];
            }
        }
        for (; b < nblk; b += nsl) a0 += (double)part[(size_t)b * (2 * C) + col];
    }
    sh[t] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    s_out = q_out = 0.0;
    c_out = t < Cb ? c_lo + t : -1;
    if (t < Cb) {
        for (int k = 0; k < nsl; k++) {
            s_out += sh[k * no + t];
            q_out += sh[k * no + Cb + t];
        }
    }
}

// sums the per-block partials in order (16 channels per block); mean / invstd / scale / shift; running statistics (nn.BatchNorm2d semantics:
// running = (1 - momentum) * running + momentum * batch, the variance unbiased) and the batch counter
__global__ __launch_bounds__(1024) void bn_fwd_finalize_kernel(const float *__restrict__ part, int nblk, long M, int C, const float *__restrict__ gamma,
                                                                 const float *__restrict__ beta, float eps, float momentum,
                                                                 float *__restrict__ running_mean, float *__restrict__ running_var,
                                                                 long long *__restrict__ num_batches, float *__restrict__ save_mean,
                                                                 float *__restrict__ save_invstd, float *__restrict__ scale_shift)
{
    __shared__ double shd[1024];
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches) *num_batches += 1;
    double s, q;
    int c;
    sum_partials(part, nblk, C, shd, s, q, c);
    if (c < 0) return;
    const double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float sc = gamma[c] * invstd;
    scale_shift[c] = sc;
    scale_shift[C + c] = __builtin_fmaf(-(float)mean, sc, beta[c]);          // (explicit: the backward re-forms scale and shift, relu_open)
    if (running_mean) {
        const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

template <typename MT, bool RELU, bool RES>
__global__ __launch_bounds__(BN_BLK) void bn_fwd_apply_kernel(const void *__restrict__ x, const void *__restrict__ res, long n16, int C,
                                                              const float *__restrict__ scale_shift, void *__restrict__ y)
{
    typedef Map8<MT> V;
    __shared__ float ss[2 * BN_MAXC];
    for (int i = threadIdx.x; i < 2 * C; i += BN_BLK) ss[i] = scale_shift[i];
    __syncthreads();
    const int ngrp = C / 8;
    const bool pow2 = (ngrp & (ngrp - 1)) == 0;
    for (long i = (long)blockIdx.x * BN_BLK + threadIdx.x; i < n16; i += (long)gridDim.x * BN_BLK) {
        const int c0 = (pow2 ? (int)(i & (ngrp - 1)) : (int)(i % ngrp)) * 8;
        const typename V::raw xv = V::ld(x, i);
        typename V::raw rv = xv;
        if (RES) rv = V::ld(res, i);
        float f[8], r[8];
        V::unpack(xv, f);
        if (RES) V::unpack(rv, r);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            float v = __builtin_fmaf(f[k], ss[c0 + k], ss[C + c0 + k]);      // (explicit: the backward recomputes this value for its ReLU mask)
            if (RES) v += r[k];
            if (RELU) v = v > 0.0f ? v : 0.0f;
            f[k] = v;
        }
        V::st(y, i, V::pack(f));
    }
}

// ReLU mask of a BatchNorm WITHOUT a skip connection, recomputed from x: the forward stored y = relu(fma(x, scale, shift)) (rounded to bf16 on
// bf16 maps), so "y > 0" is a function of x and two per-channel numbers — the backward passes then read two maps (dy, x) instead of three
// and three (dy, x -> dx) instead of four: y never crosses HBM again.  scale = gamma * invstd, shift = beta - mean * scale exactly as
// fwd_finalize_channel forms them.
template <typename MT> __device__ __forceinline__ bool relu_open(float x, float scale, float shift)
{
    const float v = __builtin_fmaf(x, scale, shift);
    if (sizeof(MT) == 2) return bf2f(f2bf(v > 0.0f ? v : 0.0f)) > 0.0f;      // what the forward's bf16 store kept of it
    return v > 0.0f;
}

// backward, pass 1: dz = dy * (y > 0) [ReLU] ; partial sums of dz and dz * xhat per channel, xhat = (x - mean) * invstd.
// y == NULL (RELU, no skip connection, beta given): the mask is recomputed from x.
template <typename MT, bool RELU>
__global__ __launch_bounds__(BN_BLK) void bn_bwd_reduce_kernel(const void *__restrict__ dy, const void *__restrict__ x, const void *__restrict__ y,
                                                               long M, int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                               const float *__restrict__ gamma, const float *__restrict__ beta, float *__restrict__ part)
{
    typedef Map8<MT> V;
    typedef typename V::raw R8;
    const int ngrp = C / 8, rpb = BN_BLK / ngrp;
    const int g = threadIdx.x % ngrp, rr = threadIdx.x / ngrp;
    const bool xmask = RELU && y == nullptr;
    float mu[8], is[8], fs[8], fh[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        mu[i] = mean[g * 8 + i];
        is[i] = invstd[g * 8 + i];
        fs[i] = fh[i] = 0.0f;
        if (xmask) {
            fs[i] = gamma[g * 8 + i] * is[i];
            fh[i] = __builtin_fmaf(-mu[i], fs[i], beta[g * 8 + i]);
        }
    }
    float acc[2][8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[0][i] = acc[1][i] = 0.0f;
    const long stride = (long)gridDim.x * rpb;
    auto row = [&](const R8 &dv, const R8 &xv, const R8 &yv) {
        float d[8], xf[8], yf[8];
        V::unpack(dv, d);
        V::unpack(xv, xf);
        if (RELU && !xmask) V::unpack(yv, yf);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool open = !RELU || (xmask ? relu_open<MT>(xf[i], fs[i], fh[i]) : yf[i] > 0.0f);
            const float dz = open ? d[i] : 0.0f;
            acc[0][i] += dz;
            acc[1][i] += dz * ((xf[i] - mu[i]) * is[i]);
        }
    };
    long r = (long)blockIdx.x * rpb + rr;
    for (; r + stride < M; r += 2 * stride) {               // two rows (six 16-byte loads) in flight per thread
        const long i0 = r * ngrp + g, i1 = (r + stride) * ngrp + g;
        const R8 d0 = V::ld(dy, i0), x0 = V::ld(x, i0), d1 = V::ld(dy, i1), x1 = V::ld(x, i1);
        R8 y0 = d0, y1 = d1;
        if (RELU && !xmask) { y0 = V::ld(y, i0); y1 = V::ld(y, i1); }
        row(d0, x0, y0);
        row(d1, x1, y1);
    }
    if (r < M) {
        const long i0 = r * ngrp + g;
        const R8 d0 = V::ld(dy, i0), x0 = V::ld(x, i0);
        R8 y0 = d0;
        if (RELU && !xmask) y0 = V::ld(y, i0);
        row(d0, x0, y0);
    }
    block_reduce_store<2>(acc, C, g, rr, rpb, part);
}

// one block: dbeta = sum dz, dgamma = sum dz * xhat; coefficients of pass 2:  dx = a * dz + b * xhat + c  with
//   a = gamma * invstd,  b = -a * dgamma / M,  c = -a * dbeta / M     (batch-statistics BN backward)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float *__restrict__ part, int nblk, long M, int C, const float *__restrict__ gamma,
                                                                 const float *__restrict__ invstd, float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                 float *__restrict__ coef)
{
    __shared__ double shd[1024];
    double s, q;
    int c;
    sum_partials(part, nblk, C, shd, s, q, c);
    if (c < 0) return;
    if (dbeta) dbeta[c] = (float)s;
    if (dgamma) dgamma[c] = (float)q;
    const float a = gamma[c] * invstd[c];
    coef[c] = a;
    coef[C + c] = (float)(-(double)a * q / (double)M);
    coef[2 * C + c] = (float)(-(double)a * s / (double)M);
}

template <typename MT, bool RELU, bool RES>
__global__ __launch_bounds__(BN_BLK) void bn_bwd_apply_kernel(const void *__restrict__ dy, const void *__restrict__ x, const void *__restrict__ y, long n16,
                                                              int C, const float *__restrict__ mean, const float *__restrict__ invstd,
                                                              const float *__restrict__ coef, const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, void *__restrict__ dx, void *__restrict__ dres)
{
    typedef Map8<MT> V;
    __shared__ float sc[7 * BN_MAXC];
    const bool xmask = RELU && y == nullptr;                      // (see bn_bwd_reduce_kernel)
    for (int i = threadIdx.x; i < C; i += BN_BLK) {
        sc[i] = coef[i];
        sc[C + i] = coef[C + i];
        sc[2 * C + i] = coef[2 * C + i];
        sc[3 * C + i] = mean[i];
        sc[4 * C + i] = invstd[i];
        if (xmask) {
            const float fs = gamma[i] * invstd[i];
            sc[5 * C + i] = fs;
            sc[6 * C + i] = __builtin_fmaf(-mean[i], fs, beta[i]);
        }
    }
    __syncthreads();
    const int ngrp = C / 8;
    const bool pow2 = (ngrp & (ngrp - 1)) == 0;
    for (long i = (long)blockIdx.x * BN_BLK + threadIdx.x; i < n16; i += (long)gridDim.x * BN_BLK) {
        const int c0 = (pow2 ? (int)(i & (ngrp - 1)) : (int)(i % ngrp)) * 8;
        const typename V::raw dv = V::ld(dy, i), xv = V::ld(x, i);
        typename V::raw yv = dv;
        if (RELU && !xmask) yv = V::ld(y, i);
        float d[8], xf[8], yf[8], o[8];
        V::unpack(dv, d);
        V::unpack(xv, xf);
        if (RELU && !xmask) V::unpack(yv, yf);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const bool open = !RELU || (xmask ? relu_open<MT>(xf[k], sc[5 * C + c0 + k], sc[6 * C + c0 + k]) : yf[k] > 0.0f);
            const float dz = open ? d[k] : 0.0f;
            d[k] = dz;
            const float xhat = (xf[k] - sc[3 * C + c0 + k]) * sc[4 * C + c0 + k];
            o[k] = sc[c0 + k] * dz + sc[C + c0 + k] * xhat + sc[2 * C + c0 + k];
        }
        V::st(dx, i, V::pack(o));
        if (RES) V::st(dres, i, V::pack(d));
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1) of the stem (torchvision resnet18 children[3], cvae.py:431-435) on NHWC bf16 maps, with the
// window position of the maximum kept as one byte per output so that the backward is a GATHER (every input looks at the <= 4 windows
// that contain it): no atomics, deterministic, one coalesced pass.  First maximum in (kh, kw) scan order wins (strict >), like
// at::max_pool2d_with_indices.  The library's NHWC backward takes 107 us for the stem's 67 MB map; this pass is bandwidth-bound.
// ------------------------------------------------------------------------------------------------
template <typename MT>
__global__ __launch_bounds__(BN_BLK) void maxpool_fwd_kernel(const void *__restrict__ x, int N, int H, int W, int C, int OH, int OW,
                                                             void *__restrict__ y, unsigned long long *__restrict__ idx)
{
    typedef Map8<MT> V;
    const int ngrp = C / 8;
    const long total = (long)N * OH * OW * ngrp;
    for (long i = (long)blockIdx.x * BN_BLK + threadIdx.x; i < total; i += (long)gridDim.x * BN_BLK) {
        const int g = (int)(i % ngrp);
        long t = i / ngrp;
        const int ow = (int)(t % OW);
        t /= OW;
        const int oh = (int)(t % OH), n = (int)(t / OH);
        float best[8];
        unsigned char bi[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { best[k] = -INFINITY; bi[k] = 0; }
        bool first = true;
#pragma unroll
        for (int kh = 0; kh < 3; kh++)
#pragma unroll
            for (int kw = 0; kw < 3; kw++) {
                const int ih = oh * 2 - 1 + kh, iw = ow * 2 - 1 + kw;
                if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
                float f[8];
                V::unpack(V::ld(x, (((long)n * H + ih) * W + iw) * ngrp + g), f);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if (first || f[k] > best[k] || f[k] != f[k]) { best[k] = f[k]; bi[k] = (unsigned char)(kh * 3 + kw); }
                first = false;
            }
        V::st(y, i, V::pack(best));
        unsigned long long pk = 0;
        if (!idx) continue;                                  // (inference: nobody will ask for the backward)
#pragma unroll
        for (int k = 0; k < 8; k++) pk |= (unsigned long long)bi[k] << (8 * k);
        idx[i] = pk;
    }
}

template <typename MT>
__global__ __launch_bounds__(BN_BLK) void maxpool_bwd_kernel(const void *__restrict__ dy, const unsigned long long *__restrict__ idx, int N, int H,
                                                             int W, int C, int OH, int OW, void *__restrict__ dx)
{
    typedef Map8<MT> V;
    const int ngrp = C / 8;
    const long total = (long)N * H * W * ngrp;
    for (long i = (long)blockIdx.x * BN_BLK + threadIdx.x; i < total; i += (long)gridDim.x * BN_BLK) {
        const int g = (int)(i % ngrp);
        long t = i / ngrp;
        const int iw = (int)(t % W);
        t /= W;
        const int ih = (int)(t % H), n = (int)(t / H);
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = 0.0f;
        // windows that contain (ih, iw): oh with oh*2 - 1 <= ih <= oh*2 + 1
        const int oh0 = ih >> 1, ow0 = iw >> 1;                 // and oh0 + 1 / ow0 + 1 when the coordinate is odd
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) {
                const int oh = oh0 + a, ow = ow0 + b;
                if ((a && !(ih & 1)) || (b && !(iw & 1)) || oh >= OH || ow >= OW) continue;
                const int pos = (ih - (oh * 2 - 1)) * 3 + (iw - (ow * 2 - 1));
                const long o = (((long)n * OH + oh) * OW + ow) * ngrp + g;
                const unsigned long long pk = idx[o];
                float d[8];
                V::unpack(V::ld(dy, o), d);
#pragma unroll
                for (int k = 0; k < 8; k++)
                    if ((int)((pk >> (8 * k)) & 0xff) == pos) acc[k] += d[k];
            }
        V::st(dx, i, V::pack(acc));
    }
}

// eval mode (generation: cvae.py TestOP runs the encoders in .eval()): y = gamma (x - running_mean) / sqrt(running_var + eps) + beta
__global__ void bn_eval_coef_kernel(const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ rmean,
                                    const float *__restrict__ rvar, float eps, int C, float *__restrict__ scale_shift)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);           // (the operator order of at::batch_norm's inference path)
    scale_shift[c] = sc;
    scale_shift[C + c] = beta[c] - rmean[c] * sc;
}

int bn_blocks(long M, int C)
{
    const int rpb = BN_BLK / (C / 8);
    long need = (M + rpb - 1) / rpb;
    long nb = need < 512 ? need : 512;            // two blocks per CU, four rows in flight per thread: 32 KB of loads in flight per CU
    return (int)(nb < 1 ? 1 : nb);
}

}  // namespace

extern "C" size_t psi_bn_workspace_floats(long M, int C) { return (size_t)bn_blocks(M, C) * 2 * C + 5 * (size_t)C + 64; }

template <typename MT>
static int bn_forward_t(const void *x, const void *residual, const float *gamma, const float *beta, float *running_mean, float *running_var,
                        long long *num_batches_tracked, long M, int C, int relu, float momentum, float eps, void *y, float *save_mean,
                        float *save_invstd, float *ws, int eval_mode, hipStream_t st)
{
    PSI_REQUIRE(x && gamma && beta && y && ws, "null pointer");
    PSI_REQUIRE(eval_mode ? (running_mean && running_var) : (save_mean && save_invstd), "null pointer (statistics)");
    PSI_REQUIRE(M > 0 && C >= 8 && C <= BN_MAXC && C % 8 == 0 && BN_BLK % (C / 8) == 0 && 1024 % (2 * C) == 0, "C must be 8, 16, 32, 64, 128 or 256");
    const int nb = bn_blocks(M, C);
    float *part = ws, *ss = ws + (size_t)nb * 2 * C;
    if (eval_mode) {
        hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(psi_cdiv(C, 256)), dim3(256), 0, st, gamma, beta, running_mean, running_var, eps, C, ss);
        PSI_CHECK_LAUNCH("bn_eval_coef_kernel");
    } else {
        hipLaunchKernelGGL(bn_stats_kernel<MT>, dim3(nb), dim3(BN_BLK), 0, st, x, M, C, part);
        PSI_CHECK_LAUNCH("bn_stats_kernel");
        psi_mark("bn_stats_kernel", st);
        hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3(bn_fin_blocks(C)), dim3(1024), 0, st, part, nb, M, C, gamma, beta, eps, momentum, running_mean,
                           running_var, num_batches_tracked, save_mean, save_invstd, ss);
        PSI_CHECK_LAUNCH("bn_fwd_finalize_kernel");
    }
    const long n16 = M * (C / 8);
    const int ga = (int)((n16 + BN_BLK - 1) / BN_BLK < 2048 ? (n16 + BN_BLK - 1) / BN_BLK : 2048);
#define PSI_BN_APPLY(R_, S_) hipLaunchKernelGGL((bn_fwd_apply_kernel<MT, R_, S_>), dim3(ga), dim3(BN_BLK), 0, st, x, residual, n16, C, ss, y)
    if (relu && residual) PSI_BN_APPLY(true, true);
    else if (relu) PSI_BN_APPLY(true, false);
    else if (residual) PSI_BN_APPLY(false, true);
    else PSI_BN_APPLY(false, false);
#undef PSI_BN_APPLY
    PSI_CHECK_LAUNCH("bn_fwd_apply_kernel");
    psi_mark("bn_fwd_apply_kernel", st);
    return 0;
}

template <typename MT>
static int bn_backward_t(const void *dy, const void *x, const void *y, const float *gamma, const float *beta, const float *save_mean,
                         const float *save_invstd, long M, int C, int relu, void *dx, void *dresidual, float *dgamma, float *dbeta, float *ws,
                         hipStream_t st)
{
    PSI_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && ws, "null pointer");
    PSI_REQUIRE(!relu || y || (beta && !dresidual), "the ReLU mask needs the forward output (or beta, when the layer had no skip connection)");
    PSI_REQUIRE(M > 0 && C >= 8 && C <= BN_MAXC && C % 8 == 0 && BN_BLK % (C / 8) == 0 && 1024 % (2 * C) == 0, "C must be 8, 16, 32, 64, 128 or 256");
    const int nb = bn_blocks(M, C);
    float *part = ws, *coef = ws + (size_t)nb * 2 * C;
    if (relu)
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<MT, true>), dim3(nb), dim3(BN_BLK), 0, st, dy, x, y, M, C, save_mean, save_invstd, gamma, beta, part);
    else
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<MT, false>), dim3(nb), dim3(BN_BLK), 0, st, dy, x, y, M, C, save_mean, save_invstd, gamma, beta, part);
    PSI_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    psi_mark("bn_bwd_reduce_kernel", st);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(bn_fin_blocks(C)), dim3(1024), 0, st, part, nb, M, C, gamma, save_invstd, dgamma, dbeta, coef);
    PSI_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    const long n16 = M * (C / 8);
    const int ga = (int)((n16 + BN_BLK - 1) / BN_BLK < 2048 ? (n16 + BN_BLK - 1) / BN_BLK : 2048);
#define PSI_BN_BAPPLY(R_, S_) hipLaunchKernelGGL((bn_bwd_apply_kernel<MT, R_, S_>), dim3(ga), dim3(BN_BLK), 0, st, dy, x, y, n16, C, save_mean, save_invstd, coef, gamma, beta, dx, dresidual)
    if (relu && dresidual) PSI_BN_BAPPLY(true, true);
    else if (relu) PSI_BN_BAPPLY(true, false);
    else if (dresidual) PSI_BN_BAPPLY(false, true);
    else PSI_BN_BAPPLY(false, false);
#undef PSI_BN_BAPPLY
    PSI_CHECK_LAUNCH("bn_bwd_apply_kernel");
    psi_mark("bn_bwd_apply_kernel", st);
    return 0;
}

template <typename MT>
static int maxpool_forward_t(const void *x, int N, int H, int W, int C, void *y, void *idx, hipStream_t st)
{
    PSI_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0, "bad arguments (C must be a multiple of 8)");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * OH * OW * (C / 8);
    const int grid = (int)((total + BN_BLK - 1) / BN_BLK < 4096 ? (total + BN_BLK - 1) / BN_BLK : 4096);
    hipLaunchKernelGGL(maxpool_fwd_kernel<MT>, dim3(grid), dim3(BN_BLK), 0, st, x, N, H, W, C, OH, OW, y, (unsigned long long *)idx);
    PSI_CHECK_LAUNCH("maxpool_fwd_kernel");
    psi_mark("maxpool_fwd_kernel", st);
    return 0;
}

template <typename MT>
static int maxpool_backward_t(const void *dy, const void *idx, int N, int H, int W, int C, void *dx, hipStream_t st)
{
    PSI_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0, "bad arguments (C must be a multiple of 8)");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * H * W * (C / 8);
    const int grid = (int)((total + BN_BLK - 1) / BN_BLK < 4096 ? (total + BN_BLK - 1) / BN_BLK : 4096);
    hipLaunchKernelGGL(maxpool_bwd_kernel<MT>, dim3(grid), dim3(BN_BLK), 0, st, dy, (const unsigned long long *)idx, N, H, W, C, OH, OW, dx);
    PSI_CHECK_LAUNCH("maxpool_bwd_kernel");
    psi_mark("maxpool_bwd_kernel", st);
    return 0;
}

// bf16 maps (the bf16 mode of the trunk)
extern "C" int psi_bn_forward(const void *x, const void *residual, const float *gamma, const float *beta, float *running_mean,
                              float *running_var, long long *num_batches_tracked, long M, int C, int relu, float momentum, float eps,
                              void *y, float *save_mean, float *save_invstd, float *ws, void *stream)
{
    return bn_forward_t<bf16raw>(x, residual, gamma, beta, running_mean, running_var, num_batches_tracked, M, C, relu, momentum, eps, y, save_mean,
                                 save_invstd, ws, 0, (hipStream_t)stream);
}

extern "C" int psi_bn_backward(const void *dy, const void *x, const void *y, const float *gamma, const float *save_mean,
                               const float *save_invstd, long M, int C, int relu, void *dx, void *dresidual, float *dgamma, float *dbeta,
                               float *ws, void *stream)
{
    return bn_backward_t<bf16raw>(dy, x, y, gamma, nullptr, save_mean, save_invstd, M, C, relu, dx, dresidual, dgamma, dbeta, ws, (hipStream_t)stream);
}

extern "C" int psi_maxpool3x3s2_forward(const void *x, int N, int H, int W, int C, void *y, void *idx, void *stream)
{
    PSI_REQUIRE(idx, "null pointer");
    return maxpool_forward_t<bf16raw>(x, N, H, W, C, y, idx, (hipStream_t)stream);
}

extern "C" int psi_maxpool3x3s2_backward(const void *dy, const void *idx, int N, int H, int W, int C, void *dx, void *stream)
{
    return maxpool_backward_t<bf16raw>(dy, idx, N, H, W, C, dx, (hipStream_t)stream);
}

// The same operators on maps of either element type (map_f32 != 0: fp32 NHWC maps — the fp32 model), and with the inference form of the
// normalisation (eval_mode != 0: running statistics, nothing is updated; save_mean / save_invstd may be NULL).  idx == NULL in
// psi_maxpool3x3s2_forward_t: inference, no window positions are kept.
extern "C" int psi_bn_forward_t(const void *x, int map_f32, const void *residual, const float *gamma, const float *beta, float *running_mean,
                                float *running_var, long long *num_batches_tracked, long M, int C, int relu, float momentum, float eps, void *y,
                                float *save_mean, float *save_invstd, float *ws, int eval_mode, void *stream)
{
    if (map_f32)
        return bn_forward_t<float>(x, residual, gamma, beta, running_mean, running_var, num_batches_tracked, M, C, relu, momentum, eps, y, save_mean,
                                   save_invstd, ws, eval_mode, (hipStream_t)stream);
    return bn_forward_t<bf16raw>(x, residual, gamma, beta, running_mean, running_var, num_batches_tracked, M, C, relu, momentum, eps, y, save_mean,
                                 save_invstd, ws, eval_mode, (hipStream_t)stream);
}

extern "C" int psi_bn_backward_t(const void *dy, int map_f32, const void *x, const void *y, const float *gamma, const float *beta,
                                 const float *save_mean, const float *save_invstd, long M, int C, int relu, void *dx, void *dresidual, float *dgamma,
                                 float *dbeta, float *ws, void *stream)
{
    if (map_f32)
        return bn_backward_t<float>(dy, x, y, gamma, beta, save_mean, save_invstd, M, C, relu, dx, dresidual, dgamma, dbeta, ws, (hipStream_t)stream);
    return bn_backward_t<bf16raw>(dy, x, y, gamma, beta, save_mean, save_invstd, M, C, relu, dx, dresidual, dgamma, dbeta, ws, (hipStream_t)stream);
}

extern "C" int psi_maxpool3x3s2_forward_t(const void *x, int map_f32, int N, int H, int W, int C, void *y, void *idx, void *stream)
{
    if (map_f32) return maxpool_forward_t<float>(x, N, H, W, C, y, idx, (hipStream_t)stream);
    return maxpool_forward_t<bf16raw>(x, N, H, W, C, y, idx, (hipStream_t)stream);
}

extern "C" int psi_maxpool3x3s2_backward_t(const void *dy, int map_f32, const void *idx, int N, int H, int W, int C, void *dx, void *stream)
{
    if (map_f32) return maxpool_backward_t<float>(dy, idx, N, H, W, C, dx, (hipStream_t)stream);
    return maxpool_backward_t<bf16raw>(dy, idx, N, H, W, C, dx, (hipStream_t)stream);
}
