// Dense layers of the CVAEs (source/net_layers.py:28-43 ResBlock = Linear-LeakyReLU-Linear-LeakyReLU + skip; the scene-feature
// `fc` layers 8192 -> 256 and 32768 -> 256, cvae.py:436-440 / net_layers.py:66,164; the other nn.Linear layers of cvae.py:474-492)
// as hand-written bf16 MFMA kernels for gfx950: v_mfma_f32_32x32x16_bf16, operands rounded to bf16 (RNE) ON LOAD from the fp32
// master weights / activations, fp32 accumulate, fp32 output with bias + LeakyReLU (+ residual) fused into the epilogue.
//
// What this replaces on the PyTorch bf16-autocast path: per layer a cast kernel for the weight (fp32 read + bf16 write), a cast
// kernel for the input, the hipBLASLt GEMM, an elementwise bias/LeakyReLU and an add for the skip connection — five launches
// and two extra passes over the weight; here one launch reads the fp32 weight exactly once.  These GEMMs have M = batch = 128
// rows, so they are bound by the weight stream (HBM/L2 bytes) and by launch latency, not by the matrix pipe: the 33.5 MB weight
// of the 32768 -> 256 layer is the only one large enough to need the whole chip (split-K over 256 workgroups).
//
// Operand layouts of v_mfma_f32_32x32x16_bf16 (wave64): A lane l holds A[i = l%32][k = 8*(l/32) .. +8], B lane l holds
// B[k = 8*(l/32) .. +8][j = l%32]; D lane l holds D[8*(r/4) + 4*(l/32) + r%4][j = l%32] for r = 0..15.
//   forward   y = x W^T  : A = x rows (8 consecutive k: two 16-byte loads), B[k][n] = W[n][k] (8 consecutive k of row n: same)
//   dX = G W             : A = G rows (8 consecutive n), B[n][k] = W[n][k] (8 rows n, column k: eight 4-byte loads, coalesced over k)
//   dW = G^T X           : A[n][m] = G[m][n], B[m][k] = X[m][k] (both eight 4-byte loads per lane, coalesced over n resp. k)
#include "psi_internal.h"
#include <math.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

constexpr int TM = 32, TN = 32, TK = 16;

__device__ __forceinline__ float act_fwd(float v, int act, float slope) { return act == 1 ? (v > 0.0f ? v : v * slope) : v; }

// eight consecutive elements of a row as bf16 (fp32 source: two 16-byte loads + RNE; bf16 source: one 16-byte load)
__device__ __forceinline__ bf16x8 load8(const float *p, bool ok)
{
    bf16x8 r;
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = (__bf16)0.0f;
        return r;
    }
    const f4 a = *(const f4 *)p, b = *(const f4 *)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) { r[i] = (__bf16)a[i]; r[i + 4] = (__bf16)b[i]; }
    return r;
}
__device__ __forceinline__ bf16x8 load8(const __bf16 *p, bool ok)
{
    if (!ok) {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = (__bf16)0.0f;
        return r;
    }
    return *(const bf16x8 *)p;
}
__device__ __forceinline__ float ldf(const float *p) { return *p; }
__device__ __forceinline__ float ldf(const __bf16 *p) { return (float)*p; }

// ------------------------------------------------------------------------------------------------
// forward: workgroup = 4 waves = (up to) 128 rows x 32 columns x one K chunk; wave w owns rows [32w, 32w+32) of the row block.
// Tiles of 64 k go through LDS: the global loads are coalesced along k (16 lanes cover 256 contiguous bytes of one row; a lane
// reading "its own row" touched a different cache line per lane and ran 4-5x slower), values are rounded to bf16 on the way into
// LDS, the MFMA operands are 16-byte LDS reads (row pitch 72 elements: conflict-free), and the next tile's global loads are in
// flight while the current tile is multiplied (register-staged double buffer, two LDS buffers, one barrier per tile).
// ksplit == 1: bias / activation / residual epilogue here; ksplit > 1: raw partial sums to part[ks][M][N] (linear_reduce_kernel).
// ------------------------------------------------------------------------------------------------
constexpr int BK = 64, PITCH = BK + 8;

template <typename XT> struct XTile;
template <> struct XTile<float> {                           // 128 x 64 fp32: 8 float4 per thread
    f4 r[8];
    __device__ __forceinline__ void load(const float *x, int M, int K, int mblk, int k0, int k_end)
    {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            const bool ok = mblk + row < M && k0 + c < k_end;
            r[i] = ok ? *(const f4 *)(x + (size_t)(mblk + row) * K + k0 + c) : (f4){0, 0, 0, 0};
        }
    }
    __device__ __forceinline__ void store(__bf16 (*As)[PITCH]) const
    {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            bf4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (__bf16)r[i][e];
            *(bf4 *)&As[row][c] = v;
        }
    }
};
template <> struct XTile<__bf16> {                          // 128 x 64 bf16: 4 x 16 bytes per thread
    bf16x8 r[4];
    __device__ __forceinline__ void load(const __bf16 *x, int M, int K, int mblk, int k0, int k_end)
    {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 3, c = (idx & 7) * 8;
            const bool ok = mblk + row < M && k0 + c < k_end;
            if (ok) r[i] = *(const bf16x8 *)(x + (size_t)(mblk + row) * K + k0 + c);
            else {
#pragma unroll
                for (int e = 0; e < 8; e++) r[i][e] = (__bf16)0.0f;
            }
        }
    }
    __device__ __forceinline__ void store(__bf16 (*As)[PITCH]) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 3, c = (idx & 7) * 8;
            *(bf16x8 *)&As[row][c] = r[i];
        }
    }
};

template <typename XT>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const XT *__restrict__ x, const float *__restrict__ W, const float *__restrict__ bias,
                                                         const float *__restrict__ residual, int M, int N, int K, int kchunk, int act,
                                                         float slope, float *__restrict__ y, float *__restrict__ act_out,
                                                         float *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) __bf16 As[2][128][PITCH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][TN][PITCH];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n0 = blockIdx.x * TN, mblk = blockIdx.y * 128, m0 = mblk + w * TM, ks = blockIdx.z;
    const int k_begin = ks * kchunk, k_end = min(K, k_begin + kchunk);
    const int li = lane & 31, kb = (lane >> 5) * 8;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    XTile<XT> xa;
    f4 wb[2];
    auto load_w = [&](int k0) {                             // 32 x 64 fp32: 2 float4 per thread
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            const bool ok = n0 + row < N && k0 + c < k_end;
            wb[i] = ok ? *(const f4 *)(W + (size_t)(n0 + row) * K + k0 + c) : (f4){0, 0, 0, 0};
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            bf4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (__bf16)wb[i][e];
            *(bf4 *)&Bs[buf][row][c] = v;
        }
    };
    xa.load(x, M, K, mblk, k_begin, k_end);
    load_w(k_begin);
    xa.store(As[0]);
    store_w(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < k_end;
        if (more) {                                         // next tile: global loads in flight during this tile's MFMAs
            xa.load(x, M, K, mblk, k0 + BK, k_end);
            load_w(k0 + BK);
        }
#pragma unroll
        for (int st = 0; st < BK / TK; st++) {
            const bf16x8 a = *(const bf16x8 *)&As[buf][w * TM + li][st * TK + kb];
            const bf16x8 b = *(const bf16x8 *)&Bs[buf][li][st * TK + kb];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (more) {
            xa.store(As[buf ^ 1]);
            store_w(buf ^ 1);
        }
        __syncthreads();
    }
    if (m0 >= M) return;
    const int j = n0 + li;
    if (j >= N) return;
    const float bj = (bias && !part) ? bias[j] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = m0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (i >= M) continue;
        const size_t o = (size_t)i * N + j;
        if (part) {
            part[((size_t)ks * M + i) * N + j] = acc[r];
        } else {
            float v = act_fwd(acc[r] + bj, act, slope);
            if (act_out) act_out[o] = v;
            y[o] = residual ? v + residual[o] : v;
        }
    }
}

__global__ __launch_bounds__(256) void linear_reduce_kernel(const float *__restrict__ part, int S, const float *__restrict__ bias,
                                                            const float *__restrict__ residual, int M, int N, int act, float slope,
                                                            float *__restrict__ y, float *__restrict__ act_out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, MN = (size_t)M * N;
    if (i >= MN) return;
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int s = 0;
    for (; s + 3 < S; s += 4) {                                  // fixed order: deterministic
        a0 += part[(size_t)s * MN + i];
        a1 += part[(size_t)(s + 1) * MN + i];
        a2 += part[(size_t)(s + 2) * MN + i];
        a3 += part[(size_t)(s + 3) * MN + i];
    }
    for (; s < S; s++) a0 += part[(size_t)s * MN + i];
    float v = (a0 + a1) + (a2 + a3);
    if (bias) v += bias[i % N];
    v = act_fwd(v, act, slope);
    if (act_out) act_out[i] = v;
    y[i] = residual ? v + residual[i] : v;
}

// G = gy * act'(pre): the activation output has the sign of the pre-activation (slope > 0)
__device__ __forceinline__ float gmask(float g, const float *act_out, size_t o, float slope)
{
    return act_out ? (act_out[o] > 0.0f ? g : g * slope) : g;
}

// ------------------------------------------------------------------------------------------------
// dX[M,K] = G[M,N] W[N,K]: workgroup = 128 rows x 32 columns of K, contraction over N
// ------------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float *__restrict__ gy, const float *__restrict__ act_out,
                                                            const float *__restrict__ W, int M, int N, int K, float slope,
                                                            XT *__restrict__ gx)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k0 = blockIdx.x * TN, m0 = blockIdx.y * 128 + w * TM;
    const int li = lane & 31, nb = (lane >> 5) * 8;
    const int row = m0 + li, col = k0 + li;
    const bool rok = row < M, cok = col < K;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    constexpr int U = 4;
    for (int n = 0; n < N; n += U * TK) {                       // N % 16 == 0 (host); U steps' loads in flight together
        f4 g0[U], g1[U], m0v[U], m1v[U];
        float wv[U][8];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int nn = n + u * TK;
            const bool in = nn < N;
            const size_t go = (size_t)(rok ? row : 0) * N + (in ? nn : 0) + nb;
            const bool ld = rok && in;
            g0[u] = ld ? *(const f4 *)(gy + go) : (f4){0, 0, 0, 0};
            g1[u] = ld ? *(const f4 *)(gy + go + 4) : (f4){0, 0, 0, 0};
            if (act_out) {
                m0v[u] = ld ? *(const f4 *)(act_out + go) : (f4){1, 1, 1, 1};
                m1v[u] = ld ? *(const f4 *)(act_out + go + 4) : (f4){1, 1, 1, 1};
            }
#pragma unroll
            for (int i = 0; i < 8; i++) wv[u][i] = (cok && in) ? W[(size_t)(nn + nb + i) * K + col] : 0.0f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            bf16x8 a, b;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                a[i] = (__bf16)(act_out ? (m0v[u][i] > 0.0f ? g0[u][i] : g0[u][i] * slope) : g0[u][i]);
                a[i + 4] = (__bf16)(act_out ? (m1v[u][i] > 0.0f ? g1[u][i] : g1[u][i] * slope) : g1[u][i]);
            }
#pragma unroll
            for (int i = 0; i < 8; i++) b[i] = (__bf16)wv[u][i];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    if (m0 >= M || col >= K) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = m0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (i < M) gx[(size_t)i * K + col] = (XT)acc[r];
    }
}

// ------------------------------------------------------------------------------------------------
// dW[N,K] = G^T[N,M] X[M,K]: wave = one 32 x 32 tile of dW, contraction over M; workgroup = 4 adjacent K tiles of one N tile.
// The workgroups with blockIdx.x == 0 also produce gbias[n] = sum_m G[m][n] for their N tile.
// ------------------------------------------------------------------------------------------------
template <typename XT>
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float *__restrict__ gy, const float *__restrict__ act_out,
                                                            const XT *__restrict__ x, int M, int N, int K, float slope,
                                                            float *__restrict__ gW, float *__restrict__ gbias)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n0 = blockIdx.y * TM, k0 = (blockIdx.x * 4 + w) * TN;
    const int li = lane & 31, mb = (lane >> 5) * 8;
    const int nrow = n0 + li, kcol = k0 + li;
    const bool nok = nrow < N, kok = kcol < K;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    float bsum = 0.0f;
    constexpr int U = 4;
    for (int m = 0; m < M; m += U * TK) {                       // U steps' loads (8 + 8 + 8 per step and lane) in flight together
        float gv[U][8], mv[U][8], xv[U][8];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int mm = m + u * TK + mb + i;
                const bool mok = mm < M;
                const size_t go = (size_t)(mok ? mm : 0) * N + (nok ? nrow : 0);
                gv[u][i] = (mok && nok) ? gy[go] : 0.0f;
                mv[u][i] = (act_out && mok && nok) ? act_out[go] : 1.0f;
                xv[u][i] = (mok && kok) ? ldf(x + (size_t)mm * K + kcol) : 0.0f;
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) {
            bf16x8 a, b;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float g = mv[u][i] > 0.0f ? gv[u][i] : gv[u][i] * slope;
                bsum += g;
                a[i] = (__bf16)g;
                b[i] = (__bf16)xv[u][i];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    if (gbias && blockIdx.x == 0 && w == 0) {                  // lanes l and l+32 hold the two halves of every 16-row step
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 32 && nok) gbias[nrow] = bsum;
    }
    if (!kok) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = n0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (i < N) gW[(size_t)i * K + kcol] = acc[r];
    }
}

int pick_ksplit(int M, int N, int K)
{
    // enough workgroups to stream a large weight from HBM with every CU; small layers stay single-pass (latency-bound anyway)
    const long tiles = (long)psi_cdiv(N, TN) * psi_cdiv(M, 128);
    if ((long)N * K < (1L << 21)) return 1;                     // weight < 8 MB
    int s = (int)(256 / tiles);
    while (s > 1 && (K / s) < 256) s >>= 1;
    if (s < 1) s = 1;
    return s;
}

}  // namespace

extern "C" size_t psi_linear_workspace_floats(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = pick_ksplit(M, N, K);
    return s > 1 ? (size_t)s * M * N : 0;
}

extern "C" int psi_linear_forward(const void *x, int x_is_bf16, const float *W, const float *bias, const float *residual, int M, int N, int K,
                                  int act, float slope, float *y, float *act_out, float *ws, void *stream)
{
    PSI_REQUIRE(x && W && y, "null pointer");
    PSI_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, "K must be a positive multiple of 16");
    PSI_REQUIRE(act == 0 || (act == 1 && slope > 0.0f), "act: 0 = none, 1 = LeakyReLU with slope > 0");
    hipStream_t st = (hipStream_t)stream;
    int S = pick_ksplit(M, N, K);
    int kchunk = K;
    if (S > 1) {
        PSI_REQUIRE(ws, "this shape is split over K: pass psi_linear_workspace_floats() floats of workspace");
        kchunk = psi_cdiv(psi_cdiv(K, S), 16) * 16;
        S = psi_cdiv(K, kchunk);
    }
    dim3 grid(psi_cdiv(N, TN), psi_cdiv(M, 128), S);
    float *part = S > 1 ? ws : nullptr;
    if (x_is_bf16)
        hipLaunchKernelGGL(linear_fwd_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16 *)x, W, bias, residual, M, N, K, kchunk, act, slope, y,
                           act_out, part);
    else
        hipLaunchKernelGGL(linear_fwd_kernel<float>, grid, dim3(256), 0, st, (const float *)x, W, bias, residual, M, N, K, kchunk, act, slope, y,
                           act_out, part);
    PSI_CHECK_LAUNCH("linear_fwd_kernel");
    if (S > 1) {
        hipLaunchKernelGGL(linear_reduce_kernel, dim3(psi_cdiv((long)M * N, 256)), dim3(256), 0, st, part, S, bias, residual, M, N, act, slope, y,
                           act_out);
        PSI_CHECK_LAUNCH("linear_reduce_kernel");
    }
    return 0;
}

extern "C" int psi_linear_backward(const float *gy, const float *act_out, const void *x, int x_is_bf16, const float *W, int M, int N, int K,
                                   float slope, void *gx, float *gW, float *gbias, void *stream)
{
    PSI_REQUIRE(gy && x && W, "null pointer");
    PSI_REQUIRE(M > 0 && N > 0 && K > 0 && N % 16 == 0, "N must be a positive multiple of 16");
    hipStream_t st = (hipStream_t)stream;
    if (gx) {
        dim3 grid(psi_cdiv(K, TN), psi_cdiv(M, 128));
        if (x_is_bf16)
            hipLaunchKernelGGL(linear_bwd_dx_kernel<__bf16>, grid, dim3(256), 0, st, gy, act_out, W, M, N, K, slope, (__bf16 *)gx);
        else
            hipLaunchKernelGGL(linear_bwd_dx_kernel<float>, grid, dim3(256), 0, st, gy, act_out, W, M, N, K, slope, (float *)gx);
        PSI_CHECK_LAUNCH("linear_bwd_dx_kernel");
    }
    if (gW) {
        dim3 grid(psi_cdiv(K, 4 * TN), psi_cdiv(N, TM));
        if (x_is_bf16)
            hipLaunchKernelGGL(linear_bwd_dw_kernel<__bf16>, grid, dim3(256), 0, st, gy, act_out, (const __bf16 *)x, M, N, K, slope, gW, gbias);
        else
            hipLaunchKernelGGL(linear_bwd_dw_kernel<float>, grid, dim3(256), 0, st, gy, act_out, (const float *)x, M, N, K, slope, gW, gbias);
        PSI_CHECK_LAUNCH("linear_bwd_dw_kernel");
    }
    return 0;
}
