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
#include <atomic>

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

// NTERM = 3 (psi_linear_forward3): the fp32 model's precision on the bf16 matrix cores — every operand is split into hi = bf16(v) and
// lo = bf16(v - hi), a product is hi*hi + hi*lo + lo*hi with fp32 accumulation (conv_gemm.hip explains and quantifies it); the lo parts
// live in a second set of LDS tiles.  `vec` = rows are 16-byte aligned (K % 4 == 0); otherwise (the 3-, 75-wide input layers) the tiles
// are gathered element by element.
template <int NTERM>
__device__ __forceinline__ void split4(const f4 &v, __bf16 *hi, __bf16 *lo)
{
    bf4 h, l;
#pragma unroll
    for (int e = 0; e < 4; e++) {
        h[e] = (__bf16)v[e];
        if (NTERM > 1) l[e] = (__bf16)(v[e] - (float)h[e]);
    }
    *(bf4 *)hi = h;
    if (NTERM > 1) *(bf4 *)lo = l;
}

__device__ __forceinline__ f4 load4_checked(const float *row, int c, int k_end, bool row_ok, bool vec)
{
    if (!row_ok || c >= k_end) return (f4){0, 0, 0, 0};
    if (vec) return *(const f4 *)(row + c);
    f4 r = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; e++)
        if (c + e < k_end) r[e] = row[c + e];
    return r;
}

template <typename XT> struct XTile;
template <> struct XTile<float> {                           // 128 x 64 fp32: 8 float4 per thread
    f4 r[8];
    __device__ __forceinline__ void load(const float *x, int M, int K, int mblk, int k0, int k_end, bool vec)
    {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            r[i] = load4_checked(x + (size_t)(mblk + row) * K, k0 + c, k_end, mblk + row < M, vec);
        }
    }
    template <int NTERM>
    __device__ __forceinline__ void store(__bf16 (*As)[PITCH], __bf16 (*Al)[PITCH]) const
    {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            split4<NTERM>(r[i], &As[row][c], &Al[row][c]);
        }
    }
};
template <> struct XTile<__bf16> {                          // 128 x 64 bf16: 4 x 16 bytes per thread (exact in bf16: no lo part)
    bf16x8 r[4];
    __device__ __forceinline__ void load(const __bf16 *x, int M, int K, int mblk, int k0, int k_end, bool)
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
    template <int NTERM>
    __device__ __forceinline__ void store(__bf16 (*As)[PITCH], __bf16 (*Al)[PITCH]) const
    {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 3, c = (idx & 7) * 8;
            *(bf16x8 *)&As[row][c] = r[i];
            if (NTERM > 1) {
                bf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; e++) z[e] = (__bf16)0.0f;
                *(bf16x8 *)&Al[row][c] = z;
            }
        }
    }
};

constexpr size_t linear_fwd_lds(int nterm) { return (size_t)2 * (128 + TN) * PITCH * 2 * (nterm > 1 ? 2 : 1); }

template <typename XT, int NTERM>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const XT *__restrict__ x, const float *__restrict__ W, const float *__restrict__ bias,
                                                         const float *__restrict__ residual, int M, int N, int K, int kchunk, int act,
                                                         float slope, float *__restrict__ y, float *__restrict__ act_out,
                                                         float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __bf16 (*Tile)[PITCH];
    // [2][128][PITCH] rows of x (hi), [2][TN][PITCH] rows of W (hi), then the same again for the lo parts (NTERM = 3)
    auto As = [&](int buf) { return (Tile)(smem + (size_t)buf * 128 * PITCH * 2); };
    auto Bs = [&](int buf) { return (Tile)(smem + (size_t)(2 * 128 + buf * TN) * PITCH * 2); };
    auto Al = [&](int buf) { return (Tile)(smem + (size_t)(2 * (128 + TN) + buf * 128) * PITCH * 2); };
    auto Bl = [&](int buf) { return (Tile)(smem + (size_t)(2 * (128 + TN) + 2 * 128 + buf * TN) * PITCH * 2); };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n0 = blockIdx.x * TN, mblk = blockIdx.y * 128, m0 = mblk + w * TM, ks = blockIdx.z;
    const int k_begin = ks * kchunk, k_end = min(K, k_begin + kchunk);
    const int li = lane & 31, kb = (lane >> 5) * 8;
    const bool vec = (K & 3) == 0;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    XTile<XT> xa;
    f4 wb[2];
    auto load_w = [&](int k0) {                             // 32 x 64 fp32: 2 float4 per thread
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            wb[i] = load4_checked(W + (size_t)(n0 + row) * K, k0 + c, k_end, n0 + row < N, vec);
        }
    };
    auto store_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            split4<NTERM>(wb[i], &Bs(buf)[row][c], &Bl(buf)[row][c]);
        }
    };
    xa.load(x, M, K, mblk, k_begin, k_end, vec);
    load_w(k_begin);
    xa.template store<NTERM>(As(0), Al(0));
    store_w(0);
    __syncthreads();
    int buf = 0;
    for (int k0 = k_begin; k0 < k_end; k0 += BK, buf ^= 1) {
        const bool more = k0 + BK < k_end;
        if (more) {                                         // next tile: global loads in flight during this tile's MFMAs
            xa.load(x, M, K, mblk, k0 + BK, k_end, vec);
            load_w(k0 + BK);
        }
#pragma unroll
        for (int st = 0; st < BK / TK; st++) {
            const bf16x8 a = *(const bf16x8 *)&As(buf)[w * TM + li][st * TK + kb];
            const bf16x8 b = *(const bf16x8 *)&Bs(buf)[li][st * TK + kb];
            if (NTERM > 1) {                                // the small terms first
                const bf16x8 al = *(const bf16x8 *)&Al(buf)[w * TM + li][st * TK + kb];
                const bf16x8 bl = *(const bf16x8 *)&Bl(buf)[li][st * TK + kb];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if (more) {
            xa.template store<NTERM>(As(buf ^ 1), Al(buf ^ 1));
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
// Backward.  Both products contract over an index along which one operand is NOT contiguous (dX = G W contracts over n: W[n][k] has
// k fastest; dW = G^T X contracts over m: G[m][n] and X[m][k] have n / k fastest), so a lane cannot read "its" eight consecutive
// contraction elements from memory.  The first version did exactly that — eight 4-byte loads per lane and operand, each lane on a
// different row — and ran 4-7x slower than the library GEMM (85 us for 128 x 512 x 512).  Here every tile is read from memory the way
// it is stored (16-byte lane loads, a wave covering whole rows), rounded to bf16, and TRANSPOSED ON THE WAY INTO LDS, so that the MFMA
// operands are contiguous LDS reads again; the activation-derivative mask G = gy * act'(.) is applied in the same pass.
// ------------------------------------------------------------------------------------------------
constexpr int DXK = 64;                  // dX: output columns (k) per workgroup
constexpr int DXN = 64;                  // dX: contraction tile (n)
constexpr int DXP = DXN + 8;             // LDS pitch (elements): 16-byte aligned rows, conflict-free 16-byte reads

// dX[M,K] = G[M,N] W[N,K]: workgroup = 128 rows x 64 k-columns, wave w = rows [32w, 32w+32) x two 32-column tiles; contraction over n in
// tiles of 64: G tile [128][64] row-major (A operand as in the forward), W tile [64 n][64 k] stored TRANSPOSED as Wt[k][n].
constexpr size_t linear_dx_lds(int nterm) { return (size_t)2 * (128 + DXK) * DXP * 2 * (nterm > 1 ? 2 : 1); }

template <typename XT, int NTERM>
__device__ __forceinline__ void linear_bwd_dx_body(unsigned char *smem, int bx, int by, int bz, const float *__restrict__ gy,
                                                   const float *__restrict__ act_out, const float *__restrict__ W, int M, int N, int K, float slope,
                                                   XT *__restrict__ gx, int nchunk, float *__restrict__ part)
{
    typedef __bf16 (*Tile)[DXP];
    // [2][128][DXP] rows of G (hi), [2][DXK][DXP] W transposed (hi), then the lo parts (NTERM = 3: psi_linear_backward3)
    auto Gs = [&](int buf) { return (Tile)(smem + (size_t)buf * 128 * DXP * 2); };
    auto Wt = [&](int buf) { return (Tile)(smem + (size_t)(2 * 128 + buf * DXK) * DXP * 2); };
    auto Gl = [&](int buf) { return (Tile)(smem + (size_t)(2 * (128 + DXK) + buf * 128) * DXP * 2); };
    auto Wl = [&](int buf) { return (Tile)(smem + (size_t)(2 * (128 + DXK) + 2 * 128 + buf * DXK) * DXP * 2); };
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k0 = bx * DXK, mblk = by * 128, m0 = mblk + w * TM;
    const int n_begin = bz * nchunk, n_end = min(N, n_begin + nchunk);      // this workgroup's slice of the contraction
    const int li = lane & 31, kb = (lane >> 5) * 8;
    const bool vecN = (N & 3) == 0, vecK = (K & 3) == 0;    // rows of G / act_out and of W are 16-byte aligned
    f16v acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[t][i] = 0.0f;
    f4 gr[8], mr[8], wr[4];
    auto load_tiles = [&](int n0) {
#pragma unroll
        for (int i = 0; i < 8; i++) {                       // G: 128 rows x 64 n = 2048 float4, 8 per thread, a wave covers 4 whole rows
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            const bool rok = mblk + row < M;
            gr[i] = load4_checked(gy + (size_t)(mblk + row) * N, n0 + c, n_end, rok, vecN);
            if (act_out) {
                mr[i] = load4_checked(act_out + (size_t)(mblk + row) * N, n0 + c, n_end, rok, vecN);
                if (!rok || n0 + c >= n_end) mr[i] = (f4){1, 1, 1, 1};
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {                       // W: 64 n-rows x 64 k = 1024 float4, 4 per thread
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            wr[i] = load4_checked(W + (size_t)(n0 + row) * K, k0 + c, K, n0 + row < n_end, vecK);
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
            f4 g;
#pragma unroll
            for (int e = 0; e < 4; e++) g[e] = act_out ? (mr[i][e] > 0.0f ? gr[i][e] : gr[i][e] * slope) : gr[i][e];
            split4<NTERM>(g, &Gs(buf)[row][c], &Gl(buf)[row][c]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {                       // transposed: element (n = row, k = c + e) -> Wt[c + e][row]
            const int idx = threadIdx.x + 256 * i, row = idx >> 4, c = (idx & 15) * 4;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const __bf16 hh = (__bf16)wr[i][e];
                Wt(buf)[c + e][row] = hh;
                if (NTERM > 1) Wl(buf)[c + e][row] = (__bf16)(wr[i][e] - (float)hh);
            }
        }
    };
    load_tiles(n_begin);
    store_tiles(0);
    __syncthreads();
    int buf = 0;
    for (int n0 = n_begin; n0 < n_end; n0 += DXN, buf ^= 1) {
        const bool more = n0 + DXN < n_end;
        if (more) load_tiles(n0 + DXN);                     // next tile's global loads in flight during this tile's MFMAs
#pragma unroll
        for (int st = 0; st < DXN / TK; st++) {
            const bf16x8 a = *(const bf16x8 *)&Gs(buf)[w * TM + li][st * TK + kb];
            bf16x8 al;
            if (NTERM > 1) al = *(const bf16x8 *)&Gl(buf)[w * TM + li][st * TK + kb];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const bf16x8 b = *(const bf16x8 *)&Wt(buf)[t * TN + li][st * TK + kb];
                if (NTERM > 1) {
                    const bf16x8 bl = *(const bf16x8 *)&Wl(buf)[t * TN + li][st * TK + kb];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc[t], 0, 0, 0);
                }
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
            }
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }
    if (m0 >= M) return;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int col = k0 + t * TN + li;
        if (col >= K) continue;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int i = m0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
            if (i >= M) continue;
            if (part) part[((size_t)bz * M + i) * K + col] = acc[t][r];
            else gx[(size_t)i * K + col] = (XT)acc[t][r];
        }
    }
}

template <typename XT, int NTERM>
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float *__restrict__ gy, const float *__restrict__ act_out,
                                                            const float *__restrict__ W, int M, int N, int K, float slope,
                                                            XT *__restrict__ gx, int nchunk, float *__restrict__ part)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    linear_bwd_dx_body<XT, NTERM>(smem, blockIdx.x, blockIdx.y, blockIdx.z, gy, act_out, W, M, N, K, slope, gx, nchunk, part);
}

// sum of the n-slices of dX in slice order (deterministic), converted to the gradient's type
template <typename XT>
__global__ __launch_bounds__(256) void linear_dx_reduce_kernel(const float *__restrict__ part, int S, size_t MK, XT *__restrict__ gx)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MK) return;
    if ((MK & 3) == 0) {
        f4 a = *(const f4 *)(part + i);
        for (int s = 1; s < S; s++) a += *(const f4 *)(part + (size_t)s * MK + i);
#pragma unroll
        for (int e = 0; e < 4; e++) gx[i + e] = (XT)a[e];
    } else {
        for (int e = 0; e < 4 && i + e < MK; e++) {
            float a = part[i + e];
            for (int s = 1; s < S; s++) a += part[(size_t)s * MK + i + e];
            gx[i + e] = (XT)a;
        }
    }
}

// dW[N,K] = G^T[N,M] X[M,K] (+ gbias[n] = sum_m G[m][n]): workgroup = 32 n-rows x 128 k-columns (wave w: k tile w), contraction over m in
// tiles of 128 — the whole batch in one tile for the CVAEs' M = 128.  Both operands are stored transposed: Gt[n][m], Xt[k][m], two
// consecutive m packed into one 32-bit LDS store (a thread loads the float4s of rows 2p and 2p + 1).
constexpr int DWK = 128;
constexpr int DWM = 128;
constexpr int DWP = DWM + 8;             // pitch in elements (16-byte aligned rows)

template <typename XT> struct XPair;     // the float4 / 4 x bf16 of rows 2p and 2p+1 at columns c..c+3 -> four packed bf16 pairs
template <> struct XPair<float> {
    // c .. c + 3 of rows o0 / o1 (row starts), columns beyond `end` are zeros; vec: the rows are 16-byte aligned
    __device__ static __forceinline__ void load(const float *r0, const float *r1, int c, int end, bool ok0, bool ok1, bool vec, float (&a)[4], float (&b)[4])
    {
        const f4 v0 = load4_checked(r0, c, end, ok0, vec), v1 = load4_checked(r1, c, end, ok1, vec);
#pragma unroll
        for (int e = 0; e < 4; e++) { a[e] = v0[e]; b[e] = v1[e]; }
    }
};
template <> struct XPair<__bf16> {
    __device__ static __forceinline__ void load(const __bf16 *r0, const __bf16 *r1, int c, int end, bool ok0, bool ok1, bool, float (&a)[4], float (&b)[4])
    {
        bf4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; e++) v0[e] = v1[e] = (__bf16)0.0f;
        if (ok0 && c < end) v0 = *(const bf4 *)(r0 + c);
        if (ok1 && c < end) v1 = *(const bf4 *)(r1 + c);
#pragma unroll
        for (int e = 0; e < 4; e++) { a[e] = (float)v0[e]; b[e] = (float)v1[e]; }
    }
};
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    const __bf16 l = (__bf16)lo, h = (__bf16)hi;
    return (unsigned)__builtin_bit_cast(unsigned short, l) | ((unsigned)__builtin_bit_cast(unsigned short, h) << 16);
}

constexpr size_t linear_dw_lds(int nterm) { return (size_t)(TM + DWK) * DWP * 2 * (nterm > 1 ? 2 : 1); }

// two values -> one packed {lo, hi} bf16 pair per part (hi part, and the split residue for NTERM = 3)
template <int NTERM>
__device__ __forceinline__ void pack_split(float v0, float v1, unsigned *hi, unsigned *lo)
{
    const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
    *hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    if (NTERM > 1) {
        const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
        *lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
}

template <typename XT, int NTERM>
__device__ __forceinline__ void linear_bwd_dw_body(unsigned char *smem, int bx, int by, const float *__restrict__ gy, const float *__restrict__ act_out,
                                                   const XT *__restrict__ x, int M, int N, int K, float slope, float *__restrict__ gW,
                                                   float *__restrict__ gbias)
{
    typedef __bf16 (*Tile)[DWP];
    const Tile Gt = (Tile)smem, Xt = (Tile)(smem + (size_t)TM * DWP * 2);                                 // [TM][DWP], [DWK][DWP]
    const Tile Gl = (Tile)(smem + (size_t)(TM + DWK) * DWP * 2), Xl = (Tile)(smem + (size_t)(2 * TM + DWK) * DWP * 2);   // lo parts (NTERM = 3)
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n0 = by * TM, k0 = bx * DWK;
    const int li = lane & 31, mb = (lane >> 5) * 8;
    const bool vecN = (N & 3) == 0, vecK = (K & 3) == 0;
    f16v acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.0f;
    float bs4[4] = {0.0f, 0.0f, 0.0f, 0.0f};               // thread (pair p, quad q): sums of G over its rows for columns 4q..4q+3 -> gbias
    for (int mt = 0; mt < M; mt += DWM) {
        // G tile: 128 m x 32 n = 64 row pairs x 8 column quads = 512 items, 2 per thread
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx & 7, p = idx >> 3;
            const int ma = mt + 2 * p, c = n0 + 4 * q;
            const bool ok0 = ma < M, ok1 = ma + 1 < M;
            float a[4], b[4], ma4[4], mb4[4];
            XPair<float>::load(gy + (size_t)ma * N, gy + (size_t)(ma + 1) * N, c, N, ok0, ok1, vecN, a, b);
            if (act_out) {
                XPair<float>::load(act_out + (size_t)ma * N, act_out + (size_t)(ma + 1) * N, c, N, ok0, ok1, vecN, ma4, mb4);
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    a[e] = (ok0 && c + e < N && !(ma4[e] > 0.0f)) ? a[e] * slope : a[e];
                    b[e] = (ok1 && c + e < N && !(mb4[e] > 0.0f)) ? b[e] * slope : b[e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                bs4[e] += a[e] + b[e];
                pack_split<NTERM>(a[e], b[e], (unsigned *)&Gt[4 * q + e][2 * p], (unsigned *)&Gl[4 * q + e][2 * p]);
            }
        }
        // X tile: 128 m x 128 k = 64 row pairs x 32 column quads = 2048 items, 8 per thread
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int idx = threadIdx.x + 256 * i, q = idx & 31, p = idx >> 5;
            const int ma = mt + 2 * p, c = k0 + 4 * q;
            const bool ok0 = ma < M, ok1 = ma + 1 < M;
            float a[4], b[4];
            XPair<XT>::load(x + (size_t)ma * K, x + (size_t)(ma + 1) * K, c, K, ok0, ok1, vecK, a, b);
#pragma unroll
            for (int e = 0; e < 4; e++) pack_split<NTERM>(a[e], b[e], (unsigned *)&Xt[4 * q + e][2 * p], (unsigned *)&Xl[4 * q + e][2 * p]);
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < DWM / TK; st++) {
            const bf16x8 a = *(const bf16x8 *)&Gt[li][st * TK + mb];
            const bf16x8 b = *(const bf16x8 *)&Xt[w * TN + li][st * TK + mb];
            if (NTERM > 1) {
                const bf16x8 al = *(const bf16x8 *)&Gl[li][st * TK + mb];
                const bf16x8 bl = *(const bf16x8 *)&Xl[w * TN + li][st * TK + mb];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    if (gbias && bx == 0) {
        // column sums of G: thread (p, q) of item slots i = 0, 1 holds partial sums of columns 4q..4q+3 over its row pairs; the 64 threads
        // sharing q (p = idx >> 3 over both slots) are combined in a fixed order through LDS (the operand tiles are no longer needed)
        float (*bred)[4] = (float (*)[4])smem;
#pragma unroll
        for (int e = 0; e < 4; e++) bred[threadIdx.x][e] = bs4[e];
        __syncthreads();
        if (threadIdx.x < TM) {
            const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
            float s = 0.0f;
            for (int t = q; t < 256; t += 8) s += bred[t][e];
            if (n0 + threadIdx.x < N) gbias[n0 + threadIdx.x] = s;
        }
    }
    const int kcol = k0 + w * TN + li;
    if (kcol >= K) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int i = n0 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
        if (i < N) gW[(size_t)i * K + kcol] = acc[r];
    }
}

template <typename XT, int NTERM>
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float *__restrict__ gy, const float *__restrict__ act_out,
                                                            const XT *__restrict__ x, int M, int N, int K, float slope,
                                                            float *__restrict__ gW, float *__restrict__ gbias)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    linear_bwd_dw_body<XT, NTERM>(smem, blockIdx.x, blockIdx.y, gy, act_out, x, M, N, K, slope, gW, gbias);
}

// Both products of a layer's backward in ONE launch: the first ndx workgroups are the input-gradient tiles (gdx x gdy x slices), the rest the
// weight-gradient tiles.  Each product alone leaves most of the chip idle (64 .. 256 workgroups, a latency chain each) and the two only share
// their input G: side by side they take the time of the longer one, and a layer's backward is one launch (plus the slice sum) instead of two.
template <typename XT, int NTERM>
__global__ __launch_bounds__(256) void linear_bwd_both_kernel(const float *__restrict__ gy, const float *__restrict__ act_out, const float *__restrict__ W,
                                                              const XT *__restrict__ x, int M, int N, int K, float slope, XT *__restrict__ gx,
                                                              int nchunk, float *__restrict__ part, float *__restrict__ gW,
                                                              float *__restrict__ gbias, int gdx, int gdy, int ndx, int gwx)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int b = blockIdx.x;
    if (b < ndx) {
        const int bx = b % gdx, r = b / gdx;
        linear_bwd_dx_body<XT, NTERM>(smem, bx, r % gdy, r / gdy, gy, act_out, W, M, N, K, slope, gx, nchunk, part);
    } else {
        b -= ndx;
        linear_bwd_dw_body<XT, NTERM>(smem, b % gwx, b / gwx, gy, act_out, x, M, N, K, slope, gW, gbias);
    }
}

int pick_ksplit(int M, int N, int K)
{
    // M is the batch (128): these GEMMs are weight streams and latency chains, not matrix-pipe work.  A single pass has N / 32 workgroups
    // walking all of K in 64-wide tiles with a barrier each (32 workgroups x 16 dependent tiles for a 1024 x 1024 layer: 14.5 us for 4 MB);
    // the contraction is split until the launch has about one workgroup per compute unit, at least two tiles per workgroup.
    const long tiles = (long)psi_cdiv(N, TN) * psi_cdiv(M, 128);
    int s = (int)(256 / tiles);
    if (s > K / (2 * BK)) s = K / (2 * BK);
    return s < 1 ? 1 : s;
}

// dX: K / 64 workgroups per 128 rows (16 for a 1024 x 1024 layer) each walking all of N; same treatment over n
int pick_nsplit(int M, int N, int K)
{
    const long tiles = (long)psi_cdiv(K, DXK) * psi_cdiv(M, 128);
    int s = (int)(256 / tiles);
    if (s > N / (2 * DXN)) s = N / (2 * DXN);
    return s < 1 ? 1 : s;
}

}  // namespace

extern "C" size_t psi_linear_workspace_floats(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = pick_ksplit(M, N, K);
    return s > 1 ? (size_t)s * M * N : 0;
}

template <typename XT, int NTERM>
static int launch_linear_fwd(const void *x, const float *W, const float *bias, const float *residual, int M, int N, int K, int kchunk, int S, int act,
                             float slope, float *y, float *act_out, float *part, hipStream_t st)
{
    constexpr size_t lds = linear_fwd_lds(NTERM);
    auto kern = linear_fwd_kernel<XT, NTERM>;
    if (lds > 48 * 1024) {                                  // the attribute is per device: set once per device
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        PSI_CHECK_HIP(hipGetDevice(&dev));
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            PSI_CHECK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            done.fetch_or(bit, std::memory_order_release);
        }
    }
    dim3 grid(psi_cdiv(N, TN), psi_cdiv(M, 128), S);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, (const XT *)x, W, bias, residual, M, N, K, kchunk, act, slope, y, act_out, part);
    PSI_CHECK_LAUNCH("linear_fwd_kernel");
    return 0;
}

static int linear_forward_any(const void *x, int x_is_bf16, const float *W, const float *bias, const float *residual, int M, int N, int K, int act,
                              float slope, float *y, float *act_out, float *ws, int nterm, hipStream_t st)
{
    int S = pick_ksplit(M, N, K);
    int kchunk = K;
    if (S > 1) {
        PSI_REQUIRE(ws, "this shape is split over K: pass psi_linear_workspace_floats() floats of workspace");
        kchunk = psi_cdiv(psi_cdiv(K, S), 16) * 16;
        S = psi_cdiv(K, kchunk);
    }
    float *part = S > 1 ? ws : nullptr;
    int rc;
    if (nterm == 3)
        rc = x_is_bf16 ? launch_linear_fwd<__bf16, 3>(x, W, bias, residual, M, N, K, kchunk, S, act, slope, y, act_out, part, st)
                       : launch_linear_fwd<float, 3>(x, W, bias, residual, M, N, K, kchunk, S, act, slope, y, act_out, part, st);
    else
        rc = x_is_bf16 ? launch_linear_fwd<__bf16, 1>(x, W, bias, residual, M, N, K, kchunk, S, act, slope, y, act_out, part, st)
                       : launch_linear_fwd<float, 1>(x, W, bias, residual, M, N, K, kchunk, S, act, slope, y, act_out, part, st);
    if (rc) return rc;
    if (S > 1) {
        hipLaunchKernelGGL(linear_reduce_kernel, dim3(psi_cdiv((long)M * N, 256)), dim3(256), 0, st, part, S, bias, residual, M, N, act, slope, y,
                           act_out);
        PSI_CHECK_LAUNCH("linear_reduce_kernel");
    }
    return 0;
}

extern "C" int psi_linear_forward(const void *x, int x_is_bf16, const float *W, const float *bias, const float *residual, int M, int N, int K,
                                  int act, float slope, float *y, float *act_out, float *ws, void *stream)
{
    PSI_REQUIRE(x && W && y, "null pointer");
    PSI_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, "K must be a positive multiple of 16");
    PSI_REQUIRE(act == 0 || (act == 1 && slope > 0.0f), "act: 0 = none, 1 = LeakyReLU with slope > 0");
    return linear_forward_any(x, x_is_bf16, W, bias, residual, M, N, K, act, slope, y, act_out, ws, 1, (hipStream_t)stream);
}

// The same layer at the fp32 model's precision (three-term split products, see the kernel) and for ANY K and N (the 3-, 72-, 75-wide layers of
// cvae.py:474-492 / net_layers.py:66-93 included: rows that are not 16-byte aligned are gathered element by element).
extern "C" int psi_linear_forward3(const void *x, int x_is_bf16, const float *W, const float *bias, const float *residual, int M, int N, int K,
                                   int act, float slope, float *y, float *act_out, float *ws, void *stream)
{
    PSI_REQUIRE(x && W && y, "null pointer");
    PSI_REQUIRE(M > 0 && N > 0 && K > 0 && !(x_is_bf16 && K % 8), "bad sizes (a bf16 input needs K % 8 == 0)");
    PSI_REQUIRE(act == 0 || (act == 1 && slope > 0.0f), "act: 0 = none, 1 = LeakyReLU with slope > 0");
    return linear_forward_any(x, x_is_bf16, W, bias, residual, M, N, K, act, slope, y, act_out, ws, 3, (hipStream_t)stream);
}

extern "C" size_t psi_linear_backward_workspace_floats(int M, int N, int K)
{
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int s = pick_nsplit(M, N, K);
    return s > 1 ? (size_t)s * M * K : 0;
}

static hipError_t set_lds_once(const void *kern, size_t lds, std::atomic<unsigned long long> &done)
{
    if (lds <= 48 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}

template <typename XT, int NTERM>
static int linear_backward_t(const float *gy, const float *act_out, const void *x, const float *W, int M, int N, int K, float slope, void *gx,
                             float *gW, float *gbias, float *ws, hipStream_t st)
{
    if (gx && gW) {                                            // the usual case: both products, one launch
        int S = pick_nsplit(M, N, K), nchunk = N;
        if (S > 1) {
            PSI_REQUIRE(ws, "this shape is split over n: pass psi_linear_backward_workspace_floats() floats of workspace");
            nchunk = psi_cdiv(psi_cdiv(N, S), DXN) * DXN;
            S = psi_cdiv(N, nchunk);
        }
        float *part = S > 1 ? ws : nullptr;
        const int gdx = psi_cdiv(K, DXK), gdy = psi_cdiv(M, 128), ndx = gdx * gdy * S, gwx = psi_cdiv(K, DWK), ndw = gwx * psi_cdiv(N, TM);
        constexpr size_t lds = linear_dx_lds(NTERM) > linear_dw_lds(NTERM) ? linear_dx_lds(NTERM) : linear_dw_lds(NTERM);
        static std::atomic<unsigned long long> a_both{0};
        PSI_CHECK_HIP(set_lds_once((const void *)linear_bwd_both_kernel<XT, NTERM>, lds, a_both));
        hipLaunchKernelGGL((linear_bwd_both_kernel<XT, NTERM>), dim3((unsigned)(ndx + ndw)), dim3(256), lds, st, gy, act_out, W, (const XT *)x, M, N, K,
                           slope, (XT *)gx, nchunk, part, gW, gbias, gdx, gdy, ndx, gwx);
        PSI_CHECK_LAUNCH("linear_bwd_both_kernel");
        if (S > 1) {
            const size_t MK = (size_t)M * K;
            const dim3 rg((unsigned)psi_cdiv((long)((MK + 3) / 4), 256));
            hipLaunchKernelGGL(linear_dx_reduce_kernel<XT>, rg, dim3(256), 0, st, (const float *)part, S, MK, (XT *)gx);
            PSI_CHECK_LAUNCH("linear_dx_reduce_kernel");
        }
        return 0;
    }
    if (gx) {
        int S = pick_nsplit(M, N, K), nchunk = N;
        if (S > 1) {
            PSI_REQUIRE(ws, "this shape is split over n: pass psi_linear_backward_workspace_floats() floats of workspace");
            nchunk = psi_cdiv(psi_cdiv(N, S), DXN) * DXN;
            S = psi_cdiv(N, nchunk);
        }
        float *part = S > 1 ? ws : nullptr;
        dim3 grid(psi_cdiv(K, DXK), psi_cdiv(M, 128), S);
        static std::atomic<unsigned long long> a_dx{0};
        PSI_CHECK_HIP(set_lds_once((const void *)linear_bwd_dx_kernel<XT, NTERM>, linear_dx_lds(NTERM), a_dx));
        hipLaunchKernelGGL((linear_bwd_dx_kernel<XT, NTERM>), grid, dim3(256), linear_dx_lds(NTERM), st, gy, act_out, W, M, N, K, slope, (XT *)gx, nchunk, part);
        PSI_CHECK_LAUNCH("linear_bwd_dx_kernel");
        if (S > 1) {
            const size_t MK = (size_t)M * K;
            const dim3 rg((unsigned)psi_cdiv((long)((MK + 3) / 4), 256));
            hipLaunchKernelGGL(linear_dx_reduce_kernel<XT>, rg, dim3(256), 0, st, (const float *)part, S, MK, (XT *)gx);
            PSI_CHECK_LAUNCH("linear_dx_reduce_kernel");
        }
    }
    if (gW) {
        dim3 grid(psi_cdiv(K, DWK), psi_cdiv(N, TM));
        static std::atomic<unsigned long long> a_dw{0};
        PSI_CHECK_HIP(set_lds_once((const void *)linear_bwd_dw_kernel<XT, NTERM>, linear_dw_lds(NTERM), a_dw));
        hipLaunchKernelGGL((linear_bwd_dw_kernel<XT, NTERM>), grid, dim3(256), linear_dw_lds(NTERM), st, gy, act_out, (const XT *)x, M, N, K, slope, gW, gbias);
        PSI_CHECK_LAUNCH("linear_bwd_dw_kernel");
    }
    return 0;
}

extern "C" int psi_linear_backward(const float *gy, const float *act_out, const void *x, int x_is_bf16, const float *W, int M, int N, int K,
                                   float slope, void *gx, float *gW, float *gbias, float *ws, void *stream)
{
    PSI_REQUIRE(gy && x && W, "null pointer");
    PSI_REQUIRE(M > 0 && N > 0 && K > 0 && N % 16 == 0 && K % 4 == 0, "N must be a positive multiple of 16 and K of 4");
    if (x_is_bf16) return linear_backward_t<__bf16, 1>(gy, act_out, x, W, M, N, K, slope, gx, gW, gbias, ws, (hipStream_t)stream);
    return linear_backward_t<float, 1>(gy, act_out, x, W, M, N, K, slope, gx, gW, gbias, ws, (hipStream_t)stream);
}

// The backward of psi_linear_forward3: the same two products with three-term split operands (the fp32 model's precision), fp32 x, ANY N and K
// (rows that are not 16-byte aligned are gathered element by element).  Same arguments and workspace as psi_linear_backward.
extern "C" int psi_linear_backward3(const float *gy, const float *act_out, const float *x, const float *W, int M, int N, int K, float slope,
                                    float *gx, float *gW, float *gbias, float *ws, void *stream)
{
    PSI_REQUIRE(gy && x && W, "null pointer");
    PSI_REQUIRE(M > 0 && N > 0 && K > 0, "bad sizes");
    return linear_backward_t<float, 3>(gy, act_out, x, W, M, N, K, slope, gx, gW, gbias, ws, (hipStream_t)stream);
}
