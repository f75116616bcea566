// Chamfer nearest-neighbour for gfx950 (wave64), replacing chamfer_pytorch/chamfer.cu of the reference.
//
// Contract (include/psi_hip.h): dist[b,j] = min_k d(j,k),
//   d = x2*x2 + y2*y2 + z2*z2,  (x2,y2,z2) = target_k - query_j, fp32, left-to-right, NOT contracted
// into FMAs; idx[b,j] = lowest k attaining the minimum (chamfer.cu:46,126 keep the first minimum).
//
// Design (not a translation of the CUDA kernel, which keeps the running minimum in global memory and
// serialises 512-target chunks through shared memory with an 11-instruction compare/select body):
//   * brute force is fp32-VALU bound (bytes are negligible), so the inner loop is cut to the 9 VALU
//     instructions the arithmetic contract needs per pair: 3 sub, 3 mul, 2 add, 1 v_min.  The argmin
//     is NOT tracked per pair; each thread remembers only WHICH 64-target chunk last lowered its
//     minimum (one compare + two selects per 64 pairs) and a resolve pass rescans that single chunk
//     with the strict-< rule to recover the lowest index.  Exact, because the identical fp32
//     expression is re-evaluated.
//   * targets are wave-uniform: they are fetched with scalar loads (s_load_dwordx*) into SGPRs and
//     fed to the VALU as scalar operands — no LDS staging, no barriers, no VGPRs spent on targets.
//   * the target range is cut into slices so that the grid has >= ~2k workgroups for 256 CUs even at
//     B*n = 65k queries; slices are combined in ascending order with strict '<' (lowest index wins).
#include "psi_internal.h"
#include <math.h>

#pragma clang fp contract(off)    // the distance expression is spelled out by PSI_SQ3 (psi_common.h) in both arithmetic modes

extern "C" int psi_chamfer_arith_mode(void)
{
#ifdef PSI_CHAMFER_FMA
    return 1;
#else
    return 0;
#endif
}

namespace {

constexpr int CH = 64;       // targets per chunk = index-resolution granule
constexpr int BLK = 256;     // threads per workgroup (4 waves)

__device__ __forceinline__ float sqdist(float tx, float ty, float tz, float qx, float qy, float qz)
{
    float x2 = tx - qx;
    float y2 = ty - qy;
    float z2 = tz - qz;
    return PSI_SQ3(x2, y2, z2);
}

template <int Q>
__global__ __launch_bounds__(BLK) void nn_partial_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                         int B, int n, int m, int total_chunks, int chunks_per_slice,
                                                         float *__restrict__ pd, int *__restrict__ pc,
                                                         const int *__restrict__ qidx, long qstride, long tstride)
{
    // queries: row qidx[j] (or j) of a [qstride/3-row] table per body; targets: tstride floats between bodies (0 = shared)
    const int b = blockIdx.z;
    const int s = blockIdx.y;
    const int jbase = blockIdx.x * (BLK * Q) + threadIdx.x;
    const float *__restrict__ qb = xyz1 + (size_t)b * qstride;
    const float *__restrict__ tb = xyz2 + (size_t)b * tstride;

    float qx[Q], qy[Q], qz[Q], best[Q];
    int bchunk[Q];
    const int c_begin = s * chunks_per_slice;
    const int c_end = min(c_begin + chunks_per_slice, total_chunks);
#pragma unroll
    for (int i = 0; i < Q; i++) {
        int j = min(jbase + i * BLK, n - 1);
        if (qidx) j = qidx[j];
        qx[i] = qb[(size_t)j * 3 + 0];
        qy[i] = qb[(size_t)j * 3 + 1];
        qz[i] = qb[(size_t)j * 3 + 2];
        best[i] = INFINITY;
        bchunk[i] = c_begin;
    }

    for (int c = c_begin; c < c_end; c++) {
        const int k0 = c * CH;
        float cm[Q];
#pragma unroll
        for (int i = 0; i < Q; i++) cm[i] = INFINITY;
        if (k0 + CH <= m) {
            const float *__restrict__ t = tb + (size_t)k0 * 3;
#pragma unroll 16
            for (int k = 0; k < CH; k++) {
                float tx = t[k * 3 + 0], ty = t[k * 3 + 1], tz = t[k * 3 + 2];
#pragma unroll
                for (int i = 0; i < Q; i++) cm[i] = fminf(cm[i], sqdist(tx, ty, tz, qx[i], qy[i], qz[i]));
            }
        } else {
            for (int k = k0; k < m; k++) {
                float tx = tb[k * 3 + 0], ty = tb[k * 3 + 1], tz = tb[k * 3 + 2];
#pragma unroll
                for (int i = 0; i < Q; i++) cm[i] = fminf(cm[i], sqdist(tx, ty, tz, qx[i], qy[i], qz[i]));
            }
        }
#pragma unroll
        for (int i = 0; i < Q; i++) {
            bool lower = cm[i] < best[i];
            best[i] = lower ? cm[i] : best[i];
            bchunk[i] = lower ? c : bchunk[i];
        }
    }
#pragma unroll
    for (int i = 0; i < Q; i++) {
        int j = jbase + i * BLK;
        if (j < n) {
            size_t o = ((size_t)s * B + b) * n + j;
            pd[o] = best[i];
            pc[o] = bchunk[i];
        }
    }
}

// Combine slices (ascending, strict '<') and rescan the winning chunk for the lowest minimiser.
// CONTACT: fused epilogue of the contact loss (fitting_proxe.py:139): f = s/(s+c), s = sqrt(d+1e-4);
// writes gq[b,j,:] = gscale * df/dd * 2 (q - t*) and one partial sum of f per workgroup.
template <bool CONTACT>
__global__ __launch_bounds__(BLK) void nn_resolve_kernel(const float *__restrict__ xyz1, const float *__restrict__ xyz2,
                                                         int B, int n, int m, int nslices,
                                                         const float *__restrict__ pd, const int *__restrict__ pc,
                                                         float *__restrict__ dist, int *__restrict__ idx,
                                                         const int *__restrict__ qidx, long qstride, long tstride,
                                                         float cconst, float gscale, float *__restrict__ gq,
                                                         float *__restrict__ fpart)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * BLK + threadIdx.x;
    float fval = 0.0f;
    if (j < n) {
        size_t o = (size_t)b * n + j;
        float best = pd[o];
        int chunk = pc[o];
        for (int s = 1; s < nslices; s++) {
            size_t os = ((size_t)s * B + b) * n + j;
            float d = pd[os];
            if (d < best) {
                best = d;
                chunk = pc[os];
            }
        }
        const size_t qrow = qidx ? (size_t)qidx[j] : (size_t)j;
        const float *__restrict__ qp = xyz1 + (size_t)b * qstride + qrow * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        const float *__restrict__ tb = xyz2 + (size_t)b * tstride;
        const int k0 = chunk * CH;
        const int kend = min(k0 + CH, m);
        float bd = 0.0f;
        int bi = k0;
        for (int k = k0; k < kend; k++) {
            float d = sqdist(tb[k * 3 + 0], tb[k * 3 + 1], tb[k * 3 + 2], qx, qy, qz);
            if (k == k0 || d < bd) {
                bd = d;
                bi = k;
            }
        }
        if (dist) dist[o] = bd;
        if (idx) idx[o] = bi;
        if (CONTACT) {
            float sq = sqrtf(bd + 1e-4f);
            float den = sq + cconst;
            fval = sq / den;
            float dfdd = cconst / (2.0f * sq * den * den);     // d/dd [ s/(s+c) ],  s = sqrt(d + 1e-4)
            float g = gscale * dfdd * 2.0f;
            gq[o * 3 + 0] = g * (qx - tb[(size_t)bi * 3 + 0]);
            gq[o * 3 + 1] = g * (qy - tb[(size_t)bi * 3 + 1]);
            gq[o * 3 + 2] = g * (qz - tb[(size_t)bi * 3 + 2]);
        }
    }
    if (CONTACT) {
        __shared__ float sh[BLK / 64];
        float v = fval;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_down(v, o2, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) fpart[(size_t)b * gridDim.x + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    }
}

// grad_q[b,j] += 2*g*(q - t[idx]);  optionally grad_t[b,idx] -= the same (hardware fp32 atomics).
__global__ __launch_bounds__(BLK) void nn_grad_kernel(const float *__restrict__ q, const float *__restrict__ t,
                                                      const float *__restrict__ gdist, const int *__restrict__ idx,
                                                      int n, int m, float *gq, float *gt)
{
    const int b = blockIdx.y;
    const int j = blockIdx.x * BLK + threadIdx.x;
    if (j >= n) return;
    size_t o = (size_t)b * n + j;
    int j2 = idx[o];
    size_t o2 = (size_t)b * m + j2;
    float g = gdist[o] * 2;
    float dx = g * (q[o * 3 + 0] - t[o2 * 3 + 0]);
    float dy = g * (q[o * 3 + 1] - t[o2 * 3 + 1]);
    float dz = g * (q[o * 3 + 2] - t[o2 * 3 + 2]);
    // the query side is owned by this thread within this launch; launches are stream-ordered
    if (gq) {
        gq[o * 3 + 0] += dx;
        gq[o * 3 + 1] += dy;
        gq[o * 3 + 2] += dz;
    }
    if (gt) {
        unsafeAtomicAdd(&gt[o2 * 3 + 0], -dx);
        unsafeAtomicAdd(&gt[o2 * 3 + 1], -dy);
        unsafeAtomicAdd(&gt[o2 * 3 + 2], -dz);
    }
}

struct NNPlan {
    int Q, qblocks, total_chunks, nslices, chunks_per_slice;
};

NNPlan plan_nn(int B, int n, int m)
{
    NNPlan p;
    p.Q = (long)B * n >= 32768 ? 2 : 1;
    p.qblocks = psi_cdiv(n, BLK * p.Q);
    p.total_chunks = psi_cdiv(m, CH);
    long blocks = (long)p.qblocks * B;
    int want = (int)((2048 + blocks - 1) / blocks);     // >= 8 workgroups per CU on 256 CUs
    int ns = want < 1 ? 1 : want;
    if (ns > p.total_chunks) ns = p.total_chunks;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    p.chunks_per_slice = psi_cdiv(p.total_chunks, ns);
    p.nslices = psi_cdiv(p.total_chunks, p.chunks_per_slice);
    return p;
}

size_t nn_ws_bytes(int B, int n, int m)
{
    NNPlan p = plan_nn(B, n, m);
    return (size_t)p.nslices * B * n * 8;
}

int launch_nn_ex(const float *q, const float *t, int B, int n, int m, float *dist, int32_t *idx, void *ws, hipStream_t st,
                 const int *qidx, long qstride, long tstride, bool contact, float cconst, float gscale, float *gq, float *fpart)
{
    NNPlan p = plan_nn(B, n, m);
    float *pd = (float *)ws;
    int *pc = (int *)(pd + (size_t)p.nslices * B * n);
    dim3 grid(p.qblocks, p.nslices, B);
    if (p.Q == 2)
        hipLaunchKernelGGL(nn_partial_kernel<2>, grid, dim3(BLK), 0, st, q, t, B, n, m, p.total_chunks, p.chunks_per_slice, pd, pc,
                           qidx, qstride, tstride);
    else
        hipLaunchKernelGGL(nn_partial_kernel<1>, grid, dim3(BLK), 0, st, q, t, B, n, m, p.total_chunks, p.chunks_per_slice, pd, pc,
                           qidx, qstride, tstride);
    PSI_CHECK_LAUNCH("nn_partial_kernel");
    psi_mark("nn_partial_kernel", st);
    dim3 rg(psi_cdiv(n, BLK), B);
    if (contact)
        hipLaunchKernelGGL(nn_resolve_kernel<true>, rg, dim3(BLK), 0, st, q, t, B, n, m, p.nslices, pd, pc, dist, idx, qidx, qstride,
                           tstride, cconst, gscale, gq, fpart);
    else
        hipLaunchKernelGGL(nn_resolve_kernel<false>, rg, dim3(BLK), 0, st, q, t, B, n, m, p.nslices, pd, pc, dist, idx, qidx, qstride,
                           tstride, 0.0f, 0.0f, nullptr, nullptr);
    PSI_CHECK_LAUNCH("nn_resolve_kernel");
    psi_mark("nn_resolve_kernel", st);
    return 0;
}

int launch_nn(const float *q, const float *t, int B, int n, int m, float *dist, int32_t *idx, void *ws, hipStream_t st)
{
    return launch_nn_ex(q, t, B, n, m, dist, idx, ws, st, nullptr, (long)n * 3, (long)m * 3, false, 0, 0, nullptr, nullptr);
}

}  // namespace

// internal (psi_internal.h): contact-loss NN for the fused fitting engine — queries gathered from the vertex table,
// one scene cloud shared by all bodies, loss epilogue fused into the resolve pass.
int psi_nn_contact(const float *verts, long vstride, const int *vid, const float *scene, int B, int n, int m, void *ws,
                   float cconst, float gscale, float *gq, float *fpart, int *idx_out, hipStream_t st)
{
    return launch_nn_ex(verts, scene, B, n, m, nullptr, idx_out, ws, st, vid, vstride, 0, true, cconst, gscale, gq, fpart);
}
int psi_nn_contact_fparts(int n) { return psi_cdiv(n, BLK); }
size_t psi_nn_ws_bytes(int B, int n, int m) { return nn_ws_bytes(B, n, m); }

extern "C" size_t psi_chamfer_workspace_bytes(int B, int n, int m)
{
    if (B <= 0 || n <= 0 || m <= 0) return 0;
    size_t a = nn_ws_bytes(B, n, m), b = nn_ws_bytes(B, m, n);
    return a > b ? a : b;
}

extern "C" int psi_chamfer_forward(const float *xyz1, const float *xyz2, int B, int n, int m,
                                   float *dist1, int32_t *idx1, float *dist2, int32_t *idx2,
                                   void *workspace, void *stream)
{
    PSI_REQUIRE(B >= 0 && n >= 0 && m >= 0, "negative size");
    PSI_REQUIRE((dist2 == nullptr) == (idx2 == nullptr), "dist2 and idx2 must both be given or both be NULL");
    if (B == 0 || n == 0 || m == 0) return 0;   // reference leaves the zero-filled outputs untouched
    PSI_REQUIRE(xyz1 && xyz2 && dist1 && idx1, "null pointer");
    PSI_REQUIRE(B <= 65535, "B exceeds grid.z");
    hipStream_t st = (hipStream_t)stream;
    void *ws = workspace;
    if (!ws) {
        ws = psi_scratch(psi_chamfer_workspace_bytes(B, n, m), st);
        if (!ws) return PSI_ENOMEM;
    }
    int rc = launch_nn(xyz1, xyz2, B, n, m, dist1, idx1, ws, st);
    if (rc) return rc;
    if (dist2) rc = launch_nn(xyz2, xyz1, B, m, n, dist2, idx2, ws, st);   // stream-ordered: ws reuse is safe
    return rc;
}

extern "C" int psi_chamfer_backward(const float *xyz1, const float *xyz2, float *gradxyz1, float *gradxyz2,
                                    const float *graddist1, const float *graddist2,
                                    const int32_t *idx1, const int32_t *idx2, int B, int n, int m, void *stream)
{
    PSI_REQUIRE(B >= 0 && n >= 0 && m >= 0, "negative size");
    if (B == 0 || n == 0 || m == 0) return 0;
    PSI_REQUIRE(xyz1 && xyz2 && gradxyz1 && graddist1 && idx1, "null pointer");
    PSI_REQUIRE((graddist2 == nullptr) == (idx2 == nullptr), "graddist2 and idx2 must both be given or both be NULL");
    PSI_REQUIRE(B <= 65535, "B exceeds grid.y");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(nn_grad_kernel, dim3(psi_cdiv(n, BLK), B), dim3(BLK), 0, st, xyz1, xyz2, graddist1, idx1, n, m, gradxyz1, gradxyz2);
    PSI_CHECK_LAUNCH("nn_grad_kernel");
    psi_mark("nn_grad_kernel", st);
    if (graddist2) {
        // direction 2 (chamfer.cu:185): own side is gradxyz2 (skipped when NULL), scatter side is gradxyz1
        hipLaunchKernelGGL(nn_grad_kernel, dim3(psi_cdiv(m, BLK), B), dim3(BLK), 0, st, xyz2, xyz1, graddist2, idx2, m, n, gradxyz2, gradxyz1);
        PSI_CHECK_LAUNCH("nn_grad_kernel(dir2)");
    psi_mark("nn_grad_kernel", st);
    }
    return 0;
}
