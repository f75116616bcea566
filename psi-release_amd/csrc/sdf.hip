// Trilinear SDF lookup + analytic gradient for gfx950, replacing the reference's
//   F.grid_sample(sdf[B,1,D,D,D], norm[:,:,[2,1,0]].view(-1,V,1,1,3), padding_mode='border')
// (fitting_proxe.py:144-151) and the mask/mean of fitting_proxe.py:155-158.
//
// One volume per SCENE ([S,D,D,D], selected by scene_id[b]) instead of the reference's per-sample
// replica (fitting_proxe.py:90): at B=32, D=256 that is 67 MB resident (fits the 256 MiB Infinity
// Cache) instead of 2.1 GB streamed.  The kernel is a latency/L2-gather kernel: 12 B in, 8 gathers,
// 4+12 B out per vertex; one thread per vertex, 8 independent loads in flight per lane.
// The gradient d sdf/d vert is produced in the SAME pass (the 8 corner values are already in
// registers), so backward never touches the volume again: grad_verts += grad_sdf * out_grad.
#include "psi_common.h"
#include <math.h>

namespace {

constexpr int BLK = 256;

struct Axis {
    int i0, i1;
    float w1, du;   // weight of the upper corner; d(u)/d(vert) (0 when clamped by the border rule)
};

__device__ __forceinline__ Axis axis_setup(float v, float mn, float mx, int D, int align_corners)
{
    // fitting_proxe.py:147: (v - min) / (max - min) * 2 - 1, in this operation order
    float nrm = (v - mn) / (mx - mn) * 2.0f - 1.0f;
    float u, scale;
    if (align_corners) {
        u = (nrm + 1.0f) / 2.0f * (float)(D - 1);
        scale = (float)(D - 1) / 2.0f;
    } else {
        u = ((nrm + 1.0f) * (float)D - 1.0f) / 2.0f;
        scale = (float)D / 2.0f;
    }
    // padding_mode='border': clip to [0, D-1]; the clipped coordinate has zero gradient
    float g = scale;
    if (!(u > 0.0f)) { u = 0.0f; g = 0.0f; }
    else if (u >= (float)(D - 1)) { u = (float)(D - 1); g = 0.0f; }
    float fl = floorf(u);
    Axis a;
    a.i0 = (int)fl;
    a.w1 = u - fl;
    a.i1 = min(a.i0 + 1, D - 1);     // upper corner of the last cell has weight 0; clamp keeps the load in bounds
    a.du = g * 2.0f / (mx - mn);
    return a;
}

__global__ __launch_bounds__(BLK) void sdf_sample_kernel(const float *__restrict__ sdf, const int *__restrict__ scene_id,
                                                         const float *__restrict__ gmin, const float *__restrict__ gmax,
                                                         const float *__restrict__ verts, int V, int D, int align_corners,
                                                         float *__restrict__ out, float *__restrict__ out_grad)
{
    const int b = blockIdx.y;
    const int v = blockIdx.x * BLK + threadIdx.x;
    if (v >= V) return;
    const int s = scene_id ? scene_id[b] : 0;
    const float *__restrict__ vol = sdf + (size_t)s * D * D * D;
    const size_t o = (size_t)b * V + v;
    Axis ax = axis_setup(verts[o * 3 + 0], gmin[s * 3 + 0], gmax[s * 3 + 0], D, align_corners);
    Axis ay = axis_setup(verts[o * 3 + 1], gmin[s * 3 + 1], gmax[s * 3 + 1], D, align_corners);
    Axis az = axis_setup(verts[o * 3 + 2], gmin[s * 3 + 2], gmax[s * 3 + 2], D, align_corners);
    const size_t x0 = (size_t)ax.i0 * D, x1 = (size_t)ax.i1 * D;
    const size_t r00 = (x0 + ay.i0) * D, r01 = (x0 + ay.i1) * D, r10 = (x1 + ay.i0) * D, r11 = (x1 + ay.i1) * D;
    // 8 gathers; the two z-neighbours of each row are adjacent dwords
    float c000 = vol[r00 + az.i0], c001 = vol[r00 + az.i1];
    float c010 = vol[r01 + az.i0], c011 = vol[r01 + az.i1];
    float c100 = vol[r10 + az.i0], c101 = vol[r10 + az.i1];
    float c110 = vol[r11 + az.i0], c111 = vol[r11 + az.i1];
    const float wx1 = ax.w1, wx0 = 1.0f - ax.w1;
    const float wy1 = ay.w1, wy0 = 1.0f - ay.w1;
    const float wz1 = az.w1, wz0 = 1.0f - az.w1;
    // interpolate along z, then y, then x
    float c00 = c000 * wz0 + c001 * wz1, c01 = c010 * wz0 + c011 * wz1;
    float c10 = c100 * wz0 + c101 * wz1, c11 = c110 * wz0 + c111 * wz1;
    float c0 = c00 * wy0 + c01 * wy1, c1 = c10 * wy0 + c11 * wy1;
    out[o] = c0 * wx0 + c1 * wx1;
    if (out_grad) {
        float gx = c1 - c0;
        float gy = (c01 - c00) * wx0 + (c11 - c10) * wx1;
        float d00 = c001 - c000, d01 = c011 - c010, d10 = c101 - c100, d11 = c111 - c110;
        float gz = (d00 * wy0 + d01 * wy1) * wx0 + (d10 * wy0 + d11 * wy1) * wx1;
        out_grad[o * 3 + 0] = gx * ax.du;
        out_grad[o * 3 + 1] = gy * ay.du;
        out_grad[o * 3 + 2] = gz * az.du;
    }
}

__global__ __launch_bounds__(BLK) void sdf_backward_kernel(const float *__restrict__ gs, const float *__restrict__ og,
                                                           long n, float *__restrict__ gv)
{
    long i = (long)blockIdx.x * BLK + threadIdx.x;
    if (i >= n) return;
    float g = gs[i];
    gv[i * 3 + 0] += g * og[i * 3 + 0];
    gv[i * 3 + 1] += g * og[i * 3 + 1];
    gv[i * 3 + 2] += g * og[i * 3 + 2];
}

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
    return x;
}

__global__ __launch_bounds__(BLK) void pen_stats_kernel(const float *__restrict__ vals, long n, float *stats)
{
    float s = 0.0f, c = 0.0f;
    for (long i = (long)blockIdx.x * BLK + threadIdx.x; i < n; i += (long)gridDim.x * BLK) {
        float x = vals[i];
        if (x < 0.0f) {
            s -= x;
            c += 1.0f;
        }
    }
    s = wave_sum(s);
    c = wave_sum(c);
    __shared__ float sh[2][BLK / 64];
    int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[0][w] = s;
        sh[1][w] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0, tc = 0;
        for (int i = 0; i < BLK / 64; i++) {
            ts += sh[0][i];
            tc += sh[1][i];
        }
        unsafeAtomicAdd(&stats[0], ts);
        unsafeAtomicAdd(&stats[1], tc);
    }
}

}  // namespace

extern "C" int psi_sdf_sample_forward(const float *sdf, const int32_t *scene_id, const float *gmin, const float *gmax,
                                      const float *verts, int B, int V, int D, int S, int align_corners,
                                      float *out_sdf, float *out_grad, void *stream)
{
    PSI_REQUIRE(B >= 0 && V >= 0, "negative size");
    if (B == 0 || V == 0) return 0;
    PSI_REQUIRE(sdf && gmin && gmax && verts && out_sdf, "null pointer");
    PSI_REQUIRE(D >= 2 && S >= 1, "grid dim must be >= 2 and at least one scene");
    PSI_REQUIRE(B <= 65535, "B exceeds grid.y");
    hipLaunchKernelGGL(sdf_sample_kernel, dim3(psi_cdiv(V, BLK), B), dim3(BLK), 0, (hipStream_t)stream,
                       sdf, scene_id, gmin, gmax, verts, V, D, align_corners, out_sdf, out_grad);
    PSI_CHECK_LAUNCH("sdf_sample_kernel");
    return 0;
}

extern "C" int psi_sdf_sample_backward(const float *grad_sdf, const float *out_grad, int B, int V,
                                       float *grad_verts, void *stream)
{
    PSI_REQUIRE(B >= 0 && V >= 0, "negative size");
    long n = (long)B * V;
    if (n == 0) return 0;
    PSI_REQUIRE(grad_sdf && out_grad && grad_verts, "null pointer");
    hipLaunchKernelGGL(sdf_backward_kernel, dim3(psi_cdiv(n, BLK)), dim3(BLK), 0, (hipStream_t)stream,
                       grad_sdf, out_grad, n, grad_verts);
    PSI_CHECK_LAUNCH("sdf_backward_kernel");
    return 0;
}

extern "C" int psi_sdf_penetration_stats(const float *sdf_vals, long n, float *stats, void *stream)
{
    PSI_REQUIRE(n >= 0, "negative size");
    if (n == 0) return 0;
    PSI_REQUIRE(sdf_vals && stats, "null pointer");
    int blocks = psi_cdiv(n, BLK * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(pen_stats_kernel, dim3(blocks), dim3(BLK), 0, (hipStream_t)stream, sdf_vals, n, stats);
    PSI_CHECK_LAUNCH("pen_stats_kernel");
    return 0;
}
