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
#include "sdf_device.h"
#include <math.h>

namespace {

constexpr int BLK = 256;

__global__ __launch_bounds__(BLK) void sdf_sample_kernel(const float *__restrict__ sdf, const int *__restrict__ scene_id,
                                                         const float *__restrict__ gmin, const float *__restrict__ gmax,
                                                         const float *__restrict__ verts, int V, int D, int align_corners,
                                                         float *__restrict__ out, float *__restrict__ out_grad)
{
    const int b = blockIdx.y;
    const int v = blockIdx.x * BLK + threadIdx.x;
    if (v >= V) return;
    const int s = scene_id ? scene_id[b] : 0;
    const size_t o = (size_t)b * V + v;
    float g[3];
    out[o] = psi_trilinear(sdf + (size_t)s * D * D * D, gmin + s * 3, gmax + s * 3, verts[o * 3 + 0], verts[o * 3 + 1],
                           verts[o * 3 + 2], D, align_corners, out_grad ? g : nullptr);
    if (out_grad) {
        out_grad[o * 3 + 0] = g[0];
        out_grad[o * 3 + 1] = g[1];
        out_grad[o * 3 + 2] = g[2];
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

// per-block partial sums in a fixed grid, then ONE block adds them in block order: the same bits on every run (the first version added
// the blocks' sums to stats[] with floating-point atomics, i.e. in arrival order)
constexpr int PEN_BLOCKS = 256;

__global__ __launch_bounds__(BLK) void pen_stats_partial_kernel(const float *__restrict__ vals, long n, float *__restrict__ part /* [PEN_BLOCKS][2] */)
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
        part[blockIdx.x * 2 + 0] = ts;
        part[blockIdx.x * 2 + 1] = tc;
    }
}

__global__ __launch_bounds__(64) void pen_stats_finalize_kernel(const float *__restrict__ part, int nblocks, float *__restrict__ stats)
{
    const int t = threadIdx.x;
    if (t >= 2) return;
    float v[PEN_BLOCKS];
#pragma unroll
    for (int b = 0; b < PEN_BLOCKS; b++) v[b] = b < nblocks ? part[b * 2 + t] : 0.0f;      // all loads first, then the sum in block order
    float s = 0.0f;
#pragma unroll
    for (int b = 0; b < PEN_BLOCKS; b++) s += v[b];
    stats[t] += s;                                                                          // the caller zeroes stats (historic contract)
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
    if (blocks > PEN_BLOCKS) blocks = PEN_BLOCKS;
    float *part = (float *)psi_scratch((size_t)PEN_BLOCKS * 2 * sizeof(float), (hipStream_t)stream);
    PSI_REQUIRE(part != nullptr, "scratch allocation failed");
    hipLaunchKernelGGL(pen_stats_partial_kernel, dim3(blocks), dim3(BLK), 0, (hipStream_t)stream, sdf_vals, n, part);
    PSI_CHECK_LAUNCH("pen_stats_partial_kernel");
    hipLaunchKernelGGL(pen_stats_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, part, blocks, stats);
    PSI_CHECK_LAUNCH("pen_stats_finalize_kernel");
    return 0;
}
