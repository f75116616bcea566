// The tail of the two scene losses of a training step — contact and penetration — as four launches instead of ~45 operators, gfx950.
//
// train_s1.py:156-204 / train_s2.py:159-202 (cal_loss), after the body mesh, the body->scene nearest-neighbour distances dist [B, n_c] of
// the contact vertices (chamfer dist1) and the trilinear SDF values sdf [B, V] are known:
//     contact = gate * w_contact   * mean( s / (s + 1) ),   s = sqrt(dist + 1e-4)                        train_s1.py:171-177
//     pene    = gate * w_collision * ( mean |sdf| over sdf < 0,  0 when no vertex penetrates )           train_s1.py:193-204
// and the gradient of both with respect to the body vertices [B, V, 3]:
//     penetration: -(gate w_collision / count) * [sdf < 0] * d sdf / d vertex        (every vertex: a dense write, no zero fill)
//     contact    : (gate w_contact / (B n_c)) * 1 / (2 s (s + 1)^2) * 2 (x - nn(x))  added to the contact vertices' rows
//                  (chamfer.cu:155-174, query side).  A vertex listed by two contact parts accumulates both (cvae.py:99-115 keeps
//                  duplicates): the slots of one vertex form a chain in ascending slot order, built by a small kernel per call, and the
//                  FIRST slot's thread adds the whole chain to the row — no atomics, the same bits on every run
// As PyTorch operators that is: a sqrt / add / div / mean chain and its autograd twin, torch.where pieces around the penetration mean,
// an advanced-indexing gather for the neighbours, an index_put_(accumulate=True) for the contact rows (61 us at batch 128: it sorts), a
// 16 MB zero fill and a 16 MB add of the two vertex gradients.
// Reductions: per-block partials in a fixed grid, summed in block order by a one-block kernel — deterministic.
#include "psi_internal.h"

namespace {

constexpr int SL_BLK = 256;
constexpr int SL_GRID = 128;            // blocks of the partial-sum kernel (also the partial count the finalize kernel reads)

__global__ __launch_bounds__(SL_BLK) void scene_loss_partial_kernel(const float *__restrict__ dist, long n_contact, const float *__restrict__ sdf,
                                                                    long n_sdf, float *__restrict__ part /* [SL_GRID][3] */)
{
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    const long stride = (long)SL_GRID * SL_BLK, t0 = (long)blockIdx.x * SL_BLK + threadIdx.x;
    for (long i = t0; i < n_contact; i += stride) {
        const float s = sqrtf(dist[i] + 1e-4f);
        a0 += s / (s + 1.0f);
    }
    // four independent loads per round: the [B, V] values are a 5 MB stream at batch 128
    long i = t0;
    for (; i + 3 * stride < n_sdf; i += 4 * stride) {
        const float v0 = sdf[i], v1 = sdf[i + stride], v2 = sdf[i + 2 * stride], v3 = sdf[i + 3 * stride];
        a1 += (v0 < 0.0f ? -v0 : 0.0f) + (v1 < 0.0f ? -v1 : 0.0f) + (v2 < 0.0f ? -v2 : 0.0f) + (v3 < 0.0f ? -v3 : 0.0f);
        a2 += (v0 < 0.0f ? 1.0f : 0.0f) + (v1 < 0.0f ? 1.0f : 0.0f) + (v2 < 0.0f ? 1.0f : 0.0f) + (v3 < 0.0f ? 1.0f : 0.0f);
    }
    for (; i < n_sdf; i += stride) {
        const float v = sdf[i];
        a1 += v < 0.0f ? -v : 0.0f;
        a2 += v < 0.0f ? 1.0f : 0.0f;
    }
    __shared__ float sh[3][SL_BLK / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_down(a0, o, 64);
        a1 += __shfl_down(a1, o, 64);
        a2 += __shfl_down(a2, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = a0;
        sh[1][threadIdx.x >> 6] = a1;
        sh[2][threadIdx.x >> 6] = a2;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0.0f;
        for (int w = 0; w < SL_BLK / 64; w++) s += sh[threadIdx.x][w];
        part[blockIdx.x * 3 + threadIdx.x] = s;
    }
}

// losses[0] = contact, losses[1] = penetration; stats[0] = sum |sdf| over sdf < 0, stats[1] = their count (kept for the backward)
__global__ __launch_bounds__(64) void scene_loss_finalize_kernel(const float *__restrict__ part, long n_contact, float w_contact, float w_collision,
                                                                 float gate, float *__restrict__ losses, float *__restrict__ stats)
{
    const int t = threadIdx.x;
    if (t >= 3) return;
    float v[SL_GRID];
#pragma unroll
    for (int b = 0; b < SL_GRID; b++) v[b] = part[b * 3 + t];          // all loads first, then the sum in block order
    float s = 0.0f;
#pragma unroll
    for (int b = 0; b < SL_GRID; b++) s += v[b];
    if (t == 0) losses[0] = gate * w_contact * (s / (float)n_contact);
    if (t >= 1) stats[t - 1] = s;
    const float cnt = __shfl(s, 2, 64), sum = __shfl(s, 1, 64);
    if (t == 1) losses[1] = gate * w_collision * (cnt > 0.0f ? sum / fmaxf(cnt, 1.0f) : 0.0f);
}

// dense part of the vertex gradient: every (b, v) is written
__global__ __launch_bounds__(SL_BLK) void scene_loss_bwd_dense_kernel(const float *__restrict__ g_losses, const float *__restrict__ stats,
                                                                      const float *__restrict__ sdf, const float *__restrict__ og, long n_sdf,
                                                                      float w_collision, float gate, float *__restrict__ g_verts)
{
    const long i = (long)blockIdx.x * SL_BLK + threadIdx.x;
    if (i >= n_sdf) return;
    const float cnt = stats[1];
    const float c = cnt > 0.0f ? -(g_losses[1] * (gate * w_collision)) / fmaxf(cnt, 1.0f) : 0.0f;
    const bool neg = sdf[i] < 0.0f;
    const float *o = og + i * 3;
    float *g = g_verts + i * 3;
    g[0] = neg ? c * o[0] : 0.0f;
    g[1] = neg ? c * o[1] : 0.0f;
    g[2] = neg ? c * o[2] : 0.0f;
}

// chain[j] = the next contact slot (> j) that lists the same vertex as slot j, or -1; chain[n + j] = 1 when no earlier slot lists it
__global__ __launch_bounds__(SL_BLK) void contact_chain_kernel(const int *__restrict__ vid, int n, int *__restrict__ chain)
{
    extern __shared__ int svid[];
    for (int i = threadIdx.x; i < n; i += SL_BLK) svid[i] = vid[i];
    __syncthreads();
    const int j = blockIdx.x * SL_BLK + threadIdx.x;
    if (j >= n) return;
    const int my = svid[j];
    int first = 1, next = -1;
    for (int k = 0; k < j; k++) first &= svid[k] != my;
    for (int k = n - 1; k > j; k--) next = svid[k] == my ? k : next;
    chain[j] = next;
    chain[n + j] = first;
}

// contact part: thread = (body, contact slot); the first slot of a vertex adds the terms of all of the vertex's slots to its row
__global__ __launch_bounds__(SL_BLK) void scene_loss_bwd_contact_kernel(const float *__restrict__ g_losses, const float *__restrict__ dist,
                                                                        const float *__restrict__ xyz1, const int *__restrict__ idx,
                                                                        const int *__restrict__ slot, const float *__restrict__ table, long m,
                                                                        const int *__restrict__ vid, const int *__restrict__ chain, int B, int n,
                                                                        int V, float w_contact, float gate, float *__restrict__ g_verts)
{
    const long i = (long)blockIdx.x * SL_BLK + threadIdx.x;
    if (i >= (long)B * n) return;
    const int b = (int)(i / n), j = (int)(i % n);
    if (!chain[n + j]) return;
    float *g = g_verts + ((size_t)b * V + vid[j]) * 3;
    float a0 = g[0], a1 = g[1], a2 = g[2];                    // the penetration part (scene_loss_bwd_dense_kernel, earlier on the stream)
    const float k0 = g_losses[0] * (gate * w_contact) / (float)((long)B * n);
    const float *pt = table + (size_t)slot[b] * m * 3;
    for (int jj = j; jj >= 0; jj = chain[jj]) {
        const long q_i = (long)b * n + jj;
        const float s = sqrtf(dist[q_i] + 1e-4f);
        const float sp = s + 1.0f;
        // d/d dist of mean(s / (s + 1)):  1 / (s + 1)^2 * 1 / (2 s) / (B n)
        const float c = k0 * (1.0f / (sp * sp)) * (0.5f / s);
        const float *q = xyz1 + q_i * 3;
        const float *p = pt + (size_t)idx[q_i] * 3;
        a0 += 2.0f * c * (q[0] - p[0]);
        a1 += 2.0f * c * (q[1] - p[1]);
        a2 += 2.0f * c * (q[2] - p[2]);
    }
    g[0] = a0; g[1] = a1; g[2] = a2;
}

} // namespace

extern "C" size_t psi_scene_losses_workspace_floats(void) { return (size_t)SL_GRID * 3; }

// The slot chain of a contact-id list (a constant of the model: built once per list, 100 us for 2048 slots — every thread scans the list)
extern "C" int psi_contact_slot_chain(const int32_t *vid, int n_c, int32_t *chain, void *stream)
{
    PSI_REQUIRE(vid && chain && n_c > 0, "bad arguments");
    PSI_REQUIRE(n_c <= 16384, "more than 16384 contact slots");          // the slot table is staged in 64 KB of LDS
    hipLaunchKernelGGL(contact_chain_kernel, dim3((unsigned)psi_cdiv(n_c, SL_BLK)), dim3(SL_BLK), (size_t)n_c * sizeof(int), (hipStream_t)stream, vid, n_c,
                       chain);
    PSI_CHECK_LAUNCH("contact_chain_kernel");
    return 0;
}

extern "C" int psi_scene_losses_forward(const float *dist, long n_contact, const float *sdf_vals, long n_sdf, float w_contact, float w_collision,
                                        float gate, float *ws, float *losses2, float *stats2, void *stream)
{
    PSI_REQUIRE(dist && sdf_vals && ws && losses2 && stats2, "null pointer");
    PSI_REQUIRE(n_contact > 0 && n_sdf > 0, "empty input");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(scene_loss_partial_kernel, dim3(SL_GRID), dim3(SL_BLK), 0, st, dist, n_contact, sdf_vals, n_sdf, ws);
    PSI_CHECK_LAUNCH("scene_loss_partial_kernel");
    hipLaunchKernelGGL(scene_loss_finalize_kernel, dim3(1), dim3(64), 0, st, ws, n_contact, w_contact, w_collision, gate, losses2, stats2);
    PSI_CHECK_LAUNCH("scene_loss_finalize_kernel");
    return 0;
}

extern "C" int psi_scene_losses_backward(const float *g_losses2, const float *stats2, const float *dist, const float *xyz1, const int32_t *idx,
                                         const int32_t *slot, const float *verts_table, long m, const int32_t *vid, const float *sdf_vals,
                                         const float *sdf_grad, int B, int V, int n_c, float w_contact, float w_collision, float gate,
                                         const int32_t *chain, float *g_verts, void *stream)
{
    PSI_REQUIRE(g_losses2 && stats2 && dist && xyz1 && idx && slot && verts_table && vid && sdf_vals && sdf_grad && chain && g_verts, "null pointer");
    PSI_REQUIRE(B > 0 && V > 0 && n_c > 0 && m > 0, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const long n_sdf = (long)B * V, n_q = (long)B * n_c;
    hipLaunchKernelGGL(scene_loss_bwd_dense_kernel, dim3((unsigned)psi_cdiv(n_sdf, (long)SL_BLK)), dim3(SL_BLK), 0, st, g_losses2, stats2, sdf_vals,
                       sdf_grad, n_sdf, w_collision, gate, g_verts);
    PSI_CHECK_LAUNCH("scene_loss_bwd_dense_kernel");
    hipLaunchKernelGGL(scene_loss_bwd_contact_kernel, dim3((unsigned)psi_cdiv(n_q, (long)SL_BLK)), dim3(SL_BLK), 0, st, g_losses2, dist, xyz1, idx,
                       slot, verts_table, m, vid, chain, B, n_c, V, w_contact, gate, g_verts);
    PSI_CHECK_LAUNCH("scene_loss_bwd_contact_kernel");
    return 0;
}
