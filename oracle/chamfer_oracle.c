/*
 * TEST INFRASTRUCTURE — CPU restatement of the reference's Chamfer nearest-neighbour op.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object; the product path (psi-release_amd/) never does.
 *
 * Reference: /root/reference/chamfer_pytorch/chamfer.cu (CUDA; cannot be compiled here:
 * no nvcc, no GPU; hipify is out of bounds for this build).  This file restates
 *   - NmDistanceKernel      chamfer.cu:12-134   -> psi_oracle_nm_distance
 *   - NmDistanceGradKernel  chamfer.cu:155-174  -> psi_oracle_nm_distance_grad
 *   - chamfer_cuda_forward  chamfer.cu:136-154  -> psi_oracle_chamfer_forward  (two directions)
 *   - chamfer_cuda_backward chamfer.cu:176-196  -> psi_oracle_chamfer_backward
 *
 * Arithmetic contract: d = x2*x2 + y2*y2 + z2*z2 with
 * (x2,y2,z2) = target - query, fp32, evaluated left to right, NO fused multiply-add
 * (compile with -ffp-contract=off; nvcc's contraction of the original is unknowable, the build
 * defines this one).  Winner = lowest target index among minimal d: inside a 512-target chunk the
 * reference keeps the first minimum (strict `d<best`, chamfer.cu:46,55,64,73,121) and across chunks
 * the earlier chunk (strict `result>best`, chamfer.cu:126), which together equal an ascending scan
 * with strict `<`.  psi_oracle_nm_distance_chunked spells the chunked form out literally so a test
 * can show the two agree.
 *
 * Parity pin: chamfer_pytorch/test_chamfer.py:35-54 (expanded-form brute force, sum of squared
 * differences < 1e-8 on rand(4,100,3)) is reproduced in tests/test_oracle_cpu.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>

#ifdef _OPENMP
#include <omp.h>
#endif
/* thread count of the OpenMP loops (bench.py sweeps it to report the best CPU configuration) */
void psi_oracle_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* The distance expression in the two arithmetic modes of the build (psi-release_amd/csrc/psi_common.h has the same macro):
 * default = the CUDA source as written with every product and sum rounded (nvcc --fmad=false); -DPSI_CHAMFER_FMA = mul, fma, fma,
 * what nvcc's default --fmad=true makes of chamfer.cu:32-35 (each `product + sum` pair contracted, left to right). */
#ifdef PSI_CHAMFER_FMA
#define PSI_SQ3(x, y, z) fmaf((z), (z), fmaf((y), (y), (x) * (x)))
#else
#define PSI_SQ3(x, y, z) ((x) * (x) + (y) * (y) + (z) * (z))
#endif

#define QB 16 /* queries per SIMD block: the inner loop over queries vectorises (fair CPU baseline) */

/* chamfer.cu:12-134 (one direction): for every query j of batch i, min over targets k of the
 * squared distance and the index of the first minimiser. */
void psi_oracle_nm_distance(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int32_t *result_i)
{
    long nblk = ((long)n + QB - 1) / QB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int i = 0; i < b; i++) {
        for (long blk = 0; blk < nblk; blk++) {
            float qx[QB], qy[QB], qz[QB], best[QB];
            int32_t besti[QB];
            int j0 = (int)(blk * QB);
            int cnt = n - j0 < QB ? n - j0 : QB;
            for (int q = 0; q < QB; q++) {
                int j = j0 + (q < cnt ? q : 0);
                qx[q] = xyz[((size_t)i * n + j) * 3 + 0];
                qy[q] = xyz[((size_t)i * n + j) * 3 + 1];
                qz[q] = xyz[((size_t)i * n + j) * 3 + 2];
                best[q] = 0.0f;
                besti[q] = 0;
            }
            const float *t = xyz2 + (size_t)i * m * 3;
            for (int k = 0; k < m; k++) {
                float tx = t[k * 3 + 0], ty = t[k * 3 + 1], tz = t[k * 3 + 2];
#pragma omp simd
                for (int q = 0; q < QB; q++) {
                    float x2 = tx - qx[q];
                    float y2 = ty - qy[q];
                    float z2 = tz - qz[q];
                    float d = PSI_SQ3(x2, y2, z2);
                    int take = (k == 0) | (d < best[q]);
                    best[q] = take ? d : best[q];
                    besti[q] = take ? k : besti[q];
                }
            }
            for (int q = 0; q < cnt; q++) {
                result[(size_t)i * n + j0 + q] = best[q];
                result_i[(size_t)i * n + j0 + q] = besti[q];
            }
        }
    }
}

/* Literal chunked form of chamfer.cu:12-134 (512-target chunks, running minimum kept in result[]
 * with strict '>' across chunks).  Scalar, slow; exists to prove equivalence with the scan above. */
void psi_oracle_nm_distance_chunked(int b, int n, const float *xyz, int m, const float *xyz2, float *result, int32_t *result_i)
{
    const int batch = 512;
    for (int i = 0; i < b; i++) {
        for (int k2 = 0; k2 < m; k2 += batch) {
            int end_k = (m < k2 + batch ? m : k2 + batch) - k2;
            const float *buf = xyz2 + ((size_t)i * m + k2) * 3;
            for (int j = 0; j < n; j++) {
                float x1 = xyz[((size_t)i * n + j) * 3 + 0];
                float y1 = xyz[((size_t)i * n + j) * 3 + 1];
                float z1 = xyz[((size_t)i * n + j) * 3 + 2];
                int best_i = 0;
                float best = 0;
                for (int k = 0; k < end_k; k++) {
                    float x2 = buf[k * 3 + 0] - x1;
                    float y2 = buf[k * 3 + 1] - y1;
                    float z2 = buf[k * 3 + 2] - z1;
                    float d = PSI_SQ3(x2, y2, z2);
                    if (k == 0 || d < best) {
                        best = d;
                        best_i = k + k2;
                    }
                }
                if (k2 == 0 || result[(size_t)i * n + j] > best) {
                    result[(size_t)i * n + j] = best;
                    result_i[(size_t)i * n + j] = best_i;
                }
            }
        }
    }
}

/* chamfer.cu:136-154: both directions. dist2/idx2 may be NULL (PSI discards them, fitting_proxe.py:136). */
int psi_oracle_chamfer_forward(const float *xyz1, const float *xyz2, int b, int n, int m,
                               float *dist1, int32_t *idx1, float *dist2, int32_t *idx2)
{
    psi_oracle_nm_distance(b, n, xyz1, m, xyz2, dist1, idx1);
    if (dist2 && idx2)
        psi_oracle_nm_distance(b, m, xyz2, n, xyz1, dist2, idx2);
    return 1;
}

/* chamfer.cu:155-174: scatter/gather gradient of one direction (accumulates into the outputs). */
void psi_oracle_nm_distance_grad(int b, int n, const float *xyz1, int m, const float *xyz2,
                                 const float *grad_dist1, const int32_t *idx1, float *grad_xyz1, float *grad_xyz2)
{
    for (int i = 0; i < b; i++) {
        for (int j = 0; j < n; j++) {
            float x1 = xyz1[((size_t)i * n + j) * 3 + 0];
            float y1 = xyz1[((size_t)i * n + j) * 3 + 1];
            float z1 = xyz1[((size_t)i * n + j) * 3 + 2];
            int j2 = idx1[(size_t)i * n + j];
            float x2 = xyz2[((size_t)i * m + j2) * 3 + 0];
            float y2 = xyz2[((size_t)i * m + j2) * 3 + 1];
            float z2 = xyz2[((size_t)i * m + j2) * 3 + 2];
            float g = grad_dist1[(size_t)i * n + j] * 2;
            grad_xyz1[((size_t)i * n + j) * 3 + 0] += g * (x1 - x2);
            grad_xyz1[((size_t)i * n + j) * 3 + 1] += g * (y1 - y2);
            grad_xyz1[((size_t)i * n + j) * 3 + 2] += g * (z1 - z2);
            grad_xyz2[((size_t)i * m + j2) * 3 + 0] += -(g * (x1 - x2));
            grad_xyz2[((size_t)i * m + j2) * 3 + 1] += -(g * (y1 - y2));
            grad_xyz2[((size_t)i * m + j2) * 3 + 2] += -(g * (z1 - z2));
        }
    }
}

/* chamfer.cu:176-196: caller passes zero-filled grads (dist_chamfer.py:40-45). */
int psi_oracle_chamfer_backward(const float *xyz1, const float *xyz2, float *gradxyz1, float *gradxyz2,
                                const float *graddist1, const float *graddist2,
                                const int32_t *idx1, const int32_t *idx2, int b, int n, int m)
{
    psi_oracle_nm_distance_grad(b, n, xyz1, m, xyz2, graddist1, idx1, gradxyz1, gradxyz2);
    if (graddist2 && idx2)
        psi_oracle_nm_distance_grad(b, m, xyz2, n, xyz1, graddist2, idx2, gradxyz2, gradxyz1);
    return 1;
}

/*
 * Trilinear SDF sample restated from the formulas of torch's grid_sample (5-D, bilinear mode,
 * padding_mode='border'), call site fitting_proxe.py:144-151.  Used to cross-check F.grid_sample in
 * the oracle tests and as a scalar CPU statement of Appendix C; the Python oracle calls
 * F.grid_sample itself.  verts [B,V,3] world coordinates, sdf [S,D,D,D] indexed [ix][iy][iz],
 * out [B,V]; grad_out [B,V,3] = d(sdf)/d(vert) (may be NULL).
 */
void psi_oracle_sdf_sample(const float *sdf, const int32_t *scene_id, const float *gmin, const float *gmax,
                           const float *verts, int B, int V, int D, int align_corners,
                           float *out, float *grad_out)
{
#pragma omp parallel for schedule(static)
    for (long t = 0; t < (long)B * V; t++) {
        int b = (int)(t / V);
        int s = scene_id ? scene_id[b] : 0;
        const float *vol = sdf + (size_t)s * D * D * D;
        float u[3], du[3];
        int i0[3], i1[3];
        float w1[3];
        for (int a = 0; a < 3; a++) {
            float mn = gmin[s * 3 + a], mx = gmax[s * 3 + a];
            float nrm = (verts[t * 3 + a] - mn) / (mx - mn) * 2.0f - 1.0f; /* fitting_proxe.py:147 */
            float uu, scale;
            if (align_corners) { uu = (nrm + 1.0f) / 2.0f * (float)(D - 1); scale = (float)(D - 1) / 2.0f; }
            else { uu = ((nrm + 1.0f) * (float)D - 1.0f) / 2.0f; scale = (float)D / 2.0f; }
            float g = scale;
            if (uu <= 0.0f) { uu = 0.0f; g = 0.0f; }            /* border: clip_coordinates_set_grad */
            else if (uu >= (float)(D - 1)) { uu = (float)(D - 1); g = 0.0f; }
            u[a] = uu;
            du[a] = g * 2.0f / (mx - mn);
            float fl = __builtin_floorf(uu);
            i0[a] = (int)fl;
            i1[a] = i0[a] + 1;
            w1[a] = uu - fl;
        }
        float val = 0.0f, gx = 0.0f, gy = 0.0f, gz = 0.0f;
        for (int cx = 0; cx < 2; cx++)
            for (int cy = 0; cy < 2; cy++)
                for (int cz = 0; cz < 2; cz++) {
                    int ix = cx ? i1[0] : i0[0], iy = cy ? i1[1] : i0[1], iz = cz ? i1[2] : i0[2];
                    if (ix > D - 1 || iy > D - 1 || iz > D - 1) continue; /* out-of-bound corner contributes 0 */
                    float wx = cx ? w1[0] : 1.0f - w1[0];
                    float wy = cy ? w1[1] : 1.0f - w1[1];
                    float wz = cz ? w1[2] : 1.0f - w1[2];
                    float sv = vol[((size_t)ix * D + iy) * D + iz];
                    val += sv * wx * wy * wz;
                    gx += sv * (cx ? 1.0f : -1.0f) * wy * wz;
                    gy += sv * (cy ? 1.0f : -1.0f) * wx * wz;
                    gz += sv * (cz ? 1.0f : -1.0f) * wx * wy;
                }
        out[t] = val;
        if (grad_out) {
            grad_out[t * 3 + 0] = gx * du[0];
            grad_out[t * 3 + 1] = gy * du[1];
            grad_out[t * 3 + 2] = gz * du[2];
        }
    }
}
