/*
 * psi_hip.h — C ABI of libpsi_hip.so, the MI355X (gfx950) implementation of PSI's
 * generation-and-fitting hot path.  Plain pointers and sizes only; every pointer is DEVICE memory
 * unless the name starts with h_.  `stream` is a hipStream_t passed as void* (NULL = default stream).
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * yz-cnsdqz/PSI-release checkout).  How a maintainer binds these from the reference's Python is
 * shown in INTEGRATION.md; this build's own binding is psi-release_amd/hip.py (ctypes).
 *
 * Error convention: every function returns 0 on success, otherwise a hipError_t value (or a
 * negative PSI_E* code for argument errors); psi_last_error() returns a static message.  The
 * reference's pybind functions returned 1/0 and printed (chamfer.cu:145-151); callers ignored it.
 * No entry point synchronises the host; all work is enqueued on `stream`.
 */
#ifndef PSI_HIP_H
#define PSI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSI_EINVAL (-22)
#define PSI_ENOMEM (-12)

const char *psi_last_error(void);
int psi_version(void);
/* Device facts for the bench roofline: CU count, wave size, clock (kHz), 1 if the device is gfx950. */
int psi_device_info(int *cu_count, int *wave_size, int *clock_khz, int *is_gfx950);

/* ---------------------------------------------------------------------------------------------
 * Chamfer nearest neighbour — replaces the `chamfer` CUDA extension
 *   chamfer.forward (xyz1, xyz2, dist1, dist2, idx1, idx2)            chamfer_cuda.cpp:17-19,30-31
 *     -> chamfer_cuda_forward / NmDistanceKernel x2                  chamfer.cu:12-154
 *   chamfer.backward(xyz1, xyz2, gradxyz1, gradxyz2, graddist1, graddist2, idx1, idx2)
 *                                                                     chamfer_cuda.cpp:22-27,32
 *     -> chamfer_cuda_backward / NmDistanceGradKernel x2             chamfer.cu:155-196
 * xyz1 [B,n,3], xyz2 [B,m,3] contiguous fp32; dist1 [B,n], dist2 [B,m] fp32; idx1, idx2 int32.
 * dist = min_k (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 evaluated in fp32 left-to-right without FMA
 * contraction; idx = lowest k attaining it (first-minimum rule of chamfer.cu:46,126).
 * dist2/idx2 may both be NULL: PSI discards that direction (fitting_proxe.py:136) and it is skipped.
 * `workspace` (device, >= psi_chamfer_workspace_bytes) holds per-slice partial minima; NULL lets the
 * library use an internal per-device buffer that grows on demand (not capturable in a hipGraph the
 * first time it grows).
 * ------------------------------------------------------------------------------------------- */
size_t psi_chamfer_workspace_bytes(int B, int n, int m);
int psi_chamfer_forward(const float *xyz1, const float *xyz2, int B, int n, int m,
                        float *dist1, int32_t *idx1, float *dist2, int32_t *idx2,
                        void *workspace, void *stream);
/* Accumulates (+=) into gradxyz1 [B,n,3] / gradxyz2 [B,m,3] exactly like the reference, which is
 * handed zero-filled buffers (dist_chamfer.py:40-45).  gradxyz2 may be NULL (scene needs no grad);
 * graddist2/idx2 may be NULL (then direction 2 contributes nothing). */
int psi_chamfer_backward(const float *xyz1, const float *xyz2, float *gradxyz1, float *gradxyz2,
                         const float *graddist1, const float *graddist2,
                         const int32_t *idx1, const int32_t *idx2, int B, int n, int m, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Trilinear SDF lookup with analytic gradient — replaces
 *   F.grid_sample(sdf[B,1,D,D,D], norm_verts[:,:,[2,1,0]].view(-1,V,1,1,3), padding_mode='border')
 *                                    fitting_proxe.py:144-151, train_s1.py:183-190, train_s2.py:182-189
 * including the normalisation (v-min)/(max-min)*2-1 of fitting_proxe.py:147.
 * sdf [S,D,D,D] fp32, element [s][ix][iy][iz] (C order, fitting_proxe.py:85) — ONE volume per scene,
 * scene_id [B] int32 selects it per batch row (NULL = scene 0); the reference replicates the volume
 * B times (fitting_proxe.py:90).  gmin,gmax [S,3]; verts [B,V,3]; out_sdf [B,V];
 * out_grad [B,V,3] = d sdf / d vert (NULL to skip).  align_corners: 1 = torch 1.2.0 (pinned)
 * semantics, 0 = torch>=1.3 default.
 * psi_sdf_sample_backward: grad_verts[b,v,:] += grad_sdf[b,v] * out_grad[b,v,:].
 * ------------------------------------------------------------------------------------------- */
int psi_sdf_sample_forward(const float *sdf, const int32_t *scene_id, const float *gmin, const float *gmax,
                           const float *verts, int B, int V, int D, int S, int align_corners,
                           float *out_sdf, float *out_grad, void *stream);
int psi_sdf_sample_backward(const float *grad_sdf, const float *out_grad, int B, int V,
                            float *grad_verts, void *stream);
/* Penetration statistics of fitting_proxe.py:155-158 without the .item() host sync:
 * stats[0] = sum_{sdf<0} |sdf|, stats[1] = count(sdf<0) (as float, exact below 2^24), over n values.
 * stats must be zeroed by the caller (it is accumulated with atomics). */
int psi_sdf_penetration_stats(const float *sdf_vals, long n, float *stats, void *stream);

/* ---------------------------------------------------------------------------------------------
 * SMPL-X linear blend skinning — replaces the body-model call
 *   smplx.create(path, model_type='smplx', ..., batch_size=B)(return_verts=True, body_pose=..., ...)
 *                                                       fitting_proxe.py:55-69,125-128; train_s1.py:66-81,150-153
 * whose arithmetic is lbs() (human_body_prior/body_model/lbs.py:34-118 == smplx 0.1.13 lbs), plus
 * `vertices + transl` and, when cam_ext is given, GeometryTransformer.verts_transform (cvae.py:141-149).
 * psi_lbs_create takes HOST arrays laid out as the SMPL-X loader leaves them:
 *   v_template [V,3]; shapedirs [V,3,NB] (betas | expression); posedirs [P,3V], P=(J-1)*9
 *   (i.e. reshape(posedirs,[-1,P]).T); J_regressor [J,V]; weights [V,J]; parents [J] (-1 for the root).
 * Forward inputs (device): betas [B,NB] (shape | expression), pose [B,J*3] axis-angle of ALL joints
 * (after hand PCA and pose_mean), transl [B,3] or NULL, cam_ext [B,4,4] row-major or NULL.
 * Outputs: verts [B,V,3]; joints [B,J,3] or NULL (posed joints + transl, no camera transform).
 * ws: psi_lbs_workspace_floats(model,B) floats of device scratch; forward saves what backward needs
 * there, so the same ws must be passed to the matching psi_lbs_backward.
 * Backward: grad_verts [B,V,3] -> grad_betas [B,NB], grad_pose [B,J*3], grad_transl [B,3] (each
 * nullable, each OVERWRITTEN).  Joints are not differentiated (PSI reads .vertices only).
 * ------------------------------------------------------------------------------------------- */
typedef struct psi_lbs_model psi_lbs_model;
int psi_lbs_create(psi_lbs_model **out, const float *h_v_template, const float *h_shapedirs,
                   const float *h_posedirs, const float *h_J_regressor, const float *h_weights,
                   const int32_t *h_parents, int V, int J, int NB);
void psi_lbs_destroy(psi_lbs_model *model);
size_t psi_lbs_workspace_floats(const psi_lbs_model *model, int B);
int psi_lbs_forward(const psi_lbs_model *model, const float *betas, const float *pose, const float *transl,
                    const float *cam_ext, int B, float *verts, float *joints, float *ws, void *stream);
int psi_lbs_backward(const psi_lbs_model *model, const float *grad_verts, const float *betas, const float *pose,
                     const float *cam_ext, int B, float *ws, float *grad_betas, float *grad_pose,
                     float *grad_transl, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PSI_HIP_H */
