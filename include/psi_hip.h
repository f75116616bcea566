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
#define PSI_ETIMEOUT (-110)   /* psi_stream_wait: the stream was still busy at the bound */
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
 * library use an internal buffer owned by (device, stream) that grows on demand — calls on different
 * streams never share it; growth is refused (PSI_ENOMEM) while the stream is being captured into a
 * hipGraph, so captured callers pass their own workspace.
 * ------------------------------------------------------------------------------------------- */
/* Arithmetic mode of the distance expression this library was built with: 0 = three products and two sums, each rounded
 * (the CUDA source as written, nvcc --fmad=false) — libpsi_hip.so; 1 = mul, fma, fma (nvcc's default --fmad=true contraction
 * of chamfer.cu:32-35) — libpsi_hip_fma.so, selected by PSI_CHAMFER_FMA=1.  Indices can differ between the two only where two
 * targets are within an ulp of the same distance. */
int psi_chamfer_arith_mode(void);
size_t psi_chamfer_workspace_bytes(int B, int n, int m);
int psi_chamfer_forward(const float *xyz1, const float *xyz2, int B, int n, int m,
                        float *dist1, int32_t *idx1, float *dist2, int32_t *idx2,
                        void *workspace, void *stream);
/* Frees the library's internal scratch buffer of `stream` on the current device (all != 0: of every stream of every device).  The
 * buffers behind `workspace = NULL` are per (device, stream) and otherwise live as long as the process; call this before destroying a
 * stream that was used with workspace = NULL. */
int psi_scratch_release(void *stream, int all);
/* Accumulates (+=) into gradxyz1 [B,n,3] / gradxyz2 [B,m,3] exactly like the reference, which is
 * handed zero-filled buffers (dist_chamfer.py:40-45).  gradxyz2 may be NULL (scene needs no grad);
 * graddist2/idx2 may be NULL (then direction 2 contributes nothing). */
int psi_chamfer_backward(const float *xyz1, const float *xyz2, float *gradxyz1, float *gradxyz2,
                         const float *graddist1, const float *graddist2,
                         const int32_t *idx1, const int32_t *idx2, int B, int n, int m, void *stream);

/* Exact NN index over a STATIC target cloud (one scene's vertices, fitting_proxe.py:93-96 loads them once per
 * FittingOP).  psi_nn_index_query returns exactly what psi_chamfer_forward returns for direction 1 against that
 * cloud (bit-identical dist1 / idx1, same first-minimum rule) at a fraction of the pair evaluations: a kd-tree
 * prunes only targets that provably cannot attain or tie the minimum.  h_points: HOST [m,3]. */
typedef struct psi_nn_index psi_nn_index;
int psi_nn_index_create(psi_nn_index **out, const float *h_points, int m);
void psi_nn_index_destroy(psi_nn_index *index);
/* hint (device int32 [B,n], nullable): warm start — on entry the target index that won for each query last time
 * (-1 = none), on exit this call's idx1.  It only seeds the search bound with a real candidate; results are unchanged. */
int psi_nn_index_query(const psi_nn_index *index, const float *xyz1, int B, int n, float *dist1, int32_t *idx1,
                       int32_t *hint, void *stream);

/* A set of scene indices queried in ONE launch with a per-body scene slot: body b is searched in indices[slot[b]]
 * (slot: device int32 [B], values in [0,S)).  This is the training-time pattern — every body of a batch brings its own
 * PROX scene (train_s1.py:159-169 gathers scene_verts [B,m,3] from batch_gen_hdf5.py:222-257 and calls chamfer.forward);
 * dist1/idx1 are bit-identical to psi_chamfer_forward on that gathered [B,m,3] tensor.  The set borrows the indices,
 * which must outlive it. */
typedef struct psi_nn_index_set psi_nn_index_set;
int psi_nn_index_set_create(psi_nn_index_set **out, const psi_nn_index *const *indices, int S);
void psi_nn_index_set_destroy(psi_nn_index_set *set);
int psi_nn_index_set_query(const psi_nn_index_set *set, const int32_t *slot, const float *xyz1, int B, int n,
                           float *dist1, int32_t *idx1, void *stream);

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
 * stats must be zeroed by the caller (the result is ADDED to it; per-block partial sums combined in block order: no atomics,
 * run-to-run bit-identical). */
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
 * Arithmetic: fp32 throughout; the blend-shape product v_template + [betas | R - I] . dirs (lbs.py:81,94-99) is formed on the fp16
 * matrix pipe as three split products of two fp16 parts per operand (22 mantissa bits, exact power-of-two scales, fp32 accumulation):
 * 2^-22 relative per product, the accuracy class of an fp32 product chain (tests/test_lbs_gpu.py holds it to fp64 with the margin an
 * fp32 evaluation needs).  |betas| beyond 4094 saturate (the feature rows' fixed scale).  The backward product g_feat = g_vposed . dirs^T
 * is formed the same way, the gradient rows scaled by their largest entry of the call (psi_lbs_backward: a row-maximum pass; the fitting
 * engine below: recorded by the kernels that write the rows).
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

/* ---------------------------------------------------------------------------------------------
 * Fused fitting engine — replaces the loop body of FittingOP.fitting / cal_loss
 *   for ii in range(num_iter): optimizer.zero_grad(); cal_loss(...); loss.backward(); optimizer.step()
 *                                                                      fitting_proxe.py:101-162,177-189
 * (fitting_habitat.py:103-208 is the same with contact_const = 1.0 and a pre-multiplied camera).
 * One iteration = ~16 stream-ordered HIP kernels, no autograd, no host sync; psi_fit_iterate replays it
 * as a hipGraph.  State (device, owned by the engine): x = xhr_rec [B,75] (transl 3 | 6D rot 6 | betas 10 |
 * VPoser latent 32 | hand PCA 12+12), Adam moments and step count (fitting_proxe.py:73-74: the optimizer
 * persists across files unless reset).
 * psi_fit_create: h_* are HOST arrays.  VPoser decoder weights in nn.Linear layout [out,in]
 * (bodyprior_dec_fc1 [512,32], _fc2 [512,512], _out [126,512], vposer_smpl.py:83-89); hand PCA components
 * [num_pca_comps,45] each; pose_mean [165]; contact vertex ids [n_contact] in the order
 * GeometryTransformer.get_contact_id returns them (cvae.py:99-115).  d_scene_verts [m,3] and d_sdf [D,D,D]
 * are DEVICE arrays owned by the caller and must outlive the engine (one copy per scene, shared by the batch).
 * Data parallel: world_size > 1 scales the loss normalisers by the global batch B*world_size; the caller
 * runs psi_fit_forward(stats) -> all-reduce(sum) of stats[0..5] -> psi_fit_backward_step(stats) per iteration
 * (stats = [sum|xhr-x|, sum z^2, sum s/(s+c), sum|sdf<0|, count(sdf<0), 0], a caller-owned device buffer).
 * ------------------------------------------------------------------------------------------- */
typedef struct psi_fit_engine psi_fit_engine;
typedef struct psi_fit_config {
    int B, n_contact, m_scene, D, align_corners, world_size, num_pca_comps, max_history;
    int nn_mode;   /* 0 = brute-force Chamfer kernel, 1 = exact kd-tree index built from the scene cloud at create */
    float w_rec, w_vposer, w_contact, w_collision, contact_const;
    float lr, beta1, beta2, eps;
    int independent_bodies;   /* != 0: the B bodies are B INDEPENDENT problems — every loss is normalised per body (rec / 75, prior / 32,
                                 contact / n_contact, penetration / that body's own count), so one engine run over B bodies equals B runs of
                                 the reference's loop at batch size 1 (one generated-body file each, fitting_proxe.py:252-263).  world_size 1. */
    int concurrent_engines;   /* how many engines the caller keeps in flight on this GPU at the same time (other streams; 0 or 1 = this one
                                 alone).  The per-body head / tail kernels spread a body over up to 8 workgroups so that a lone engine with
                                 few bodies still covers the chip; engines that share the chip are told not to (B x engines x width <= 256). */
    double lr_d, beta1_d, beta2_d;   /* the Adam hyper-parameters as DOUBLES, the way torch.optim.Adam (fitting_proxe.py:73-74) holds them: it forms
                                 1 - beta and the bias corrections 1 - beta^t in double precision before rounding to fp32 (1 - 0.999 -> 0.001f;
                                 1.0f - 0.999f would be 0.00099998713f).  0 = use the fp32 fields above. */
} psi_fit_config;
int psi_fit_create(psi_fit_engine **out, const psi_lbs_model *lbs, const psi_fit_config *cfg,
                   const float *h_w1, const float *h_b1, const float *h_w2, const float *h_b2,
                   const float *h_w3, const float *h_b3, const float *h_lh_comp, const float *h_rh_comp,
                   const float *h_pose_mean, const int32_t *h_contact_ids,
                   const float *d_scene_verts, const float *d_sdf, const float *h_gmin, const float *h_gmax);
void psi_fit_destroy(psi_fit_engine *engine);
/* xhr = target body vectors [B,75]; x_init = starting parameters (NULL: start at xhr, fitting_proxe.py:175);
 * cam_ext [B,4,4]; reset_optimizer != 0 zeroes the Adam state and step count. */
int psi_fit_set_problem(psi_fit_engine *engine, const float *d_xhr, const float *d_x_init, const float *d_cam_ext,
                        int reset_optimizer, void *stream);
/* use_graph != 0: each half is captured once into its own hipGraph (stats pointer baked in) and replayed. */
int psi_fit_forward(psi_fit_engine *engine, float *d_stats, int use_graph, void *stream);
int psi_fit_backward_step(psi_fit_engine *engine, const float *d_stats, int use_graph, void *stream);
/* n_iter full iterations on one GPU; use_graph != 0 captures the iteration once (stream must not be the NULL stream)
 * and replays it with hipGraphLaunch — a 20-iteration graph for every full twenty, a 10-iteration graph for a remaining ten, a
 * single-iteration graph for the rest
 * (the iteration keeps no host-side state, so the result does not depend on how n_iter is split into calls). */
int psi_fit_iterate(psi_fit_engine *engine, int n_iter, int use_graph, void *stream);
/* Copies x [B,75] and the first n_hist rows of the loss-history RING [max_history,4] (row = (adam_step-1) % max_history) =
 * (l_rec, l_vposer, l_contact, l_collision as printed by fitting_proxe.py:184-186) to device buffers;
 * h_step (host, nullable) receives the Adam step count and forces a stream sync; with h_step the call also checks the engine's error
 * word and returns non-zero if one of the in-kernel cluster exchanges of the per-body kernels gave up waiting (results invalid). */
int psi_fit_read(psi_fit_engine *engine, float *d_x_out, float *d_history_out, int n_hist, int *h_step, void *stream);
/* The four loss values recorded by Adam step `adam_step` (1-based; the row (adam_step-1) % max_history of the ring) -> d_out4
 * (device, 4 floats): what the reference prints per iteration (fitting_proxe.py:184-186) without copying the whole ring.
 * The caller must not ask for a step more than max_history steps in the past (the ring has wrapped). */
int psi_fit_read_losses(psi_fit_engine *engine, int adam_step, float *d_out4, void *stream);
/* Differentiable body decode of the CVAE training losses: x75 [B,75] = [transl | 6D global rot | betas 10 | VPoser latent 32 |
 * hand PCA 12+12] -> camera-frame vertices [B,V,3].  One call replaces the reference's chain
 *   convert_to_3D_rot (cvae.py:117-139) -> body_params_encapsulate_batch (cvae.py:221-251) -> vposer.decode(..., 'aa')
 *   (vposer_smpl.py:156-167) -> body_mesh_model(...) (train_s1.py:143-150) -> verts_transform (cvae.py:141-149),
 * train_s1.py:136-157, with the engine's own head / LBS kernels (B = the engine's batch size).  The forward leaves its
 * activations in the engine; psi_fit_decode_backward must follow it (before the next forward) and returns dL/dx75.
 * The engine's fitting state (x, Adam moments) is overwritten: use a dedicated engine. */
int psi_fit_decode_forward(psi_fit_engine *engine, const float *d_x75, const float *d_cam_ext, float *d_verts, void *stream);
int psi_fit_decode_backward(psi_fit_engine *engine, const float *d_grad_verts, float *d_grad_x75, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Data-parallel fitting with the collective issued from C (RCCL over xGMI; one process per GPU).
 * The reference has no multi-GPU code (cluster_mpi/htcondor_submission.sub:15 submits independent processes); what is sharded
 * here is the batch of FittingOP.fitting (fitting_proxe.py:167-195): rank r owns its rows of xhr_rec and their Adam state, the
 * loss normalisers of fitting_proxe.py:105-110,139,155-158 run over the GLOBAL batch through one 6-float all-reduce per iteration.
 * A communicator is created once per process: rank 0 calls psi_dp_unique_id and hands the 128 bytes to every rank (any side channel:
 * torch.distributed's store, MPI, a file), then every rank calls psi_dp_comm_create with its rank — a collective call, made with the
 * rank's GPU current.  psi_fit_iterate_dp runs n_iter iterations of [forward kernels + the joint-side backward contraction, which leaves
 * the rank's LOCAL loss sums in stats; ncclAllReduce(stats[0..5]); the reduction that applies the global penetration count + the tail] on
 * `stream`; with use_graph != 0 as 10-iteration hipGraphs that CONTAIN the RCCL kernel (no host code between iterations).  The engine
 * must have been created with world_size == the communicator's size.  d_stats (device, >= 8 floats, nullable = engine-owned) is the
 * buffer the all-reduce runs in place on.
 * ------------------------------------------------------------------------------------------- */
#define PSI_DP_ID_BYTES 128
typedef struct psi_dp_comm psi_dp_comm;
int psi_dp_unique_id(char *h_id128);
int psi_dp_comm_create(psi_dp_comm **out, const char *h_id128, int rank, int world);
void psi_dp_comm_destroy(psi_dp_comm *comm);
int psi_dp_comm_info(const psi_dp_comm *comm, int *rank, int *world, int *rccl_version);
/* In-place sum over the ranks of d_buf[0..n) on `stream` (the collective psi_fit_iterate_dp issues; exported for tests and for the
 * training path's scalar reductions). */
int psi_dp_allreduce_sum(psi_dp_comm *comm, float *d_buf, int n, void *stream);
int psi_fit_iterate_dp(psi_fit_engine *engine, psi_dp_comm *comm, int n_iter, int use_graph, float *d_stats, void *stream);
/* How this engine's data-parallel iterations have been launched so far: 0 = none yet, 1 = captured hipGraphs (the collective inside),
 * 2 = eager launches from C (capturing the collective was refused once; results are the same). */
int psi_fit_dp_mode(const psi_fit_engine *engine);
/* Host-side wait for everything enqueued on `stream` with a bound (hipStreamQuery polls): 0 = done, PSI_ETIMEOUT = still busy after
 * timeout_ms — the watchdog of the data-parallel loop (a collective that cannot complete becomes an error, not a hang). */
int psi_stream_wait(void *stream, int timeout_ms);

/* Per-kernel timing of one fitting iteration with HIP events on the launch stream (an event is recorded right after
 * every kernel launch of the sequence psi_fit_iterate runs; ungraphed), averaged over n_rep iterations.  Advances the
 * optimisation by n_rep steps.  h_names: [max_stages][name_stride] chars, h_ms: [max_stages] milliseconds. */
int psi_fit_profile(psi_fit_engine *engine, int n_rep, char *h_names, int name_stride, float *h_ms, int max_stages,
                    int *h_n_stages, void *stream);
/* Test/diagnostic copy of an engine-owned device buffer by name ("verts" [B,V,3], "g_verts", "pose" [B,165],
 * "g_pose", "g_rot" [B,55,9], "stats" [8], "adam_m"/"adam_v" [B,75]; the backward's intermediates "gA" [B,64,16], "gfeat" [B,Kpad],
 * "g_transl" [B,3], "gl" / "g_vp" / "v_posed" [B,Npad], and — engines whose per-vertex backward runs inside the scene launch — the
 * contact class of rows "glc" / "gvpc" / "vpc" [B,3 ncp] in slot order) into d_out (device). */
int psi_fit_copy_buffer(psi_fit_engine *engine, const char *name, float *d_out, long n_floats, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Dense layers of the CVAEs on the matrix cores — replaces nn.Linear (+ LeakyReLU, + skip connection) of
 *   ResBlock                 net_layers.py:28-43  (fc1 -> LeakyReLU -> fc2 -> LeakyReLU -> + x0)
 *   scene feature `fc`       cvae.py:436-440, net_layers.py:66,164  (8192 -> 256, 32768 -> 256)
 *   linear_in / mu_enc / logvar_enc / linear_latent / linear_out, torso_linear, pose_linear, mean/log_var_linear, decode.*
 *                            cvae.py:441-452,474-492; net_layers.py:67-86,165-184
 * y[M,N] = act(x[M,K] W[N,K]^T + bias[N]) (+ residual[M,N]);  W is the nn.Linear weight ([out,in], fp32 master copy).
 * Arithmetic: x and W are rounded to bf16 (RNE) as they are loaded, products accumulate in fp32 on v_mfma_f32_32x32x16_bf16,
 * y is fp32 — the precision of the reference trained under bf16 autocast with an fp32 output.  x: fp32 [M,K], or bf16 [M,K]
 * when x_is_bf16 != 0 (the NHWC bf16 conv feature map flattened).  K % 16 == 0 (forward) and N % 16 == 0 (backward).
 * act: 0 none, 1 LeakyReLU(slope).  act_out (nullable, [M,N]) receives act(...) BEFORE the residual is added: the backward
 * needs its sign.  ws: psi_linear_workspace_floats(M,N,K) floats (0 for small layers; split-K partials for large weights).
 * Backward: gy [M,N] = dL/dy; act_out as saved by the forward (NULL when act == 0); outputs gx [M,K] (dtype of x; nullable),
 * gW [N,K] fp32 and gbias [N] fp32 (nullable), all OVERWRITTEN.  The residual's gradient is gy itself (caller adds it).
 * ws of the backward: psi_linear_backward_workspace_floats(M,N,K) floats (0 for small layers; partial dX of the n-slices otherwise;
 * may be NULL when gx is NULL).  Partials of either direction are summed in slice order: deterministic.
 * ------------------------------------------------------------------------------------------- */
size_t psi_linear_workspace_floats(int M, int N, int K);
int psi_linear_forward(const void *x, int x_is_bf16, const float *W, const float *bias, const float *residual, int M, int N, int K,
                       int act, float slope, float *y, float *act_out, float *ws, void *stream);
/* psi_linear_forward3: the same layer at the fp32 model's precision (cvae.py:474-492 as shipped, no autocast) — three-term split products
 * (see psi_conv2d_forward) — and for ANY K and N (the 3-, 72-, 75-wide layers included; rows that are not 16-byte aligned are gathered
 * element by element).  Same arguments, same workspace. */
int psi_linear_forward3(const void *x, int x_is_bf16, const float *W, const float *bias, const float *residual, int M, int N, int K,
                        int act, float slope, float *y, float *act_out, float *ws, void *stream);
size_t psi_linear_backward_workspace_floats(int M, int N, int K);
/* psi_linear_backward3: the backward of psi_linear_forward3 — dX = G W and dW = G^T X (+ gbias) with three-term split operands, fp32 x, ANY N
 * and K; arguments and workspace as psi_linear_backward. */
int psi_linear_backward3(const float *gy, const float *act_out, const float *x, const float *W, int M, int N, int K, float slope, float *gx,
                         float *gW, float *gbias, float *ws, void *stream);
int psi_linear_backward(const float *gy, const float *act_out, const void *x, int x_is_bf16, const float *W, int M, int N, int K,
                        float slope, void *gx, float *gW, float *gbias, float *ws, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Batch normalisation of the scene trunk with its ReLU and skip connection fused in — replaces, inside the BasicBlocks and the stem of
 * the ResNet-18 prefix the reference builds (cvae.py:427-435; torchvision resnet18 children[1:6] = bn1, relu, maxpool, layer1, layer2),
 *   out = relu(bn1(conv1(x)))          out = relu(bn2(conv2(out)) + identity)          identity = downsample_bn(downsample_conv(x))
 * in TRAINING mode (train_s1.py / train_s2.py call model_h.train(): batch statistics, running statistics updated with `momentum`,
 * unbiased running variance, num_batches_tracked += 1 — nn.BatchNorm2d semantics).
 * x, residual (nullable), y: bf16 feature maps [M, C] with the channel fastest (M = N*H*W: an NHWC / torch channels_last tensor),
 * C % 8 == 0, C <= 256.  y = act(gamma * (x - mean) * invstd + beta (+ residual)), act = ReLU when relu != 0.  save_mean / save_invstd
 * [C] fp32 are kept for the backward.  running_mean / running_var / num_batches_tracked (int64) may be NULL.
 * Backward: dy [M,C] bf16 = dL/dy -> dx [M,C] bf16, dresidual [M,C] bf16 (nullable: the gradient of the skip input = dy masked by the
 * ReLU), dgamma / dbeta [C] fp32 (OVERWRITTEN).  y is the forward output (needed only when relu != 0, for the mask).
 * ws: psi_bn_workspace_floats(M, C) floats of device scratch per call (per-block partial sums; deterministic summation order).
 * ------------------------------------------------------------------------------------------- */
size_t psi_bn_workspace_floats(long M, int C);
int psi_bn_forward(const void *x, const void *residual, const float *gamma, const float *beta, float *running_mean,
                   float *running_var, long long *num_batches_tracked, long M, int C, int relu, float momentum, float eps,
                   void *y, float *save_mean, float *save_invstd, float *ws, void *stream);
int psi_bn_backward(const void *dy, const void *x, const void *y, const float *gamma, const float *save_mean,
                    const float *save_invstd, long M, int C, int relu, void *dx, void *dresidual, float *dgamma, float *dbeta,
                    float *ws, void *stream);

/* ---------------------------------------------------------------------------------------------
 * 3x3 convolution, stride 1, padding 1 (nn.Conv2d(C, C', 3, 1, 1)) of the scene trunk on the bf16 matrix cores — replaces the library
 * convolution for layer1 / layer2 of the ResNet-18 prefix (cvae.py:427-435) and the 128 -> 128 head convolution (net_layers.py:160-164),
 * forward and input gradient (the input gradient is the same convolution of dY with the weight rotated by 180 degrees and its channel
 * axes swapped: psi_conv3x3_rotate_weight).  x [N,H,W,Cin] bf16 (NHWC), w [Cout][3][3][Cin] bf16 (a channels_last Conv2d weight),
 * bias [Cout] fp32 or NULL, y [N,H,W,Cout] bf16; fp32 accumulation.  Covered shapes (psi_conv3x3_supported != 0):
 * Cin = 64 with Cout % 64 == 0, H % 8 == 0, W % 32 == 0;  Cin = 128 with Cout % 128 == 0, H % 8 == 0, W % 16 == 0.
 * ------------------------------------------------------------------------------------------- */
int psi_conv3x3_supported(int Cin, int Cout, int H, int W);
int psi_conv3x3_forward(const void *x, const void *w, const float *bias, int N, int H, int W, int Cin, int Cout, void *y, void *stream);
/* Weight gradient of the same convolution: gw [Cout][3][3][Cin] fp32 (OVERWRITTEN) = sum over all pixels of dY (x) X per filter tap; covered
 * when the image width is 16 or 32, H % (128 / W) == 0 and both channel counts are multiples of 64.  ws: psi_conv3x3_wrw_workspace_floats()
 * floats (per-split partial results, summed in a fixed order: deterministic, unlike the library's atomic split-K kernels). */
size_t psi_conv3x3_wrw_workspace_floats(int N, int H, int W, int Cin, int Cout);
int psi_conv3x3_weight_grad(const void *x, const void *dy, int N, int H, int W, int Cin, int Cout, float *gw, float *ws, void *stream);
/* w [Cout][3][3][Cin] -> wt [Cin][3][3][Cout] with wt[ci][kh][kw][co] = w[co][2-kh][2-kw][ci]:  dX = psi_conv3x3_forward(dY, wt) */
int psi_conv3x3_rotate_weight(const void *w, int Cin, int Cout, void *wt, void *stream);
/* fp32 master weight w32 [Cout][3][3][Cin] (the channels_last memory of an nn.Conv2d weight) -> bf16 in both layouts in one launch:
 * wb [Cout][3][3][Cin] (rounded to nearest even, what autocast's cast produces) and, when wt != NULL, wt = the rotated layout above. */
int psi_conv3x3_prepare_weight(const float *w32, int Cin, int Cout, void *wb, void *wt, void *stream);

/* MaxPool2d(kernel_size=3, stride=2, padding=1) of the trunk's stem (torchvision resnet18 children[3]; cvae.py:431-435) on an NHWC bf16 map
 * x [N,H,W,C] -> y [N,OH,OW,C], OH = (H - 1) / 2 + 1; idx [N,OH,OW,C] uint8 = position kh * 3 + kw of the maximum inside its window (first
 * maximum in scan order, like at::max_pool2d_with_indices).  Backward: dx [N,H,W,C] bf16 (OVERWRITTEN) gathers dy through idx. */
int psi_maxpool3x3s2_forward(const void *x, int N, int H, int W, int C, void *y, void *idx, void *stream);
int psi_maxpool3x3s2_backward(const void *dy, const void *idx, int N, int H, int W, int C, void *dx, void *stream);

/* ---- the scene encoders at the REFERENCE'S PRECISION (fp32 model: cvae.py:427-455; torchvision resnet18 children[1:6]; the head
 * convolutions cvae.py:436, net_layers.py:64,162) on hand-written kernels — replaces nn.Conv2d / nn.BatchNorm2d / nn.MaxPool2d of an fp32
 * HumanCVAES1 / S2 (F.conv2d -> MIOpen, at::batch_norm, at::max_pool2d) on the GPU.
 * psi_conv2d_forward: ANY 2-D convolution of that trunk (7x7 / stride 2 stem, stride-1 and stride-2 3x3, 1x1 / stride 2 downsample, 3x3 heads)
 *   as ONE implicit-GEMM kernel (csrc/conv_gemm.hip).  x [N,H,W,Cin] NHWC, fp32 or bf16 (x_bf16); w [Cout,KH,KW,Cin] FP32 (the memory of a
 *   channels_last Conv2d weight: the master weight is read directly, no prepared copy); bias [Cout] fp32 or NULL; y [N,OH,OW,Cout] NHWC fp32 or
 *   bf16 (y_bf16), OH = (H + 2 pad - KH) / stride + 1.  nterm = 3: every product is a three-term bf16 split (hi*hi + hi*lo + lo*hi) with
 *   fp32 accumulation — 0.6-3.2e-5 of the reference's recorded fp32 forward passes (tests/golden/cvae.npz; bound of the tests 2e-4);
 *   nterm = 1: operands rounded to bf16 (the bf16 mode of the trunk: the convolutions psi_conv3x3_forward does not cover).
 *   psi_conv2d_supported: Cout % 32 == 0 and (Cin % 16 == 0, or Cin * KH * KW <= 4096: element-gather paths; the 2 -> 64 channel 7x7 stride-2
 *   stem is routed to its own kernels, csrc/conv_stem.hip).
 * psi_bn_forward_t / psi_bn_backward_t / psi_maxpool3x3s2_forward_t / _backward_t: the operators above on maps of either element type
 *   (map_f32 != 0: fp32 NHWC maps), psi_bn_forward_t also in the INFERENCE form (eval_mode != 0: y = gamma (x - running_mean) /
 *   sqrt(running_var + eps) + beta, nothing updated, save_mean / save_invstd may be NULL) — the generation drivers run the encoders in
 *   .eval() (test_habitat_s2.py:155-170).  idx == NULL in psi_maxpool3x3s2_forward_t: no window positions are kept (inference). */
int psi_conv2d_supported(int Cin, int Cout, int KH, int KW, int stride, int pad);
int psi_conv2d_forward(const void *x, int x_bf16, const float *w, const float *bias, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                       int stride, int pad, void *y, int y_bf16, int nterm, void *stream);
/* psi_conv2d_input_grad: dX of the same convolution — dy [N,OH,OW,Cout] -> dx [N,H,W,Cin] (OVERWRITTEN); wt = the weight re-laid out as
 *   [Cin][KH][KW][Cout] fp32 (torch: weight.permute(1,2,3,0).contiguous()); the forward kernel in its transposed-gather form, any stride
 *   (Cin % 32 == 0; Cout % 64 == 0 or Cout * KH * KW <= 4096).  psi_conv2d_weight_grad: gw [Cout,KH,KW,Cin] fp32 (OVERWRITTEN) = sum over pixels of dy x im2col(x); the
 *   pixel range is split over workgroups, partial tiles summed in slice order (deterministic); ws: psi_conv2d_wgrad_workspace_floats floats.
 *   Both replace aten::convolution_backward (MIOpen igemm_bwd / igemm_wrw) for the trunk's convolutions; nterm as in the forward. */
int psi_conv2d_input_grad(const void *dy, int dy_bf16, const float *wt, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          void *dx, int dx_bf16, int nterm, void *stream);
/* Prepared weights: psi_conv2d_prepare_weight rounds / splits the fp32 master weight ONCE per layer and step into bf16 parts (hi parts, for
 *   nterm = 3 followed by the lo parts; Cout*KH*KW*Cin elements each) in the forward layout `wf` [Cout][KH][KW][Cin] and / or the input
 *   gradient's layout `wt` [Cin][KH][KW][Cout] (either may be NULL) — instead of a re-layout copy for the backward and a re-split of every
 *   64 x 64 weight tile in every workgroup of both kernels.  psi_conv2d_forward_p / psi_conv2d_input_grad_p take those buffers in place of
 *   the fp32 weight (same results, bit for bit); psi_conv2d_prepared_ok: the layers they cover (Cin % 16 == 0: all but the stem). */
int psi_conv2d_prepared_ok(int Cin, int Cout, int KH, int KW, int stride, int pad);
int psi_conv2d_prepare_weight(const float *w, int Cout, int KH, int KW, int Cin, int nterm, void *wf, void *wt, void *stream);
int psi_conv2d_forward_p(const void *x, int x_bf16, const void *wf, const float *bias, int N, int H, int W, int Cin, int Cout, int KH, int KW,
                         int stride, int pad, void *y, int y_bf16, int nterm, void *stream);
int psi_conv2d_input_grad_p(const void *dy, int dy_bf16, const void *wt, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                            void *dx, int dx_bf16, int nterm, void *stream);
size_t psi_conv2d_wgrad_workspace_floats(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int psi_conv2d_weight_grad(const void *x, int x_bf16, const void *dy, int dy_bf16, int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride,
                           int pad, float *gw, float *ws, int nterm, void *stream);
/* psi_adam_step: ONE Adam step over all `count` fp32 parameter tensors of a model (train_s1.py:229 / train_s2.py:295-296:
 *   optim.Adam(model_h.parameters(), lr) stepped once per batch) — replaces torch.optim.Adam's multi-tensor launches (_foreach_add on the
 *   step counters + 4 x multi_tensor_apply for HumanCVAES2's 122 tensors) by one or two hand-written launches (80 tensors each; addresses in
 *   the kernel arguments, so a hipGraph captures them by value).  p / g / m / v: HOST arrays of `count` device pointers (parameter,
 *   gradient, exp_avg, exp_avg_sq), n: element counts; step: device fp32 scalar = steps taken so far, shared by all tensors, incremented
 *   by the last workgroup to finish; ticket: device uint32, zero before the first call.  Arithmetic: the operation order of PyTorch's fused
 *   Adam (moments in double, rounded to fp32 once); weight_decay is the L2 form (grad += wd * p), no amsgrad / maximize. */
int psi_adam_step(void *const *p, const void *const *g, void *const *m, void *const *v, const long *n, int count, float *step, unsigned *ticket,
                  double lr, double beta1, double beta2, double eps, double weight_decay, void *stream);
int psi_bn_forward_t(const void *x, int map_f32, const void *residual, const float *gamma, const float *beta, float *running_mean,
                     float *running_var, long long *num_batches_tracked, long M, int C, int relu, float momentum, float eps, void *y,
                     float *save_mean, float *save_invstd, float *ws, int eval_mode, void *stream);
/* psi_bn_backward_t: y may be NULL for a ReLU layer WITHOUT a skip connection when beta is given — the mask "y > 0" is then recomputed from x
 *   (y = relu(fma(x, gamma * invstd, beta - mean * gamma * invstd)), rounded as the forward stored it): both backward passes read one map less. */
int psi_bn_backward_t(const void *dy, int map_f32, const void *x, const void *y, const float *gamma, const float *beta, const float *save_mean,
                      const float *save_invstd, long M, int C, int relu, void *dx, void *dresidual, float *dgamma, float *dbeta, float *ws,
                      void *stream);
int psi_maxpool3x3s2_forward_t(const void *x, int map_f32, int N, int H, int W, int C, void *y, void *idx, void *stream);
int psi_maxpool3x3s2_backward_t(const void *dy, int map_f32, const void *idx, int N, int H, int W, int C, void *dx, void *stream);

/* ---- body-vector glue of a CVAE training step (train_s1.py:95-133 / train_s2.py:102-139 cal_loss) ------------------------------
 * psi_cvae_target: out75[b] = convert_to_6D_rot(normalize_global_T(xh72[b], cam_int[b], max_d[b]))   (cvae.py:176-199 + 118-127, the
 *   torchgeometry 0.1.2 angle_axis_to_rotation_matrix with its first-order branch at theta^2 <= 1e-6): [B,72] -> [B,75].
 * psi_cvae_losses_forward: xh_rec75 = recover_global_T(rec75, cam_int, max_d) (cvae.py:153-172; only the translation changes) and
 *   losses5 = { w_rec (0.5 L1(rec[:, :3], target[:, :3]) + 0.5 L1(xh_rec[:, :3], xh[:, :3])),  w_rec L1(rec[:, 3:], target[:, 3:]),
 *               KL of latent 0, KL of latent 1 (fca^2 w_kl 0.5 mean(exp(logvar) + mu^2 - 1 - logvar); 0 when mu == NULL),
 *               w_vposer mean(xh_rec[:, 19:51]^2) }                                            (train_s2.py:122-139)
 *   fca: the KL annealing factor, read from fca_dev (a device float, so that a captured step can change it) when that is not NULL.
 * psi_cvae_losses_backward: gradients of  sum_k g_losses5[k] * losses5[k]  +  <g_xh_rec75, xh_rec75>  (g_xh_rec75 may be NULL) with
 *   respect to rec75, mu, logvar (all OVERWRITTEN).  All tensors fp32, contiguous, device memory. */
size_t psi_cvae_losses_workspace_floats(void);      /* ws of psi_cvae_losses_forward */
int psi_cvae_target(const float *xh72, const float *cam_int, const float *max_d, int B, float *out75, void *stream);
int psi_cvae_losses_forward(const float *rec75, const float *target75, const float *xh72, const float *cam_int, const float *max_d,
                            const float *mu0, const float *logvar0, int nz0, const float *mu1, const float *logvar1, int nz1, int B,
                            float w_rec, float w_kl, float w_vposer, float fca, const float *fca_dev, float *ws, float *xh_rec75,
                            float *losses5, void *stream);
int psi_cvae_losses_backward(const float *rec75, const float *target75, const float *xh72, const float *cam_int, const float *max_d,
                             const float *mu0, const float *logvar0, int nz0, const float *mu1, const float *logvar1, int nz1, int B,
                             float w_rec, float w_kl, float w_vposer, float fca, const float *fca_dev, const float *xh_rec75,
                             const float *g_losses5, const float *g_xh_rec75, float *g_rec75, float *g_mu0, float *g_logvar0,
                             float *g_mu1, float *g_logvar1, void *stream);

/* ---- tail of the two scene losses of a training step (train_s1.py:156-204 / train_s2.py:159-202) ----------------------------------
 * forward : losses2[0] = gate w_contact mean(s / (s + 1)), s = sqrt(dist + 1e-4) over the n_contact = B * n_c body->scene distances;
 *           losses2[1] = gate w_collision * (mean |sdf| over the negative ones of the n_sdf = B * V values, 0 if there is none);
 *           stats2 = { sum |sdf| over sdf < 0, their count } (input of the backward).  ws: psi_scene_losses_workspace_floats() floats.
 * backward: g_verts [B,V,3] (OVERWRITTEN) = d(g_losses2[0] losses2[0] + g_losses2[1] losses2[1]) / d body vertices, given xyz1 [B,n_c,3] =
 *           the contact vertices that were queried (rows vid [n_c] of the body), their nearest scene points verts_table[slot[b]][idx[b,j]]
 *           (verts_table [S,m,3], chamfer.cu:155-174) and sdf_grad [B,V,3] = d sdf / d vertex (psi_sdf_sample_forward's out_grad).
 *           chain: psi_contact_slot_chain(vid) — the slots of a vertex that is listed more than once (cvae.py:99-115 keeps duplicates) are
 *           chained in slot order and added by one thread: no atomics, run-to-run bit-identical. */
size_t psi_scene_losses_workspace_floats(void);
int psi_scene_losses_forward(const float *dist, long n_contact, const float *sdf_vals, long n_sdf, float w_contact, float w_collision,
                             float gate, float *ws, float *losses2, float *stats2, void *stream);
int psi_scene_losses_backward(const float *g_losses2, const float *stats2, const float *dist, const float *xyz1, const int32_t *idx,
                              const int32_t *slot, const float *verts_table, long m, const int32_t *vid, const float *sdf_vals,
                              const float *sdf_grad, int B, int V, int n_c, float w_contact, float w_collision, float gate,
                              const int32_t *chain, float *g_verts, void *stream);
/* chain [2 * n_c] of a contact-id list vid [n_c] (device): chain[j] = the next slot > j that lists the same vertex as slot j (-1: none),
 * chain[n_c + j] = 1 when no earlier slot lists it.  A constant of the list: build it once, pass it to every backward. */
int psi_contact_slot_chain(const int32_t *vid, int n_c, int32_t *chain, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PSI_HIP_H */
