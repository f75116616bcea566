// Fused fitting engine for gfx950: ONE fitting iteration of the reference's loop
//   for ii in range(num_iter): zero_grad; cal_loss; (loss_rec+loss_vposer+loss_contact+loss_collision).backward(); Adam.step()
//                                                                                        source/fitting_proxe.py:101-162,177-189
// as a fixed sequence of HIP kernels with a hand-derived backward, replayable as a hipGraph (no host sync, no autograd).
//
// Seven launches per iteration (single process, B <= 128: skin_fwd<SdfPen> and kd_query<CONTACT> share ONE launch, fwd_scene_kernel,
// in which the NN-search workgroups skin their own contact vertices; eight launches above that):
//   head_fwd          per body: L1 / latent-prior partial sums (fitting_proxe.py:105-110); convert_to_3D_rot (cvae.py:128-137);
//                     VPoser.decode 32->512->512->126 -> 21 x (6D -> R -> angle-axis) (vposer_smpl.py:107-121,152-161);
//                     SMPL-X hand PCA + pose_mean (smplx 0.1.13 forward); then the LBS pose stage of the
//                     body (Rodrigues, joints, kinematic chain: lbs_device.h)
//   blend_fwd         v_posed = v_t + feat @ dirs (MFMA)                                  (lbs.hip)
//   skin_fwd<SdfPen>  skinning incl. cam_ext, with the trilinear SDF lookup + analytic gradient + penetration partial sums
//                     as its per-vertex epilogue (fitting_proxe.py:144-158)               (lbs_device.h template)
//   kd_query<CONTACT> exact NN of the gathered contact vertices + contact-loss epilogue   (nnindex.hip; fitting_proxe.py:131-139;
//                     nn_mode 0: the brute-force kernels of chamfer.hip)
//   skin_bwd_v<Grad>  prologue: statistics stats[6] = [sum|dx|, sum z^2, sum f, sum|sdf-|, N, 0] and d loss / d verts =
//                     penetration part (needs the GLOBAL count N) + contact part, built on the fly; then the skinning backward
//   bwd_joint         skin_bwd_A and blend_bwd (both MFMA) as one heterogeneous grid       (lbs.hip)
//   reduce_partials   sums of the split-contraction partials                               (lbs.hip)
//   head_bwd_adam     per body: LBS pose backward, Gram-Schmidt / VPoser-MLP / hand-PCA backward, + L1 and prior gradients,
//                     Adam update (torch.optim.Adam defaults)
// Data-parallel runs add loss_finalize after kd_query, all-reduce `stats` there (one 6-float RCCL all-reduce per iteration)
// and skin_bwd_v reads the reduced values.  psi_fit_decode_forward/backward drive the head / LBS kernels alone (training).
//
// Gradient through "6D -> R -> angle-axis -> Rodrigues -> R'": the forward evaluates the reference's chain literally;
// the backward uses that R' == R on SO(3) and that Gram-Schmidt only moves along SO(3), so J_GS^T dL/dR' is the exact
// gradient (requires pose_mean == 0 for global_orient/body joints, which holds for SMPL-X; checked at creation).
#include "psi_internal.h"
#ifdef PSI_HEAD_STOPS
__device__ int psi_dbg_sstop;            // dev: leave the skinning / scene kernels at this point (tools/head_stops.sh)
#define PSI_SSTOP(k) do { if (psi_dbg_sstop == (k)) return; } while (0)
__device__ int psi_dbg_pstop;            // dev: end the tail kernel inside the pose-backward stage (PSI_TAIL_STOP = 20 + k)
#define PSI_PSTOP(k) do { if (psi_dbg_pstop == (k)) asm volatile("s_endpgm"); } while (0)
#define PSI_TRACE(lo, hi) PsiBlockTrace trace_((lo), (hi))
// dev (PSI_SKIN_STOP=9): the workgroup timeline of the LAST fwd_scene launch — {start, end} in 10 ns wall-clock ticks, the hardware id
// words and the kind of workgroup, one record per workgroup (tools/timeline.py draws it)
__device__ unsigned long long psi_dbg_tl[4 * 8192];
struct PsiBlockTrace {
    unsigned long long t0;
    int kind;
    int sel_lo, sel_hi;
    __device__ PsiBlockTrace(int lo = 9, int hi = 10) : t0(wall_clock64()), kind(0), sel_lo(lo), sel_hi(hi) {}
    __device__ ~PsiBlockTrace()
    {
        const unsigned bid = blockIdx.y * gridDim.x + blockIdx.x;
        if (psi_dbg_sstop < sel_lo || psi_dbg_sstop > sel_hi || threadIdx.x != 0 || bid >= 8192) return;
        unsigned long long *o = psi_dbg_tl + 4 * (size_t)bid;
        o[0] = t0;
        o[1] = wall_clock64();
        o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        o[3] = (unsigned long long)kind;
    }
};
extern "C" int psi_dbg_timeline(unsigned long long *out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psi_dbg_tl), sizeof(unsigned long long) * 4 * (size_t)(nblocks < 8192 ? nblocks : 8192));
}
#endif
#include "lbs_device.h"
#include "lbs_joint_device.h"
#include "sdf_device.h"
#ifndef PSI_SDF_CELLS
#define PSI_SDF_CELLS 1      // the engine's copy of the SDF volume: 1 = cell-major records (two 16-byte gathers per sample), 0 = apron bricks (four 8-byte)
#endif
#include "nnindex_device.h"
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace {

constexpr int NH = 512;        // VPoser hidden width (vposer_smpl.py:75-89 with num_neurons=512)
constexpr int NZ = 32;         // VPoser latent
constexpr int NJ6 = 126;       // 21 joints x 6D
constexpr int XD = 75;         // body vector with 6D global rotation (cvae.py:117-126)
#ifndef PSI_HEAD_THREADS
#define PSI_HEAD_THREADS 512
#endif
constexpr int HB = PSI_HEAD_THREADS;   // threads per body in the head kernels: 128 output quads x KQ K-splits
static_assert(HB >= 384 && HB % 128 == 0, "the head / tail kernels place work on threads up to 256 + 75");
constexpr int KQ = HB / 128;            // K-splits of the 512-wide layers
constexpr int KS3 = HB / 32;            // K-slices of fc3 (32 output quads)
constexpr int OS1 = HB / 8;             // output slices of W1^T (8 latent quads)
typedef float f4 __attribute__((ext_vector_type(4)));

struct FitDev {
    int B, V, Vpad, J, NB, n_c, m, D, align_corners, world, ncomp, nfp, nsdfblk;
    int indep;                                        // 1: the B bodies are B independent problems (per-body loss normalisers)
    float w_rec, w_vp, w_contact, w_col, cconst;
    float lr, beta1, beta2, eps;
    // torch.optim.Adam takes its hyper-parameters as Python doubles: `mul_(beta2)` rounds beta2 to fp32, `value = 1 - beta2` is formed
    // in DOUBLE and then rounded (0.001f, not 1.0f - 0.999f = 0.00099998713f), and the bias corrections 1 - beta^t are double arithmetic
    float one_m_beta1, one_m_beta2;
    double lr_d, beta1_d, beta2_d;
    // model constants
    const float *W1T, *b1, *W2T, *b2, *W3T, *b3;      // transposed [in][out] for the forward
    const float *W1, *W2, *W3;                        // original [out][in] for the backward
    const float *lhc, *rhc, *pose_mean;               // [ncomp][45] x2, [J*3]
    const int *vid;                                   // [n_c] contact vertex ids
    const int *cs_ptr, *cs_idx;                       // vertex -> contact slots (CSR, V+1 / n_c)
    const int *cs_first;                              // [V] first contact slot of the vertex | number of its slots << 24 (0: none)
    const float *scene, *sdf, *gmin, *gmax;           // scene cloud [m,3], volume [D^3], bounds [3]
    const float *sdf_brick;                           // engine-owned copy of the volume in apron-brick order (sdf_device.h; nullptr: D % 4 != 0)
    const float *Wct;                                 // [n_c][64] skinning weights of the contact vertices, one row per contact slot
    // state
    float *x, *xhr, *cam, *adam_m, *adam_v;
    int *step;
    // per-iteration buffers
    float *h1, *h2, *o6, *betas20, *pose, *transl, *verts, *og, *gq, *fpart, *penpart, *recpart, *vppart;
    float *cverts;                        // [B][n_c][3] the contact rows of `verts` in SLOT order (large batches: what the separate NN search reads)
    float *g_betas, *g_pose, *g_transl, *g_rot;
    unsigned long long *hx_o6, *hx_gh1;   // head / tail cluster exchange words {value, tag}: partial fc3 outputs [B][C][128], partial W2^T products [B][C][512]
    unsigned *hx_epoch;      // [2][B] launch counts of the head / tail kernel per body (the exchange tag)
    int *hx_err;             // != 0: a cluster exchange gave up waiting (psi_fit_read reports it)
    int hc;                  // workgroups per body in the head / tail kernels (1, 2, 4 or 8)
#ifdef PSI_HEAD_STOPS
    int stop_h, stop_t;      // dev: leave the head / tail kernel at this point (differential timing; tools/head_stops.sh)
#define HSTOP(k) do { if (f.stop_h == (k)) return; } while (0)
#define TSTOP(k) do { if (f.stop_t == (k)) return; } while (0)
#else
#define HSTOP(k)
#define TSTOP(k)
#endif
    int *nn_hint;        // [B,n_c] previous nearest-neighbour indices (warm start of the kd-tree search), -1 = none
    float *history;      // [max_hist][4] loss values per iteration
    int max_hist;
    // ---- the skinning backward inside fwd_scene (fused_bwd; see the comment in front of fit_bwd_joint_kernel).  The contact slots are a second
    // class of "vertices" of the joint-side contractions: ncp = n_c padded to whole 256-slot slices, rows [B][3 ncp] in slot order
    int fused_bwd, ncp, ncp3;
    float *glc, *gvpc, *vpc;              // [B][ncp3] contact part of g_local / g_vposed, and the posed contact vertices (padding slots stay zero)
    float *gtc_part;                      // [B][nfp][4] translation-gradient partials of the search workgroups (contact part)
#ifndef PSI_GV_SLOTS
#define PSI_GV_SLOTS 1024
#endif
    unsigned *gvbits;                     // [2][PSI_GV_SLOTS] bit patterns of max |g_vposed| over the penetration / the contact class of rows of this iteration, in
                                          // slots per class (integer atomicMax by their producers, slot = workgroup % PSI_GV_SLOTS: no hot address; read and combined by
                                          // fit_bwd_joint_kernel for the fp16 parts' scale; zeroed by fit_reduce_kernel)
    const float *dirs_ch;                 // the contact slots' blend-shape columns, one copy per slot, as two fp16 parts per entry in LbsDev::dirs_bh's operand order (12.6 MB at n_c = 2048)
    const float *WTt_c;                   // [ncp/64][PSI_JP][64] skinning weights of the contact slots, tiled per wave like LbsDev::WTt
    float *gA_part, *gfeat_part;          // [nsv + nsv_c][B][JP][16], [nsn_m + nsn_c][B][Kpad] split-contraction partials of both classes
    float *spb;                           // [B] independent-bodies mode: -w_col / N_b (0 when N_b == 0) written by the statistics workgroup
    int nsv, nsv_c, nsn_m, nsn_c, spm, spc;   // slices of the model's vertices / the contact slots (skin_bwd_A), of their columns (blend_bwd) and steps per slice
};

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.0f ? v : v * slope; }

// ContinousRotReprDecoder.decode (cvae.py:58-68): a = view(3,2); columns b1,b2,b3
__device__ __forceinline__ void gs_forward(const float *x6, float *R)
{
    float a1[3] = {x6[0], x6[2], x6[4]}, a2[3] = {x6[1], x6[3], x6[5]};
    float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    for (int r = 0; r < 3; r++) {
        R[r * 3 + 0] = b1[r];
        R[r * 3 + 1] = b2[r];
        R[r * 3 + 2] = b3[r];
    }
}

// gradient of gs_forward: gR [3x3 row-major] -> g6
__device__ __forceinline__ void gs_backward(const float *x6, const float *gR, float *g6)
{
    float a1[3] = {x6[0], x6[2], x6[4]}, a2[3] = {x6[1], x6[3], x6[5]};
    float n1 = fmaxf(sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]), 1e-12f);
    float b1[3] = {a1[0] / n1, a1[1] / n1, a1[2] / n1};
    float d = b1[0] * a2[0] + b1[1] * a2[1] + b1[2] * a2[2];
    float u[3] = {a2[0] - d * b1[0], a2[1] - d * b1[1], a2[2] - d * b1[2]};
    float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
    float gb1[3] = {gR[0], gR[3], gR[6]}, gb2[3] = {gR[1], gR[4], gR[7]}, gb3[3] = {gR[2], gR[5], gR[8]};
    // b3 = b1 x b2:  gb1 += b2 x gb3,  gb2 += gb3 x b1
    gb1[0] += b2[1] * gb3[2] - b2[2] * gb3[1];
    gb1[1] += b2[2] * gb3[0] - b2[0] * gb3[2];
    gb1[2] += b2[0] * gb3[1] - b2[1] * gb3[0];
    gb2[0] += gb3[1] * b1[2] - gb3[2] * b1[1];
    gb2[1] += gb3[2] * b1[0] - gb3[0] * b1[2];
    gb2[2] += gb3[0] * b1[1] - gb3[1] * b1[0];
    // b2 = u / |u|
    float p = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
    float gu[3] = {(gb2[0] - b2[0] * p) / n2, (gb2[1] - b2[1] * p) / n2, (gb2[2] - b2[2] * p) / n2};
    // u = a2 - (b1.a2) b1
    float q = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
    float ga2[3] = {gu[0] - b1[0] * q, gu[1] - b1[1] * q, gu[2] - b1[2] * q};
    for (int i = 0; i < 3; i++) gb1[i] += -d * gu[i] - q * a2[i];
    // b1 = a1 / |a1|
    float r = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
    float ga1[3] = {(gb1[0] - b1[0] * r) / n1, (gb1[1] - b1[1] * r) / n1, (gb1[2] - b1[2] * r) / n1};
    g6[0] = ga1[0]; g6[2] = ga1[1]; g6[4] = ga1[2];
    g6[1] = ga2[0]; g6[3] = ga2[1]; g6[5] = ga2[2];
}

// torchgeometry 0.1.2 rotation_matrix_to_angle_axis (quaternion route); R row-major 3x3
__device__ __forceinline__ void rotmat_to_aa(const float *R, float *aa)
{
    // m = R^T
    const float m00 = R[0], m01 = R[3], m02 = R[6], m10 = R[1], m11 = R[4], m12 = R[7], m20 = R[2], m21 = R[5], m22 = R[8];
    float q[4], t;
    if (m22 < 1e-6f) {
        if (m00 > m11) {
            t = 1 + m00 - m11 - m22;
            q[0] = m12 - m21; q[1] = t; q[2] = m01 + m10; q[3] = m20 + m02;
        } else {
            t = 1 - m00 + m11 - m22;
            q[0] = m20 - m02; q[1] = m01 + m10; q[2] = t; q[3] = m12 + m21;
        }
    } else {
        if (m00 < -m11) {
            t = 1 - m00 - m11 + m22;
            q[0] = m01 - m10; q[1] = m20 + m02; q[2] = m12 + m21; q[3] = t;
        } else {
            t = 1 + m00 + m11 + m22;
            q[0] = t; q[1] = m12 - m21; q[2] = m20 - m02; q[3] = m01 - m10;
        }
    }
    float sc = sqrtf(t);
    for (int i = 0; i < 4; i++) q[i] = q[i] / sc * 0.5f;
    float s2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
    float s = sqrtf(s2);
    float two_theta = 2.0f * (q[0] < 0.0f ? atan2f(-s, -q[0]) : atan2f(s, q[0]));
    float k = s2 > 0.0f ? two_theta / s : 2.0f;
    aa[0] = q[1] * k; aa[1] = q[2] * k; aa[2] = q[3] * k;
}

__device__ __forceinline__ float block_sum(float v, float *sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) s += sh[i];
    return s;
}

// two block sums in one round (one pair of barriers instead of two)
__device__ __forceinline__ void block_sum2(float &a, float &c, float *sh2)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        c += __shfl_down(c, o, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        sh2[2 * (threadIdx.x >> 6)] = a;
        sh2[2 * (threadIdx.x >> 6) + 1] = c;
    }
    __syncthreads();
    float sa = 0, sc = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) {
        sa += sh2[2 * i];
        sc += sh2[2 * i + 1];
    }
    a = sa;
    c = sc;
}

// ------------------------------------------------------------------------------------------------
// Cluster exchange (head / tail kernels below).  A value travels between two workgroups of one launch as ONE 64-bit word
// {float value, 32-bit tag of this launch}, written and read with relaxed device-scope atomics: a word either carries this launch's tag
// — then its value is the one stored with it — or it does not yet, and the reader looks again.  No fences (on gfx950 a device-scope
// release / acquire pair is a write-back and an invalidate walk of the XCD's whole L2: 8-11 us per exchange, measured), no counters.
// The reader is the LAST workgroup of its cluster in dispatch order, so every workgroup it waits for was dispatched before it and runs
// (or has run) regardless of how full the chip is.  The tag is the body's launch count, kept in memory by the reader.
__device__ __forceinline__ void hx_put(unsigned long long *p, float v, unsigned tag)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long hx_peek(unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The wait is bounded (about a second of polling): if a producer never shows up — which in-order dispatch rules out — the reader
// raises the engine's error word (reported by psi_fit_read) and goes on with whatever the word holds, instead of hanging the GPU.
constexpr int HX_MAX_POLLS = 1 << 20;
__device__ __forceinline__ float hx_value(unsigned long long *p, unsigned long long w, unsigned tag, int *err)
{
    int polls = 0;
    while ((unsigned)(w >> 32) != tag) {
        if (++polls > HX_MAX_POLLS) {
            *err = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        w = hx_peek(p);
    }
    return __uint_as_float((unsigned)w);
}

// ------------------------------------------------------------------------------------------------
// Head kernel: VPoser decoder MLP, rotations, hand PCA and the LBS pose stage of one body (Rodrigues, joints, kinematic chain).
//
// A body is worked on by a CLUSTER of C workgroups (C = 8 at the BASELINE batch of 32: 256 workgroups instead of 32).  What a single
// workgroup per body cannot avoid is dragging the 1 MB fc2 matrix (and 0.25 MB of fc3) through ONE compute unit's L1: 13.3 + 4.1 us of
// a 32 us kernel, measured with in-kernel clocks.  In a cluster, workgroup c computes fc1 in full (64 KB, redundant), the
// outputs [c*512/C, (c+1)*512/C) of fc2 (1/C of the matrix), and — fc3 being linear in its input — the PARTIAL fc3 output of exactly
// those activations (1/C of fc3's rows): one exchange of C x 128 floats per body.  The last workgroup of the cluster sums the partials
// in cluster order and carries on with the per-body tail.  With C > 1 a thread's whole share of the three matrices is 28 16-byte loads:
// they are issued at the top of the kernel, before the body vector is even read, so the weight fetch overlaps everything else.
// C = 1 (B > 128: the grid fills the chip anyway) compiles to the exchange-free kernel.  Workgroup id = c * B + b: for B % 8 == 0 a
// body's cluster shares an XCD (and its L2).
template <int C>
__global__ __launch_bounds__(HB) void head_fwd_kernel(FitDev f, PsiLbsView lv)
{
    constexpr int NS = NH / C;              // fc2 outputs of this workgroup
    constexpr int NQ = NS / 4;              // ... in quads
    constexpr int KSPL = HB / NQ;           // K-splits of fc2
    constexpr int KPER = NH / KSPL;         // k per split (>= 16)
    constexpr int KPER3 = NS / KS3;         // k per fc3 slice
    constexpr int K1 = NZ / KQ;             // k per fc1 split
    constexpr int PRE2 = C > 1 ? 16 : 0;    // rows of fc2 / fc3 held in registers from the top of the kernel
    constexpr int PRE3 = C > 1 ? (KPER3 < 4 ? KPER3 : 4) : 0;
    static_assert(KPER % 16 == 0 && KPER3 >= 1 && NS >= 64, "cluster too wide for the thread layout");
    const int b = blockIdx.x / C, c = blockIdx.x % C, t = threadIdx.x;
    const bool last = c == C - 1;           // the workgroup that carries on after the exchange
    __shared__ float sx[XD + 5], sh1[NH], sh2[NS], so6[128], red[HB / 64], spose[PSI_JP * 3], sbetas[32], sJ[PSI_JP][3];
    __shared__ f4 part4[HB], part3[KS3][32];
    // ---- loads that depend on nothing computed here
    const int og1 = t & 127, kq1 = t >> 7, og2 = t % NQ, ks2 = t / NQ, og3 = t & 31, ks3 = t >> 5;
    const float *w1 = f.W1T + (size_t)(kq1 * K1) * NH + og1 * 4;
    const float *w2 = f.W2T + (size_t)(ks2 * KPER) * NH + c * NS + og2 * 4;
    const float *w3 = f.W3T + (size_t)(c * NS + ks3 * KPER3) * 128 + og3 * 4;
    f4 w1p[K1], w2p[PRE2 ? PRE2 : 1], w3p[PRE3 ? PRE3 : 1];
#pragma unroll
    for (int k = 0; k < K1; k++) w1p[k] = *(const f4 *)(w1 + (size_t)k * NH);
#pragma unroll
    for (int k = 0; k < PRE2; k++) w2p[k] = *(const f4 *)(w2 + (size_t)k * NH);
#pragma unroll
    for (int k = 0; k < PRE3; k++) w3p[k] = *(const f4 *)(w3 + (size_t)k * 128);
    const float *x = f.x + (size_t)b * XD;
    if (t < XD) sx[t] = x[t];
    if (t >= 64 && t < 96) sbetas[t - 64] = t - 64 < 10 ? x[9 + (t - 64)] : 0.0f;
    unsigned tag = 0;
    float pm[3] = {0, 0, 0}, b3v = 0.0f;
    PsiJump jp;
    for (int r = 0; r < PSI_NJUMP; r++) jp.a[r] = -1;
    if (C > 1) tag = f.hx_epoch[b] + 1u;
    f4 b1v = {0, 0, 0, 0};
    if (t < 128) b1v = *(const f4 *)(f.b1 + t * 4);
    const float b2v = t < NS ? f.b2[c * NS + t] : 0.0f;
    if (last) {
        if (t < 22)
            for (int e = 0; e < 3; e++) pm[e] = f.pose_mean[t * 3 + e];
        if (t < 128) b3v = f.b3[t];
        jp = psi_load_jump(lv.m);
    }
    __syncthreads();
    HSTOP(1);
    if (c == 0) {
        // loss partial sums (fitting_proxe.py:105, :109-110)
        float dr = t < XD ? fabsf(f.xhr[(size_t)b * XD + t] - sx[t]) : 0.0f;
        float sr = block_sum(dr, red);
        float dz = t < NZ ? sx[19 + t] * sx[19 + t] : 0.0f;     // latent = xh_rec[:,16:48] = x[:,19:51] in the 75-D layout
        float sz = block_sum(dz, red);
        if (t == 0) {
            f.recpart[b] = sr;
            f.vppart[b] = sz;
        }
    }
    if (last) {
        // the part of the per-body tail that needs the body vector only: hand PCA, jaw / eyes, rest joints — into LDS; what the
        // later kernels need of it is stored at the very end (a wait for a load also waits for the wave's earlier stores)
        if (t >= 64 && t < 64 + 9) {
            int e = 66 + (t - 64);                               // jaw, leye, reye: zero parameters + mean
            spose[e] = f.pose_mean[e];
        } else if (t >= 128 && t < 128 + 90) {
            int e = t - 128;                                     // hand PCA: 12 -> 45 per hand
            const float *comp = e < 45 ? f.lhc : f.rhc;
            const float *hx = sx + (e < 45 ? 51 : 63);
            int cc = e < 45 ? e : e - 45;
            float a = 0;
            for (int i = 0; i < f.ncomp; i++) a += hx[i] * comp[i * 45 + cc];
            spose[75 + e] = a + f.pose_mean[75 + e];
        }
        psi_pose_fwd_rest(lv.m, sbetas, sJ);
    }
    HSTOP(2);
    // VPoser decoder.  Each thread owns 4 adjacent outputs (one 16-byte weight load per k) and one slice of K; the K-slices are
    // summed through LDS.
    const float *z = sx + 19;
    {   // fc1: 32 -> 512, K-quarter = 8 (every workgroup of the cluster, in full)
        f4 a = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < K1; k++) a += w1p[k] * z[kq1 * K1 + k];
        part4[kq1 * 128 + og1] = a;
    }
    __syncthreads();
    if (t < 128) {
        f4 a = b1v;
#pragma unroll
        for (int q = 0; q < KQ; q++) a += part4[q * 128 + t];
        for (int e = 0; e < 4; e++) sh1[t * 4 + e] = leaky(a[e], 0.2f);
    }
    __syncthreads();
    HSTOP(3);
    {   // fc2: 512 -> this workgroup's NS outputs: NQ output quads x KSPL K-splits
        f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        const float *h = sh1 + ks2 * KPER;
#pragma unroll
        for (int k = 0; k < PRE2; k += 2) {
            a0 += w2p[k] * h[k];
            a1 += w2p[k + 1] * h[k + 1];
        }
#pragma unroll 8
        for (int k = PRE2; k < KPER; k += 2) {
            a0 += *(const f4 *)(w2 + (size_t)k * NH) * h[k];
            a1 += *(const f4 *)(w2 + (size_t)(k + 1) * NH) * h[k + 1];
        }
        part4[ks2 * NQ + og2] = a0 + a1;
    }
    __syncthreads();
    if (t < NS) {
        float a = b2v;
        const float *p = (const float *)part4;
#pragma unroll
        for (int q = 0; q < KSPL; q++) a += p[q * NS + t];
        a = leaky(a, 0.2f);
        sh2[t] = a;
        f.h2[(size_t)b * NH + c * NS + t] = a;
    }
    if (c == 0)
        for (int o = t; o < NH; o += HB) f.h1[(size_t)b * NH + o] = sh1[o];
    __syncthreads();
    HSTOP(4);
    {   // fc3 over this workgroup's activations: NS -> 126 (rows padded to 128): 32 output quads x KS3 K-slices
        f4 a = {0, 0, 0, 0};
        const float *h = sh2 + ks3 * KPER3;
#pragma unroll
        for (int k = 0; k < PRE3; k++) a += w3p[k] * h[k];
#pragma unroll
        for (int k = PRE3; k < KPER3; k++) a += *(const f4 *)(w3 + (size_t)k * 128) * h[k];
        part3[ks3][og3] = a;
    }
    __syncthreads();
    HSTOP(5);
    if (t < 128) {
        float a = 0.0f;
        const float *p = (const float *)part3;
#pragma unroll
        for (int ks = 0; ks < KS3; ks++) a += p[ks * 128 + t];
        if (C == 1) {
            so6[t] = a + b3v;
        } else if (!last) {
            hx_put(f.hx_o6 + ((size_t)b * C + c) * 128 + t, a, tag);
        } else {
            unsigned long long *p8 = f.hx_o6 + (size_t)b * C * 128 + t;
            unsigned long long w[C - 1 ? C - 1 : 1];
#pragma unroll
            for (int cc = 0; cc < C - 1; cc++) w[cc] = hx_peek(p8 + cc * 128);
            float o = b3v;
#pragma unroll
            for (int cc = 0; cc < C - 1; cc++) o += hx_value(p8 + cc * 128, w[cc], tag, f.hx_err);
            so6[t] = o + a;
        }
    }
    if (C > 1 && !last) return;
    __syncthreads();
    HSTOP(6);
    if (t < NJ6) f.o6[(size_t)b * 128 + t] = so6[t];
    // rotations: thread 0 = global orient (x[3:9]), threads 1..21 = VPoser body joints
    if (t < 22) {
        float R[9], aa[3];
        gs_forward(t == 0 ? sx + 3 : so6 + (t - 1) * 6, R);
        rotmat_to_aa(R, aa);
        for (int e = 0; e < 3; e++) spose[t * 3 + e] = aa[e] + pm[e];
    }
    __syncthreads();
    HSTOP(7);
    psi_pose_fwd_chain(lv.m, spose, nullptr, f.B, b, sJ, jp, lv.feat, lv.R, lv.G, lv.A, nullptr);
    HSTOP(8);
    // what the skinning kernels and the backward read of the body-vector part
    psi_pose_fwd_rest_store(lv.m, sbetas, f.B, b, sJ, lv.feat, lv.Jl);
    for (int i = t; i < f.J * 3; i += HB) f.pose[(size_t)b * f.J * 3 + i] = spose[i];
    if (t < f.NB) f.betas20[(size_t)b * f.NB + t] = sbetas[t];
    if (t >= 64 && t < 67) f.transl[(size_t)b * 3 + (t - 64)] = sx[t - 64];
    if (C > 1 && t == 0) f.hx_epoch[b] = tag;
}

// ------------------------------------------------------------------------------------------------
// Epilogue of the skinning kernel (lbs_device.h): the trilinear SDF lookup (fitting_proxe.py:144-158) happens while the
// vertex is still in registers; per workgroup it leaves sum(-sdf) and the count over penetrating vertices, per vertex
// the SDF gradient masked to sdf < 0.
struct SdfPenEpilogue {
    PsiSdfGrid G;             // sampling constants of the bricked volume (G.brick == nullptr: the plain volume below)
    const float *sdf, *gmin, *gmax;
    float *og, *penpart;
    int D, align_corners, Vpad;
    const int *contact_of;        // != nullptr: only the vertices that are contact queries (cs_first[v] != 0) are stored, to `cverts`
    float s[2];
    bool neg[2];                  // per body of the workgroup (the skinning kernel handles one or two)
    float *cverts;                // [B][n_c][3]: contact rows in SLOT order (+ the CSR for a vertex that several contact parts list)
    const int *cs_ptr, *cs_idx;
    int n_c;
    // != nullptr: the skinning backward of the vertex happens HERE (backward()), on the UNSCALED masked SDF gradient — d loss / d sdf = -w / N
    // needs the batch-global count N (fitting_proxe.py:155-158), but everything behind dL/dverts is linear in it, so the 1 / N is applied where
    // the split contractions are summed (fit_reduce_kernel): g_local, g_vposed [B][Npad] and the translation partials [nvb][B][4] instead of `og`
    float *gl, *gvp, *gtp;
    int Npad, B;
    float gm[2][3], gs[2][3];
    float gvmax;                  // max |g_vposed entry| this lane has stored (-> gmaxp, one integer atomicMax per workgroup)
    unsigned *gmaxp;
    // the vertex store of the skinning kernel.  All vertices: [B][V][3] as always.  Contact vertices only (large batches, where the NN search is
    // a launch of its own and reads them): the rows go to their contact SLOT, not to their vertex — the slot list follows the vertex order
    // within a contact part, so the 12-byte pieces of neighbouring lanes are neighbours in memory again (scattered through [B][V][3] they
    // cost as much as storing every row), and the search reads row j instead of chasing vid[j]
    __device__ __forceinline__ void store(float *verts, size_t body_off, int b, int v, unsigned v12, float x, float y, float z) const
    {
        if (!contact_of) {
            if (verts) psi_st(verts + body_off, v12, psi_p3{x, y, z});
            return;
        }
        const int cw = psi_ld<int>(contact_of, (unsigned)v * 4u);
        if (!(cw >> 24)) return;
        float *row = cverts + (size_t)b * n_c * 3;
        psi_st(row, (unsigned)(cw & 0xffffff) * 12u, psi_p3{x, y, z});
        if ((cw >> 24) > 1)
            for (int ci = cs_ptr[v] + 1; ci < cs_ptr[v + 1]; ci++) psi_st(row, (unsigned)cs_idx[ci] * 12u, psi_p3{x, y, z});
    }
    __device__ __forceinline__ void vertex(int n, int b, int v, float x, float y, float z, bool live)
    {
        s[n] = 0.0f;
        neg[n] = false;
#pragma unroll
        for (int a = 0; a < 3; a++) gm[n][a] = 0.0f;
        if (!live) return;
        float g[3];
        if (G.brick) {
            bool in[3];
#if PSI_SDF_CELLS
            const float val = psi_sdf_sample_cells(G, x, y, z, g, in);
#else
            const float val = psi_sdf_sample_fast(G, x, y, z, g, in);
#endif
            neg[n] = val < 0.0f;
#pragma unroll
            for (int a = 0; a < 3; a++) g[a] = (neg[n] && in[a]) ? g[a] : 0.0f;
            s[n] = neg[n] ? -val : 0.0f;
        } else {
            const float val = psi_trilinear(sdf, gmin, gmax, x, y, z, D, align_corners, g);
            neg[n] = val < 0.0f;
#pragma unroll
            for (int a = 0; a < 3; a++) g[a] = neg[n] ? g[a] : 0.0f;
            s[n] = neg[n] ? -val : 0.0f;
        }
        if (gl) {
#pragma unroll
            for (int a = 0; a < 3; a++) gm[n][a] = g[a];
            return;
        }
        // [B][Vpad][4]: one ALIGNED 16-byte store per lane, a wave = 1 KB = eight whole cache lines (a [B][V][3] row starts 4 bytes past a
        // line boundary for every body but the first, and every wave's 768 bytes then end in two partially written lines)
        psi_st(og + (size_t)b * Vpad * 4, (unsigned)v * 16u, f4{g[0], g[1], g[2], 0.0f});
    }
    // the vertex's skinning backward (what psi_skin_bwd_v_kernel does in a launch of its own, with the same expressions): g_local = R_c^T g,
    // g_vposed = T_R^T g_local — the lane still holds the blended transform.  Padding lanes store zeros (the contractions read whole slices).
    __device__ __forceinline__ void backward(int n, int b, unsigned v12, const psi_f2 (&T2)[6], const float *C)
    {
        if (!gl) return;
        const float gx = gm[n][0], gy = gm[n][1], gz = gm[n][2];
        float lx = gx, ly = gy, lz = gz;
        if (C) {
            lx = psi_dot3(C[0], C[4], C[8], gx, gy, gz);
            ly = psi_dot3(C[1], C[5], C[9], gx, gy, gz);
            lz = psi_dot3(C[2], C[6], C[10], gx, gy, gz);
        }
        psi_st(gl + (size_t)b * Npad, v12, psi_p3{lx, ly, lz});
        const float vx = psi_dot3(T2[0].x, T2[2].x, T2[4].x, lx, ly, lz), vy = psi_dot3(T2[0].y, T2[2].y, T2[4].y, lx, ly, lz),
                    vz = psi_dot3(T2[1].x, T2[3].x, T2[5].x, lx, ly, lz);
        psi_st(gvp + (size_t)b * Npad, v12, psi_p3{vx, vy, vz});
        gvmax = fmaxf(gvmax, fmaxf(fabsf(vx), fmaxf(fabsf(vy), fabsf(vz))));
        gs[n][0] = lx; gs[n][1] = ly; gs[n][2] = lz;
    }
    __device__ __forceinline__ void finish(int n, int b, int vblock, int nvb)
    {
        // per workgroup: sum(-sdf) over the penetrating vertices by DPP adds, their count from the lane mask (scalar popcount)
        __shared__ psi_f2 red[PSI_SKIN_BLK / 64];
        __shared__ float red3[PSI_SKIN_BLK / 64][3];
        __shared__ float redm[PSI_SKIN_BLK / 64];
        const float ws = psi_wave_sum(s[n]);
        const float wc = (float)(int)__builtin_popcountll(__builtin_amdgcn_ballot_w64(neg[n]));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = (psi_f2){ws, wc};
        if (gl) {
            const float sx = psi_wave_sum(gs[n][0]), sy = psi_wave_sum(gs[n][1]), sz = psi_wave_sum(gs[n][2]);
            float mx = gvmax;
#pragma unroll
            for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
            if ((threadIdx.x & 63) == 0) {
                red3[threadIdx.x >> 6][0] = sx;
                red3[threadIdx.x >> 6][1] = sy;
                red3[threadIdx.x >> 6][2] = sz;
                redm[threadIdx.x >> 6] = mx;
            }
        }
        __syncthreads();
        if (gl && threadIdx.x == 64) {
            float mx = redm[0];
#pragma unroll
            for (int w = 1; w < PSI_SKIN_BLK / 64; w++) mx = fmaxf(mx, redm[w]);
#ifndef PSI_NO_GVMAX_ATOMIC
            if (mx > 0.0f) atomicMax(gmaxp + (blockIdx.x & (PSI_GV_SLOTS - 1)), __float_as_uint(mx));
#endif       // (non-negative floats order like their bit patterns; max is order-independent)
        }
        if (threadIdx.x == 0) {
            psi_f2 a = red[0];
#pragma unroll
            for (int w = 1; w < PSI_SKIN_BLK / 64; w++) a += red[w];
            *(psi_f2 *)(penpart + ((size_t)b * nvb + vblock) * 2) = a;
        }
        if (gl && threadIdx.x < 3) {
            float a = 0.0f;
#pragma unroll
            for (int w = 0; w < PSI_SKIN_BLK / 64; w++) a += red3[w][threadIdx.x];
            gtp[((size_t)vblock * B + b) * 4 + threadIdx.x] = a;
        }
    }
};

static inline SdfPenEpilogue make_sdf_epilogue(const FitDev &f, const PsiSdfGrid &G, bool contact_vertices_only = false, const PsiLbsView *bwd = nullptr)
{
    SdfPenEpilogue e = {G, f.sdf, f.gmin, f.gmax, f.og, f.penpart, f.D, f.align_corners, f.Vpad, contact_vertices_only ? f.cs_first : nullptr,
                        {0.0f, 0.0f}, {false, false}, f.cverts, f.cs_ptr, f.cs_idx, f.n_c, nullptr, nullptr, nullptr, 0, f.B, {}, {}, 0.0f, f.gvbits};
    if (bwd) {
        e.gl = bwd->gl;
        e.gvp = bwd->g_vp;
        e.gtp = bwd->gt_part_w;
        e.Npad = bwd->m.Npad;
    }
    return e;
}

// ------------------------------------------------------------------------------------------------
// Query source of the NN search inside the fused forward launch: the contact vertex is SKINNED HERE, by the query's own lane group,
// instead of being read from `verts` — so the search does not depend on the skinning kernel and both run in ONE launch
// (fwd_scene_kernel).  Lanes 0..2 of the 4-lane group each build one row of the blended transform (all joints, ascending, the
// same packed fma as psi_blend_transforms: bit-identical to the vertex the skinning workgroups write), the three coordinates are
// exchanged with shuffles and every lane applies translation and camera.  Weights come from a compact [n_c][64] table (one
// contiguous 256-byte row per contact slot; zero beyond J), the body's joint transforms from LDS.
struct ContactSkinSrc {
    FitDev f;
    LbsDev m;
    const float *As, *v_posed;
    psi_f2 (*sA)[6];
    // The source's loads in TWO dependent rounds (round 3 had six in a row in front of the search: winner index, winner coordinates, two
    // load -> LDS-store trips of the transform staging loop, then contact id + weight row, then posed vertex + translation + camera):
    //   issue()   this thread's two pieces of the body's transforms, the slot's vertex id            (with the caller's hint load)
    //             + one entry of the body's camera / translation per thread 128 .. 142
    //   fetch()   the posed vertex (needs the id)                                                    (with the caller's winner coordinates)
    //   prepare() transforms, camera, translation -> LDS, barrier;   point() the weight row (slot-indexed) and the arithmetic
    // = two rounds + the weight row in front of the search where there were six (camera and translation come back from LDS: as per-lane
    // loads the scheduler sank them behind the blend, as scalar loads the kernel spilled 221 registers).
    psi_f2 st[2];
    float ct;                     // threads 128 .. 142 of the workgroup: one entry of the body's camera (12) / translation (3), staged with the transforms
    int v;
    float px, py, pz;
    float *sCT;
    static constexpr int NWQ = PSI_JP / 4 - 2, NWE = 10;      // weight-row quads: 14 (56 joints); the first NWE are requested in fetch()
    f4 wq[NWE];
    int jslot;
    // fused_bwd: the query's lane group also carries the gradient of its contact term back through its own skinning (contact_post): the
    // rotation part of the blended transform waits in LDS (each of lanes 0..2 holds one row; held in registers across the search they would
    // cost the launch an occupancy step), the camera is there already
    float (*sTR)[9] = nullptr;    // [QPB][9] row-major rotation part of each query's blended transform
    float gvm = 0.0f;                     // max |g_vposed entry| of this lane's contact row
    float gs[3] = {0.0f, 0.0f, 0.0f};     // this lane's g_local (lane 0 of a group with a query; 0 elsewhere): summed per workgroup for the translation gradient
    int bq = 0;
    __device__ __forceinline__ void issue(int b, int j)
    {
        jslot = j;
        bq = b;
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int idx = threadIdx.x + q * 256;
            st[q] = idx < m.J * 6 ? psi_ld<psi_f2>(As + (size_t)b * m.J * 12, (unsigned)idx * 8u) : (psi_f2){0.0f, 0.0f};
        }
        const int k = (int)threadIdx.x - 128;                // (the second transform piece of these threads is beyond J: they have a load slot free)
        ct = 0.0f;
        if (k >= 0 && k < 12) ct = f.cam[(size_t)b * 16 + k];
        else if (k >= 12 && k < 15) ct = f.transl[(size_t)b * 3 + (k - 12)];
        v = f.vid[j];
    }
    __device__ __forceinline__ void fetch(int b)
    {
        const psi_p3 p = psi_ld<psi_p3>(v_posed + (size_t)b * m.Npad, (unsigned)v * 12u);
        px = p.x; py = p.y; pz = p.z;
        // ... and as much of the slot's weight row as the register budget holds (it depends on the slot only): in flight across the LDS
        // staging and the barrier of prepare() instead of a round of its own behind them
        const f4 *wrow = (const f4 *)(f.Wct + (size_t)jslot * PSI_JP);
#pragma unroll
        for (int q = 0; q < NWE; q++) wq[q] = wrow[q];
    }
    __device__ __forceinline__ void prepare(int)
    {
        psi_f2 (*sA_)[6] = psi_transform_stage<1>()[0];           // (shared with the skinning workgroups of the launch: lbs_device.h)
        __shared__ float sCT_[16];
        __shared__ float sTR_[psikd::QPB][9];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int idx = threadIdx.x + q * 256;
            if (idx < PSI_JP * 6) (&sA_[0][0])[idx] = st[q];             // rows beyond J: zeros (issue)
        }
        const int k = (int)threadIdx.x - 128;
        if (k >= 0 && k < 16) sCT_[k] = ct;
        __syncthreads();
        sA = sA_;
        sCT = sCT_;
        sTR = sTR_;
    }
    __device__ __forceinline__ void point(int b, int j, int c, float &qx, float &qy, float &qz) const
    {
        const int r = c < 2 ? c : 2;                          // lane 3 repeats row 2 (its result is not used)
        const f4 *wrow = (const f4 *)(f.Wct + (size_t)j * PSI_JP);
        psi_f2 T0 = {0.0f, 0.0f}, T1 = {0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < NWQ; q++) {                        // 56 joints: J = 55 (SMPL-X) + one zero row; psi_fit_create checks J <= 56
            const f4 w4 = q < NWE ? wq[q] : wrow[q];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const psi_f2 w2 = {w4[t], w4[t]};
                T0 = __builtin_elementwise_fma(w2, sA[q * 4 + t][2 * r], T0);
                T1 = __builtin_elementwise_fma(w2, sA[q * 4 + t][2 * r + 1], T1);
            }
        }
        if (f.fused_bwd && c < 3) {
            float *tr = &sTR[threadIdx.x / psikd::LPQ][3 * c];
            tr[0] = T0.x; tr[1] = T0.y; tr[2] = T1.x;
        }
        float xr = psi_dot3p(T0.x, T0.y, T1.x, T1.y, px, py, pz) + sCT[12 + r];
        const int base = (threadIdx.x & 63) & ~3;
        const float x = __shfl(xr, base, 64), y = __shfl(xr, base + 1, 64), z = __shfl(xr, base + 2, 64);
        const float *C = sCT;                                  // camera rows from LDS (broadcast reads)
        qx = psi_dot3p(C[0], C[1], C[2], C[3], x, y, z);
        qy = psi_dot3p(C[4], C[5], C[6], C[7], x, y, z);
        qz = psi_dot3p(C[8], C[9], C[10], C[11], x, y, z);
    }
    // lane 0 of a group with a query, after the search: (gx, gy, gz) = d(contact term) / d(query point), already scaled (gscale).  The same
    // two maps as the skinning workgroups' backward, through the group's own blend: g_local = R_c^T g, g_vposed = T_R^T g_local, stored in
    // SLOT order next to the posed vertex itself (skin_bwd_A's second operand) — the contact slots are a class of "vertices" of their own
    // in the joint-side contractions (fit_bwd_joint_kernel), never merged with the penetration rows, which still lack their 1 / N
    __device__ __forceinline__ void contact_post(size_t, float gx, float gy, float gz)
    {
        if (!f.fused_bwd) return;
        const float *C = sCT;
        const float lx = psi_dot3(C[0], C[4], C[8], gx, gy, gz), ly = psi_dot3(C[1], C[5], C[9], gx, gy, gz), lz = psi_dot3(C[2], C[6], C[10], gx, gy, gz);
        const float *R = sTR[threadIdx.x / psikd::LPQ];
        const size_t row = (size_t)bq * f.ncp3;
        const unsigned off = (unsigned)jslot * 12u;
        psi_st(f.glc + row, off, psi_p3{lx, ly, lz});
        const float vx = psi_dot3(R[0], R[3], R[6], lx, ly, lz), vy = psi_dot3(R[1], R[4], R[7], lx, ly, lz), vz = psi_dot3(R[2], R[5], R[8], lx, ly, lz);
        psi_st(f.gvpc + row, off, psi_p3{vx, vy, vz});
        gvm = fmaxf(fabsf(vx), fmaxf(fabsf(vy), fabsf(vz)));
        psi_st(f.vpc + row, off, psi_p3{px, py, pz});
        gs[0] = lx; gs[1] = ly; gs[2] = lz;
    }
    __device__ __forceinline__ void contact_finish(int b, int bx, int nbx)
    {
        if (!f.fused_bwd) return;
        __shared__ float wsum3[psikd::QBLK / 64][3];
        __shared__ float wmax[psikd::QBLK / 64];
        const float sx = psi_wave_sum(gs[0]), sy = psi_wave_sum(gs[1]), sz = psi_wave_sum(gs[2]);
        float mx = gvm;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
        if ((threadIdx.x & 63) == 0) {
            wsum3[threadIdx.x >> 6][0] = sx;
            wsum3[threadIdx.x >> 6][1] = sy;
            wsum3[threadIdx.x >> 6][2] = sz;
            wmax[threadIdx.x >> 6] = mx;
        }
        __syncthreads();
        if (threadIdx.x == 64) {
            float m2 = wmax[0];
#pragma unroll
            for (int w = 1; w < psikd::QBLK / 64; w++) m2 = fmaxf(m2, wmax[w]);
#ifndef PSI_NO_GVMAX_ATOMIC
            if (m2 > 0.0f) atomicMax(f.gvbits + PSI_GV_SLOTS + (blockIdx.x & (PSI_GV_SLOTS - 1)), __float_as_uint(m2));
#endif
        }
        if (threadIdx.x < 3) {
            float a = 0.0f;
#pragma unroll
            for (int w = 0; w < psikd::QBLK / 64; w++) a += wsum3[w][threadIdx.x];
            f.gtc_part[((size_t)b * nbx + bx) * 4 + threadIdx.x] = a;
        }
    }
};

#ifdef PSI_HEAD_STOPS
extern "C" int psi_dbg_kd_mark(unsigned long long *out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psi_kd_mark), sizeof(unsigned long long) * 4 * (size_t)(nblocks < 8192 ? nblocks : 8192));
}
extern "C" int psi_dbg_kd_reason(int *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psi_kd_reason), sizeof(int) * 8); }
extern "C" int psi_dbg_kd_stat(int *out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psi_kd_stat), sizeof(int) * 4 * (size_t)(nblocks < 8192 ? nblocks : 8192));
}
#endif
// ONE launch for the two scene terms of the forward pass: block ids [0, n_kd) are the NN-search workgroups (64 contact queries of
// one body each; long, VALU-issue-bound pointer chases — dispatched first), the rest are the skinning + SDF workgroups (256
// vertices of one body each; gather-latency-bound).  As two launches they ran back to back (23 + 18 us); they depend on the same
// inputs only, so in one grid their waves share the SIMDs and hide each other's stalls.
template <int NB>
__global__ __launch_bounds__(256, 6) void fwd_scene_kernel(FitDev f, LbsDev m, const float *__restrict__ As, const float *__restrict__ v_posed,
                                                           psikd::KdDev T, int n_kd, int nqb, int rows, float gscale, int skin_first, SdfPenEpilogue epi)
{
    extern __shared__ int smem_i[];
#ifdef PSI_HEAD_STOPS
    if (psi_dbg_sstop == 1 && (int)blockIdx.x >= n_kd) return;      // NN-search workgroups only
    if (psi_dbg_sstop == 2 && (int)blockIdx.x < n_kd) return;       // skinning + SDF workgroups only
    if (psi_dbg_sstop == 3) return;                                  // neither: the launch itself
    PsiBlockTrace trace;
    if (threadIdx.x < 4 && blockIdx.x < 8192) psi_kd_stat[4 * blockIdx.x + threadIdx.x] = 0;     // (the search's atomics come after a barrier)
    trace.kind = skin_first ? ((int)blockIdx.x >= (int)gridDim.x - n_kd) : ((int)blockIdx.x < n_kd);
#endif
    // The NN-search workgroups come FIRST in the grid: they are the long ones.  Measured with the skinning workgroups first the launch
    // takes 43.6 us instead of 36.6 (HIP-event), with the two kinds spread evenly through the grid 45.8.
    int bid = blockIdx.x;
    bool is_kd;
    if (skin_first) {                                         // the LONGER kind of workgroup is dispatched first (see the comment above)
        const int n_sk = (int)gridDim.x - n_kd;
        is_kd = bid >= n_sk;
        if (is_kd) bid -= n_sk;
    } else {
        is_kd = bid < n_kd;
        if (!is_kd) bid -= n_kd;
    }
    if (is_kd) {
        const int b = bid / nqb, bx = bid % nqb;
        psikd::kd_query_body<true, false>(T, ContactSkinSrc{f, m, As, v_posed, nullptr, {}, 0.0f, 0, 0.0f, 0.0f, 0.0f, nullptr, {}, 0}, f.n_c, (float *)nullptr, (int *)nullptr, f.cconst, gscale,
                                          f.fused_bwd ? (float *)nullptr : f.gq, f.fpart, f.nn_hint, rows, (const psikd::KdDev *)nullptr, (const int *)nullptr, bx, b, nqb, smem_i);
    } else {
        const int i = bid;
        psi_skin_fwd_body<NB, PsiBlendCompact>(m, As, v_posed, f.transl, f.cam, f.B, f.verts, epi, i % f.nsdfblk, (i / f.nsdfblk) * NB, f.nsdfblk);
    }
}

__global__ void contact_weight_table_kernel(const float *__restrict__ WT, int Vpad, const int *__restrict__ vid, int n_c, int J, float *__restrict__ Wct)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_c * PSI_JP) return;
    const int slot = i / PSI_JP, j = i % PSI_JP;
    Wct[i] = j < J ? WT[(size_t)j * Vpad + vid[slot]] : 0.0f;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void loss_finalize_kernel(FitDev f, float *stats)
{
    // one workgroup of 1024 threads; every thread keeps four independent partial sums per quantity so that its loads are all in
    // flight together (a 256-thread version with one accumulator took 44 us for the 21k partial pairs of B = 512)
    __shared__ float red[16];
    const int t = threadIdx.x, nt = 1024;
    auto total = [&](const float *p, int n, int stride) {
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int i = t;
        for (; i + 3 * nt < n; i += 4 * nt) {
            a0 += p[(size_t)i * stride];
            a1 += p[(size_t)(i + nt) * stride];
            a2 += p[(size_t)(i + 2 * nt) * stride];
            a3 += p[(size_t)(i + 3 * nt) * stride];
        }
        for (; i < n; i += nt) a0 += p[(size_t)i * stride];
        return block_sum((a0 + a1) + (a2 + a3), red);
    };
    float s_rec = total(f.recpart, f.B, 1);
    float s_vp = total(f.vppart, f.B, 1);
    float s_f = total(f.fpart, f.B * f.nfp, 1);
    float s_pen = total(f.penpart, f.B * f.nsdfblk, 2);
    float n_pen = total(f.penpart + 1, f.B * f.nsdfblk, 2);
    if (t == 0) {
        stats[0] = s_rec; stats[1] = s_vp; stats[2] = s_f; stats[3] = s_pen; stats[4] = n_pen; stats[5] = 0.0f;
        *f.step += 1;
    }
}

// ------------------------------------------------------------------------------------------------
// The backward of a batch of <= 128 bodies WITHOUT a per-vertex launch (fused_bwd).  d loss / d verts = -w_col / N * (masked SDF gradient)
// + (contact gradient) needs the batch-global penetration count N (fitting_proxe.py:155-158), which exists only when the last skinning
// workgroup of the forward has finished — the reason psi_skin_bwd_v_kernel used to be a launch of its own between fwd_scene and the joint-side
// contractions, re-reading the weight rows and the posed vertices the forward had just held in registers.  But everything between dL/dverts
// and the Adam update of the body vector is LINEAR in the gradient (lbs.py:108-116 and :94-99 backward, the pose backward, the VPoser / 6D
// chain rule), so the two parts travel separately up to the point where they are sums of a few slices:
//   * the skinning workgroups of fwd_scene do the vertex's backward on the spot for the UNSCALED penetration part (SdfPenEpilogue::backward);
//   * the search lanes, which skin their own contact vertex, do the same for the contact part (ContactSkinSrc::contact_post), into rows in
//     SLOT order — a vertex listed by several contact parts simply has several slots;
//   * fit_bwd_joint_kernel runs the two contractions over BOTH classes of rows in one grid: skin_bwd_A over the model's 41 vertex slices and
//     the n_c / 256 slot slices (weights: the wave-tiled copies WTt / WTt_c), blend_bwd over the model's columns and the 3 n_c contact
//     columns (dirs_bh / dirs_ch: the contact vertices' blend-shape columns gathered once per engine, 12.6 MB at n_c = 2048), the column
//     slices sized so that the 256 stream workgroups carry equal shares; one more workgroup produces the iteration's statistics;
//   * fit_reduce_kernel sums the slices of each class in slice order and combines  sp * (penetration class) + (contact class),
//     sp = -w_col / N  (per body in the independent-bodies mode).
// Data-parallel runs need the all-reduced N only in that last step: the collective overlaps the joint kernel instead of stalling the iteration.
// ------------------------------------------------------------------------------------------------
// the iteration's statistics [sum|dx|, sum z^2, sum f, sum|sdf-|, N, 0] from the per-workgroup partials; one workgroup, every thread four
// independent partial sums per quantity (all loads in flight together); also advances the Adam step
__device__ __forceinline__ void fit_stats_body(const FitDev &f, float *__restrict__ stats)
{
    __shared__ float red[16];
    const int t = threadIdx.x, nt = blockDim.x;
    if (f.indep) {
        // every body is its own problem (its own file in the reference's loop): the penetration mean runs over THIS body's penetrating
        // vertices; the recorded loss values are means over the bodies of each body's loss
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int b = t; b < f.B; b += nt) {
            a0 += f.recpart[b];
            a1 += f.vppart[b];
            float sf = 0, sp = 0, cn = 0;
            for (int k = 0; k < f.nfp; k++) sf += f.fpart[(size_t)b * f.nfp + k];
            for (int k = 0; k < f.nsdfblk; k++) {
                sp += f.penpart[2 * ((size_t)b * f.nsdfblk + k)];
                cn += f.penpart[2 * ((size_t)b * f.nsdfblk + k) + 1];
            }
            a2 += sf;
            a3 += cn > 0.0f ? sp / cn : 0.0f;
            f.spb[b] = cn > 0.0f ? -f.w_col / cn : 0.0f;
        }
        const float s0 = block_sum(a0, red), s1 = block_sum(a1, red), s2 = block_sum(a2, red), s3 = block_sum(a3, red);
        if (t == 0) {
            stats[0] = s0; stats[1] = s1; stats[2] = s2; stats[3] = s3; stats[4] = 0.0f; stats[5] = 0.0f;
            *f.step += 1;
        }
        return;
    }
    auto total = [&](const float *p, int n, int stride) {
        float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int i = t;
        for (; i + 3 * nt < n; i += 4 * nt) {
            a0 += p[(size_t)i * stride];
            a1 += p[(size_t)(i + nt) * stride];
            a2 += p[(size_t)(i + 2 * nt) * stride];
            a3 += p[(size_t)(i + 3 * nt) * stride];
        }
        for (; i < n; i += nt) a0 += p[(size_t)i * stride];
        return block_sum((a0 + a1) + (a2 + a3), red);
    };
    const float s_rec = total(f.recpart, f.B, 1);
    const float s_vp = total(f.vppart, f.B, 1);
    const float s_f = total(f.fpart, f.B * f.nfp, 1);
    const float s_pen = total(f.penpart, f.B * f.nsdfblk, 2);
    const float n_pen = total(f.penpart + 1, f.B * f.nsdfblk, 2);
    if (t == 0) {
        stats[0] = s_rec; stats[1] = s_vp; stats[2] = s_f; stats[3] = s_pen; stats[4] = n_pen; stats[5] = 0.0f;
        *f.step += 1;
    }
}

// grid: [n_ska skin_bwd_A workgroups: (nsv + nsv_c) slices x body groups] [n_blend stream workgroups: k-groups x (nsn_m + nsn_c) column slices x
// body groups] [one statistics workgroup when with_stats] — the short kinds first, as in bwd_joint_kernel (lbs.hip), whose bodies these are
template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void fit_bwd_joint_kernel(
    FitDev f, LbsDev m, const float *__restrict__ g_vp, const float *__restrict__ gl, const float *__restrict__ v_posed, int n_ska, int n_blend,
    int kgroups, int nbody, float *__restrict__ stats)
{
    constexpr int SMEM_B = psi_blend_bwd_h_smem_f4<(MT + 1) / 2>();
    __shared__ f4 smem[SMEM_B > SKA_SMEM_F4 ? SMEM_B : SKA_SMEM_F4];
    const int bid = blockIdx.x;
#ifdef PSI_HEAD_STOPS
    PsiBlockTrace trace(11, 11);                         // dev (PSI_SKIN_STOP=11): this launch's workgroup timeline instead of fwd_scene's (tools/timeline_joint.py)
    trace.kind = bid < n_ska ? (bid % (f.nsv + f.nsv_c) < f.nsv ? 0 : 1) : (bid < n_ska + n_blend ? 2 : 4);
#endif
    if (bid < n_ska) {
        const int nsl = f.nsv + f.nsv_c, sl = bid % nsl, b0 = (bid / nsl) * nbody;
        float *part = f.gA_part + (size_t)sl * f.B * PSI_JP * 16;
        if (sl < f.nsv) {
            const PsiSkaSlice o = {m.WTt + (size_t)sl * 4 * PSI_JP * 64, gl + (size_t)sl * 768, v_posed + (size_t)sl * 768, (size_t)m.Npad, part};
            skin_bwd_A_dispatch(o, f.B, b0, nbody, smem);
        } else {
            const int c = sl - f.nsv;
            const PsiSkaSlice o = {f.WTt_c + (size_t)c * 4 * PSI_JP * 64, f.glc + (size_t)c * 768, f.vpc + (size_t)c * 768, (size_t)f.ncp3, part};
            skin_bwd_A_dispatch(o, f.B, b0, nbody, smem);
        }
    } else if (bid < n_ska + n_blend) {
        int kg, slice, bg;
        psi_blend_bwd_place(bid - n_ska, kgroups, f.nsn_m + f.nsn_c, kg, slice, bg);
#ifdef PSI_HEAD_STOPS
        trace.kind = slice < f.nsn_m ? 2 : 3;
#endif
        float *part = f.gfeat_part + (size_t)slice * f.B * m.Kpad;
        // (the rows' fp16 scale from the class's largest entry of THIS iteration, recorded by fwd_scene_kernel; the matrix's own scale is static)
        const float dsc_inv = m.dirs_unscale * PSI_FEAT_SCALE;
        unsigned cbits = 0u;
#pragma unroll
        for (int q = 0; q < PSI_GV_SLOTS / 64; q++) cbits = max(cbits, f.gvbits[(slice < f.nsn_m ? 0 : PSI_GV_SLOTS) + q * 64 + (threadIdx.x & 63)]);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) cbits = max(cbits, (unsigned)__shfl_xor((int)cbits, o2, 64));
        if (slice < f.nsn_m) {
            const float gsc = psi_fp16_class_scale(cbits);
            const PsiBlendBwdColsH o = {m.dirs_bh, g_vp, (size_t)m.Npad, m.Kpad, m.Npad / 16, gsc, dsc_inv / gsc};
            blend_bwd_h_body<(MT + 1) / 2>(o, f.B, slice * f.spm, (slice + 1) * f.spm, part, kg, bg, smem);
        } else {
            const int c = slice - f.nsn_m;
            const float gsc = psi_fp16_class_scale(cbits);
            const PsiBlendBwdColsH o = {f.dirs_ch, f.gvpc, (size_t)f.ncp3, m.Kpad, f.ncp3 / 16, gsc, dsc_inv / gsc};
            blend_bwd_h_body<(MT + 1) / 2>(o, f.B, c * f.spc, (c + 1) * f.spc, part, kg, bg, smem);
        }
    } else {
        fit_stats_body(f, stats);
    }
}

// Sums of the split-contraction partials of both classes, in slice order, and their combination  sp * (penetration) + (contact):
// gA [B][JP][16], g_feat [B][Kpad], g_transl [B][3] — what the tail kernel reads.  One output per PSI_RSPL threads (lbs_device.h).
// `stats` is final here (single process: written by the joint kernel's statistics workgroup; data parallel: all-reduced): thread 0 records
// the iteration's loss values.
__global__ __launch_bounds__(256) void fit_reduce_kernel(FitDev f, PsiLbsView lv, const float *__restrict__ stats)
{
    const int B = f.B, Kpad = lv.m.Kpad;
    const long nA = (long)B * PSI_JP * 16, nF = (long)B * Kpad, nT = (long)B * 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = t / PSI_RSPL;
    const int s0 = (int)(t % PSI_RSPL);
    float spg = 0.0f;
    if (!f.indep) {
        const float N = stats[4];
        spg = N > 0.0f ? -f.w_col / N : 0.0f;          // d/d sdf_k of w * sum(-sdf) / N on the penetrating entries
    }
    if (i < nA) {
        const int b = (int)(i / (PSI_JP * 16));
        const float sp = f.indep ? f.spb[b] : spg;
        float r = 0.0f;
        if ((i & 15) < 12) {        // (entries 12..15 of a joint's 16 do not exist: skin_bwd_A writes the 3 x 4 gradient only)
            const float pen = psi_sum_slices_split<12>(f.gA_part + i, (size_t)nA, f.nsv, s0);
            const float con = psi_sum_slices_split<2>(f.gA_part + (size_t)f.nsv * nA + i, (size_t)nA, f.nsv_c, s0);
            r = __builtin_fmaf(sp, pen, con);
        }
        if (s0 == 0) ((float *)lv.gA)[i] = r;
    } else if (i < nA + nF) {
        const long k = i - nA;
        const int b = (int)(k / Kpad);
        const float sp = f.indep ? f.spb[b] : spg;
        const float pen = psi_sum_slices_split<8>(f.gfeat_part + k, (size_t)nF, f.nsn_m, s0);
        const float con = psi_sum_slices_split<2>(f.gfeat_part + (size_t)f.nsn_m * nF + k, (size_t)nF, f.nsn_c, s0);
        if (s0 == 0) ((float *)lv.gfeat)[k] = __builtin_fmaf(sp, pen, con);
    } else if (i < nA + nF + nT) {
        const long k = i - nA - nF;
        const int b = (int)(k >> 2), c = (int)(k & 3), cc = c < 3 ? c : 0;
        const float sp = f.indep ? f.spb[b] : spg;
        const float pen = psi_sum_slices_split<12>(lv.gt_part + (size_t)b * 4 + cc, (size_t)B * 4, lv.nvb, s0);
        const float con = psi_sum_slices_split<8>(f.gtc_part + (size_t)b * f.nfp * 4 + cc, (size_t)4, f.nfp, s0);
        if (s0 == 0 && c < 3) f.g_transl[(size_t)b * 3 + c] = __builtin_fmaf(sp, pen, con);
    }
    if (t >= 64 && t < 64 + 2 * PSI_GV_SLOTS) f.gvbits[t - 64] = 0u;         // (read by fit_bwd_joint_kernel, the launch before this one: free for the next iteration's producers)
    if (t == 0) {
        const int it = *f.step - 1;
        if (it >= 0) {
            float *h = f.history + (size_t)(it % f.max_hist) * 4;     // ring buffer
            if (f.indep) {          // mean over the bodies of each body's loss
                h[0] = f.w_rec * stats[0] / ((float)B * XD);
                h[1] = f.w_vp * stats[1] / ((float)B * NZ);
                h[2] = f.w_contact * stats[2] / ((float)B * f.n_c);
                h[3] = f.w_col * stats[3] / (float)B;
            } else {
                const float Bg = (float)B * (float)f.world, N = stats[4];
                h[0] = f.w_rec * stats[0] / (Bg * XD);
                h[1] = f.w_vp * stats[1] / (Bg * NZ);
                h[2] = f.w_contact * stats[2] / (Bg * f.n_c);
                h[3] = N > 0.0f ? f.w_col * stats[3] / N : 0.0f;
            }
        }
    }
}

// one-off (psi_fit_create): the contact slots' skinning weights in LbsDev::WTt's wave tiles, and their blend-shape columns in dirs_bh's operand order
__global__ void contact_weight_tiles_kernel(const float *__restrict__ Wct, int n_c, int ncp, float *__restrict__ WTt_c)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ncp * PSI_JP) return;
    const int tile = i / (PSI_JP * 64), j = (i / 64) % PSI_JP, s = tile * 64 + (i & 63);
    WTt_c[i] = s < n_c ? Wct[(size_t)s * PSI_JP + j] : 0.0f;
}
// the contact slots' blend-shape columns as two fp16 parts per entry, in blend_bwd_h_body's operand order (from the model's own fp16 parts)
__global__ void contact_dirs_h_kernel(const float *__restrict__ dirs_bh, int Kpad, const int *__restrict__ vid, int n_c, int ncp3, float *__restrict__ dirs_ch)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n_all = (size_t)ncp3 * Kpad;
    if (i >= n_all) return;
    // i enumerates the DESTINATION's fp16 pairs {hi, lo} in (n-step, k-tile, n half, k, n) order
    const int ne = (int)(i & 7), kr = (int)((i >> 3) & 31), nh = (int)((i >> 8) & 1), KT = Kpad / 32;
    const int kt = (int)((i >> 9) % KT), s = (int)((i >> 9) / KT);
    const int n = s * 16 + nh * 8 + ne;
    const int slot = n / 3, comp = n - 3 * slot;
    _Float16 hi = (_Float16)0.0f, lo = (_Float16)0.0f;
    if (slot < n_c) {
        const int sn = 3 * vid[slot] + comp;
        const size_t so = ((((size_t)(sn >> 4) * KT + kt) * 2) * 2 + ((sn >> 3) & 1)) * 256 + (size_t)kr * 8 + (sn & 7);
        hi = ((const _Float16 *)dirs_bh)[so];
        lo = ((const _Float16 *)dirs_bh)[so + 512];
    }
    const size_t o = ((((size_t)s * KT + kt) * 2) * 2 + nh) * 256 + (size_t)kr * 8 + ne;
    ((_Float16 *)dirs_ch)[o] = hi;
    ((_Float16 *)dirs_ch)[o + 512] = lo;
}
// Gradient source of the skinning backward (lbs_device.h): dL/dverts[b][v] = penetration part (global count) + contact
// part (vertex -> contact slots, CSR), assembled on the fly; workgroup (0,0) records the loss values of the iteration.
// LOCAL (single process): every workgroup re-derives the global penetration count from the per-workgroup partials (a
// 10 KB L2-resident read) and workgroup (0,0) also produces the iteration's statistics — there is no separate
// loss_finalize launch.  Data-parallel runs (!LOCAL) read the all-reduced statistics instead.
template <bool LOCAL>
struct FitGradSource {
    FitDev f;
    float *stats;
    float N;
    float *sNb;                    // independent-bodies mode: per-body penetration counts of this workgroup's bodies (LDS)
    int b0;
    // Split form of load() for the one-body-per-workgroup kernel, which requests everything it will need before it waits for anything:
    // issue() = the vertex's SDF gradient, its contact-slot word and (coupled single-process mode) this thread's first partial pairs of the
    // penetration statistics; issue_late() = the contact gradient, whose address needs the slot word; take() = load()'s arithmetic.
    static constexpr int NPF = 4;  // partial pairs per thread requested ahead (B * 41 / 256 = 5.1 strides at B = 32; the rest is looped)
    psi_f2 pp[NPF];
    bool pp_valid;
    struct Pre { float og[3], q[3]; int cw; };
    __device__ __forceinline__ Pre issue(int b, int v, bool live)
    {
        Pre p;
        p.cw = live ? psi_ld<int>(f.cs_first, (unsigned)v * 4u) : 0;
        for (int e = 0; e < 3; e++) { p.og[e] = 0.0f; p.q[e] = 0.0f; }
        if (live) {
            const f4 o = psi_ld<f4>(f.og + (size_t)b * f.Vpad * 4, (unsigned)v * 16u);         // body row base + lane offset: one aligned 16-byte load
            p.og[0] = o.x; p.og[1] = o.y; p.og[2] = o.z;
        }
        pp_valid = LOCAL && !f.indep;
        if (pp_valid) {
            const int n = f.B * f.nsdfblk;
#pragma unroll
            for (int k = 0; k < NPF; k++) {
                const int i = threadIdx.x + k * PSI_SKIN_BLK;
                pp[k] = i < n ? *(const psi_f2 *)(f.penpart + 2 * (size_t)i) : (psi_f2){0.0f, 0.0f};
            }
        }
        return p;
    }
    __device__ __forceinline__ void issue_late(Pre &p, int b) const
    {
        if (p.cw >> 24) {
            const psi_p3 q = psi_ld<psi_p3>(f.gq + (size_t)b * f.n_c * 3, (unsigned)(p.cw & 0xffffff) * 12u);
            p.q[0] = q.x; p.q[1] = q.y; p.q[2] = q.z;
        }
    }
    __device__ __forceinline__ void take(const Pre &p, int b, int v, float &gx, float &gy, float &gz) const
    {
        const float Nb = sNb ? sNb[b - b0] : N;
        const float sp = Nb > 0.0f ? -f.w_col / Nb : 0.0f;
        gx = sp * p.og[0]; gy = sp * p.og[1]; gz = sp * p.og[2];
        const int cnt = p.cw >> 24;
        if (cnt) { gx += p.q[0]; gy += p.q[1]; gz += p.q[2]; }
        if (cnt > 1)                                            // a vertex listed more than once among the contact ids: the rest of its slots
            for (int ci = f.cs_ptr[v] + 1; ci < f.cs_ptr[v + 1]; ci++) {
                const float *q = f.gq + ((size_t)b * f.n_c + f.cs_idx[ci]) * 3;
                gx += q[0]; gy += q[1]; gz += q[2];
            }
    }
    __device__ __forceinline__ void prepare(int b0_, int nb)
    {
        const int t = threadIdx.x;
        const bool first = blockIdx.x == 0 && blockIdx.y == 0;
        b0 = b0_;
        sNb = nullptr;
        __shared__ float red[4];
        if (f.indep) {
            // every body is its own problem (its own file in the reference's loop): the penetration mean runs over THIS body's
            // penetrating vertices — nsdfblk partial pairs per body, summed by one wave per body
            __shared__ float sN[PSI_SKIN_MB];
            for (int i0 = 0; i0 < nb; i0 += PSI_SKIN_BLK / 64) {
                const int i = i0 + (t >> 6);
                float c = 0.0f;
                if (i < nb)
                    for (int k = t & 63; k < f.nsdfblk; k += 64) c += f.penpart[2 * ((size_t)(b0 + i) * f.nsdfblk + k) + 1];
                c = psi_wave_sum(c);
                if (i < nb && (t & 63) == 0) sN[i] = c;
            }
            __syncthreads();
            sNb = sN;
            N = 0.0f;
            if (first) {                                        // history of the iteration: mean over the bodies of each body's loss
                float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                for (int b = t; b < f.B; b += PSI_SKIN_BLK) {
                    a0 += f.recpart[b];
                    a1 += f.vppart[b];
                    float sf = 0, sp = 0, cn = 0;
                    for (int k = 0; k < f.nfp; k++) sf += f.fpart[(size_t)b * f.nfp + k];
                    for (int k = 0; k < f.nsdfblk; k++) {
                        sp += f.penpart[2 * ((size_t)b * f.nsdfblk + k)];
                        cn += f.penpart[2 * ((size_t)b * f.nsdfblk + k) + 1];
                    }
                    a2 += sf;
                    a3 += cn > 0.0f ? sp / cn : 0.0f;
                }
                const float s0 = block_sum(a0, red), s1 = block_sum(a1, red), s2 = block_sum(a2, red), s3 = block_sum(a3, red);
                if (t == 0) {
                    if (LOCAL) {
                        stats[0] = s0; stats[1] = s1; stats[2] = s2; stats[3] = s3; stats[4] = 0.0f; stats[5] = 0.0f;
                        *f.step += 1;
                    }
                    const int it = *f.step - 1;
                    if (it >= 0) {
                        float *h = f.history + (size_t)(it % f.max_hist) * 4;
                        h[0] = f.w_rec * s0 / ((float)f.B * XD);
                        h[1] = f.w_vp * s1 / ((float)f.B * NZ);
                        h[2] = f.w_contact * s2 / ((float)f.B * f.n_c);
                        h[3] = f.w_col * s3 / (float)f.B;
                    }
                }
            }
            return;
        }
        float st[5];
        if (LOCAL) {
            float a = 0, c = 0;
            int i = t;
            if (pp_valid) {                                     // the first NPF strides were requested by issue()
#pragma unroll
                for (int k = 0; k < NPF; k++) { a += pp[k].x; c += pp[k].y; }
                i += NPF * PSI_SKIN_BLK;
            }
            for (; i < f.B * f.nsdfblk; i += PSI_SKIN_BLK) {
                a += f.penpart[2 * i];
                c += f.penpart[2 * i + 1];
            }
            __shared__ float red2[2 * PSI_SKIN_BLK / 64];
            block_sum2(a, c, red2);
            st[3] = a;
            st[4] = c;
            if (first) {                                        // same reductions as loss_finalize_kernel
                a = 0;
                for (int i = t; i < f.B; i += PSI_SKIN_BLK) a += f.recpart[i];
                st[0] = block_sum(a, red);
                a = 0;
                for (int i = t; i < f.B; i += PSI_SKIN_BLK) a += f.vppart[i];
                st[1] = block_sum(a, red);
                a = 0;
                for (int i = t; i < f.B * f.nfp; i += PSI_SKIN_BLK) a += f.fpart[i];
                st[2] = block_sum(a, red);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 5; i++) st[i] = stats[i];
        }
        N = st[4];
        if (first && t == 0) {
            if (LOCAL) {
                stats[0] = st[0]; stats[1] = st[1]; stats[2] = st[2]; stats[3] = st[3]; stats[4] = st[4]; stats[5] = 0.0f;
                *f.step += 1;
            }
            const float Bg = (float)f.B * (float)f.world;
            int it = *f.step - 1;
            if (it >= 0) {
                float *h = f.history + (size_t)(it % f.max_hist) * 4;     // ring buffer
                h[0] = f.w_rec * st[0] / (Bg * XD);
                h[1] = f.w_vp * st[1] / (Bg * NZ);
                h[2] = f.w_contact * st[2] / (Bg * f.n_c);
                h[3] = N > 0.0f ? f.w_col * st[3] / N : 0.0f;
            }
        }
    }
    __device__ __forceinline__ void load(int b, int v, float &gx, float &gy, float &gz) const
    {
        const float Nb = sNb ? sNb[b - b0] : N;
        const float sp = Nb > 0.0f ? -f.w_col / Nb : 0.0f;     // d/d sdf_k of w * sum(-sdf)/N on the penetrating entries
        const f4 o = *(const f4 *)(f.og + ((size_t)b * f.Vpad + v) * 4);
        gx = sp * o.x; gy = sp * o.y; gz = sp * o.z;
        for (int ci = f.cs_ptr[v]; ci < f.cs_ptr[v + 1]; ci++) {
            const float *q = f.gq + ((size_t)b * f.n_c + f.cs_idx[ci]) * 3;
            gx += q[0]; gy += q[1]; gz += q[2];
        }
    }
};

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// One workgroup per body: (1) the LBS pose-backward stage on this body's reduced gradients, (2) back-propagation through
// the 6D rotations / hand PCA / VPoser decoder, (3) the Adam update.  (The split-contraction partials are summed by a
// separate all-CU kernel: 32 workgroups pulling 7.5 MB of freshly written partials through 32 CUs took 15 us.)
// ADAM = false (psi_fit_decode_backward): the chain-rule gradient wrt the 75-D body vector is written to g_out instead of
// being combined with the fitting losses' own terms and applied.
// Tail kernel, the head kernel's mirror: LBS pose backward, Gram-Schmidt / hand-PCA backward, VPoser MLP backward, Adam.  Same
// clusters: every workgroup of a body's cluster runs the (cheap, latency-bound) pose backward redundantly, then takes the columns
// [c*512/C, ...) of W3^T — which gives it the gradient of exactly its own slice of fc2's output — and from those the PARTIAL
// product with the matching rows of W2 (1/C of the 1 MB matrix); one exchange of C x 512 floats per body, the last workgroup sums
// them in cluster order and finishes (W1^T, Adam).  The weight shares (and the last workgroup's Adam operands) are loaded at the top.
template <int C, bool ADAM>
__global__ __launch_bounds__(HB) void head_bwd_adam_kernel(FitDev f, PsiLbsView lv, float *__restrict__ g_out)
{
    constexpr int NS = NH / C, NQ = NS / 4;
    constexpr int OSPL = HB / NQ;           // splits of W3's 126 rows
    constexpr int OPER = 128 / OSPL;        // rows per split
    constexpr int RPER = NS / KQ;           // rows of W2 per split
    constexpr int O1 = NH / OS1;            // rows of W1 per slice
    constexpr int PRE3 = C > 1 ? (OPER < 16 ? OPER : 16) : 0;
    constexpr int PRE2 = C > 1 ? 16 : 0;
    static_assert(RPER % 16 == 0, "cluster too wide for the thread layout");
    const int b = blockIdx.x / C, c = blockIdx.x % C, t = threadIdx.x;
    const bool last = c == C - 1;
    __shared__ float sx[XD + 5], sg6[128], sga2[NS], sga1[NH], sgx[XD + 5];
    __shared__ float sgbetas[32], sgpose[PSI_JP * 3], sgrot[PSI_JP * 9];
    __shared__ f4 part4[HB], part1[OS1][8];
    // ---- loads that depend on nothing computed here
    const int og3 = t % NQ, os3 = t / NQ, og2 = t & 127, kq2 = t >> 7, kg1 = t & 7, os1 = t >> 3;
    const int o0 = os3 * OPER, o1 = min(o0 + OPER, NJ6);
    const float *w3 = f.W3 + c * NS + og3 * 4;
    const float *w2 = f.W2 + (size_t)(c * NS + kq2 * RPER) * NH + og2 * 4;
    const float *w1 = f.W1 + (size_t)(os1 * O1) * NZ + kg1 * 4;
    f4 w3p[PRE3 ? PRE3 : 1], w2p[PRE2 ? PRE2 : 1], w1p[O1];
#pragma unroll
    for (int i = 0; i < PRE3; i++) w3p[i] = *(const f4 *)(w3 + (size_t)min(o0 + i, NJ6 - 1) * NH);   // rows past 125 meet sg6 = 0
#pragma unroll
    for (int i = 0; i < PRE2; i++) w2p[i] = *(const f4 *)(w2 + (size_t)i * NH);
    const float *h1 = f.h1 + (size_t)b * NH, *h2 = f.h2 + (size_t)b * NH;
    const float h2v = t < NS ? h2[c * NS + t] : 0.0f;
    unsigned tag = 0;
    if (C > 1) tag = f.hx_epoch[f.B + b] + 1u;
    float h1v = 0.0f, xhrv = 0.0f, am = 0.0f, av = 0.0f, step_size = 0.0f, bc2_sqrt = 1.0f;
    int step = 0;
    const int ta = (int)threadIdx.x - 256;       // the Adam update of entry ta is done by thread 256 + ta (waves 4-5)
    float o6v[6] = {0, 0, 0, 0, 0, 0}, gtv = 0.0f;
    if (t >= 1 && t < 22)
        for (int i = 0; i < 6; i++) o6v[i] = f.o6[(size_t)b * 128 + (t - 1) * 6 + i];
    if (t >= 128 && t < 128 + 3) gtv = f.g_transl[(size_t)b * 3 + (t - 128)];
    TSTOP(1);
    if (last) {
#pragma unroll
        for (int i = 0; i < O1; i++) w1p[i] = *(const f4 *)(w1 + (size_t)i * NZ);
        h1v = h1[t];
        if (ADAM && ta >= 0 && ta < XD) {
            xhrv = f.xhr[(size_t)b * XD + ta];
            am = f.adam_m[(size_t)b * XD + ta];
            av = f.adam_v[(size_t)b * XD + ta];
            step = *f.step;
            // the bias corrections (double-precision pow: a few hundred instructions) are computed HERE, by waves that idle through
            // the pose backward, not at the end of the kernel's critical path
            const double bc1 = 1.0 - pow(f.beta1_d, (double)step);
            const double bc2 = 1.0 - pow(f.beta2_d, (double)step);
            step_size = (float)(f.lr_d / bc1);
            bc2_sqrt = (float)sqrt(bc2);
        }
    }
    psi_pose_bwd_body(lv.m, f.pose + (size_t)b * f.J * 3, lv.R, lv.Jl, lv.G, lv.gA + (size_t)b * PSI_JP * 16, lv.gfeat + (size_t)b * lv.m.Kpad, b,
                      sgbetas, sgpose, sgrot);
    const float *x = f.x + (size_t)b * XD;
    if (t < XD) { sx[t] = x[t]; sgx[t] = 0.0f; }
    if (t < 128) sg6[t] = 0.0f;
    __syncthreads();                                         // g_betas / g_pose / g_rot of this body are in LDS
    TSTOP(2);
    if (c == 0) {                                            // inspection copies (psi_fit_copy_buffer)
        for (int i = t; i < f.J * 9; i += HB) f.g_rot[(size_t)b * f.J * 9 + i] = sgrot[i];
        for (int i = t; i < f.J * 3; i += HB) f.g_pose[(size_t)b * f.J * 3 + i] = sgpose[i];
        if (t < f.NB) f.g_betas[(size_t)b * f.NB + t] = sgbetas[t];
    }
    if (t == 0) {
        float g6[6];
        gs_backward(sx + 3, sgrot, g6);
        for (int i = 0; i < 6; i++) sgx[3 + i] = g6[i];
    } else if (t >= 1 && t < 22) {
        float g6[6];
        gs_backward(o6v, sgrot + t * 9, g6);
        for (int i = 0; i < 6; i++) sg6[(t - 1) * 6 + i] = g6[i];
    } else if (t >= 64 && t < 64 + 2 * f.ncomp) {
        int i = t - 64;                                   // hand PCA backward
        const float *comp = i < f.ncomp ? f.lhc : f.rhc;
        int ii = i < f.ncomp ? i : i - f.ncomp;
        const float *gp = sgpose + (i < f.ncomp ? 75 : 120);
        float a = 0;
        for (int e = 0; e < 45; e++) a += comp[ii * 45 + e] * gp[e];
        sgx[(i < f.ncomp ? 51 : 63) + ii] = a;
    } else if (t >= 128 && t < 128 + 3) {
        sgx[t - 128] = gtv;
    } else if (t >= 160 && t < 160 + 10) {
        sgx[9 + (t - 160)] = sgbetas[t - 160];
    }
    __syncthreads();
    TSTOP(3);
    // VPoser MLP backward (weights are constants): same 16-byte / split scheme on the [out][in] layouts
    {   // g_h2[k] = sum_o W3[o][k] g6[o] for this workgroup's k: NQ k-quads x OSPL row-splits (126 rows)
        f4 a = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < PRE3; i++) a += w3p[i] * sg6[o0 + i];
#pragma unroll 8
        for (int o = o0 + PRE3; o < o1; o++) a += *(const f4 *)(w3 + (size_t)o * NH) * sg6[o];
        part4[os3 * NQ + og3] = a;
    }
    __syncthreads();
    if (t < NS) {
        float a = 0.0f;
        const float *p = (const float *)part4;
#pragma unroll
        for (int q = 0; q < OSPL; q++) a += p[q * NS + t];
        sga2[t] = a * (h2v > 0.0f ? 1.0f : 0.2f);
    }
    __syncthreads();
    TSTOP(4);
    {   // g_h1[k] (partial over this workgroup's rows o) = sum_o W2[o][k] g_a2[o]: 128 k-quads x KQ row-splits
        f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        const float *g = sga2 + kq2 * RPER;
#pragma unroll
        for (int o = 0; o < PRE2; o += 2) {
            a0 += w2p[o] * g[o];
            a1 += w2p[o + 1] * g[o + 1];
        }
#pragma unroll 8
        for (int o = PRE2; o < RPER; o += 2) {
            a0 += *(const f4 *)(w2 + (size_t)o * NH) * g[o];
            a1 += *(const f4 *)(w2 + (size_t)(o + 1) * NH) * g[o + 1];
        }
        part4[kq2 * 128 + og2] = a0 + a1;
    }
    __syncthreads();
    TSTOP(5);
    {
        // every thread sums one output over the KQ row-splits
        const float *p = (const float *)part4;
        float a = p[t];
#pragma unroll
        for (int q = 1; q < KQ; q++) a += p[q * NH + t];
        if (C == 1) {
            sga1[t] = a * (h1[t] > 0.0f ? 1.0f : 0.2f);
        } else if (!last) {
            hx_put(f.hx_gh1 + ((size_t)b * C + c) * NH + t, a, tag);
            return;
        } else {
            unsigned long long *p8 = f.hx_gh1 + (size_t)b * C * NH + t;
            unsigned long long w[C - 1 ? C - 1 : 1];
#pragma unroll
            for (int cc = 0; cc < C - 1; cc++) w[cc] = hx_peek(p8 + (size_t)cc * NH);
            float o = 0.0f;
#pragma unroll
            for (int cc = 0; cc < C - 1; cc++) o += hx_value(p8 + (size_t)cc * NH, w[cc], tag, f.hx_err);
            sga1[t] = (o + a) * (h1v > 0.0f ? 1.0f : 0.2f);
        }
    }
    __syncthreads();
    TSTOP(6);
    {   // g_z[k] = sum_o W1[o][k] g_a1[o]: 8 k-quads x 64 o-slices of 8
        f4 a = {0, 0, 0, 0};
        if (C == 1) {
#pragma unroll
            for (int i = 0; i < O1; i++) a += *(const f4 *)(w1 + (size_t)i * NZ) * sga1[os1 * O1 + i];
        } else {
#pragma unroll
            for (int i = 0; i < O1; i++) a += w1p[i] * sga1[os1 * O1 + i];
        }
        part1[os1][kg1] = a;
    }
    __syncthreads();
    if (t < 8) {
        f4 a = {0, 0, 0, 0};
        for (int os = 0; os < OS1; os++) a += part1[os][t];
        for (int e = 0; e < 4; e++) sgx[19 + t * 4 + e] = a[e];
    }
    __syncthreads();
    TSTOP(7);
    if (C > 1 && t == 0) f.hx_epoch[f.B + b] = tag;
    if (!ADAM) {
        if (t < XD) g_out[(size_t)b * XD + t] = sgx[t];
        return;
    }
    if (ta >= 0 && ta < XD) {
        const float Bg = f.indep ? 1.0f : (float)f.B * (float)f.world;
        float g = sgx[ta];
        // d/dx of w_rec * mean|xhr - x|  (fitting_proxe.py:105)
        float df = xhrv - sx[ta];
        g += f.w_rec / (Bg * XD) * (df > 0.0f ? -1.0f : (df < 0.0f ? 1.0f : 0.0f));
        // d/dz of w_vp * mean(z^2)       (fitting_proxe.py:109-110)
        if (ta >= 19 && ta < 19 + NZ) g += f.w_vp / (Bg * NZ) * 2.0f * sx[ta];
        // torch.optim.Adam (defaults: amsgrad False, weight_decay 0), fitting_proxe.py:73-74 (step_size, bc2_sqrt: top of the kernel)
        size_t o = (size_t)b * XD + ta;
        float m = am * f.beta1 + f.one_m_beta1 * g;
        float v = av * f.beta2 + f.one_m_beta2 * g * g;
        f.adam_m[o] = m;
        f.adam_v[o] = v;
        float denom = sqrtf(v) / bc2_sqrt + f.eps;
        f.x[o] = sx[ta] - step_size * (m / denom);
    }
}

// one-off re-layout of the caller's [ix][iy][iz] volume into the engine's brick order (sdf_device.h)
__global__ void sdf_to_bricks_kernel(const float *__restrict__ src, float *__restrict__ dst, int D)
{
    const int nbr = D >> 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)nbr * nbr * nbr * PSI_BRICK_FLOATS;
    if (i >= n) return;
    const int e = (int)(i % PSI_BRICK_FLOATS);
    const size_t br = i / PSI_BRICK_FLOATS;
    const int bz = (int)(br % nbr), by = (int)((br / nbr) % nbr), bx = (int)(br / ((size_t)nbr * nbr));
    float v = 0.0f;
    if (e < 125) {
        const int ix = min(4 * bx + e / 25, D - 1), iy = min(4 * by + (e / 5) % 5, D - 1), iz = min(4 * bz + e % 5, D - 1);
        v = src[((size_t)ix * D + iy) * D + iz];
    }
    dst[i] = v;
}

// ... and into the cell-major order (sdf_device.h): one thread per stored float, [brick][lx][ly][lz][dx][dy][dz]
__global__ void sdf_to_cells_kernel(const float *__restrict__ src, float *__restrict__ dst, int D)
{
    const int nbr = D >> 2;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n = (size_t)D * D * D * 8;
    if (i >= n) return;
    const int c = (int)(i & 7), l = (int)((i >> 3) & 63);
    const size_t br = i >> 9;
    const int bz = (int)(br % nbr), by = (int)((br / nbr) % nbr), bx = (int)(br / ((size_t)nbr * nbr));
    const int ix = min(4 * bx + (l >> 4) + (c >> 2), D - 1), iy = min(4 * by + ((l >> 2) & 3) + ((c >> 1) & 1), D - 1),
              iz = min(4 * bz + (l & 3) + (c & 1), D - 1);
    dst[i] = src[((size_t)ix * D + iy) * D + iz];
}

__global__ void adam_reset_kernel(float *m, float *v, int *step, int n)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { m[i] = 0.0f; v[i] = 0.0f; }
    if (i == 0) *step = 0;
}

}  // namespace

#ifndef PSI_GRAPH_UNROLL
#define PSI_GRAPH_UNROLL 10
#endif
constexpr int GRAPH_UNROLL = PSI_GRAPH_UNROLL;

struct psi_fit_engine {
    FitDev d;
    PsiSdfGrid grid;              // sampling constants of the bricked SDF volume (sdf_device.h)
    const psi_lbs_model *lbs;
    psi_nn_index *nn_index;
    float *lbs_ws;
    PsiLbsView lv;                // pointers into lbs_ws for the pose stages fused into the head / tail kernels
    void *nn_ws;
    char *blob;
    float *stats_local;           // engine-owned stats buffer (single-GPU path)
    bool merged_scene;            // skinning + SDF and the NN search in one launch (kd-tree mode, J <= 56, 4 lanes per query)
    bool self_skin;               // the NN search can skin its own contact vertices (ContactSkinSrc): it does not read `verts`
    bool keep_verts;              // the forward skinning kernel stores the camera-frame vertices (only needed when !self_skin)
    bool scene_skin_first;        // block order inside that launch: skinning + SDF workgroups before the NN-search workgroups
    bool fused_bwd;               // the skinning backward rides on fwd_scene (see fit_bwd_joint_kernel): six launches per iteration, no per-vertex backward launch
    int skin_nb;                  // bodies per workgroup of the forward skinning + SDF kernel (1 or 2: lbs_device.h)
    hipGraph_t graph, graphN, graph2N;     // one iteration / GRAPH_UNROLL iterations / twice that
    hipGraphExec_t graph_exec, graphN_exec, graph2N_exec;
    bool graph_ready, graphN_ready, graph2N_ready;
    // data-parallel path: forward and backward halves captured separately (the all-reduce sits between them)
    hipGraph_t g_half[2];
    hipGraphExec_t ge_half[2];
    bool half_ready[2];
    const float *half_stats[2];
    // data-parallel path with the collective issued from C (psi_fit_iterate_dp): whole iterations — forward, RCCL all-reduce, backward —
    // captured like the single-process graphs; the captured graphs bake the communicator and the statistics pointer in
    hipGraph_t g_dp[2];           // [0] one iteration, [1] GRAPH_UNROLL iterations
    hipGraphExec_t ge_dp[2];
    bool dp_ready[2];
    const void *dp_comm;
    const float *dp_stats;
    bool dp_warm;                 // the communicator has run a collective for this engine outside capture
    bool dp_no_graph;             // capturing the RCCL collective failed once on this engine: the loop stays on eager launches (still from C)
};

// head / tail launches: grid = B bodies x hc workgroups per body (template instance per cluster width)
static void launch_head_fwd(const FitDev &f, const PsiLbsView &lv, hipStream_t st)
{
    const dim3 g(f.B * f.hc), blk(HB);
    switch (f.hc) {
    case 8: hipLaunchKernelGGL(head_fwd_kernel<8>, g, blk, 0, st, f, lv); break;
    case 4: hipLaunchKernelGGL(head_fwd_kernel<4>, g, blk, 0, st, f, lv); break;
    case 2: hipLaunchKernelGGL(head_fwd_kernel<2>, g, blk, 0, st, f, lv); break;
    default: hipLaunchKernelGGL(head_fwd_kernel<1>, g, blk, 0, st, f, lv); break;
    }
}

template <bool ADAM>
static void launch_head_bwd(const FitDev &f, const PsiLbsView &lv, float *g_out, hipStream_t st)
{
    const dim3 g(f.B * f.hc), blk(HB);
    switch (f.hc) {
    case 8: hipLaunchKernelGGL((head_bwd_adam_kernel<8, ADAM>), g, blk, 0, st, f, lv, g_out); break;
    case 4: hipLaunchKernelGGL((head_bwd_adam_kernel<4, ADAM>), g, blk, 0, st, f, lv, g_out); break;
    case 2: hipLaunchKernelGGL((head_bwd_adam_kernel<2, ADAM>), g, blk, 0, st, f, lv, g_out); break;
    default: hipLaunchKernelGGL((head_bwd_adam_kernel<1, ADAM>), g, blk, 0, st, f, lv, g_out); break;
    }
}

// local: single-process iteration — the statistics are produced inside the backward's first kernel (no loss_finalize launch)
// Single-process iterations fold the statistics into the first backward kernel (every workgroup re-derives the global
// penetration count from the per-workgroup partials: no loss_finalize launch).  That re-derivation reads B * 41 partial pairs per
// workgroup — O(B^2) in total — so from B = 128 on the separate one-workgroup statistics kernel is used instead.
static inline bool fit_use_local_stats(const FitDev &f, bool local) { return local && f.B < PSI_SKIN_MB_MIN_B; }

// finalize = false (fused_bwd only): the caller produces the statistics itself (the data-parallel loop does it on a side stream)
static int fit_forward(psi_fit_engine *e, float *stats, hipStream_t st, bool local = false, bool finalize = true)
{
    FitDev &f = e->d;
    // fused_bwd: a single-process iteration takes its statistics from the joint kernel's statistics workgroup at every batch size
    local = e->fused_bwd ? (local || !finalize) : fit_use_local_stats(f, local);
    launch_head_fwd(f, e->lv, st);
    PSI_CHECK_LAUNCH("head_fwd_kernel");
    psi_mark("head_fwd_kernel", st);
    int rc = psi_lbs_blend_forward(e->lbs, f.B, e->lbs_ws, st);
    if (rc) return rc;
    float gscale = f.w_contact / ((f.indep ? 1.0f : (float)f.B * (float)f.world) * (float)f.n_c);
    if (e->nn_index && e->merged_scene) {
        // skinning + SDF and the NN search of the contact vertices as ONE launch (fwd_scene_kernel)
        const psikd::KdDev T = psi_nn_index_dev(e->nn_index);
        const int nqb = f.nfp, n_kd = nqb * f.B;
        FitDev fk = f;
        if (!e->keep_verts) fk.verts = nullptr;
        if (e->skin_nb == 2)
            hipLaunchKernelGGL(fwd_scene_kernel<2>, dim3(n_kd + f.nsdfblk * psi_cdiv(f.B, 2)), dim3(256), psikd::kd_lds_bytes(T.rows), st, fk, e->lv.m,
                               e->lv.A, e->lv.v_posed, T, n_kd, nqb, T.rows, gscale, e->scene_skin_first ? 1 : 0, make_sdf_epilogue(f, e->grid, false, e->fused_bwd ? &e->lv : nullptr));
        else
            hipLaunchKernelGGL(fwd_scene_kernel<1>, dim3(n_kd + f.nsdfblk * f.B), dim3(256), psikd::kd_lds_bytes(T.rows), st, fk, e->lv.m, e->lv.A,
                               e->lv.v_posed, T, n_kd, nqb, T.rows, gscale, e->scene_skin_first ? 1 : 0, make_sdf_epilogue(f, e->grid, false, e->fused_bwd ? &e->lv : nullptr));
        PSI_CHECK_LAUNCH("fwd_scene_kernel");
        psi_mark("fwd_scene_kernel", st);
        if (local) return 0;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, f, stats);
        PSI_CHECK_LAUNCH("loss_finalize_kernel");
        psi_mark("loss_finalize_kernel", st);
        return 0;
    }
    // (Round 3 tried the dense blend of this kernel on the matrix cores at large batches — v_mfma_f32_16x16x4_f32 tiles of 16 vertices with
    // the weights as the register-resident B operand and a body's transforms as the LDS-staged A operand, bit-identical results — on the
    // grounds that the kernel is vector-ALU bound at B = 512 (805 VALU instructions per wave, 660 of them the blend; profiles/
    // r03_pmc_skin_fwd_sdf_b512.txt).  It measured 191 us against 165: the fp32 MFMA runs at the vector FLOP rate and, as far as these timings
    // show, does not overlap the other waves' vector instructions, so the blend's cycles moved but did not disappear.)
    // separate launches (large batches): the search reads its contact vertices from `verts` — letting its lane groups skin them
    // themselves, as in the shared launch, measured 157 us against 113 for the search at B = 512 — so the skinning kernel stores those
    // rows, and ONLY those (2048 of 10475: the rest of the 64 MB was written for nobody; PSI_KEEP_VERTS=1 stores all of them)
    const bool all_verts = !e->nn_index || (getenv("PSI_KEEP_VERTS") && getenv("PSI_KEEP_VERTS")[0] == '1');
    if (e->skin_nb == 2)
        hipLaunchKernelGGL((psi_skin_fwd_kernel<SdfPenEpilogue, 2>), dim3(f.nsdfblk, psi_cdiv(f.B, 2)), dim3(PSI_SKIN_BLK), 0, st, e->lv.m, e->lv.A,
                           e->lv.v_posed, f.transl, f.cam, f.B, f.verts, make_sdf_epilogue(f, e->grid, !all_verts));
    else
        hipLaunchKernelGGL((psi_skin_fwd_kernel<SdfPenEpilogue, 1>), dim3(f.nsdfblk, f.B), dim3(PSI_SKIN_BLK), 0, st, e->lv.m, e->lv.A,
                           e->lv.v_posed, f.transl, f.cam, f.B, f.verts, make_sdf_epilogue(f, e->grid, !all_verts));
    PSI_CHECK_LAUNCH("skin_fwd_sdf_kernel");
    psi_mark("skin_fwd_sdf_kernel", st);
    if (e->nn_index && !all_verts)                               // the contact rows in slot order: query j is row j
        rc = psi_nn_index_contact(e->nn_index, f.cverts, (long)f.n_c * 3, nullptr, f.B, f.n_c, f.cconst, gscale, f.gq, f.fpart, f.nn_hint, st);
    else if (e->nn_index)
        rc = psi_nn_index_contact(e->nn_index, f.verts, (long)f.V * 3, f.vid, f.B, f.n_c, f.cconst, gscale, f.gq, f.fpart, f.nn_hint, st);
    else
        rc = psi_nn_contact(f.verts, (long)f.V * 3, f.vid, f.scene, f.B, f.n_c, f.m, e->nn_ws, f.cconst, gscale, f.gq, f.fpart, nullptr, st);
    if (rc) return rc;
    if (local) return 0;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, f, stats);
    PSI_CHECK_LAUNCH("loss_finalize_kernel");
    psi_mark("loss_finalize_kernel", st);
    return 0;
}

static int fit_launch_stats(psi_fit_engine *e, float *stats, hipStream_t st)
{
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1024), 0, st, e->d, stats);
    PSI_CHECK_LAUNCH("loss_finalize_kernel");
    psi_mark("loss_finalize_kernel", st);
    return 0;
}

// fused_bwd: the joint-side contractions over both classes of rows (+ the statistics workgroup when `local`); the rest of the backward
// (fit_backward_tail) needs the FINAL statistics
static int fit_backward_joint(psi_fit_engine *e, float *stats, hipStream_t st, bool local)
{
    FitDev &f = e->d;
    const LbsDev &m = e->lv.m;
    const int kgroups = m.Kpad / 64;
    const int mt = f.B > 32 ? 4 : (f.B > 16 ? 2 : 1);
    const int bgroups = psi_cdiv(f.B, 16 * mt);
    const int n_blend = kgroups * (f.nsn_m + f.nsn_c) * bgroups;
    // bodies per skin_bwd_A workgroup: about one such workgroup per CU beside its stream workgroup
    const int nsl = f.nsv + f.nsv_c;
    // (AT MOST one: the body groups are whole, so 32 bodies in groups of 6 are 6 groups, not 5.33 — 45 slices x 6 = 270 workgroups put two on
    // 14 CUs and the launch waited for those: bwd_joint 37.6 us at n_c = 1024 against 28.4 at 2048)
    int nbody = psi_cdiv((long)f.B * nsl, 256);
    if (nbody < 1) nbody = 1;
    while (nbody < SKA_NBODY && (long)nsl * psi_cdiv(f.B, nbody) > 256) nbody++;
    if (nbody > SKA_NBODY) nbody = SKA_NBODY;
    if (const char *ev = getenv("PSI_SKA_NBODY")) { int v = atoi(ev); if (v >= 1 && v <= SKA_NBODY) nbody = v; }
    const int n_ska = nsl * psi_cdiv(f.B, nbody);
    const dim3 grid(n_ska + n_blend + (local ? 1 : 0));
#define PSI_LAUNCH_FIT_JOINT(MT_)                                                                                                  \
    hipLaunchKernelGGL(fit_bwd_joint_kernel<MT_>, grid, dim3(256), 0, st, f, m, e->lv.g_vp, e->lv.gl, e->lv.v_posed, n_ska, n_blend, kgroups, nbody, stats)
    if (mt == 4) PSI_LAUNCH_FIT_JOINT(4);
    else if (mt == 2) PSI_LAUNCH_FIT_JOINT(2);
    else PSI_LAUNCH_FIT_JOINT(1);
#undef PSI_LAUNCH_FIT_JOINT
    PSI_CHECK_LAUNCH("fit_bwd_joint_kernel");
    psi_mark("bwd_joint_kernel", st);
    return 0;
}

static int fit_backward_tail(psi_fit_engine *e, float *stats, hipStream_t st)
{
    FitDev &f = e->d;
    const long nred = (long)f.B * PSI_JP * 16 + (long)f.B * e->lv.m.Kpad + (long)f.B * 4;
    hipLaunchKernelGGL(fit_reduce_kernel, dim3(psi_cdiv(nred * PSI_RSPL, 256)), dim3(256), 0, st, f, e->lv, stats);
    PSI_CHECK_LAUNCH("fit_reduce_kernel");
    psi_mark("reduce_partials_kernel", st);
    launch_head_bwd<true>(f, e->lv, nullptr, st);
    PSI_CHECK_LAUNCH("head_bwd_adam_kernel");
    psi_mark("head_bwd_adam_kernel", st);
    return 0;
}

static int fit_backward(psi_fit_engine *e, float *stats, hipStream_t st, bool local = false)
{
    FitDev &f = e->d;
    if (e->fused_bwd) {
        int rc = fit_backward_joint(e, stats, st, local);
        return rc ? rc : fit_backward_tail(e, stats, st);
    }
    local = fit_use_local_stats(f, local);
    // large batches: compressed rows keep the multi-body kernel (a lane's weights in registers, eight bodies per workgroup); DENSE rows
    // take one body per workgroup with the pipelined scalar-cache blend — the multi-body kernel reads the transforms as LDS broadcasts
    // and is bound by the LDS return path with 55-joint rows (PSI_BWDV_MB=1: the multi-body kernel for dense rows too)
    static const bool dense_mb = getenv("PSI_BWDV_MB") && getenv("PSI_BWDV_MB")[0] == '1';
    const bool big = f.B >= PSI_SKIN_MB_MIN_B;
    const bool mb = big && (e->lv.m.Wc || dense_mb);
    const dim3 bgrid(f.nsdfblk, mb ? psi_cdiv(f.B, PSI_SKIN_MB) : f.B);
    if (local && mb)
        psi_launch_skin_bwd_v_mb(e->lv.m, e->lv.A, FitGradSource<true>{f, stats, 0.0f, nullptr, 0, {}, false}, f.cam, f.B, e->lv.gl, e->lv.g_vp, e->lv.gt_part_w, st);
    else if (local)
        hipLaunchKernelGGL(psi_skin_bwd_v_kernel<FitGradSource<true>>, bgrid, dim3(PSI_SKIN_BLK), 0, st, e->lv.m, e->lv.A,
                           FitGradSource<true>{f, stats, 0.0f, nullptr, 0, {}, false}, f.cam, f.B, e->lv.gl, e->lv.g_vp, e->lv.gt_part_w);
    else if (mb)
        psi_launch_skin_bwd_v_mb(e->lv.m, e->lv.A, FitGradSource<false>{f, stats, 0.0f, nullptr, 0, {}, false}, f.cam, f.B, e->lv.gl, e->lv.g_vp, e->lv.gt_part_w, st);
    else if (big)
        hipLaunchKernelGGL((psi_skin_bwd_v_kernel<FitGradSource<false>, PsiBlendPipelined>), bgrid, dim3(PSI_SKIN_BLK), 0, st, e->lv.m, e->lv.A,
                           FitGradSource<false>{f, stats, 0.0f, nullptr, 0, {}, false}, f.cam, f.B, e->lv.gl, e->lv.g_vp, e->lv.gt_part_w);
    else
        hipLaunchKernelGGL(psi_skin_bwd_v_kernel<FitGradSource<false>>, bgrid, dim3(PSI_SKIN_BLK), 0, st, e->lv.m, e->lv.A,
                           FitGradSource<false>{f, stats, 0.0f, nullptr, 0, {}, false}, f.cam, f.B, e->lv.gl, e->lv.g_vp, e->lv.gt_part_w);
    PSI_CHECK_LAUNCH("skin_bwd_v_grad_kernel");
    psi_mark("skin_bwd_v_grad_kernel", st);
    int rc = psi_lbs_backward_joint_parts(e->lbs, f.B, e->lbs_ws, f.g_transl, st);
    if (rc) return rc;
    launch_head_bwd<true>(f, e->lv, nullptr, st);
    PSI_CHECK_LAUNCH("head_bwd_adam_kernel");
    psi_mark("head_bwd_adam_kernel", st);
    return 0;
}

extern "C" int psi_fit_create(psi_fit_engine **out, const psi_lbs_model *lbs, const psi_fit_config *cfg,
                              const float *h_w1, const float *h_b1, const float *h_w2, const float *h_b2,
                              const float *h_w3, const float *h_b3, const float *h_lh_comp, const float *h_rh_comp,
                              const float *h_pose_mean, const int32_t *h_contact_ids,
                              const float *d_scene_verts, const float *d_sdf, const float *h_gmin, const float *h_gmax)
{
    PSI_REQUIRE(out && lbs && cfg && h_w1 && h_b1 && h_w2 && h_b2 && h_w3 && h_b3 && h_lh_comp && h_rh_comp && h_pose_mean &&
                h_contact_ids && d_scene_verts && d_sdf && h_gmin && h_gmax, "null pointer");
    int V, J, NB;
    psi_lbs_dims(lbs, &V, &J, &NB);
    PSI_REQUIRE(J == 55 && NB >= 10 && NB <= 32, "the fused engine is SMPL-X shaped (J=55, 10..32 betas)");
    PSI_REQUIRE(cfg->B > 0 && cfg->n_contact > 0 && cfg->n_contact < (1 << 24) && cfg->m_scene > 0 && cfg->D >= 2 && cfg->world_size >= 1, "bad sizes");
    PSI_REQUIRE(cfg->num_pca_comps > 0 && cfg->num_pca_comps <= 12, "1..12 hand PCA components");
    for (int i = 0; i < 66; i++) PSI_REQUIRE(h_pose_mean[i] == 0.0f, "pose_mean must be zero for global_orient/body joints");
    for (int i = 0; i < cfg->n_contact; i++) PSI_REQUIRE(h_contact_ids[i] >= 0 && h_contact_ids[i] < V, "contact id out of range");
    psi_fit_engine *e = new psi_fit_engine;
    memset(e, 0, sizeof(*e));
    e->lbs = lbs;
    FitDev &f = e->d;
    f.B = cfg->B; f.V = V; f.J = J; f.NB = NB; f.n_c = cfg->n_contact; f.m = cfg->m_scene; f.D = cfg->D;
    f.align_corners = cfg->align_corners; f.world = cfg->world_size; f.ncomp = cfg->num_pca_comps;
    f.indep = cfg->independent_bodies ? 1 : 0;
    PSI_REQUIRE(!(f.indep && cfg->world_size > 1), "independent bodies have no cross-rank coupling: use world_size 1");
    f.w_rec = cfg->w_rec; f.w_vp = cfg->w_vposer; f.w_contact = cfg->w_contact; f.w_col = cfg->w_collision; f.cconst = cfg->contact_const;
    f.lr = cfg->lr; f.beta1 = cfg->beta1; f.beta2 = cfg->beta2; f.eps = cfg->eps;
    // the hyper-parameters as the doubles the caller's optimiser holds (0: only the fp32 fields were filled in)
    f.lr_d = cfg->lr_d != 0.0 ? cfg->lr_d : (double)cfg->lr;
    f.beta1_d = cfg->beta1_d != 0.0 ? cfg->beta1_d : (double)cfg->beta1;
    f.beta2_d = cfg->beta2_d != 0.0 ? cfg->beta2_d : (double)cfg->beta2;
    f.one_m_beta1 = (float)(1.0 - f.beta1_d);
    f.one_m_beta2 = (float)(1.0 - f.beta2_d);
    // one launch for both scene terms up to B = 128 (measured: 0.1695 -> 0.1611 ms per iteration at B = 32, 0.2342 -> 0.2258 at 64, 0.3957 ->
    // 0.3922 at 128; at 256 and above both parts are throughput-bound and the shared launch is 1-2 % slower); PSI_SPLIT_SCENE=1: two launches.
    // (Round 3 also tried the opposite arrangement — no search workgroups at all, every skinning workgroup answering the contact queries of
    // its own 256 vertices from LDS after the SDF lookup: 1312 workgroups = one occupancy round instead of two, no second skinning of the
    // query vertices.  It measured 42 us against 33: the search became a serial tail of every workgroup instead of running beside them.)
    e->self_skin = cfg->nn_mode == 1 && J <= PSI_JP - 8 && psikd::LPQ == 4;      // the search lanes can skin their own contact vertex
    e->merged_scene = e->self_skin && cfg->B <= 128 && !(getenv("PSI_SPLIT_SCENE") && getenv("PSI_SPLIT_SCENE")[0] == '1');
    // the vertices themselves are an output nobody reads when the search skins its own queries (the shared launch): not stored there
    // (psi_fit_copy_buffer("verts") produces them on demand); with separate launches the search reads its contact rows, which are the
    // only ones stored (fit_forward); PSI_KEEP_VERTS=1 stores all of them in every iteration
    e->keep_verts = !e->merged_scene || (getenv("PSI_KEEP_VERTS") && getenv("PSI_KEEP_VERTS")[0] == '1');
    f.nsdfblk = psi_cdiv(V, 256);
    f.Vpad = f.nsdfblk * 256;
    f.nfp = cfg->nn_mode == 1 ? psi_nn_index_fparts(f.n_c) : psi_nn_contact_fparts(f.n_c);
    f.max_hist = cfg->max_history > 0 ? cfg->max_history : 1024;
    // workgroups per body in the head / tail kernels: enough to put ~256 workgroups on the chip, none once the bodies alone do
    {
        const long bodies = (long)cfg->B * (cfg->concurrent_engines > 1 ? cfg->concurrent_engines : 1);
        f.hc = bodies <= 32 ? 8 : bodies <= 64 ? 4 : bodies <= 128 ? 2 : 1;
    }
#ifdef PSI_HEAD_STOPS
    f.stop_h = getenv("PSI_HEAD_STOP") ? atoi(getenv("PSI_HEAD_STOP")) : 0;
    f.stop_t = getenv("PSI_TAIL_STOP") ? atoi(getenv("PSI_TAIL_STOP")) : 0;
    {
        int sst = getenv("PSI_SKIN_STOP") ? atoi(getenv("PSI_SKIN_STOP")) : 0;
        (void)hipMemcpyToSymbol(HIP_SYMBOL(psi_dbg_sstop), &sst, sizeof(int));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(psi_dbg_pstop), &f.stop_t, sizeof(int));
    }
#endif
    if (const char *hcv = getenv("PSI_HEAD_CLUSTER")) {
        const int v = atoi(hcv);
        if (v == 1 || v == 2 || v == 4 || v == 8) f.hc = v;
    }
    f.scene = d_scene_verts;
    f.sdf = d_sdf;
    const int B = f.B;
    // host staging of constants
    std::vector<float> W1T((size_t)NZ * NH), W2T((size_t)NH * NH), W3T((size_t)NH * 128, 0.0f), b3p(128, 0.0f);
    memcpy(b3p.data(), h_b3, NJ6 * 4);
    for (int o = 0; o < NH; o++) for (int k = 0; k < NZ; k++) W1T[(size_t)k * NH + o] = h_w1[(size_t)o * NZ + k];
    for (int o = 0; o < NH; o++) for (int k = 0; k < NH; k++) W2T[(size_t)k * NH + o] = h_w2[(size_t)o * NH + k];
    for (int o = 0; o < NJ6; o++) for (int k = 0; k < NH; k++) W3T[(size_t)k * 128 + o] = h_w3[(size_t)o * NH + k];
    std::vector<int> cs_ptr(V + 1, 0), cs_idx(f.n_c), cs_first(V, 0);
    for (int i = 0; i < f.n_c; i++) cs_ptr[h_contact_ids[i] + 1]++;
    for (int v = 0; v < V; v++) cs_ptr[v + 1] += cs_ptr[v];
    {
        std::vector<int> fill(cs_ptr.begin(), cs_ptr.end() - 1);
        for (int i = 0; i < f.n_c; i++) cs_idx[fill[h_contact_ids[i]]++] = i;   // ascending slot order per vertex
        for (int v = 0; v < V; v++)
            if (cs_ptr[v + 1] > cs_ptr[v]) cs_first[v] = cs_idx[cs_ptr[v]] | (std::min(cs_ptr[v + 1] - cs_ptr[v], 127) << 24);
    }
    struct Item { const void *src; size_t bytes; size_t off; };
    std::vector<Item> items;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    auto cst = [&](const void *src, size_t bytes) { size_t r = take(bytes); items.push_back({src, bytes, r}); return r; };
    size_t o_w1t = cst(W1T.data(), W1T.size() * 4), o_b1 = cst(h_b1, NH * 4), o_w2t = cst(W2T.data(), W2T.size() * 4), o_b2 = cst(h_b2, NH * 4),
           o_w3t = cst(W3T.data(), W3T.size() * 4), o_b3 = cst(b3p.data(), 128 * 4), o_w1 = cst(h_w1, (size_t)NH * NZ * 4),
           o_w2 = cst(h_w2, (size_t)NH * NH * 4), o_w3 = cst(h_w3, (size_t)NJ6 * NH * 4), o_lh = cst(h_lh_comp, (size_t)f.ncomp * 45 * 4),
           o_rh = cst(h_rh_comp, (size_t)f.ncomp * 45 * 4), o_pm = cst(h_pose_mean, (size_t)J * 3 * 4), o_vid = cst(h_contact_ids, (size_t)f.n_c * 4),
           o_cp = cst(cs_ptr.data(), cs_ptr.size() * 4), o_ci = cst(cs_idx.data(), cs_idx.size() * 4), o_cf = cst(cs_first.data(), cs_first.size() * 4), o_gmin = cst(h_gmin, 12), o_gmax = cst(h_gmax, 12);
    size_t zero_begin = o;
    size_t o_x = take((size_t)B * XD * 4), o_xhr = take((size_t)B * XD * 4), o_cam = take((size_t)B * 16 * 4), o_am = take((size_t)B * XD * 4),
           o_av = take((size_t)B * XD * 4), o_step = take(256), o_h1 = take((size_t)B * NH * 4), o_h2 = take((size_t)B * NH * 4),
           o_o6 = take((size_t)B * 128 * 4), o_b20 = take((size_t)B * NB * 4), o_pose = take((size_t)B * J * 3 * 4), o_tr = take((size_t)B * 3 * 4),
           o_verts = take((size_t)B * V * 3 * 4), o_cv = take((size_t)B * f.n_c * 3 * 4), o_og = take((size_t)B * psi_cdiv(V, 256) * 256 * 4 * 4),
           o_gq = take((size_t)B * f.n_c * 3 * 4), o_fp = take((size_t)B * f.nfp * 4), o_pp = take((size_t)B * f.nsdfblk * 2 * 4),
           o_rp = take((size_t)B * 4), o_vp = take((size_t)B * 4), o_gb = take((size_t)B * NB * 4), o_gp = take((size_t)B * J * 3 * 4),
           o_gt = take((size_t)B * 3 * 4), o_gr = take((size_t)B * J * 9 * 4), o_hist = take((size_t)f.max_hist * 4 * 4), o_stats = take(256), o_hint = take((size_t)B * f.n_c * 4);
    size_t o_hxo = take((size_t)B * f.hc * 128 * 8), o_hxg = take((size_t)B * f.hc * NH * 8), o_hxc = take((size_t)2 * B * 4);
    size_t o_wct = take((size_t)f.n_c * PSI_JP * 4);
    // the contact slots as a second class of rows of the joint-side contractions (fused_bwd)
    e->fused_bwd = e->merged_scene && !(getenv("PSI_FIT_FUSED_BWD") && getenv("PSI_FIT_FUSED_BWD")[0] == '0');
    f.fused_bwd = e->fused_bwd ? 1 : 0;
    f.ncp = psi_cdiv(f.n_c, 256) * 256;
    f.ncp3 = 3 * f.ncp;
    size_t o_glc = 0, o_gvpc = 0, o_vpc = 0, o_gtc = 0, o_gvb = 0, o_wttc = 0, o_dirsc = 0, o_gap = 0, o_gfp = 0, o_spb = 0;
    {
        // slice counts of the model's own rows: from the LBS workspace layout (offsets only: no memory is touched through this view)
        PsiLbsView lv0;
        if (int rcv = psi_lbs_view(lbs, B, reinterpret_cast<float *>((uintptr_t)4096), &lv0)) { delete e; return rcv; }
        const int Kpad = lv0.m.Kpad, Npad = lv0.m.Npad, nsn = lv0.nsn;
        const int SM = Npad / 16, SC = f.ncp3 / 16;
        f.nsv = lv0.nsv;
        f.nsv_c = f.ncp / 256;
        // column slices of the stream workgroups, split between the two classes so that the LONGEST slice is as short as possible (a
        // proportional split left a 512-slot contact set with one 96-step slice beside 64-step ones: bwd_joint 42 us instead of 28)
        f.nsn_c = 0;
        for (int c = 1, best = INT_MAX; c < nsn; c++) {
            const int longest = std::max(psi_cdiv(SM, nsn - c), psi_cdiv(SC, c));
            if (longest < best) { best = longest; f.nsn_c = c; }
        }
        f.nsn_m = nsn - f.nsn_c;
        f.spm = psi_cdiv(SM, f.nsn_m);
        f.spc = f.nsn_c ? psi_cdiv(SC, f.nsn_c) : 0;
        if (e->fused_bwd && !f.nsn_c) { e->fused_bwd = false; f.fused_bwd = 0; }
        if (e->fused_bwd) {
            o_glc = take((size_t)B * f.ncp3 * 4); o_gvpc = take((size_t)B * f.ncp3 * 4); o_vpc = take((size_t)B * f.ncp3 * 4);
            o_gtc = take((size_t)B * f.nfp * 4 * 4); o_gvb = take(2 * PSI_GV_SLOTS * 4); o_wttc = take((size_t)f.ncp * PSI_JP * 4); o_dirsc = take((size_t)f.ncp3 * Kpad * 4);
            o_gap = take((size_t)(f.nsv + f.nsv_c) * B * PSI_JP * 16 * 4); o_gfp = take((size_t)(f.nsn_m + f.nsn_c) * B * Kpad * 4);
            o_spb = take((size_t)B * 4);
        }
    }
    // (the bricked copy is addressed with 32-bit byte offsets: 512 bytes x (D / 4)^3 must stay below 4 GB, D <= 800)
    const bool bricks = (cfg->D % 4 == 0) && cfg->D <= (PSI_SDF_CELLS ? 480 : 800) && !(getenv("PSI_SDF_LINEAR") && getenv("PSI_SDF_LINEAR")[0] == '1');
#if PSI_SDF_CELLS
    size_t o_brick = bricks ? take((size_t)f.D * f.D * f.D * 32) : 0;             // D <= 480 keeps the byte offsets below 4 GB
#else
    size_t o_brick = bricks ? take((size_t)(f.D / 4) * (f.D / 4) * (f.D / 4) * PSI_BRICK_FLOATS * 4) : 0;
#endif
    size_t lbs_floats = psi_lbs_workspace_floats(lbs, B);
    size_t o_lws = take(lbs_floats * 4), o_nws = take(psi_nn_ws_bytes(B, f.n_c, f.m));
    hipError_t err = hipMalloc((void **)&e->blob, o);
    if (err != hipSuccess) {
        delete e;
        psi_set_error("psi_fit_create: hipMalloc(%zu) failed: %s", o, hipGetErrorString(err));
        return (int)err;
    }
    err = hipMemset(e->blob + zero_begin, 0, o - zero_begin);
    if (err == hipSuccess) err = hipMemset(e->blob + o_hint, 0xff, (size_t)B * f.n_c * 4);   // -1 = no hint
    for (auto &it : items)
        if (err == hipSuccess) err = hipMemcpy(e->blob + it.off, it.src, it.bytes, hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        (void)hipFree(e->blob);
        delete e;
        psi_set_error("psi_fit_create: upload failed: %s", hipGetErrorString(err));
        return (int)err;
    }
    char *bl = e->blob;
    auto F = [&](size_t off) { return (float *)(bl + off); };
    f.W1T = F(o_w1t); f.b1 = F(o_b1); f.W2T = F(o_w2t); f.b2 = F(o_b2); f.W3T = F(o_w3t); f.b3 = F(o_b3);
    f.W1 = F(o_w1); f.W2 = F(o_w2); f.W3 = F(o_w3); f.lhc = F(o_lh); f.rhc = F(o_rh); f.pose_mean = F(o_pm);
    f.vid = (const int *)(bl + o_vid); f.cs_ptr = (const int *)(bl + o_cp); f.cs_idx = (const int *)(bl + o_ci); f.cs_first = (const int *)(bl + o_cf);
    f.gmin = F(o_gmin); f.gmax = F(o_gmax);
    f.x = F(o_x); f.xhr = F(o_xhr); f.cam = F(o_cam); f.adam_m = F(o_am); f.adam_v = F(o_av); f.step = (int *)(bl + o_step); f.hx_err = f.step + 1;
    f.h1 = F(o_h1); f.h2 = F(o_h2); f.o6 = F(o_o6); f.betas20 = F(o_b20); f.pose = F(o_pose); f.transl = F(o_tr);
    f.verts = F(o_verts); f.cverts = F(o_cv); f.og = F(o_og); f.gq = F(o_gq); f.fpart = F(o_fp); f.penpart = F(o_pp);
    f.recpart = F(o_rp); f.vppart = F(o_vp); f.g_betas = F(o_gb); f.g_pose = F(o_gp); f.g_transl = F(o_gt); f.g_rot = F(o_gr);
    f.history = F(o_hist);
    f.nn_hint = (int *)(bl + o_hint);
    f.hx_o6 = (unsigned long long *)(bl + o_hxo); f.hx_gh1 = (unsigned long long *)(bl + o_hxg); f.hx_epoch = (unsigned *)(bl + o_hxc);
    e->stats_local = F(o_stats);
    e->lbs_ws = F(o_lws);
    {
        int rcv = psi_lbs_view(lbs, B, e->lbs_ws, &e->lv);
        if (rcv) { (void)hipFree(e->blob); delete e; return rcv; }
    }
    f.Wct = F(o_wct);
    if (e->fused_bwd) {
        f.glc = F(o_glc); f.gvpc = F(o_gvpc); f.vpc = F(o_vpc); f.gtc_part = F(o_gtc); f.WTt_c = F(o_wttc); f.dirs_ch = F(o_dirsc); f.gvbits = (unsigned *)(bl + o_gvb);
        f.gA_part = F(o_gap); f.gfeat_part = F(o_gfp); f.spb = F(o_spb);
    }
    e->scene_skin_first = getenv("PSI_SCENE_ORDER") && getenv("PSI_SCENE_ORDER")[0] == '1';
    // two bodies per skinning workgroup share one pass over the vertex's weight row: from the batch size at which the kernel is
    // throughput-bound (its own launch, B > 128); PSI_SKIN_NB=1|2 overrides
    // (compressed rows are read once per lane either way and measured 2-3 % slower with two bodies: 114.7 vs 112.3 us at B = 512)
    e->skin_nb = cfg->B > 128 && !e->lv.m.Wc ? 2 : 1;
    if (const char *nbv = getenv("PSI_SKIN_NB")) e->skin_nb = atoi(nbv) == 2 ? 2 : 1;
    hipLaunchKernelGGL(contact_weight_table_kernel, dim3(psi_cdiv((long)f.n_c * PSI_JP, 256)), dim3(256), 0, 0, e->lv.m.WT, e->lv.m.Vpad, f.vid,
                       f.n_c, J, (float *)f.Wct);
    if (e->fused_bwd) {
        hipLaunchKernelGGL(contact_weight_tiles_kernel, dim3(psi_cdiv((long)f.ncp * PSI_JP, 256)), dim3(256), 0, 0, f.Wct, f.n_c, f.ncp, (float *)f.WTt_c);
        hipLaunchKernelGGL(contact_dirs_h_kernel, dim3(psi_cdiv((long)f.ncp3 * e->lv.m.Kpad, 256)), dim3(256), 0, 0, e->lv.m.dirs_bh, e->lv.m.Kpad, f.vid,
                           f.n_c, f.ncp3, (float *)f.dirs_ch);
    }
    f.sdf_brick = nullptr;
    if (bricks) {
#if PSI_SDF_CELLS
        const size_t n = (size_t)f.D * f.D * f.D * 8;
        hipLaunchKernelGGL(sdf_to_cells_kernel, dim3((unsigned)psi_cdiv((long)n, 256)), dim3(256), 0, 0, d_sdf, F(o_brick), f.D);
#else
        const size_t n = (size_t)(f.D / 4) * (f.D / 4) * (f.D / 4) * PSI_BRICK_FLOATS;
        hipLaunchKernelGGL(sdf_to_bricks_kernel, dim3((unsigned)psi_cdiv((long)n, 256)), dim3(256), 0, 0, d_sdf, F(o_brick), f.D);
#endif
        f.sdf_brick = F(o_brick);
    }
    e->grid = psi_sdf_grid_make(f.sdf_brick, h_gmin, h_gmax, f.D, f.align_corners);
    err = hipDeviceSynchronize();                                // the two one-off layout kernels above ran on the NULL stream
    if (err == hipSuccess) err = hipGetLastError();
    if (err != hipSuccess) {
        (void)hipFree(e->blob);
        delete e;
        psi_set_error("psi_fit_create: layout kernels failed: %s", hipGetErrorString(err));
        return (int)err;
    }
    e->nn_ws = bl + o_nws;
    if (cfg->nn_mode == 1) {
        std::vector<float> hs((size_t)f.m * 3);
        err = hipMemcpy(hs.data(), d_scene_verts, hs.size() * 4, hipMemcpyDeviceToHost);
        int rc = err == hipSuccess ? psi_nn_index_create(&e->nn_index, hs.data(), f.m) : (int)err;
        if (rc) {
            (void)hipFree(e->blob);
            delete e;
            return rc;
        }
    }
    *out = e;
    return 0;
}

extern "C" void psi_fit_destroy(psi_fit_engine *e)
{
    if (!e) return;
    if (e->nn_index) psi_nn_index_destroy(e->nn_index);
    if (e->graph_ready) {
        (void)hipGraphExecDestroy(e->graph_exec);
        (void)hipGraphDestroy(e->graph);
    }
    if (e->graphN_ready) {
        (void)hipGraphExecDestroy(e->graphN_exec);
        (void)hipGraphDestroy(e->graphN);
    }
    if (e->graph2N_ready) {
        (void)hipGraphExecDestroy(e->graph2N_exec);
        (void)hipGraphDestroy(e->graph2N);
    }
    for (int i = 0; i < 2; i++) {
        if (e->half_ready[i]) {
            (void)hipGraphExecDestroy(e->ge_half[i]);
            (void)hipGraphDestroy(e->g_half[i]);
        }
        if (e->dp_ready[i]) {
            (void)hipGraphExecDestroy(e->ge_dp[i]);
            (void)hipGraphDestroy(e->g_dp[i]);
        }
    }
    (void)hipFree(e->blob);
    delete e;
}

extern "C" int psi_fit_set_problem(psi_fit_engine *e, const float *d_xhr, const float *d_x_init, const float *d_cam_ext,
                                   int reset_optimizer, void *stream)
{
    PSI_REQUIRE(e && d_xhr && d_cam_ext, "null pointer");
    FitDev &f = e->d;
    hipStream_t st = (hipStream_t)stream;
    size_t nb = (size_t)f.B * XD * 4;
    PSI_CHECK_HIP(hipMemcpyAsync(f.xhr, d_xhr, nb, hipMemcpyDeviceToDevice, st));
    PSI_CHECK_HIP(hipMemcpyAsync(f.x, d_x_init ? d_x_init : d_xhr, nb, hipMemcpyDeviceToDevice, st));
    PSI_CHECK_HIP(hipMemcpyAsync(f.cam, d_cam_ext, (size_t)f.B * 16 * 4, hipMemcpyDeviceToDevice, st));
    PSI_CHECK_HIP(hipMemsetAsync(f.nn_hint, 0xff, (size_t)f.B * f.n_c * 4, st));
    if (reset_optimizer) {
        hipLaunchKernelGGL(adam_reset_kernel, dim3(psi_cdiv((long)f.B * XD, 256)), dim3(256), 0, st, f.adam_m, f.adam_v, f.step, f.B * XD);
        PSI_CHECK_LAUNCH("adam_reset_kernel");
    psi_mark("adam_reset_kernel", st);
    }
    return 0;
}

static int fit_half(psi_fit_engine *e, int which, float *stats, int use_graph, hipStream_t st)
{
    if (!use_graph) return which == 0 ? fit_forward(e, stats, st) : fit_backward(e, stats, st);
    if (e->half_ready[which] && e->half_stats[which] != stats) {       // the captured graph bakes the stats pointer in
        (void)hipGraphExecDestroy(e->ge_half[which]);
        (void)hipGraphDestroy(e->g_half[which]);
        e->half_ready[which] = false;
    }
    if (!e->half_ready[which]) {
        PSI_REQUIRE(st != nullptr, "graph capture needs a non-default stream");
        PSI_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        int rc = which == 0 ? fit_forward(e, stats, st) : fit_backward(e, stats, st);
        hipError_t ce = hipStreamEndCapture(st, &e->g_half[which]);
        if (rc) return rc;
        PSI_CHECK_HIP(ce);
        PSI_CHECK_HIP(hipGraphInstantiate(&e->ge_half[which], e->g_half[which], nullptr, nullptr, 0));
        e->half_ready[which] = true;
        e->half_stats[which] = stats;
    }
    PSI_CHECK_HIP(hipGraphLaunch(e->ge_half[which], st));
    return 0;
}

extern "C" int psi_fit_forward(psi_fit_engine *e, float *d_stats, int use_graph, void *stream)
{
    PSI_REQUIRE(e, "null engine");
    return fit_half(e, 0, d_stats ? d_stats : e->stats_local, use_graph, (hipStream_t)stream);
}

extern "C" int psi_fit_backward_step(psi_fit_engine *e, const float *d_stats, int use_graph, void *stream)
{
    PSI_REQUIRE(e, "null engine");
    return fit_half(e, 1, (float *)(d_stats ? d_stats : e->stats_local), use_graph, (hipStream_t)stream);
}

extern "C" int psi_fit_iterate(psi_fit_engine *e, int n_iter, int use_graph, void *stream)
{
    PSI_REQUIRE(e && n_iter >= 0, "bad arguments");
    PSI_REQUIRE(e->d.world == 1, "psi_fit_iterate is the single-process path; data-parallel runs call forward / all-reduce / backward_step");
    hipStream_t st = (hipStream_t)stream;
    if (!use_graph) {
        for (int i = 0; i < n_iter; i++) {
            int rc = fit_forward(e, e->stats_local, st, true);
            if (rc) return rc;
            rc = fit_backward(e, e->stats_local, st, true);
            if (rc) return rc;
        }
        return 0;
    }
    // Two graphs: one iteration, and GRAPH_UNROLL iterations back to back.  Launching a graph costs ~8 us on this stack
    // (tools/ubench_graph.hip: 9.8 us for a 1-kernel graph, +1.5-2.7 us per further kernel), i.e. 4 % of an iteration when every
    // iteration is its own launch; the iteration has no host-side state (step count, history and statistics live on the
    // device), so a 100-iteration fit is 10 launches of the long graph.
    auto capture = [&](int iters, hipGraph_t *g, hipGraphExec_t *ge) -> int {
        PSI_REQUIRE(st != nullptr, "graph capture needs a non-default stream");
        PSI_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        int rc = 0;
        for (int i = 0; i < iters && !rc; i++) {
            rc = fit_forward(e, e->stats_local, st, true);
            if (!rc) rc = fit_backward(e, e->stats_local, st, true);
        }
        hipError_t ce = hipStreamEndCapture(st, g);
        if (rc) return rc;
        PSI_CHECK_HIP(ce);
        PSI_CHECK_HIP(hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
        return 0;
    };
    int done = 0;
    // (a call of 2 GRAPH_UNROLL iterations or more — the reference's 100-iteration loop, the bench's 20-step blocks — takes the long graph
    // first: one launch per 20 iterations instead of two, 0.1115 -> 0.1111 ms per iteration, A/B on one box)
    if (n_iter >= 2 * GRAPH_UNROLL) {
        if (!e->graph2N_ready) {
            int rc = capture(2 * GRAPH_UNROLL, &e->graph2N, &e->graph2N_exec);
            if (rc) return rc;
            e->graph2N_ready = true;
        }
        for (; done + 2 * GRAPH_UNROLL <= n_iter; done += 2 * GRAPH_UNROLL) PSI_CHECK_HIP(hipGraphLaunch(e->graph2N_exec, st));
    }
    if (n_iter - done >= GRAPH_UNROLL) {
        if (!e->graphN_ready) {
            int rc = capture(GRAPH_UNROLL, &e->graphN, &e->graphN_exec);
            if (rc) return rc;
            e->graphN_ready = true;
        }
        for (; done + GRAPH_UNROLL <= n_iter; done += GRAPH_UNROLL) PSI_CHECK_HIP(hipGraphLaunch(e->graphN_exec, st));
    }
    if (done < n_iter && !e->graph_ready) {
        int rc = capture(1, &e->graph, &e->graph_exec);
        if (rc) return rc;
        e->graph_ready = true;
    }
    for (; done < n_iter; done++) PSI_CHECK_HIP(hipGraphLaunch(e->graph_exec, st));
    return 0;
}

// Data-parallel iterations with the collective issued from C: forward half -> ncclAllReduce(stats[0..5], sum) -> backward half, n_iter
// times, on ONE stream.  use_graph: 10-iteration hipGraphs (+ single-iteration graphs for the remainder) that contain the RCCL kernel,
// so a 100-iteration fit is 10 graph launches per rank and no host code runs between iterations.  The first iteration an engine runs
// with a given communicator is launched eagerly: RCCL connects its channels lazily at a communicator's first collective, and that
// setup (allocations, host synchronisation) is illegal under stream capture.
extern "C" int psi_fit_iterate_dp(psi_fit_engine *e, psi_dp_comm *comm, int n_iter, int use_graph, float *d_stats, void *stream)
{
    PSI_REQUIRE(e && comm && n_iter >= 0, "bad arguments");
    PSI_REQUIRE(psi_dp_world(comm) == e->d.world, "the communicator's size differs from psi_fit_config.world_size");
    hipStream_t st = (hipStream_t)stream;
    float *stats = d_stats ? d_stats : e->stats_local;
    auto one = [&]() -> int {
        if (e->fused_bwd) {
            // The global penetration count is needed only where the slices are summed (fit_reduce_kernel), so the statistics come from the joint
            // kernel's statistics workgroup (no separate launch) and the ONE collective of the iteration sits between the joint kernel and the
            // reduction: forward -> joint kernel -> ncclAllReduce(stats) -> reduction -> tail, all on one stream / one chain of the graph.
            // (Round 6 also built the overlapped form — statistics kernel + collective on a side stream BESIDE the joint kernel, a fork / join of
            // the captured graph: 0.1237 ms per iteration over a 1-rank RCCL group against 0.1102 for the serial chain and 0.1041 single-process
            // on the same box, profiles/r06_ab_dp_overlap.txt: a cross-stream dependency inside a hipGraph costs ~6 us on this stack, twice,
            // which is more than the 6-float collective it hides until the all-reduce itself takes longer than that.)
            int rc = fit_forward(e, stats, st, false, false);
            if (!rc) rc = fit_backward_joint(e, stats, st, true);
            if (!rc) rc = psi_dp_allreduce_sum(comm, stats, 6, st);
            return rc ? rc : fit_backward_tail(e, stats, st);
        }
        int rc = fit_forward(e, stats, st);
        if (!rc) rc = psi_dp_allreduce_sum(comm, stats, 6, st);
        if (!rc) rc = fit_backward(e, stats, st);
        return rc;
    };
    if (e->dp_comm != comm || e->dp_stats != stats) {
        for (int i = 0; i < 2; i++)
            if (e->dp_ready[i]) {
                (void)hipGraphExecDestroy(e->ge_dp[i]);
                (void)hipGraphDestroy(e->g_dp[i]);
                e->dp_ready[i] = false;
            }
        e->dp_warm = e->dp_warm && e->dp_comm == comm;
        e->dp_comm = comm;
        e->dp_stats = stats;
    }
    int done = 0;
    if (!use_graph || (!e->dp_warm && n_iter > 0)) {
        const int n_eager = use_graph ? 1 : n_iter;
        for (; done < n_eager; done++) {
            int rc = one();
            if (rc) return rc;
        }
        e->dp_warm = true;
        if (!use_graph) return 0;
    }
    // capture() = 0: graph ready; 1: the capture itself was refused (the stream's capture was invalidated by a call that is illegal under
    // capture — an RCCL build / transport that needs host work per collective); anything else: a genuine error of the sequence, with the
    // message of the call that failed, to be PROPAGATED (round 3 treated every failure as "cannot capture" and retried eagerly on a
    // communicator that may already have been aborted, overwriting the error)
    bool capture_refused = false;
    auto capture = [&](int iters, int slot) -> int {
        PSI_REQUIRE(st != nullptr, "graph capture needs a non-default stream");
        PSI_CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        int rc = 0;
        for (int i = 0; i < iters && !rc; i++) rc = one();
        hipGraph_t g = nullptr;
        const hipError_t ce = hipStreamEndCapture(st, &g);
        const bool invalidated = ce >= hipErrorStreamCaptureUnsupported && ce <= hipErrorStreamCaptureWrongThread;   // 900 .. 908
        // ... or RCCL itself declined the enqueue under capture without touching the stream (ncclInvalidArgument / ncclInvalidUsage, returned
        // synchronously by ncclAllReduce: nothing was launched, the communicator is intact) — the same "cannot capture here" answer
        const bool declined = rc == 1000 + 4 || rc == 1000 + 5;
        if (rc || ce != hipSuccess) {
            if (g) (void)hipGraphDestroy(g);
            if (invalidated || declined) {
                (void)hipGetLastError();
                capture_refused = true;
                return 1;
            }
            if (!rc) {
                psi_set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce));
                rc = (int)ce;
            }
            return rc;
        }
        e->g_dp[slot] = g;
        PSI_CHECK_HIP(hipGraphInstantiate(&e->ge_dp[slot], e->g_dp[slot], nullptr, nullptr, 0));
        e->dp_ready[slot] = true;
        return 0;
    };
    // A collective that cannot be captured must not cost the run: the first refused capture switches this engine to eager launches of
    // the same sequence — every rank takes the same decision at the same iteration, because they run the same code on the same communicator
    auto eager_rest = [&]() -> int {
        for (; done < n_iter; done++) {
            int rc = one();
            if (rc) return rc;
        }
        return 0;
    };
    if (e->dp_no_graph) return eager_rest();
    if (n_iter - done >= GRAPH_UNROLL) {
        if (!e->dp_ready[1]) {
            const int rc = capture(GRAPH_UNROLL, 1);
            if (rc && !capture_refused) return rc;
            if (rc) {
                e->dp_no_graph = true;
                return eager_rest();
            }
        }
        for (; done + GRAPH_UNROLL <= n_iter; done += GRAPH_UNROLL) PSI_CHECK_HIP(hipGraphLaunch(e->ge_dp[1], st));
    }
    if (done < n_iter && !e->dp_ready[0]) {
        const int rc = capture(1, 0);
        if (rc && !capture_refused) return rc;
        if (rc) {
            e->dp_no_graph = true;
            return eager_rest();
        }
    }
    for (; done < n_iter; done++) PSI_CHECK_HIP(hipGraphLaunch(e->ge_dp[0], st));
    return 0;
}

extern "C" int psi_fit_dp_mode(const psi_fit_engine *e)
{
    if (!e || !e->dp_warm) return 0;
    return e->dp_no_graph ? 2 : 1;
}

// ---- differentiable body decode for the CVAE training losses (train_s1.py:136-170): the same head / LBS kernels, driven
// from outside; forward state (activations, rotations, transforms) stays in the engine for the matching backward
extern "C" int psi_fit_decode_forward(psi_fit_engine *e, const float *d_x75, const float *d_cam_ext, float *d_verts, void *stream)
{
    PSI_REQUIRE(e && d_x75 && d_cam_ext && d_verts, "null pointer");
    FitDev &f = e->d;
    hipStream_t st = (hipStream_t)stream;
    PSI_CHECK_HIP(hipMemcpyAsync(f.x, d_x75, (size_t)f.B * XD * 4, hipMemcpyDeviceToDevice, st));
    PSI_CHECK_HIP(hipMemcpyAsync(f.cam, d_cam_ext, (size_t)f.B * 16 * 4, hipMemcpyDeviceToDevice, st));
    launch_head_fwd(f, e->lv, st);
    PSI_CHECK_LAUNCH("head_fwd_kernel");
    int rc = psi_lbs_blend_forward(e->lbs, f.B, e->lbs_ws, st);
    if (rc) return rc;
    hipLaunchKernelGGL(psi_skin_fwd_kernel<PsiSkinNoEpilogue>, dim3(f.nsdfblk, f.B), dim3(PSI_SKIN_BLK), 0, st, e->lv.m, e->lv.A,
                       e->lv.v_posed, f.transl, f.cam, f.B, d_verts, PsiSkinNoEpilogue());
    PSI_CHECK_LAUNCH("skin_fwd_kernel");
    return 0;
}

extern "C" int psi_fit_decode_backward(psi_fit_engine *e, const float *d_grad_verts, float *d_grad_x75, void *stream)
{
    PSI_REQUIRE(e && d_grad_verts && d_grad_x75, "null pointer");
    FitDev &f = e->d;
    hipStream_t st = (hipStream_t)stream;
    if (f.B >= PSI_SKIN_MB_MIN_B)
        psi_launch_skin_bwd_v_mb(e->lv.m, e->lv.A, PsiGradFromMemory{d_grad_verts, f.V}, f.cam, f.B, e->lv.gl, e->lv.g_vp, e->lv.gt_part_w, st);
    else
        hipLaunchKernelGGL(psi_skin_bwd_v_kernel<PsiGradFromMemory>, dim3(f.nsdfblk, f.B), dim3(PSI_SKIN_BLK), 0, st, e->lv.m, e->lv.A,
                           PsiGradFromMemory{d_grad_verts, f.V}, f.cam, f.B, e->lv.gl, e->lv.g_vp, e->lv.gt_part_w);
    PSI_CHECK_LAUNCH("skin_bwd_v_kernel");
    int rc = psi_lbs_backward_joint_parts(e->lbs, f.B, e->lbs_ws, f.g_transl, st);
    if (rc) return rc;
    launch_head_bwd<false>(f, e->lv, d_grad_x75, st);
    PSI_CHECK_LAUNCH("head_bwd_kernel");
    return 0;
}

extern "C" int psi_fit_profile(psi_fit_engine *e, int n_rep, char *h_names, int name_stride, float *h_ms, int max_stages,
                               int *h_n_stages, void *stream)
{
    // Per-kernel durations of ONE fitting iteration measured with HIP events recorded on the launch stream right after
    // every kernel launch (ungraphed replay of exactly the sequence psi_fit_iterate runs), averaged over n_rep iterations.
    PSI_REQUIRE(e && h_names && h_ms && h_n_stages && n_rep > 0 && max_stages > 0 && name_stride >= 16, "bad arguments");
    PSI_REQUIRE(e->d.world == 1, "profile the single-process path");
    hipStream_t st = (hipStream_t)stream;
    PsiStageTimer tm;
    memset(&tm, 0, sizeof(tm));
    for (int i = 0; i < 48; i++) PSI_CHECK_HIP(hipEventCreate(&tm.ev[i]));
    std::vector<double> acc(48, 0.0);
    int n_st = 0, rc = 0;
    for (int r = 0; r < n_rep && !rc; r++) {
        tm.n = 0;
        (void)hipEventRecord(tm.ev[0], st);
        tm.name[0] = "start";
        tm.n = 1;
        g_psi_timer = &tm;
        rc = fit_forward(e, e->stats_local, st, true);
        if (!rc) rc = fit_backward(e, e->stats_local, st, true);
        g_psi_timer = nullptr;
        if (rc) break;
        PSI_CHECK_HIP(hipStreamSynchronize(st));
        n_st = tm.n - 1;
        for (int i = 1; i < tm.n; i++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, tm.ev[i - 1], tm.ev[i]);
            acc[i - 1] += ms;
        }
    }
    for (int i = 0; i < 48; i++) (void)hipEventDestroy(tm.ev[i]);
    if (rc) return rc;
    if (n_st > max_stages) n_st = max_stages;
    for (int i = 0; i < n_st; i++) {
        h_ms[i] = (float)(acc[i] / n_rep);
        strncpy(h_names + (size_t)i * name_stride, tm.name[i + 1], name_stride - 1);
        h_names[(size_t)i * name_stride + name_stride - 1] = 0;
    }
    *h_n_stages = n_st;
    return 0;
}

extern "C" int psi_fit_read(psi_fit_engine *e, float *d_x_out, float *d_history_out, int n_hist, int *h_step, void *stream)
{
    PSI_REQUIRE(e, "null engine");
    FitDev &f = e->d;
    hipStream_t st = (hipStream_t)stream;
    if (d_x_out) PSI_CHECK_HIP(hipMemcpyAsync(d_x_out, f.x, (size_t)f.B * XD * 4, hipMemcpyDeviceToDevice, st));
    if (d_history_out && n_hist > 0) {
        if (n_hist > f.max_hist) n_hist = f.max_hist;
        PSI_CHECK_HIP(hipMemcpyAsync(d_history_out, f.history, (size_t)n_hist * 16, hipMemcpyDeviceToDevice, st));
    }
    if (h_step) {
        int hs[2] = {0, 0};                                      // {Adam step, cluster-exchange error word}
        PSI_CHECK_HIP(hipMemcpyAsync(hs, f.step, 8, hipMemcpyDeviceToHost, st));
        PSI_CHECK_HIP(hipStreamSynchronize(st));
        *h_step = hs[0];
        if (hs[1]) {
            (void)hipMemsetAsync(f.hx_err, 0, 4, st);
            psi_set_error("psi_fit_read: a head / tail cluster exchange timed out (a producer workgroup never published its partials); "
                          "the results of this fit are invalid.  PSI_HEAD_CLUSTER=1 runs the kernels without clusters");
            return 902;
        }
    }
    return 0;
}

extern "C" int psi_fit_read_losses(psi_fit_engine *e, int adam_step, float *d_out4, void *stream)
{
    PSI_REQUIRE(e && d_out4 && adam_step >= 1, "bad arguments");
    FitDev &f = e->d;
    PSI_CHECK_HIP(hipMemcpyAsync(d_out4, f.history + (size_t)((adam_step - 1) % f.max_hist) * 4, 16, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}

extern "C" int psi_fit_copy_buffer(psi_fit_engine *e, const char *name, float *d_out, long n_floats, void *stream)
{
    PSI_REQUIRE(e && name && d_out && n_floats >= 0, "bad arguments");
    FitDev &f = e->d;
    const float *src = nullptr;
    long cap = 0;
    if (!strcmp(name, "verts")) {
        src = f.verts; cap = (long)f.B * f.V * 3;
        if (e->nn_index) {           // not (or only partly) stored by the iteration: skinned on demand from the last forward's v_posed and transforms
            hipLaunchKernelGGL(psi_skin_fwd_kernel<PsiSkinNoEpilogue>, dim3(f.nsdfblk, f.B), dim3(PSI_SKIN_BLK), 0, (hipStream_t)stream, e->lv.m, e->lv.A,
                               e->lv.v_posed, f.transl, f.cam, f.B, f.verts, PsiSkinNoEpilogue());
            PSI_CHECK_LAUNCH("skin_fwd_kernel");
        }
    }
    else if (!strcmp(name, "pose")) { src = f.pose; cap = (long)f.B * f.J * 3; }
    else if (!strcmp(name, "g_pose")) { src = f.g_pose; cap = (long)f.B * f.J * 3; }
    else if (!strcmp(name, "g_rot")) { src = f.g_rot; cap = (long)f.B * f.J * 9; }
    else if (!strcmp(name, "stats")) { src = e->stats_local; cap = 8; }
    else if (!strcmp(name, "adam_m")) { src = f.adam_m; cap = (long)f.B * XD; }
    else if (!strcmp(name, "adam_v")) { src = f.adam_v; cap = (long)f.B * XD; }
    // the reduced gradients the tail kernel reads, and the per-vertex rows of the skinning backward (fused_bwd: the UNSCALED penetration part in
    // gl / g_vp, the contact part in slot order in glc / gvpc)
    else if (!strcmp(name, "gA")) { src = e->lv.gA; cap = (long)f.B * PSI_JP * 16; }
    else if (!strcmp(name, "gfeat")) { src = e->lv.gfeat; cap = (long)f.B * e->lv.m.Kpad; }
    else if (!strcmp(name, "g_transl")) { src = f.g_transl; cap = (long)f.B * 3; }
    else if (!strcmp(name, "gl")) { src = e->lv.gl; cap = (long)f.B * e->lv.m.Npad; }
    else if (!strcmp(name, "g_vp")) { src = e->lv.g_vp; cap = (long)f.B * e->lv.m.Npad; }
    else if (!strcmp(name, "v_posed")) { src = e->lv.v_posed; cap = (long)f.B * e->lv.m.Npad; }
    else if (!strcmp(name, "glc") && e->fused_bwd) { src = f.glc; cap = (long)f.B * f.ncp3; }
    else if (!strcmp(name, "gvpc") && e->fused_bwd) { src = f.gvpc; cap = (long)f.B * f.ncp3; }
    else if (!strcmp(name, "vpc") && e->fused_bwd) { src = f.vpc; cap = (long)f.B * f.ncp3; }
    PSI_REQUIRE(src != nullptr, "unknown buffer name");
    PSI_REQUIRE(n_floats <= cap, "buffer is smaller than requested");
    PSI_CHECK_HIP(hipMemcpyAsync(d_out, src, (size_t)n_floats * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return 0;
}
