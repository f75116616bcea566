// Device-side pieces of the SMPL-X LBS operator that more than one translation unit needs: the model descriptor, the
// workspace view, and the per-body pose stages (executed by one workgroup per body), which the fused fitting engine
// (fit.hip) inlines into its head / tail kernels.  Reference arithmetic: human_body_prior/body_model/lbs.py:165-262.
#pragma once
#ifndef PSI_TRACE
#define PSI_TRACE(lo, hi)
#endif
#ifndef PSI_SSTOP
#define PSI_SSTOP(k)
#endif
#ifndef PSI_PSTOP
#define PSI_PSTOP(k)          // dev: end the program inside the pose-backward stage at point k (fit.hip, -DPSI_HEAD_STOPS)
#endif
#include <hip/hip_runtime.h>

typedef float psi_f4 __attribute__((ext_vector_type(4)));

constexpr int PSI_JP = 64;          // padded joint count
constexpr int PSI_SUB_MAX = PSI_JP * (PSI_JP + 1) / 2;   // most subtree members over all joints (a 64-joint chain)
constexpr int PSI_ITEM_MAX = 2 * PSI_JP;                 // most chunks (psi_lbs_create sizes the chunks for this)
constexpr int PSI_NJUMP = 6;       // ceil(log2(PSI_JP)) rounds of pointer jumping cover any tree
constexpr int PSI_WNZ = 8;          // compressed skinning rows are used when no vertex has more non-zero weights than this
constexpr int PSI_A_TAIL = 16;      // zero transform rows kept behind the LAST body's A block (PsiBlendPipelined reads up to 12 ceil(J/12) + 2 rows per body)

struct LbsDev {
    int V, J, NB, P, K, Kpad, N, Npad, Vpad, maxlevel, njump;
    int dirs_tile;                                   // floats (4-byte units) between consecutive 32-column tiles of `dirs` (lbs.hip)
    float dirs_unscale;                              // 1 / (scale of the forward copy's fp16 parts x PSI_FEAT_SCALE): a power of two (lbs.hip)
    const float *dirs, *v_template, *WT, *J_t, *J_s; // dirs: the blend-shape matrix as two fp16 parts per entry in blend_fwd's operand order (lbs.hip)
    const float *dirs_bh;                            // ... and in the backward product's operand order (lbs_joint_device.h: blend_bwd_h_body)
    const float *WTt;                                // the same weights as [Vpad/64][PSI_JP][64]: a wave's 64 vertices x all joints = one contiguous 16 KB tile
    const float *Wc;                                 // compressed rows [PSI_WNZ][Vpad]: the k-th non-zero weight of each vertex (ascending joint), or nullptr
    const unsigned *Wj;                              //                 [PSI_WNZ/4][Vpad]: their joint indices, one byte each (padding: weight 0, joint 0)
    const int *parents, *level, *child_ptr, *child_idx;
    const int *jump;                                 // [PSI_NJUMP][PSI_JP]: the 2^r-th ancestor of joint j (-1: none) — pointer jumping down the chain
    // subtree sets for the chain backward (psi_pose_bwd_body): the members of subtree(j), ascending, cut into chunks of <= sub_chunk
    const unsigned char *sub_list;                   // [n_sub]    subtree members of joint 0, joint 1, ... back to back
    const unsigned int *sub_item;                    // [n_items]  one chunk: offset into sub_list | count << 16
    const unsigned char *sub_first;                  // [J + 1]    first chunk of joint j (chunks of a joint are consecutive)
    int n_sub, n_items;
};

// Pointers into an LBS workspace (psi_lbs_workspace_floats) for a batch of B bodies
struct PsiLbsView {
    LbsDev m;
    float *feat, *R, *Jl, *G, *A, *v_posed, *gl, *g_vp, *gt_part_w;
    const float *gA_part, *gfeat_part, *gt_part;     // split-contraction partials written by skin_bwd_A / blend_bwd / skin_bwd_v
    const float *gA, *gfeat;                         // their sums (reduce_partials_kernel): [B][JP][16], [B][Kpad]
    int nsv, nsn, nvb;                               // their slice counts
};

__device__ __forceinline__ void psi_rodrigues(const float *aa, float *R)
{
    // lbs.py:177-191: angle = ||aa + 1e-8||, dir = aa / angle, R = I + sin K + (1 - cos) K K
    float x = aa[0] + 1e-8f, y = aa[1] + 1e-8f, z = aa[2] + 1e-8f;
    float angle = sqrtf(x * x + y * y + z * z);
    float rx = aa[0] / angle, ry = aa[1] / angle, rz = aa[2] / angle;
    float s = sinf(angle), c1 = 1.0f - cosf(angle);
    R[0] = 1.0f + c1 * (-(ry * ry + rz * rz));
    R[1] = s * (-rz) + c1 * (rx * ry);
    R[2] = s * ry + c1 * (rx * rz);
    R[3] = s * rz + c1 * (rx * ry);
    R[4] = 1.0f + c1 * (-(rx * rx + rz * rz));
    R[5] = s * (-rx) + c1 * (ry * rz);
    R[6] = s * (-ry) + c1 * (rx * rz);
    R[7] = s * rx + c1 * (ry * rz);
    R[8] = 1.0f + c1 * (-(rx * rx + ry * ry));
}

// Pose-forward stage of body b, executed by a whole workgroup, in two parts so that a caller can run the first (which only needs
// the shape coefficients) ahead of the second (which needs the pose).  betas_b [NB] and pose_b [J*3] are THIS body's rows (global
// memory in pose_fwd_kernel, LDS in the fused fitting head kernel); sJ is the caller's LDS array for the rest joints.
// Callers: pose_fwd_kernel (lbs.hip) and the fused fitting head kernel (fit.hip).
//
// part 1: rest joints J = J_t + J_s betas into LDS (no global stores: on gfx950 a later wait for ANY load also waits for every
// earlier store of the wave, so a caller with loads still to come keeps the stores for last) ...
__device__ __forceinline__ void psi_pose_fwd_rest(const LbsDev &m, const float *betas_b, float (*sJ)[3])
{
    // one (joint, axis) pair per thread, all threads of the workgroup take part
    for (int q = threadIdx.x; q < m.J * 3; q += blockDim.x) {
        float a = m.J_t[q];
        const float *js = m.J_s + (size_t)q * m.NB;
        for (int l = 0; l < m.NB; l++) a += js[l] * betas_b[l];
        (&sJ[0][0])[q] = a;
    }
}

// The blend-shape feature row of body b as blend_fwd's MFMA operand (lbs.hip: three-term fp16 split products at fp32-class accuracy):
// entry k = v is stored as TWO fp16 parts of x = v * PSI_FEAT_SCALE — hi = fp16(x), lo = fp16((x - hi) * 2^11), 22 mantissa bits
// together — in v_mfma_f32_32x32x16_f16 operand order: [32-body tile][Kpad/16 k-steps][part][k half][32 bodies][8 k], 2 bytes each.
// The scale is a power of two (exact); |v| up to 4094 stays finite, |v| below 4e-6 goes subnormal with an ABSOLUTE error below 4e-9.
constexpr float PSI_FEAT_SCALE = 16.0f;
__device__ __forceinline__ void psi_feat_store(const LbsDev &m, float *__restrict__ feat, int b, int k, float v)
{
    float x = v * PSI_FEAT_SCALE;
    x = x > 65504.0f ? 65504.0f : (x < -65504.0f ? -65504.0f : x);      // saturates beyond |v| = 4094 instead of producing infinities; a NaN stays a NaN
    const _Float16 hi = (_Float16)x;
    const _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
    _Float16 *o = (_Float16 *)feat + (((size_t)(b >> 5) * (m.Kpad >> 4) + (k >> 4)) * 4 + ((k >> 3) & 1)) * 256 + (size_t)(b & 31) * 8 + (k & 7);
    o[0] = hi;
    o[512] = lo;                                     // the lo part: 2 k halves x 32 bodies x 8 = 512 entries on
}

// ... and its outputs: the rest joints, the shape entries / zero tail of the blend-shape feature row
__device__ __forceinline__ void psi_pose_fwd_rest_store(const LbsDev &m, const float *betas_b, int B, int b, const float (*sJ)[3],
                                                        float *__restrict__ feat, float *__restrict__ Jls)
{
    const int j = threadIdx.x, nthr = blockDim.x;
    for (int q = j; q < m.J * 3; q += nthr) Jls[(size_t)b * m.J * 3 + q] = (&sJ[0][0])[q];
    for (int l = j; l < m.NB; l += nthr) psi_feat_store(m, feat, b, l, betas_b[l]);
    for (int l = m.K + j; l < m.Kpad; l += nthr) psi_feat_store(m, feat, b, l, 0.0f);
}

// part 2: Rodrigues, pose feature, kinematic chain, skinning transforms.  sJ must be complete (the first barrier below orders it
// when part 1 ran in the same workgroup just before).
// The chain G_j = G_parent(j) [R_j | J_j - J_parent] is evaluated by POINTER JUMPING instead of a sweep over the tree levels: every
// joint starts with its local transform and in round r multiplies it from the left by the current value of its 2^r-th ancestor, so
// ceil(log2(depth)) = 4 rounds for SMPL-X's 11 levels instead of 10 (same products, associated differently).  All joints live in wave 0
// (J <= 64) and a wave's LDS operations execute in order: reads of a round precede its writes, no workgroup barrier in the loop.
struct PsiJump { int a[PSI_NJUMP]; };      // this thread's row of m.jump (a[0] = parent), loaded by the caller ahead of time

__device__ __forceinline__ PsiJump psi_load_jump(const LbsDev &m)
{
    PsiJump jp;
#pragma unroll
    for (int r = 0; r < PSI_NJUMP; r++) jp.a[r] = (threadIdx.x < m.J && r < m.njump) ? m.jump[r * PSI_JP + threadIdx.x] : -1;
    return jp;
}

__device__ __forceinline__ void psi_pose_fwd_chain(const LbsDev &m, const float *pose_b, const float *__restrict__ transl, int B, int b,
                                                   const float (*sJ)[3], const PsiJump &jp, float *__restrict__ feat,
                                                   float *__restrict__ Rs, float *__restrict__ Gs, float *__restrict__ As,
                                                   float *__restrict__ joints)
{
    const int j = threadIdx.x;
    __shared__ float sG[PSI_JP][12];
    __shared__ int sJmp[PSI_NJUMP][PSI_JP];          // the jump rows, so that the round loop below stays a (compact) loop
    const bool act = j < m.J;
    float R[9], Jl[3] = {0, 0, 0};
    if (j < PSI_JP)
#pragma unroll
        for (int r = 0; r < PSI_NJUMP; r++) sJmp[r][j] = jp.a[r];
    if (act) {
        psi_rodrigues(pose_b + j * 3, R);
        psi_f4 *Ro = (psi_f4 *)(Rs + ((size_t)b * m.J + j) * 12);      // rows padded to 4: three 16-byte stores
        Ro[0] = psi_f4{R[0], R[1], R[2], 0.0f};
        Ro[1] = psi_f4{R[3], R[4], R[5], 0.0f};
        Ro[2] = psi_f4{R[6], R[7], R[8], 0.0f};
        if (j >= 1)
            for (int e = 0; e < 9; e++) {
                int k = m.NB + (j - 1) * 9 + e;
                psi_feat_store(m, feat, b, k, R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f));
            }
    }
    __syncthreads();
    float G[12];   // row-major 3x4: [R | t]
    if (act) {
        const int par = jp.a[0];
        for (int c = 0; c < 3; c++) Jl[c] = sJ[j][c];
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) G[r * 4 + c] = R[r * 3 + c];
            G[r * 4 + 3] = par >= 0 ? Jl[r] - sJ[par][r] : Jl[r];
        }
        for (int e = 0; e < 12; e++) sG[j][e] = G[e];
    }
#pragma nounroll
    for (int r = 0; r < m.njump && j < 64; r++) {
        __builtin_amdgcn_wave_barrier();
        const int a = sJmp[r][j];
        float P[12];
        if (act && a >= 0)
            for (int e = 0; e < 12; e++) P[e] = sG[a][e];
        __builtin_amdgcn_wave_barrier();
        if (act && a >= 0) {
            float N[12];
            for (int q = 0; q < 3; q++) {
                for (int c = 0; c < 3; c++) N[q * 4 + c] = P[q * 4 + 0] * G[0 * 4 + c] + P[q * 4 + 1] * G[1 * 4 + c] + P[q * 4 + 2] * G[2 * 4 + c];
                N[q * 4 + 3] = P[q * 4 + 0] * G[3] + P[q * 4 + 1] * G[7] + P[q * 4 + 2] * G[11] + P[q * 4 + 3];
            }
            for (int e = 0; e < 12; e++) { G[e] = N[e]; sG[j][e] = N[e]; }
        }
    }
    if (act) {
        psi_f4 *Go = (psi_f4 *)(Gs + ((size_t)b * m.J + j) * 12), *Ao = (psi_f4 *)(As + ((size_t)b * m.J + j) * 12);
        for (int r = 0; r < 3; r++) {
            Go[r] = psi_f4{G[r * 4 + 0], G[r * 4 + 1], G[r * 4 + 2], G[r * 4 + 3]};
            // lbs.py:258-260: A = G - pad(G [J;0])
            Ao[r] = psi_f4{G[r * 4 + 0], G[r * 4 + 1], G[r * 4 + 2],
                           G[r * 4 + 3] - (G[r * 4 + 0] * Jl[0] + G[r * 4 + 1] * Jl[1] + G[r * 4 + 2] * Jl[2])};
        }
        if (joints)
            for (int r = 0; r < 3; r++) joints[((size_t)b * m.J + j) * 3 + r] = G[r * 4 + 3] + (transl ? transl[(size_t)b * 3 + r] : 0.0f);
    }
    // The pipelined dense blend pads J to whole groups of 12 joints and multiplies ZERO weights with the transform rows behind a body's
    // block: the next body's rows (finite) — or, for the last body, whatever lies behind the A array.  Those PSI_A_TAIL rows belong to
    // the array (ws_layout) and are written here on every forward, so that a recycled / caller-provided workspace cannot put a NaN
    // bit pattern under a zero weight.
    if (b == B - 1)
        for (int q = j; q < PSI_A_TAIL * 3; q += blockDim.x) ((psi_f4 *)(As + (size_t)B * m.J * 12))[q] = psi_f4{0.0f, 0.0f, 0.0f, 0.0f};
}


__device__ __forceinline__ void psi_pose_fwd_body(const LbsDev &m, const float *betas_b, const float *pose_b,
                                                  const float *__restrict__ transl, int B, int b, float *__restrict__ feat,
                                                  float *__restrict__ Rs, float *__restrict__ Jls, float *__restrict__ Gs,
                                                  float *__restrict__ As, float *__restrict__ joints)
{
    __shared__ float sJ[PSI_JP][3];
    const PsiJump jp = psi_load_jump(m);
    psi_pose_fwd_rest(m, betas_b, sJ);
    __syncthreads();
    psi_pose_fwd_rest_store(m, betas_b, B, b, sJ, feat, Jls);
    psi_pose_fwd_chain(m, pose_b, transl, B, b, sJ, jp, feat, Rs, Gs, As, joints);
}


// Body of the pose-backward stage for body b (whole workgroup).  gA_b [PSI_JP][16] and gfeat_b [Kpad] are THIS body's reduced
// gradients, pose_b [J*3] its pose row; the outputs g_betas_b [NB], g_pose_b [J*3], g_rot_b [J*9] are this body's rows too (global
// memory in pose_bwd_kernel, LDS in the fused fitting tail kernel).
//
// Gradient through the kinematic chain WITHOUT a sweep over the tree levels.  With G_d = G_j M_jd for every joint d in the subtree of
// j (M_jd = G_j^-1 G_d, rotations orthonormal), the gradient reaching G_j = [GR_j | Gt_j] from everything below it is
//     g(G_j).t = sum_d w_d,                  g(G_j).R = [ sum_d U_d  -  (sum_d w_d) Gt_j^T ] GR_j,      d over subtree(j)
//     w_d = gown(G_d).t,  U_d = gown(G_d).R GR_d^T + w_d Gt_d^T,    gown = the gradient G_d receives directly from its own A_d
// i.e. twelve numbers per joint summed over subtrees.  The subtree sets are constants of the model, stored as chunks of at most eight
// members: all chunks are summed at once (one thread per chunk and component), then every joint adds up its chunks — two short
// rounds over all threads of the workgroup instead of ten dependent rounds on one wave, in a fixed order.
// The stage in two parts: psi_pose_bwd_issue() requests everything that does NOT depend on the reduced gradients (this body's pose row,
// transforms, rest joints, the model's subtree / child tables), psi_pose_bwd_finish() does the arithmetic once gA_b / gfeat_b exist.  A
// caller whose gradients arrive late (the fused tail + head kernel of the fitting engine: they come through a workgroup exchange) calls the
// first part at the top of the kernel, so that those loads are in flight while it waits; psi_pose_bwd_body() = the two back to back.
constexpr int PSI_POSE_JSP = 8;                          // J_s entries of the g_betas part held in registers
struct PsiPoseBwdPre {
    float aa[3], js[PSI_POSE_JSP], G[12], Jl[3], P[12];
    int cp0, cp1, par;
};
struct PsiPoseBwdShared {
    float sX[PSI_JP][12];     // U_d (3x3 row-major), w_d
    float sS[PSI_JP][12];     // their sums over subtree(j)
    float sP[PSI_ITEM_MAX][12];               // ... per chunk
    unsigned int sItem[PSI_ITEM_MAX];
    unsigned char sList[PSI_SUB_MAX], sFirst[PSI_JP + 1];
    float sgJ[PSI_JP][3];     // gradient wrt the rest joint location J_j
    float sgrel[PSI_JP][3];
    int sChild[PSI_JP];       // child lists (CSR)
    float sgb[32][32];
};

__device__ __forceinline__ PsiPoseBwdPre psi_pose_bwd_issue(const LbsDev &m, const float *pose_b, const float *__restrict__ Jls,
                                                            const float *__restrict__ Gs, int b, bool want_betas, bool want_pose,
                                                            PsiPoseBwdShared &sh)
{
    const int j = threadIdx.x, nthr = blockDim.x;
    const bool act = j < m.J;
    PsiPoseBwdPre p;
    for (int e = 0; e < 3; e++) p.aa[e] = 0.0f;
    if (act && want_pose)
        for (int e = 0; e < 3; e++) p.aa[e] = pose_b[j * 3 + e];
    constexpr int JSP = PSI_POSE_JSP;
    const int nq = m.J * 3;
    const int npart = m.NB > 0 ? min(32, max(1, nthr / m.NB)) : 1;
    const int per = (nq + npart - 1) / npart;
    const bool par_ok = m.NB <= 32, js_pre = want_betas && par_ok && per <= JSP && j < npart * m.NB;
    const int bpart = m.NB > 0 ? j / m.NB : 0, bl = j - bpart * m.NB;
    for (int i = 0; i < JSP; i++) p.js[i] = 0.0f;
    if (js_pre)
        for (int i = 0; i < JSP; i++) {
            const int q = bpart * per + i;
            if (i < per && q < nq) p.js[i] = m.J_s[(size_t)q * m.NB + bl];
        }
    p.cp0 = act ? m.child_ptr[j] : 0;
    p.cp1 = act ? m.child_ptr[j + 1] : 0;
    p.par = act ? m.parents[j] : -1;
    if (j < m.J - 1) sh.sChild[j] = m.child_idx[j];
    for (int i = j; i < m.n_sub; i += nthr) sh.sList[i] = m.sub_list[i];
    for (int i = j; i < m.n_items; i += nthr) sh.sItem[i] = m.sub_item[i];
    for (int i = j; i <= m.J; i += nthr) sh.sFirst[i] = m.sub_first[i];
    for (int e = 0; e < 12; e++) { p.G[e] = 0.0f; p.P[e] = 0.0f; }
    for (int c = 0; c < 3; c++) p.Jl[c] = 0.0f;
    if (act) {
        const psi_f4 *Gp = (const psi_f4 *)(Gs + ((size_t)b * m.J + j) * 12);
        for (int r = 0; r < 3; r++) {
            const psi_f4 gg = Gp[r];
            for (int c = 0; c < 4; c++) p.G[r * 4 + c] = gg[c];
        }
        for (int c = 0; c < 3; c++) p.Jl[c] = Jls[((size_t)b * m.J + j) * 3 + c];
    }
    // the parent's transform for the local gradients
    if (act && p.par >= 0) {
        const psi_f4 *Pp = (const psi_f4 *)(Gs + ((size_t)b * m.J + p.par) * 12);
        for (int r = 0; r < 3; r++) {
            const psi_f4 pr = Pp[r];
            for (int c = 0; c < 4; c++) p.P[r * 4 + c] = pr[c];
        }
    }
    return p;
}

__device__ __forceinline__ void psi_pose_bwd_finish(const LbsDev &m, const PsiPoseBwdPre &pre, PsiPoseBwdShared &sh, const float *gA_b,
                                                    const float *gfeat_b, float *g_betas_b, float *g_pose_b, float *g_rot_b)
{
    const int j = threadIdx.x, nthr = blockDim.x;
    const bool act = j < m.J;
    float (&sX)[PSI_JP][12] = sh.sX;
    float (&sS)[PSI_JP][12] = sh.sS;
    float (&sP)[PSI_ITEM_MAX][12] = sh.sP;
    unsigned int (&sItem)[PSI_ITEM_MAX] = sh.sItem;
    unsigned char (&sList)[PSI_SUB_MAX] = sh.sList;
    unsigned char (&sFirst)[PSI_JP + 1] = sh.sFirst;
    float (&sgJ)[PSI_JP][3] = sh.sgJ;
    float (&sgrel)[PSI_JP][3] = sh.sgrel;
    int (&sChild)[PSI_JP] = sh.sChild;
    float gf9[9];
    const float *aa = pre.aa;
    for (int e = 0; e < 9; e++) gf9[e] = 0.0f;
    if (act && (g_pose_b || g_rot_b) && j >= 1)
        for (int e = 0; e < 9; e++) gf9[e] = gfeat_b[m.NB + (j - 1) * 9 + e];
    constexpr int JSP = PSI_POSE_JSP;
    const int nq = m.J * 3;
    const int npart = m.NB > 0 ? min(32, max(1, nthr / m.NB)) : 1;
    const int per = (nq + npart - 1) / npart;
    const bool par_ok = m.NB <= 32, js_pre = g_betas_b && par_ok && per <= JSP && j < npart * m.NB;
    const int bpart = m.NB > 0 ? j / m.NB : 0, bl = j - bpart * m.NB;
    const float *js = pre.js;
    const int cp0 = pre.cp0, cp1 = pre.cp1, par = pre.par;
    float G[12], gJ[3] = {0, 0, 0};
    for (int e = 0; e < 12; e++) G[e] = pre.G[e];
    if (act) {
        const psi_f4 *Ap = (const psi_f4 *)(gA_b + j * 16);
        float gA[12], Jl[3];
        for (int r = 0; r < 3; r++) {
            const psi_f4 aa4 = Ap[r];
            for (int c = 0; c < 4; c++) gA[r * 4 + c] = aa4[c];
        }
        for (int c = 0; c < 3; c++) Jl[c] = pre.Jl[c];
        // A = [G_R | G_t - G_R J]: own gradient of G_j and the direct part of the rest-joint gradient
        float go[9], w[3];
        for (int r = 0; r < 3; r++) {
            w[r] = gA[r * 4 + 3];
            for (int c = 0; c < 3; c++) go[r * 3 + c] = gA[r * 4 + c] - w[r] * Jl[c];
        }
        for (int c = 0; c < 3; c++)
            gJ[c] = -(G[0 * 4 + c] * gA[0 * 4 + 3] + G[1 * 4 + c] * gA[1 * 4 + 3] + G[2 * 4 + c] * gA[2 * 4 + 3]);
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++)
                sX[j][r * 3 + c] = go[r * 3 + 0] * G[c * 4 + 0] + go[r * 3 + 1] * G[c * 4 + 1] + go[r * 3 + 2] * G[c * 4 + 2] + w[r] * G[c * 4 + 3];
            sX[j][9 + r] = w[r];
        }
    }
    const float *P = pre.P;
    __syncthreads();
    PSI_PSTOP(21);
    for (int i = j; i < m.n_items * 12; i += nthr) {
        const int it = i / 12, e = i - it * 12;
        const unsigned int d = sItem[it];
        const unsigned char *l = sList + (d & 0xffffu);
        const int cnt = (int)(d >> 16);
        float a = 0.0f;
        for (int q = 0; q < cnt; q++) a += sX[l[q]][e];
        sP[it][e] = a;
    }
    __syncthreads();
    for (int i = j; i < m.J * 12; i += nthr) {
        const int jj = i / 12, e = i - jj * 12;
        float a = 0.0f;
        for (int it = sFirst[jj]; it < sFirst[jj + 1]; it++) a += sP[it][e];
        sS[jj][e] = a;
    }
    __syncthreads();
    PSI_PSTOP(22);
    // local gradients: gR_j = P_R^T gG_j.R, grel_j = P_R^T gG_j.t  (P = parent's G; root: identity)
    float gR[9], grel[3] = {0, 0, 0};
    for (int e = 0; e < 9; e++) gR[e] = 0.0f;
    if (act) {
        float gG[12];
        for (int r = 0; r < 3; r++) {
            const float wr = sS[j][9 + r];
            float T[3];
            for (int k = 0; k < 3; k++) T[k] = sS[j][r * 3 + k] - wr * G[k * 4 + 3];
            for (int c = 0; c < 3; c++) gG[r * 4 + c] = T[0] * G[0 * 4 + c] + T[1] * G[1 * 4 + c] + T[2] * G[2 * 4 + c];
            gG[r * 4 + 3] = wr;
        }
        if (par >= 0) {
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) gR[r * 3 + c] = P[0 * 4 + r] * gG[0 * 4 + c] + P[1 * 4 + r] * gG[1 * 4 + c] + P[2 * 4 + r] * gG[2 * 4 + c];
                grel[r] = P[0 * 4 + r] * gG[0 * 4 + 3] + P[1 * 4 + r] * gG[1 * 4 + 3] + P[2 * 4 + r] * gG[2 * 4 + 3];
            }
        } else {
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) gR[r * 3 + c] = gG[r * 4 + c];
                grel[r] = gG[r * 4 + 3];
            }
        }
        for (int c = 0; c < 3; c++) sgrel[j][c] = grel[c];
    }
    __syncthreads();
    if (act) {
        // rel_j = J_j - J_parent: own +grel, minus the children's
        for (int c = 0; c < 3; c++) gJ[c] += grel[c];
        for (int ci = cp0; ci < cp1; ci++)
            for (int c = 0; c < 3; c++) gJ[c] -= sgrel[sChild[ci]][c];
        for (int c = 0; c < 3; c++) sgJ[j][c] = gJ[c];
    }
    __syncthreads();
    PSI_PSTOP(23);
    // feature gradient (reduced over n-slices): betas part and pose-feature part
    if (g_betas_b) {
        // g_betas[l] = g_feat[l] + sum_q gJ[q] J_s[q][l]: the (joint, axis) range is cut into nthr/NB parts summed through LDS,
        // so a thread has only a handful of independent loads (they were 165 dependent rounds for NB threads before)
        float (&sgb)[32][32] = sh.sgb;
        if (js_pre) {
            float a = 0.0f;
#pragma unroll
            for (int i = 0; i < JSP; i++) {
                const int q = bpart * per + i;
                if (i < per && q < nq) a += (&sgJ[0][0])[q] * js[i];
            }
            sgb[bpart][bl] = a;
        }
        for (int i = j; i < npart * m.NB && par_ok && per > JSP; i += nthr) {
            const int part = i / m.NB, l = i - part * m.NB;
            const int q1 = min(nq, (part + 1) * per);
            float a = 0.0f;
#pragma unroll 8
            for (int q = part * per; q < q1; q++) a += (&sgJ[0][0])[q] * m.J_s[(size_t)q * m.NB + l];
            sgb[part][l] = a;
        }
        __syncthreads();
        for (int l = j; l < m.NB; l += nthr) {
            float a = gfeat_b[l];
            if (par_ok) {
                for (int part = 0; part < npart; part++) a += sgb[part][l];
            } else {
                for (int q = 0; q < nq; q++) a += (&sgJ[0][0])[q] * m.J_s[(size_t)q * m.NB + l];
            }
            g_betas_b[l] = a;
        }
    }
    PSI_PSTOP(24);
    if (act && (g_pose_b || g_rot_b)) {
        if (j >= 1)
            for (int e = 0; e < 9; e++) gR[e] += gf9[e];
        if (g_rot_b)
            for (int e = 0; e < 9; e++) g_rot_b[j * 9 + e] = gR[e];
    }
    if (act && g_pose_b) {
        // Rodrigues backward (lbs.py:177-191)
        float x = aa[0] + 1e-8f, y = aa[1] + 1e-8f, z = aa[2] + 1e-8f;
        float th = sqrtf(x * x + y * y + z * z);
        float d[3] = {aa[0] / th, aa[1] / th, aa[2] / th};
        float s = sinf(th), c = cosf(th), c1 = 1.0f - c;
        float K[9] = {0, -d[2], d[1], d[2], 0, -d[0], -d[1], d[0], 0};
        float KK[9];
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) KK[r * 3 + q] = K[r * 3 + 0] * K[0 * 3 + q] + K[r * 3 + 1] * K[1 * 3 + q] + K[r * 3 + 2] * K[2 * 3 + q];
        float gs = 0, gc1 = 0;
        for (int e = 0; e < 9; e++) { gs += gR[e] * K[e]; gc1 += gR[e] * KK[e]; }
        float gK[9];
        for (int r = 0; r < 3; r++)
            for (int q = 0; q < 3; q++) {
                float a = 0;
                for (int k = 0; k < 3; k++) a += gR[r * 3 + k] * K[q * 3 + k] + K[k * 3 + r] * gR[k * 3 + q];   // gR K^T + K^T gR
                gK[r * 3 + q] = s * gR[r * 3 + q] + c1 * a;
            }
        float gd[3] = {gK[7] - gK[5], gK[2] - gK[6], gK[3] - gK[1]};
        float gth = gs * c + gc1 * s;                      // d sin = cos, d(1-cos) = sin
        gth -= (gd[0] * aa[0] + gd[1] * aa[1] + gd[2] * aa[2]) / (th * th);
        float ga[3] = {gd[0] / th + gth * x / th, gd[1] / th + gth * y / th, gd[2] / th + gth * z / th};
        for (int q = 0; q < 3; q++) g_pose_b[j * 3 + q] = ga[q];
    }
}

__device__ __forceinline__ void psi_pose_bwd_body(const LbsDev &m, const float *pose_b,
                                                  const float *__restrict__ Rs, const float *__restrict__ Jls,
                                                  const float *__restrict__ Gs, const float *gA_b, const float *gfeat_b, int b,
                                                  float *g_betas_b, float *g_pose_b, float *g_rot_b)
{
    (void)Rs;
    __shared__ PsiPoseBwdShared sh;
    const PsiPoseBwdPre pre = psi_pose_bwd_issue(m, pose_b, Jls, Gs, b, g_betas_b != nullptr, g_pose_b != nullptr, sh);
    psi_pose_bwd_finish(m, pre, sh, gA_b, gfeat_b, g_betas_b, g_pose_b, g_rot_b);
}


// Sums of the split-contraction partials of the LBS backward (skin_bwd_A's vertex slices, blend_bwd's column slices, skin_bwd_v's
// translation partials): used by reduce_partials_kernel (lbs.hip) and by the fitting engine's fused tail + head kernel (fit.hip), which
// must add the slices in the same order.
// PSI_RSPL threads share one output: thread s of the group adds slices s, s + PSI_RSPL, ... (all requested before the first add) and the group is
// combined in lane order with shuffles — four times as many workgroups pulling the fresh partials (they were just written by other XCDs'
// workgroups: a CU gets only ~15 GB/s of such data, so the kernel is bound by how many CUs pull at once, not by arithmetic).
constexpr int PSI_RSPL = 4;
// (two halves, so that a caller with several outputs per thread can request all of their slices before the first add)
template <int MAXS>
__device__ __forceinline__ void psi_slices_load(const float *__restrict__ p, size_t stride, int n, int s0, float (&v)[MAXS])
{
#pragma unroll
    for (int k = 0; k < MAXS; k++) {
        const int sl = s0 + k * PSI_RSPL;
        v[k] = sl < n ? p[(size_t)sl * stride] : 0.0f;
    }
}
template <int MAXS>
__device__ __forceinline__ float psi_slices_sum(const float (&v)[MAXS], const float *__restrict__ p, size_t stride, int n, int s0)
{
    float a0 = 0, a1 = 0;
#pragma unroll
    for (int k = 0; k < MAXS; k += 2) { a0 += v[k]; a1 += v[k + 1]; }
    float r = a0 + a1;
    for (int sl = s0 + MAXS * PSI_RSPL; sl < n; sl += PSI_RSPL) r += p[(size_t)sl * stride];
    // lanes s0 = 0..3 of the group -> ((r0 + r1) + (r2 + r3)), the same value in all four lanes
    r += __shfl_xor(r, 1, 64);
    r += __shfl_xor(r, 2, 64);
    return r;
}
template <int MAXS>
__device__ __forceinline__ float psi_sum_slices_split(const float *__restrict__ p, size_t stride, int n, int s0)
{
    float v[MAXS];
    psi_slices_load<MAXS>(p, stride, n, s0, v);
    return psi_slices_sum<MAXS>(v, p, stride, n, s0);
}


// ------------------------------------------------------------------------------------------------
// Skinning kernels (templates: the fused fitting engine instantiates them with its own per-vertex hooks, so the SDF
// lookup rides on skin_fwd and the loss-gradient assembly on skin_bwd_v instead of being separate passes over [B,V]).
// Grid (Vpad / 256, B): one workgroup = 256 vertices of ONE body (measured at B = 32: 1 body per workgroup 12.4 us,
// 2 -> 13.4, 4 -> 21.7, 8 -> 60: more, smaller workgroups win on this latency-bound kernel).
// ------------------------------------------------------------------------------------------------
typedef float psi_f2 __attribute__((ext_vector_type(2)));
constexpr int PSI_SKIN_BLK = 256;

// Sum over the 64 lanes of a wave, returned in every lane.  Data-parallel primitives instead of shuffles: a `__shfl_down` is a
// ds_bpermute plus its lane-address arithmetic (4-5 vector instructions per step and an LDS round trip); a DPP-modified add is ONE
// instruction.  Fixed order (xor 1, xor 2, rotate 4, rotate 8 inside each row of 16, then the rows), so results repeat bit for bit.
template <int CTRL, int ROWS>
__device__ __forceinline__ float psi_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROWS, 0xf, false));
}
__device__ __forceinline__ float psi_wave_sum(float x)
{
    x += psi_dpp<0xB1, 0xf>(x);      // quad_perm [1,0,3,2]
    x += psi_dpp<0x4E, 0xf>(x);      // quad_perm [2,3,0,1]
    x += psi_dpp<0x124, 0xf>(x);     // row_ror 4
    x += psi_dpp<0x128, 0xf>(x);     // row_ror 8: every lane holds its row's total
    x += psi_dpp<0x142, 0xa>(x);     // row_bcast 15 into rows 1 and 3
    x += psi_dpp<0x143, 0xc>(x);     // row_bcast 31 into rows 2 and 3: lane 63 holds the wave's total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}

// Addressing of the per-vertex streams of the skinning kernels: address = a wave-uniform base (row / body offset added in SCALAR
// arithmetic) + ONE 32-bit unsigned lane offset, which the compiler turns into the `global_load v, v_off, s[base]` form.  A
// `p[(size_t)row * stride + v]` access costs a 64-bit vector add per load instead (55 of them in the dense blend, plus 64-bit
// multiplies for the [B, V, 3] rows).  (The raw-buffer builtins would do the same with a scalar offset operand, but the b64 / b96
// forms are miscompiled by this toolchain's demanded-elements narrowing — element 0 is returned for every element — so only
// the 32-bit form is used anywhere in this library.)
struct psi_p3 { float x, y, z; };                               // 12-byte record: one global_load / global_store_dwordx3
template <class T>
__device__ __forceinline__ T psi_ld(const void *uniform_base, unsigned lane_off)
{
    return *(const T *)((const char *)uniform_base + lane_off);
}
template <class T>
__device__ __forceinline__ void psi_st(void *uniform_base, unsigned lane_off, const T &val)
{
    *(T *)((char *)uniform_base + lane_off) = val;
}

// Blend a body's joint transforms with this lane's skinning weights: T = sum_j w_j A_j (3x4 as six float pairs), packed
// accumulation (v_pk_fma_f32), for NB bodies at once: the lane's weights are loaded ONCE and every body's transforms are blended
// with them (NB = 2 halves the weight traffic — 2.7 MB per body from L2, what bounds the dense kernel at large batches — and the
// number of skinning workgroups).
//
// The skinning kernels are latency-bound at the BASELINE batch, and what they read at the start — the transforms to stage, the
// lane's weight row, its vertex / gradient operands — are independent of each other: PsiBlend splits the blend into issue() (all
// the loads, to be called together with the caller's own first loads), commit() (LDS writes + barrier) and blend().  Requested
// one after the other, as a staged-then-blend function does, these were three to six dependent L2 round trips per workgroup.
//
// Dense rows: the transforms are the same for every lane of the workgroup — read through the SCALAR cache (48 bytes per joint and
// wave) instead of LDS broadcasts (3 KB per joint and wave: at 55 joints the LDS return path, 128 B/clk per CU, bounded the
// skinning kernels: 26 k cycles per CU = 11 us for the 20 resident waves).  Weights come from the wave-tiled copy WTt through a
// buffer descriptor: tile offset (scalar) + lane * 4, the joint as an instruction IMMEDIATE (a `WT[(size_t)j * Vpad + v]` access was a
// 64-bit vector add per joint, 55 of the kernel's 805 vector instructions per wave in round 3).  A weight register is re-requested
// for the next group right after its joint has been accumulated (the last group re-requests its own rows: first-level cache hits).
// Two forms of the joint loop (rocprofv3, profiles/r04_ab_blend_loop.txt):
//   PsiBlendCompact    one group of 11 joints of code in a loop, no scheduling fences: the scheduler hoists a group's scalar loads
//                      itself (a few scalar registers spill).  The smallest instruction footprint and the fewest scalar round trips in
//                      a row — what the latency-bound BASELINE batch wants (skin_bwd_v 14.0 us; with a fence every 4 joints 15.2, every
//                      2 joints 16.3; round 3's loop 14.7).
//   PsiBlendPipelined  groups of 12 joints (J padded: the extra joints have zero weights and read the finite rows behind the body's
//                      transforms) in sets of PS joints whose transforms sit in TWO alternating sets of scalar registers: the scalar
//                      loads of set i + 1 are issued BEFORE set i is accumulated, so a wave hides them behind its own packed FMAs
//                      instead of draining the scalar-load counter (it cannot count out-of-order returns: every wait is "all of
//                      them") in front of every set.  SMPL-X's five groups are straight-line code: at a loop header the compiler waits
//                      for EVERY outstanding weight load, i.e. for the two requested a few cycles earlier; in straight-line code its
//                      waits are exact.  What the throughput-bound large batches want (skin_fwd_sdf at B = 512, dense rows: 136 us
//                      against 150 for the same code as a loop and 158 for the compact form).
#ifndef PSI_DENSE_UNROLL
#define PSI_DENSE_UNROLL 11
#endif
enum PsiBlendForm { PsiBlendCompact = 0, PsiBlendPipelined = 1 };
// LDS home of a workgroup's staged joint transforms ([body][joint][6 float pairs]): ONE array per kernel for every user — the skinning
// workgroups (PsiBlendN::commit, compressed rows) and, in the fused fitting engine's shared launch, the search workgroups that skin their own
// contact vertex (fit.hip: ContactSkinSrc) are never the same workgroup, and two separate 3 KB arrays put that launch over an occupancy step
// of its LDS budget (6 -> 5 workgroups per CU: measured in round 6 as +4 us on the skinning workgroups).
template <int NB>
__device__ __forceinline__ psi_f2 (*psi_transform_stage())[PSI_JP][6]
{
    __shared__ psi_f2 sA[NB][PSI_JP][6];
    return sA;
}
template <int NB>
struct PsiBlendN {
    psi_f2 st[NB][2];             // this thread's share of the bodies' transforms (J * 6 float pairs over 256 threads, J <= 85)
    float wk[PSI_WNZ];            // compressed weight row of the lane's vertex (when the model has one)
    unsigned jk[PSI_WNZ / 4];     // ... and its joint indices, a byte each
    const float *As_b[NB];        // the bodies' transforms in global memory (set by issue)
    typedef psi_f2 (*Staged)[PSI_JP][6];
    __device__ __forceinline__ void issue(const LbsDev &m, const float *__restrict__ As, const int (&b)[NB], int v)
    {
#pragma unroll
        for (int n = 0; n < NB; n++) As_b[n] = As + (size_t)b[n] * m.J * 12;
        if (m.Wc) {
#pragma unroll
            for (int n = 0; n < NB; n++)
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int idx = threadIdx.x + q * PSI_SKIN_BLK;
                    st[n][q] = idx < m.J * 6 ? *(const psi_f2 *)(As_b[n] + idx * 2) : (psi_f2){0.0f, 0.0f};
                }
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)m.Wc, 0, PSI_WNZ * m.Vpad * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rj = __builtin_amdgcn_make_buffer_rsrc((void *)m.Wj, 0, PSI_WNZ / 4 * m.Vpad * 4, 0x00020000);
#pragma unroll
            for (int k = 0; k < PSI_WNZ; k++)
                wk[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, (unsigned)v * 4u, (unsigned)(k * m.Vpad) * 4u, 0));
#pragma unroll
            for (int k = 0; k < PSI_WNZ / 4; k++) jk[k] = __builtin_amdgcn_raw_buffer_load_b32(rj, (unsigned)v * 4u, (unsigned)(k * m.Vpad) * 4u, 0);
        }
    }
    __device__ __forceinline__ Staged commit(const LbsDev &m)
    {
        const Staged sA = psi_transform_stage<NB>();
        if (!m.Wc) return sA;                                   // dense rows read the transforms through the scalar cache (blend)
#pragma unroll
        for (int n = 0; n < NB; n++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int idx = threadIdx.x + q * PSI_SKIN_BLK;
                if (idx < m.J * 6) (&sA[n][0][0])[idx] = st[n][q];     // [joint][6] pairs = the rows as they lie in memory
            }
        __syncthreads();
        return sA;
    }
    template <PsiBlendForm FORM = PsiBlendCompact>
    __device__ __forceinline__ void blend(const LbsDev &m, Staged sA, int v, psi_f2 (&T2)[NB][6]) const
    {
#pragma unroll
        for (int n = 0; n < NB; n++)
#pragma unroll
            for (int e = 0; e < 6; e++) T2[n][e] = (psi_f2){0.0f, 0.0f};
        if (m.Wc) {
            // compressed rows (real SMPL-X weight rows have a handful of non-zeros): the same sum with the exact zeros skipped,
            // in ascending joint order — bit-identical to the dense loop, 8 instead of 55 terms
#pragma unroll
            for (int k = 0; k < PSI_WNZ; k++) {
                // a slot that is zero in every lane of the wave ends the row for all of them (rows are filled front to back, and adding
                // w = 0 terms changes nothing: T starts at +0): 5 of 8 slots at SMPL-X's 4-5 non-zeros — the blend is LDS-bandwidth-bound
                if (__builtin_amdgcn_ballot_w64(wk[k] != 0.0f) == 0) break;
                psi_f2 w2 = {wk[k], wk[k]};
                const int jj = (jk[k >> 2] >> (8 * (k & 3))) & 0xff;
#pragma unroll
                for (int n = 0; n < NB; n++)
#pragma unroll
                    for (int e = 0; e < 6; e++) T2[n][e] = __builtin_elementwise_fma(w2, sA[n][jj][e], T2[n][e]);
            }
            return;
        }
        const psi_f2 *Ab[NB];
#pragma unroll
        for (int n = 0; n < NB; n++) Ab[n] = (const psi_f2 *)As_b[n];
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void *)m.WTt, 0, PSI_JP * m.Vpad * 4, 0x00020000);
        const unsigned v4 = (unsigned)(v & 63) * 4u;
        const unsigned tile_off = (unsigned)__builtin_amdgcn_readfirstlane(v >> 6) * (unsigned)(PSI_JP * 64 * 4);   // a wave = 64 consecutive vertices
        auto wload = [&](int k, unsigned ro) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rw, v4 + k * 256, ro, 0)); };
        unsigned ro = tile_off;                                      // scalar offset of the group being requested
        // (the pipelined form walks whole groups of 12 joints: beyond 60 joints its sixth group would leave the 64-joint weight tile)
        if (FORM == PsiBlendCompact || m.J > 60) {
            constexpr int GJ = PSI_DENSE_UNROLL;
            static_assert(GJ * 256 <= 4096, "joint offsets of a group must fit the load's 12-bit immediate");
            float w[GJ];
#pragma unroll
            for (int k = 0; k < GJ; k++) w[k] = wload(k, ro);
            auto fma6 = [&](float wj, int j) {
                psi_f2 w2 = {wj, wj};
#pragma unroll
                for (int n = 0; n < NB; n++)
#pragma unroll
                    for (int e = 0; e < 6; e++) T2[n][e] = __builtin_elementwise_fma(w2, Ab[n][j * 6 + e], T2[n][e]);
            };
            int j0 = 0;
#pragma nounroll
            for (; j0 + GJ <= m.J; j0 += GJ) {
                ro += j0 + GJ < m.J ? GJ * 256 : 0;
#pragma unroll
                for (int k = 0; k < GJ; k++) {
                    fma6(w[k], j0 + k);
                    w[k] = wload(k, ro);                             // joint j0 + GJ + k, for the next trip
                }
            }
#pragma unroll
            for (int k = 0; k < GJ; k++)                             // J % GJ joints are left (none for J = 55)
                if (j0 + k < m.J) fma6(w[k], j0 + k);
        } else {
            constexpr int G2 = 12, PS = NB == 1 ? 2 : 1;
            static_assert(G2 % (2 * PS) == 0, "a group is a whole number of set pairs");
            float w[G2];
#pragma unroll
            for (int k = 0; k < G2; k++) w[k] = wload(k, ro);
            psi_f2 P0[NB][PS][6], P1[NB][PS][6];
            auto aload = [&](psi_f2 (&P)[NB][PS][6], int j) {
#pragma unroll
                for (int n = 0; n < NB; n++)
#pragma unroll
                    for (int q = 0; q < PS; q++)
#pragma unroll
                        for (int e = 0; e < 6; e++) P[n][q][e] = Ab[n][(j + q) * 6 + e];
            };
            auto fmas = [&](const psi_f2 (&P)[NB][PS][6], const float *wk) {
#pragma unroll
                for (int q = 0; q < PS; q++) {
                    const psi_f2 w2 = {wk[q], wk[q]};
#pragma unroll
                    for (int n = 0; n < NB; n++)
#pragma unroll
                        for (int e = 0; e < 6; e++) T2[n][e] = __builtin_elementwise_fma(w2, P[n][q][e], T2[n][e]);
                }
            };
            aload(P0, 0);
            const int ngroups = (m.J + G2 - 1) / G2;
            auto group = [&](int g) {
                const int j0 = g * G2;
                ro += g + 1 < ngroups ? G2 * 256 : 0;
#pragma unroll
                for (int k = 0; k < G2; k += 2 * PS) {
                    aload(P1, j0 + k + PS);
                    __builtin_amdgcn_sched_barrier(0);
                    fmas(P0, w + k);
#pragma unroll
                    for (int q = 0; q < PS; q++) w[k + q] = wload(k + q, ro);
                    __builtin_amdgcn_sched_barrier(0);
                    aload(P0, j0 + k + 2 * PS);                      // (the last one of the last group reads rows that are never used)
                    __builtin_amdgcn_sched_barrier(0);
                    fmas(P1, w + k + PS);
#pragma unroll
                    for (int q = 0; q < PS; q++) w[k + PS + q] = wload(k + PS + q, ro);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (ngroups == 5) {                                      // SMPL-X
#pragma unroll
                for (int g = 0; g < 5; g++) group(g);
            } else {
#pragma nounroll
                for (int g = 0; g < ngroups; g++) group(g);
            }
        }
    }
};
typedef PsiBlendN<1> PsiBlend;

// The per-vertex affine maps of the skinning kernels with an EXPLICIT operation order (fma chains), so that every instantiation
// — one body or several per workgroup, dense or compressed rows — rounds identically (left to the compiler's contraction the
// kernels differed in the last bit).
__device__ __forceinline__ float psi_dot3p(float a, float b, float c, float d, float x, float y, float z)
{
    return __builtin_fmaf(c, z, __builtin_fmaf(b, y, a * x)) + d;           // ((a x + b y) + c z) + d
}
__device__ __forceinline__ float psi_dot3(float a, float b, float c, float x, float y, float z)
{
    return __builtin_fmaf(c, z, __builtin_fmaf(b, y, a * x));               // (a x + b y) + c z
}

// Epilogue hook of skin_fwd: store(...) writes vertex v's result (to `verts`, or wherever the epilogue keeps the rows it needs); vertex(n, b, v, ...) sees every lane's final world-space vertex of the workgroup's n-th body b (live = false
// for padding lanes), backward(n, b, v12, T2, C) follows it with the lane's blended transform and the body's camera still in registers (an
// epilogue that knows dL/dvertex at this point does the vertex's skinning backward on the spot: the fused fitting engine, fit.hip),
// finish(n, b, vblock, nvb) runs once per workgroup and body (vertex block vblock of nvb) with all threads present.
struct PsiSkinNoEpilogue {
    __device__ __forceinline__ void backward(int, int, unsigned, const psi_f2 (&)[6], const float *) {}
    __device__ __forceinline__ void store(float *verts, size_t body_off, int, int, unsigned v12, float x, float y, float z) const
    {
        if (verts) psi_st(verts + body_off, v12, psi_p3{x, y, z});
    }
    __device__ __forceinline__ void vertex(int, int, int, float, float, float, bool) {}
    __device__ __forceinline__ void finish(int, int, int, int) {}
};

// verts = cam_ext * (sum_j W_j A_j [v_posed;1] + transl)      (lbs.py:108-116, cvae.py:141-149)
// (a device function so that the fused fitting engine can run it inside a launch it shares with the NN search: fit.hip)
// NB bodies per workgroup (b0, b0 + 1, ...; a body index beyond B - 1 repeats the last body and stores nothing): every lane blends
// its vertex for all of them with ONE pass over its weights, then runs transform + epilogue body by body.
template <int NB, PsiBlendForm FORM, class Epi>
__device__ __forceinline__ void psi_skin_fwd_body(const LbsDev &m, const float *__restrict__ As, const float *__restrict__ v_posed,
                                                  const float *__restrict__ transl, const float *__restrict__ cam_ext, int B,
                                                  float *__restrict__ verts, Epi &epi, int vblock, int b0, int nvb)
{
    const int v = vblock * PSI_SKIN_BLK + threadIdx.x;
    const bool live = v < m.V;
    const unsigned v12 = (unsigned)v * 12u;                     // lane offset of every [.., V, 3] stream of this kernel
    int bs[NB];
#pragma unroll
    for (int n = 0; n < NB; n++) bs[n] = min(b0 + n, B - 1);
    // all first loads in one go: transforms to stage, weight row, posed vertices (one 12-byte load each; rows are padded to Npad >= 3 Vpad)
    PsiBlendN<NB> bl;
    bl.issue(m, As, bs, v);
    psi_p3 pp[NB];
#pragma unroll
    for (int n = 0; n < NB; n++) pp[n] = psi_ld<psi_p3>(v_posed + (size_t)bs[n] * m.Npad, v12);
    psi_f2 T2[NB][6];
    bl.template blend<FORM>(m, bl.commit(m), v, T2);
#pragma unroll
    for (int n = 0; n < NB; n++) {
        const int b = bs[n];
        const bool on = n == 0 || b0 + n < B;                    // wave-uniform
        const float px = pp[n].x, py = pp[n].y, pz = pp[n].z;
        float x = psi_dot3p(T2[n][0].x, T2[n][0].y, T2[n][1].x, T2[n][1].y, px, py, pz);
        float y = psi_dot3p(T2[n][2].x, T2[n][2].y, T2[n][3].x, T2[n][3].y, px, py, pz);
        float z = psi_dot3p(T2[n][4].x, T2[n][4].y, T2[n][5].x, T2[n][5].y, px, py, pz);
        if (transl) {
            x += transl[(size_t)b * 3 + 0];
            y += transl[(size_t)b * 3 + 1];
            z += transl[(size_t)b * 3 + 2];
        }
        if (cam_ext) {   // cvae.py:141-149: [v,1] @ cam_ext^T, drop w
            const float *C = cam_ext + (size_t)b * 16;
            float X = psi_dot3p(C[0], C[1], C[2], C[3], x, y, z);
            float Y = psi_dot3p(C[4], C[5], C[6], C[7], x, y, z);
            float Z = psi_dot3p(C[8], C[9], C[10], C[11], x, y, z);
            x = X; y = Y; z = Z;
        }
        epi.vertex(n, b, v, x, y, z, live && on);
        if (on) epi.backward(n, b, v12, T2[n], cam_ext ? cam_ext + (size_t)b * 16 : nullptr);
        // stored AFTER the epilogue's lookups: a wait for a load also waits for the wave's earlier stores (one counter on gfx950).
        // verts == nullptr: the caller has no use for the vertices themselves (the fused fitting iteration: its search lanes skin their own
        // contact vertices and the backward reads the epilogue's outputs — 64 MB of stores per launch at B = 512 for nobody)
        if (live && on) epi.store(verts, (size_t)b * m.V * 3, b, v, v12, x, y, z);
    }
#pragma unroll
    for (int n = 0; n < NB; n++) {
        if (n > 0) __syncthreads();                              // the previous body's reduction has been read
        if (n == 0 || b0 + n < B) epi.finish(n, bs[n], vblock, nvb);
    }
}

// (its own launch = the throughput-bound large batches: the pipelined form of the dense blend)
template <class Epi, int NB = 1>
__global__ __launch_bounds__(PSI_SKIN_BLK, 6) void psi_skin_fwd_kernel(LbsDev m, const float *__restrict__ As, const float *__restrict__ v_posed,
                                                                     const float *__restrict__ transl, const float *__restrict__ cam_ext,
                                                                     int B, float *__restrict__ verts, Epi epi)
{
    psi_skin_fwd_body<NB, PsiBlendPipelined>(m, As, v_posed, transl, cam_ext, B, verts, epi, (int)blockIdx.x, (int)blockIdx.y * NB, (int)gridDim.x);
}

// Gradient source of skin_bwd_v: where dL/dverts[b][v] comes from.
struct PsiGradFromMemory {
    const float *g_verts;     // [B,V,3]
    int V;
    __device__ __forceinline__ void prepare(int, int) {}
    __device__ __forceinline__ void load(int b, int v, float &gx, float &gy, float &gz) const
    {
        const float *g = g_verts + ((size_t)b * V + v) * 3;
        gx = g[0]; gy = g[1]; gz = g[2];
    }
    // split form for kernels that request their operands ahead of prepare(): issue() = the loads, take() = their use
    struct Pre { float g[3]; };
    __device__ __forceinline__ Pre issue(int b, int v, bool live) const
    {
        Pre p = {{0.0f, 0.0f, 0.0f}};
        if (live) load(b, v, p.g[0], p.g[1], p.g[2]);
        return p;
    }
    __device__ __forceinline__ void issue_late(Pre &, int) const {}
    __device__ __forceinline__ void take(const Pre &p, int, int, float &gx, float &gy, float &gz) const { gx = p.g[0]; gy = p.g[1]; gz = p.g[2]; }
};

// per-vertex part of the skinning backward: g_local = R_c^T g_verts;  g_vposed = T_R^T g_local;  partial g_transl
template <class Src, PsiBlendForm FORM = PsiBlendCompact>
__global__ __launch_bounds__(PSI_SKIN_BLK, 6) void psi_skin_bwd_v_kernel(LbsDev m, const float *__restrict__ As, Src src,
                                                                       const float *__restrict__ cam_ext, int B, float *__restrict__ gl,
                                                                       float *__restrict__ g_vp, float *__restrict__ gt_part)
{
    const int v = blockIdx.x * PSI_SKIN_BLK + threadIdx.x;
    const int b = blockIdx.y;
    PSI_TRACE(30, 30);                                       // (dev: workgroup timeline, tools/timeline.py)
    // all first loads in one go: transforms to stage, weight row, the gradient source's operands and its statistics inputs
    PSI_SSTOP(11);
    typename Src::Pre pre = src.issue(b, v, v < m.V);
    PsiBlend bl;
    const int bs1[1] = {b};
    bl.issue(m, As, bs1, v);
    src.issue_late(pre, b);
    PSI_SSTOP(12);
    // the blend first (its weight loads run while the statistics inputs requested above are still in flight), then the statistics
    psi_f2 T2n[1][6];
    bl.template blend<FORM>(m, bl.commit(m), v, T2n);
    const psi_f2 (&T2)[6] = T2n[0];
    PSI_SSTOP(13);
    src.prepare(b, 1);
    PSI_SSTOP(14);
    __shared__ float sh[PSI_SKIN_BLK / 64][3];
    float lx = 0, ly = 0, lz = 0;
    if (v < m.V) {
        float gx, gy, gz;
        src.take(pre, b, v, gx, gy, gz);
        if (cam_ext) {   // g_local = R_c^T g
            const float *C = cam_ext + (size_t)b * 16;
            lx = psi_dot3(C[0], C[4], C[8], gx, gy, gz);
            ly = psi_dot3(C[1], C[5], C[9], gx, gy, gz);
            lz = psi_dot3(C[2], C[6], C[10], gx, gy, gz);
        } else {
            lx = gx; ly = gy; lz = gz;
        }
    }
    {
        // wave-uniform row base + the 32-bit lane offset v * 12: two 12-byte stores
        psi_st(gl + (size_t)b * m.Npad, (unsigned)v * 12u, psi_p3{lx, ly, lz});
        psi_st(g_vp + (size_t)b * m.Npad, (unsigned)v * 12u,      // T_R^T g_local (rotation part of T, row-major 3x3)
               psi_p3{psi_dot3(T2[0].x, T2[2].x, T2[4].x, lx, ly, lz), psi_dot3(T2[0].y, T2[2].y, T2[4].y, lx, ly, lz),
                      psi_dot3(T2[1].x, T2[3].x, T2[5].x, lx, ly, lz)});
    }
    float sx = psi_wave_sum(lx), sy = psi_wave_sum(ly), sz = psi_wave_sum(lz);
    if ((threadIdx.x & 63) == 0) {
        sh[threadIdx.x >> 6][0] = sx;
        sh[threadIdx.x >> 6][1] = sy;
        sh[threadIdx.x >> 6][2] = sz;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        float s = 0;
        for (int ww = 0; ww < PSI_SKIN_BLK / 64; ww++) s += sh[ww][threadIdx.x];
        gt_part[((size_t)blockIdx.x * B + b) * 4 + threadIdx.x] = s;
    }
}


// ------------------------------------------------------------------------------------------------
// Multi-body skinning backward for large batches (B >= PSI_SKIN_MB_MIN_B).  One body per workgroup re-reads the vertex's skinning
// weights from L2 for every body — 2.3 MB x B, 1.2 GB at B = 512.  Here a lane loads its vertex's weights ONCE into registers and
// walks PSI_SKIN_MB bodies: per body only the 2.6 KB of joint transforms are staged (double-buffered in LDS, one barrier per
// body): 349 -> 281 us at B = 512.  At B = 32 this shape is slower (fewer, longer workgroups on a latency-bound launch: measured
// 1.8x at 4 bodies), so the single-body kernel stays the default for small batches.  The FORWARD kernel keeps one body per
// workgroup at every batch size: with the SDF lookup fused in it is bound by the eight-corner gathers (272 us multi-body vs 222 us
// single-body at B = 512: the lower occupancy of the register-resident weights costs more gather latency than the weights save).
// ------------------------------------------------------------------------------------------------
constexpr int PSI_SKIN_MB = 8;            // bodies per workgroup
constexpr int PSI_SKIN_MB_MIN_B = 128;    // batch size from which the multi-body kernels are used

template <bool COMPRESSED>
struct PsiLaneWeights;
template <>
struct PsiLaneWeights<false> {
    psi_f2 w2[PSI_JP / 2];                // dense row of this lane's vertex, two joints per 64-bit register pair (zero beyond J)
    __device__ __forceinline__ void load(const LbsDev &m, int v)
    {
        // WT is [PSI_JP][Vpad] with zero rows beyond J.  Buffer loads: ONE lane offset register + a scalar row offset per joint.
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)m.WT, 0, PSI_JP * m.Vpad * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < PSI_JP / 2; j++) {
            w2[j].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, v * 4, (2 * j) * m.Vpad * 4, 0));
            w2[j].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, v * 4, (2 * j + 1) * m.Vpad * 4, 0));
        }
    }
    // same sum, same order as psi_blend_transforms (ascending joints, packed fma): bit-identical results.  The packed fma takes
    // the weight of BOTH result lanes from one half of the weight pair (op_sel / op_sel_hi on src0), so a weight costs one
    // register — the compiler's own form of `{w,w} * A + T` materialises a 2-register broadcast per weight (128 VGPRs for 64
    // joints, which spilled).
    __device__ __forceinline__ void blend(const LbsDev &m, const psi_f2 (*sA)[6], psi_f2 (&T2)[6]) const
    {
#pragma unroll
        for (int e = 0; e < 6; e++) T2[e] = (psi_f2){0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < PSI_JP / 8; c++) {
            if (c * 8 < m.J) {                    // whole groups of 8 joints (padding joints: weight 0, transform 0 -> exact zeros)
#pragma unroll
                for (int jj = c * 4; jj < c * 4 + 4; jj++) {
#pragma unroll
                    for (int e = 0; e < 6; e++) {
                        const psi_f2 a = sA[2 * jj][e];
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(T2[e]) : "v"(w2[jj]), "v"(a));
                    }
#pragma unroll
                    for (int e = 0; e < 6; e++) {
                        const psi_f2 a = sA[2 * jj + 1][e];
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(T2[e]) : "v"(w2[jj]), "v"(a));
                    }
                }
            }
        }
    }
};
template <>
struct PsiLaneWeights<true> {
    float wk[PSI_WNZ];                    // compressed row: the k-th non-zero weight and its joint (a byte each)
    unsigned jk[PSI_WNZ / 4];
    __device__ __forceinline__ void load(const LbsDev &m, int v)
    {
#pragma unroll
        for (int k = 0; k < PSI_WNZ; k++) wk[k] = m.Wc[(size_t)k * m.Vpad + v];
#pragma unroll
        for (int k = 0; k < PSI_WNZ / 4; k++) jk[k] = m.Wj[(size_t)k * m.Vpad + v];
    }
    __device__ __forceinline__ void blend(const LbsDev &, const psi_f2 (*sA)[6], psi_f2 (&T2)[6]) const
    {
#pragma unroll
        for (int e = 0; e < 6; e++) T2[e] = (psi_f2){0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < PSI_WNZ; k++) {
            if (__builtin_amdgcn_ballot_w64(wk[k] != 0.0f) == 0) break;      // (see PsiBlend::blend)
            psi_f2 w2 = {wk[k], wk[k]};
            const int jj = (jk[k >> 2] >> (8 * (k & 3))) & 0xff;
#pragma unroll
            for (int e = 0; e < 6; e++) T2[e] = __builtin_elementwise_fma(w2, sA[jj][e], T2[e]);
        }
    }
};

__device__ __forceinline__ void psi_stage_transforms(const LbsDev &m, const float *__restrict__ As, int b, psi_f2 (*sA)[6])
{
    for (int idx = threadIdx.x; idx < PSI_JP * 6; idx += PSI_SKIN_BLK)
        sA[idx / 6][idx % 6] = idx < m.J * 6 ? *(const psi_f2 *)(As + ((size_t)b * m.J) * 12 + idx * 2) : (psi_f2){0.0f, 0.0f};
}

template <class Src, bool COMPRESSED>
__global__ __launch_bounds__(PSI_SKIN_BLK, 4) void psi_skin_bwd_v_mb_kernel(LbsDev m, const float *__restrict__ As, Src src,
                                                                          const float *__restrict__ cam_ext, int B, float *__restrict__ gl,
                                                                          float *__restrict__ g_vp, float *__restrict__ gt_part)
{
    const int v = blockIdx.x * PSI_SKIN_BLK + threadIdx.x;
    const int b0 = blockIdx.y * PSI_SKIN_MB;
    src.prepare(b0, min(PSI_SKIN_MB, B - b0));
    __shared__ psi_f2 sA[2][PSI_JP][6];
    __shared__ float sh[2][PSI_SKIN_BLK / 64][3];
    PsiLaneWeights<COMPRESSED> lw;
    lw.load(m, v);
    const int nb = min(PSI_SKIN_MB, B - b0);
#pragma nounroll
    for (int bb = 0; bb < nb; bb++) {
        const int b = b0 + bb;
        psi_stage_transforms(m, As, b, sA[bb & 1]);
        float lx = 0, ly = 0, lz = 0;
        if (v < m.V) {
            float gx, gy, gz;
            src.load(b, v, gx, gy, gz);
            if (cam_ext) {
                const float *C = cam_ext + (size_t)b * 16;
                lx = psi_dot3(C[0], C[4], C[8], gx, gy, gz);
                ly = psi_dot3(C[1], C[5], C[9], gx, gy, gz);
                lz = psi_dot3(C[2], C[6], C[10], gx, gy, gz);
            } else {
                lx = gx; ly = gy; lz = gz;
            }
        }
        float sx = psi_wave_sum(lx), sy = psi_wave_sum(ly), sz = psi_wave_sum(lz);
        if ((threadIdx.x & 63) == 0) {
            sh[bb & 1][threadIdx.x >> 6][0] = sx;
            sh[bb & 1][threadIdx.x >> 6][1] = sy;
            sh[bb & 1][threadIdx.x >> 6][2] = sz;
        }
        __syncthreads();
        psi_f2 T2[6];
        lw.blend(m, sA[bb & 1], T2);
        {
            float *o = gl + (size_t)b * m.Npad + (size_t)v * 3;
            o[0] = lx; o[1] = ly; o[2] = lz;
            float *p = g_vp + (size_t)b * m.Npad + (size_t)v * 3;
            p[0] = psi_dot3(T2[0].x, T2[2].x, T2[4].x, lx, ly, lz);
            p[1] = psi_dot3(T2[0].y, T2[2].y, T2[4].y, lx, ly, lz);
            p[2] = psi_dot3(T2[1].x, T2[3].x, T2[5].x, lx, ly, lz);
        }
        if (threadIdx.x < 3) {
            float s = 0;
            for (int ww = 0; ww < PSI_SKIN_BLK / 64; ww++) s += sh[bb & 1][ww][threadIdx.x];
            gt_part[((size_t)blockIdx.x * B + b) * 4 + threadIdx.x] = s;
        }
    }
}


// launch helpers: pick the dense / compressed-row instantiation
template <class Src>
static inline void psi_launch_skin_bwd_v_mb(const LbsDev &m, const float *As, Src src, const float *cam_ext, int B, float *gl, float *g_vp,
                                            float *gt_part, hipStream_t st)
{
    const dim3 grid(m.Vpad / PSI_SKIN_BLK, (B + PSI_SKIN_MB - 1) / PSI_SKIN_MB);
    if (m.Wc)
        hipLaunchKernelGGL((psi_skin_bwd_v_mb_kernel<Src, true>), grid, dim3(PSI_SKIN_BLK), 0, st, m, As, src, cam_ext, B, gl, g_vp, gt_part);
    else
        hipLaunchKernelGGL((psi_skin_bwd_v_mb_kernel<Src, false>), grid, dim3(PSI_SKIN_BLK), 0, st, m, As, src, cam_ext, B, gl, g_vp, gt_part);
}
