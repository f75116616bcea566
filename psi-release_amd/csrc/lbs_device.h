// Device-side pieces of the SMPL-X LBS operator that more than one translation unit needs: the model descriptor, the
// workspace view, and the per-body pose stages (executed by one workgroup per body), which the fused fitting engine
// (fit.hip) inlines into its head / tail kernels.  Reference arithmetic: human_body_prior/body_model/lbs.py:165-262.
#pragma once
#include <hip/hip_runtime.h>

typedef float psi_f4 __attribute__((ext_vector_type(4)));

constexpr int PSI_JP = 64;          // padded joint count

struct LbsDev {
    int V, J, NB, P, K, Kpad, N, Npad, Vpad, maxlevel;
    const float *dirs, *v_template, *WT, *J_t, *J_s;
    const int *parents, *level, *child_ptr, *child_idx;
};

// Pointers into an LBS workspace (psi_lbs_workspace_floats) for a batch of B bodies
struct PsiLbsView {
    LbsDev m;
    float *feat, *R, *Jl, *G, *A;
    const float *gA_part, *gfeat_part, *gt_part;     // split-contraction partials written by skin_bwd_A / blend_bwd / skin_bwd_v
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

// Body of the pose-forward stage for body b, executed by a whole workgroup (threads >= J idle through the joint part).
// Callers: pose_fwd_kernel (lbs.hip) and the fused fitting head kernel (fit.hip).
__device__ __forceinline__ void psi_pose_fwd_body(const LbsDev &m, const float *__restrict__ betas, const float *__restrict__ pose,
                                                  const float *__restrict__ transl, int B, int b, float *__restrict__ feat,
                                                  float *__restrict__ Rs, float *__restrict__ Jls, float *__restrict__ Gs,
                                                  float *__restrict__ As, float *__restrict__ joints)
{
    const int j = threadIdx.x, nthr = blockDim.x;
    const int Bpad = (B + 15) & ~15;
    __shared__ float sJ[PSI_JP][3];
    __shared__ float sG[PSI_JP][12];
    const bool act = j < m.J;
    // rest joints J = J_t + J_s betas: one (joint, axis) pair per thread, all threads of the workgroup take part
    for (int q = j; q < m.J * 3; q += nthr) {
        float a = m.J_t[q];
        const float *js = m.J_s + (size_t)q * m.NB;
        for (int l = 0; l < m.NB; l++) a += js[l] * betas[(size_t)b * m.NB + l];
        (&sJ[0][0])[q] = a;
        Jls[(size_t)b * m.J * 3 + q] = a;
    }
    float R[9], Jl[3] = {0, 0, 0};
    if (act) {
        psi_rodrigues(pose + ((size_t)b * m.J + j) * 3, R);
        psi_f4 *Ro = (psi_f4 *)(Rs + ((size_t)b * m.J + j) * 12);      // rows padded to 4: three 16-byte stores
        Ro[0] = psi_f4{R[0], R[1], R[2], 0.0f};
        Ro[1] = psi_f4{R[3], R[4], R[5], 0.0f};
        Ro[2] = psi_f4{R[6], R[7], R[8], 0.0f};
        if (j >= 1)
            for (int e = 0; e < 9; e++) {
                int k = m.NB + (j - 1) * 9 + e;
                feat[((size_t)(k >> 2) * Bpad + b) * 4 + (k & 3)] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.0f : 0.0f);
            }
    }
    {   // betas and the zero tail of the feature row (feat is stored as k-quads: [Kpad/4][Bpad][4])
        for (int l = j; l < m.NB; l += nthr) feat[((size_t)(l >> 2) * Bpad + b) * 4 + (l & 3)] = betas[(size_t)b * m.NB + l];
        for (int l = m.K + j; l < m.Kpad; l += nthr) feat[((size_t)(l >> 2) * Bpad + b) * 4 + (l & 3)] = 0.0f;
    }
    __syncthreads();
    const int par = act ? m.parents[j] : -1;
    const int lvl = act ? m.level[j] : -1;
    if (act)
        for (int c = 0; c < 3; c++) Jl[c] = sJ[j][c];
    float rel[3] = {Jl[0], Jl[1], Jl[2]};
    if (act && par >= 0)
        for (int c = 0; c < 3; c++) rel[c] = Jl[c] - sJ[par][c];
    float G[12];   // row-major 3x4: [R | t]
    if (act && lvl == 0) {
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) G[r * 4 + c] = R[r * 3 + c];
            G[r * 4 + 3] = rel[r];
        }
        for (int e = 0; e < 12; e++) sG[j][e] = G[e];
    }
    for (int L = 1; L <= m.maxlevel; L++) {
        __syncthreads();
        if (act && lvl == L) {
            float P[12];
            for (int e = 0; e < 12; e++) P[e] = sG[par][e];
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++)
                    G[r * 4 + c] = P[r * 4 + 0] * R[0 * 3 + c] + P[r * 4 + 1] * R[1 * 3 + c] + P[r * 4 + 2] * R[2 * 3 + c];
                G[r * 4 + 3] = P[r * 4 + 0] * rel[0] + P[r * 4 + 1] * rel[1] + P[r * 4 + 2] * rel[2] + P[r * 4 + 3];
            }
            for (int e = 0; e < 12; e++) sG[j][e] = G[e];
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
}


// Body of the pose-backward stage for body b (whole workgroup).  gA_b [PSI_JP][16] and gfeat_b [Kpad] are THIS body's reduced
// gradients (global memory in pose_bwd_kernel, LDS in the fused fitting tail kernel).
__device__ __forceinline__ void psi_pose_bwd_body(const LbsDev &m, const float *__restrict__ betas, const float *__restrict__ pose,
                                                  const float *__restrict__ Rs, const float *__restrict__ Jls,
                                                  const float *__restrict__ Gs, const float *gA_b, const float *gfeat_b, int b,
                                                  float *__restrict__ g_betas, float *__restrict__ g_pose, float *__restrict__ g_rot)
{
    const int j = threadIdx.x, nthr = blockDim.x;
    const bool act = j < m.J;
    __shared__ float sgG[PSI_JP][12];    // gradient wrt G_j (3x4)
    __shared__ float sgJ[PSI_JP][3];     // gradient wrt the rest joint location J_j
    __shared__ float sgrel[PSI_JP][3];
    __shared__ float sRel[PSI_JP][3];
    __shared__ float sR[PSI_JP][9];
    __shared__ float sJ[PSI_JP][3];
    __shared__ int sChild[PSI_JP];       // child lists (CSR) staged once: the level sweep must not wait on global loads
    float R[9], Jl[3], G[12], gG[12], gJ[3] = {0, 0, 0};
    for (int e = 0; e < 12; e++) gG[e] = 0.0f;
    const int cp0 = act ? m.child_ptr[j] : 0, cp1 = act ? m.child_ptr[j + 1] : 0;
    if (j < m.J - 1) sChild[j] = m.child_idx[j];
    if (act) {
        const psi_f4 *Rp = (const psi_f4 *)(Rs + ((size_t)b * m.J + j) * 12);
        const psi_f4 *Gp = (const psi_f4 *)(Gs + ((size_t)b * m.J + j) * 12);
        const psi_f4 *Ap = (const psi_f4 *)(gA_b + j * 16);
        float gA[12];
        for (int r = 0; r < 3; r++) {
            const psi_f4 rr = Rp[r], gg = Gp[r], aa4 = Ap[r];
            for (int c = 0; c < 3; c++) { R[r * 3 + c] = rr[c]; sR[j][r * 3 + c] = rr[c]; }
            for (int c = 0; c < 4; c++) { G[r * 4 + c] = gg[c]; gA[r * 4 + c] = aa4[c]; }
        }
        for (int c = 0; c < 3; c++) { Jl[c] = Jls[((size_t)b * m.J + j) * 3 + c]; sJ[j][c] = Jl[c]; }
        // A = [G_R | G_t - G_R J]
        for (int r = 0; r < 3; r++) {
            float gt = gA[r * 4 + 3];
            for (int c = 0; c < 3; c++) gG[r * 4 + c] = gA[r * 4 + c] - gt * Jl[c];
            gG[r * 4 + 3] = gt;
        }
        for (int c = 0; c < 3; c++)
            gJ[c] = -(G[0 * 4 + c] * gA[0 * 4 + 3] + G[1 * 4 + c] * gA[1 * 4 + 3] + G[2 * 4 + c] * gA[2 * 4 + 3]);
        for (int e = 0; e < 12; e++) sgG[j][e] = gG[e];
    }
    __syncthreads();
    const int par = act ? m.parents[j] : -1;
    const int lvl = act ? m.level[j] : -1;
    if (act) {
        for (int c = 0; c < 3; c++) sRel[j][c] = (par >= 0) ? Jl[c] - sJ[par][c] : Jl[c];
    }
    __syncthreads();
    // reverse sweep over levels: a joint first gathers from its children (whose gG are final), then publishes its own
    for (int L = m.maxlevel - 1; L >= 0; L--) {
        if (act && lvl == L) {
            for (int ci = cp0; ci < cp1; ci++) {
                int ch = sChild[ci];
                // G_ch.R = G_j.R R_ch ; G_ch.t = G_j.R rel_ch + G_j.t
                for (int r = 0; r < 3; r++) {
                    for (int c = 0; c < 3; c++) {
                        float a = 0;
                        for (int k = 0; k < 3; k++) a += sgG[ch][r * 4 + k] * sR[ch][c * 3 + k];   // gG_ch.R R_ch^T
                        gG[r * 4 + c] += a + sgG[ch][r * 4 + 3] * sRel[ch][c];
                    }
                    gG[r * 4 + 3] += sgG[ch][r * 4 + 3];
                }
            }
            for (int e = 0; e < 12; e++) sgG[j][e] = gG[e];
        }
        __syncthreads();
    }
    // local gradients: gR_j = P_R^T gG_j.R, grel_j = P_R^T gG_j.t  (P = parent's G; root: identity)
    float gR[9], grel[3] = {0, 0, 0};
    for (int e = 0; e < 9; e++) gR[e] = 0.0f;
    if (act) {
        if (par >= 0) {
            const psi_f4 *Pp = (const psi_f4 *)(Gs + ((size_t)b * m.J + par) * 12);
            float P[12];
            for (int r = 0; r < 3; r++) {
                const psi_f4 pr = Pp[r];
                for (int c = 0; c < 4; c++) P[r * 4 + c] = pr[c];
            }
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
    // feature gradient (reduced over n-slices): betas part and pose-feature part
    if (g_betas) {
        // g_betas[l] = g_feat[l] + sum_q gJ[q] J_s[q][l]: the (joint, axis) range is cut into 8 parts summed through LDS
        __shared__ float sgb[8][32];
        const int nq = m.J * 3, per = (nq + 7) / 8;
        for (int i = j; i < 8 * m.NB && m.NB <= 32; i += nthr) {
            const int part = i / m.NB, l = i - part * m.NB;
            const int q1 = min(nq, (part + 1) * per);
            float a = 0.0f;
            for (int q = part * per; q < q1; q++) a += (&sgJ[0][0])[q] * m.J_s[(size_t)q * m.NB + l];
            sgb[part][l] = a;
        }
        __syncthreads();
        for (int l = j; l < m.NB; l += nthr) {
            float a = gfeat_b[l];
            if (m.NB <= 32) {
                for (int part = 0; part < 8; part++) a += sgb[part][l];
            } else {
                for (int q = 0; q < nq; q++) a += (&sgJ[0][0])[q] * m.J_s[(size_t)q * m.NB + l];
            }
            g_betas[(size_t)b * m.NB + l] = a;
        }
    }
    if (act && (g_pose || g_rot)) {
        if (j >= 1)
            for (int e = 0; e < 9; e++) gR[e] += gfeat_b[m.NB + (j - 1) * 9 + e];
        if (g_rot)
            for (int e = 0; e < 9; e++) g_rot[((size_t)b * m.J + j) * 9 + e] = gR[e];
    }
    if (act && g_pose) {
        // Rodrigues backward (lbs.py:177-191)
        const float *aa = pose + ((size_t)b * m.J + j) * 3;
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
        for (int q = 0; q < 3; q++) g_pose[((size_t)b * m.J + j) * 3 + q] = ga[q];
    }
}

