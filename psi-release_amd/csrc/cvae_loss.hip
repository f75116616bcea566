// The body-vector glue around the CVAE in a training step, as four launches instead of ~190 elementwise ones, gfx950.
//
// Per optimiser step the trainers (train_s1.py:95-133, train_s2.py:102-139 `cal_loss`) run, on [B, 72..75] tensors:
//   target   : xhnr = convert_to_6D_rot(normalize_global_T(xh, cam_int, max_d))                      cvae.py:118-127, 176-199
//              (torchgeometry 0.1.2 angle_axis_to_rotation_matrix)                -> psi_cvae_target
//   losses   : xh_rec = recover_global_T(xhnr_rec, cam_int, max_d)                                   cvae.py:153-172
//              rec_t  = w_rec (0.5 L1(xhnr_rec[:, :3], xhnr[:, :3]) + 0.5 L1(xh_rec[:, :3], xh[:, :3]))   train_s2.py:122-123
//              rec_p  = w_rec L1(xhnr_rec[:, 3:], xhnr[:, 3:])                                            train_s2.py:124
//              KL     = fca^2 w_kl 0.5 mean(exp(logsigma2) + mu^2 - 1 - logsigma2)   (one per latent)     train_s2.py:126-133
//              vposer = w_vp mean(xh_rec[:, 19:51]^2)        (the VPoser latent of the 75-D layout)       train_s2.py:135-139
//              and their backward                                                      -> psi_cvae_losses_forward / _backward
// Each of these is a handful of flops per element; as PyTorch operators they are ~55 (target), ~45 (losses) and ~90 (autograd) launches
// of 4-5 us each inside the captured step.  The arithmetic below follows the operator sequence of psi_release_amd/geometry.py (same
// association, no FMA contraction: this file is compiled with -ffp-contract=off), so results agree with the operator path to the last
// bits of the transcendental functions; the reductions are fixed-order (per-block partials, summed in block order).
#include "psi_internal.h"

namespace {

struct CvaeLossDev {
    const float *rec, *tgt, *xh, *cam_int, *max_d;     // [B,75] (grad), [B,75], [B,72], [B,9], [B]
    const float *mu[2], *lv[2];                        // [B, nz[k]] (nullptr: latent absent)
    int nz[2];
    int B;
    float w_rec, w_kl, w_vp, fca;
    const float *fca_dev;                              // device scalar (captured steps) or nullptr -> fca
    float *xh_rec;                                     // [B,75]
    float *loss[5];                                    // rec_t, rec_p, KL0, KL1, vposer (one float each)
    // backward
    const float *g_loss[5];                            // upstream gradients of the five scalars (nullptr: none)
    const float *g_xh_rec;                             // [B,75] or nullptr
    float *g_rec, *g_mu[2], *g_lv[2];
};

__device__ __forceinline__ float sgn(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// cvae.py:176-199 + cvae.py:118-127: camera-normalised translation, global orientation as the first two columns of its rotation matrix
__global__ void cvae_target_kernel(const float *__restrict__ xh, const float *__restrict__ cam_int, const float *__restrict__ max_d, int B,
                                   float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 75) return;
    const int b = i / 75, j = i % 75;
    const float *x = xh + (size_t)b * 72;
    if (j >= 9) {
        out[i] = x[j - 3];
        return;
    }
    if (j < 3) {
        const float *K = cam_int + (size_t)b * 9;
        const float s_ = 1.0f / fmaxf(K[2], K[5]);
        if (j == 0) out[i] = s_ * x[0] * K[0] / (x[2] + 1e-6f);
        else if (j == 1) out[i] = s_ * x[1] * K[4] / (x[2] + 1e-6f);
        else out[i] = 2.0f * x[2] / max_d[b] - 1.0f;
        return;
    }
    // torchgeometry 0.1.2 angle_axis_to_rotation_matrix; element (r, c) of the 3x3 block, c < 2
    const int r = (j - 3) / 2, c = (j - 3) % 2;
    const float rx = x[3], ry = x[4], rz = x[5];
    const float theta2 = rx * rx + ry * ry + rz * rz;
    float v;
    if (theta2 > 1e-6f) {
        const float theta = sqrtf(theta2);
        const float wx = rx / (theta + 1e-6f), wy = ry / (theta + 1e-6f), wz = rz / (theta + 1e-6f);
        const float cs = cosf(theta), sn = sinf(theta), k = 1.0f - cs;
        switch (r * 3 + c) {
        case 0: v = cs + wx * wx * k; break;
        case 1: v = wx * wy * k - wz * sn; break;
        case 3: v = wz * sn + wx * wy * k; break;
        case 4: v = cs + wy * wy * k; break;
        case 6: v = -wy * sn + wx * wz * k; break;
        default: v = wx * sn + wy * wz * k; break;        // 7
        }
    } else {
        switch (r * 3 + c) {
        case 0: v = 1.0f; break;
        case 1: v = -rz; break;
        case 3: v = rz; break;
        case 4: v = 1.0f; break;
        case 6: v = -ry; break;
        default: v = rx; break;
        }
    }
    out[i] = v;
}

// Forward in two launches: CL_GRID blocks leave fixed partial sums of the six quantities (and write xh_rec), one small block adds them in
// block order.  (One 1024-thread block doing all of it took 27 us at batch 128: 64 dependent rounds of loads per thread.)
constexpr int CL_GRID = 64;

__global__ __launch_bounds__(256) void cvae_losses_partial_kernel(CvaeLossDev a, float *__restrict__ part /* [CL_GRID][6] */)
{
    __shared__ float sh[6][4];
    float acc[6] = {0, 0, 0, 0, 0, 0};          // |rec - tgt| over [:, :3], |xh_rec - xh| over [:, :3], |rec - tgt| over [:, 3:], KL0, KL1, latent^2
    const int n = a.B * 75, stride = CL_GRID * 256, t0 = blockIdx.x * 256 + threadIdx.x;
    for (int i = t0; i < n; i += stride) {
        const int b = i / 75, j = i % 75;
        const float r = a.rec[i];
        const float d = fabsf(r - a.tgt[i]);
        if (j < 3) {
            // recover_global_T, cvae.py:153-172
            const float *K = a.cam_int + (size_t)b * 9;
            const float s_ = 1.0f / fmaxf(K[2], K[5]);
            const float z = (a.rec[(size_t)b * 75 + 2] + 1.0f) / 2.0f * a.max_d[b];
            float v;
            if (j == 0) v = r * z / s_ / K[0];
            else if (j == 1) v = r * z / s_ / K[4];
            else v = z;
            a.xh_rec[i] = v;
            acc[0] += d;
            acc[1] += fabsf(v - a.xh[(size_t)b * 72 + j]);
        } else {
            a.xh_rec[i] = r;
            acc[2] += d;
            if (j >= 19 && j < 51) acc[5] += r * r;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; k++)
        if (a.mu[k]) {
            const int nk = a.B * a.nz[k];
            for (int i = t0; i < nk; i += stride) {
                const float m = a.mu[k][i], l = a.lv[k][i];
                acc[3 + k] += expf(l) + m * m - 1.0f - l;
            }
        }
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[q] += __shfl_down(acc[q], o, 64);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 6; q++) sh[q][threadIdx.x >> 6] = acc[q];
    __syncthreads();
    if (threadIdx.x < 6) part[blockIdx.x * 6 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}

__global__ __launch_bounds__(64) void cvae_losses_finalize_kernel(CvaeLossDev a, const float *__restrict__ part)
{
    const int t = threadIdx.x;
    float v[CL_GRID];
#pragma unroll
    for (int b = 0; b < CL_GRID; b++) v[b] = t < 6 ? part[b * 6 + t] : 0.0f;      // all loads first, then the sum in block order
    float s = 0.0f;
#pragma unroll
    for (int b = 0; b < CL_GRID; b++) s += v[b];
    float acc[6];
#pragma unroll
    for (int q = 0; q < 6; q++) acc[q] = __shfl(s, q, 64);
    if (t == 0) {
        const float fB = (float)a.B;
        const float fca = a.fca_dev ? *a.fca_dev : a.fca;
        *a.loss[0] = a.w_rec * (0.5f * (acc[0] / (3.0f * fB)) + 0.5f * (acc[1] / (3.0f * fB)));
        *a.loss[1] = a.w_rec * (acc[2] / (72.0f * fB));
        for (int k = 0; k < 2; k++) *a.loss[2 + k] = a.mu[k] ? fca * fca * a.w_kl * 0.5f * (acc[3 + k] / (fB * (float)a.nz[k])) : 0.0f;
        *a.loss[4] = a.w_vp * (acc[5] / (32.0f * fB));
    }
}

__global__ void cvae_losses_bwd_kernel(CvaeLossDev a, int n_max)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_max) return;
    const float fB = (float)a.B;
    const float g_t = a.g_loss[0] ? *a.g_loss[0] : 0.0f, g_p = a.g_loss[1] ? *a.g_loss[1] : 0.0f, g_v = a.g_loss[4] ? *a.g_loss[4] : 0.0f;
    if (i < a.B * 75) {
        const int b = i / 75, j = i % 75;
        const float r = a.rec[i];
        const float gx = a.g_xh_rec ? a.g_xh_rec[i] : 0.0f;
        if (j >= 3) {
            float g = g_p * a.w_rec / (72.0f * fB) * sgn(r - a.tgt[i]) + gx;
            if (j >= 19 && j < 51) g += g_v * a.w_vp * 2.0f * r / (32.0f * fB);
            a.g_rec[i] = g;
        } else {
            const float *K = a.cam_int + (size_t)b * 9;
            const float s_ = 1.0f / fmaxf(K[2], K[5]);
            const float *t = a.rec + (size_t)b * 75;
            const float hmd = a.max_d[b] / 2.0f;
            const float z = (t[2] + 1.0f) * hmd;
            const float ct = g_t * a.w_rec * 0.5f / (3.0f * fB);
            const float own = ct * sgn(r - a.tgt[i]);
            // gradient arriving at xh_rec[b, k]: the second L1 term of rec_t and whatever the scene losses sent back
            float G[3];
#pragma unroll
            for (int k = 0; k < 3; k++)
                G[k] = ct * sgn(a.xh_rec[(size_t)b * 75 + k] - a.xh[(size_t)b * 72 + k]) + (a.g_xh_rec ? a.g_xh_rec[(size_t)b * 75 + k] : 0.0f);
            float g;
            if (j == 0) g = G[0] * z / s_ / K[0];
            else if (j == 1) g = G[1] * z / s_ / K[4];
            else g = (G[0] * t[0] / s_ / K[0] + G[1] * t[1] / s_ / K[4] + G[2]) * hmd;
            a.g_rec[i] = own + g;
        }
    }
    const float fca = a.fca_dev ? *a.fca_dev : a.fca;
#pragma unroll
    for (int k = 0; k < 2; k++)
        if (a.mu[k] && i < a.B * a.nz[k]) {
            const float gk = a.g_loss[2 + k] ? *a.g_loss[2 + k] : 0.0f;
            const float c = gk * (fca * fca * a.w_kl * 0.5f) / (fB * (float)a.nz[k]);
            a.g_mu[k][i] = c * 2.0f * a.mu[k][i];
            a.g_lv[k][i] = c * (expf(a.lv[k][i]) - 1.0f);
        }
}

} // namespace

extern "C" int psi_cvae_target(const float *xh72, const float *cam_int, const float *max_d, int B, float *out75, void *stream)
{
    PSI_REQUIRE(xh72 && cam_int && max_d && out75, "null pointer");
    PSI_REQUIRE(B > 0 && B <= (1 << 20), "batch size out of range");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cvae_target_kernel, dim3(psi_cdiv(B * 75, 256)), dim3(256), 0, st, xh72, cam_int, max_d, B, out75);
    PSI_CHECK_LAUNCH("cvae_target_kernel");
    return 0;
}

static int cvae_fill(CvaeLossDev &a, const float *rec75, const float *target75, const float *xh72, const float *cam_int, const float *max_d,
                     const float *mu0, const float *logvar0, int nz0, const float *mu1, const float *logvar1, int nz1, int B, float w_rec,
                     float w_kl, float w_vposer, float fca, const float *fca_dev)
{
    PSI_REQUIRE(rec75 && target75 && xh72 && cam_int && max_d, "null pointer");
    PSI_REQUIRE(B > 0 && B <= (1 << 20), "batch size out of range");
    PSI_REQUIRE((mu0 != nullptr) == (logvar0 != nullptr) && (mu1 != nullptr) == (logvar1 != nullptr), "a latent needs both mu and logsigma2");
    PSI_REQUIRE((!mu0 || nz0 > 0) && (!mu1 || nz1 > 0), "latent width must be positive");
    a.rec = rec75; a.tgt = target75; a.xh = xh72; a.cam_int = cam_int; a.max_d = max_d;
    a.mu[0] = mu0; a.lv[0] = logvar0; a.nz[0] = mu0 ? nz0 : 0;
    a.mu[1] = mu1; a.lv[1] = logvar1; a.nz[1] = mu1 ? nz1 : 0;
    a.B = B; a.w_rec = w_rec; a.w_kl = w_kl; a.w_vp = w_vposer; a.fca = fca; a.fca_dev = fca_dev;
    return 0;
}

extern "C" size_t psi_cvae_losses_workspace_floats(void) { return (size_t)CL_GRID * 6; }

extern "C" int psi_cvae_losses_forward(const float *rec75, const float *target75, const float *xh72, const float *cam_int, const float *max_d,
                                       const float *mu0, const float *logvar0, int nz0, const float *mu1, const float *logvar1, int nz1, int B,
                                       float w_rec, float w_kl, float w_vposer, float fca, const float *fca_dev, float *ws, float *xh_rec75,
                                       float *losses5, void *stream)
{
    CvaeLossDev a = {};
    int rc = cvae_fill(a, rec75, target75, xh72, cam_int, max_d, mu0, logvar0, nz0, mu1, logvar1, nz1, B, w_rec, w_kl, w_vposer, fca, fca_dev);
    if (rc) return rc;
    PSI_REQUIRE(xh_rec75 && losses5 && ws, "null output / workspace");
    a.xh_rec = xh_rec75;
    for (int k = 0; k < 5; k++) a.loss[k] = losses5 + k;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cvae_losses_partial_kernel, dim3(CL_GRID), dim3(256), 0, st, a, ws);
    PSI_CHECK_LAUNCH("cvae_losses_partial_kernel");
    hipLaunchKernelGGL(cvae_losses_finalize_kernel, dim3(1), dim3(64), 0, st, a, (const float *)ws);
    PSI_CHECK_LAUNCH("cvae_losses_finalize_kernel");
    return 0;
}

extern "C" int psi_cvae_losses_backward(const float *rec75, const float *target75, const float *xh72, const float *cam_int, const float *max_d,
                                        const float *mu0, const float *logvar0, int nz0, const float *mu1, const float *logvar1, int nz1, int B,
                                        float w_rec, float w_kl, float w_vposer, float fca, const float *fca_dev, const float *xh_rec75,
                                        const float *g_losses5, const float *g_xh_rec75, float *g_rec75, float *g_mu0, float *g_logvar0,
                                        float *g_mu1, float *g_logvar1, void *stream)
{
    CvaeLossDev a = {};
    int rc = cvae_fill(a, rec75, target75, xh72, cam_int, max_d, mu0, logvar0, nz0, mu1, logvar1, nz1, B, w_rec, w_kl, w_vposer, fca, fca_dev);
    if (rc) return rc;
    PSI_REQUIRE(xh_rec75 && g_losses5 && g_rec75, "null pointer");
    PSI_REQUIRE((!mu0 || (g_mu0 && g_logvar0)) && (!mu1 || (g_mu1 && g_logvar1)), "a latent needs both gradient outputs");
    a.xh_rec = const_cast<float *>(xh_rec75);
    for (int k = 0; k < 5; k++) a.g_loss[k] = g_losses5 + k;
    a.g_xh_rec = g_xh_rec75;
    a.g_rec = g_rec75;
    a.g_mu[0] = g_mu0; a.g_lv[0] = g_logvar0; a.g_mu[1] = g_mu1; a.g_lv[1] = g_logvar1;
    int n_max = B * 75;
    for (int k = 0; k < 2; k++) n_max = a.mu[k] && B * a.nz[k] > n_max ? B * a.nz[k] : n_max;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(cvae_losses_bwd_kernel, dim3(psi_cdiv(n_max, 256)), dim3(256), 0, st, a, n_max);
    PSI_CHECK_LAUNCH("cvae_losses_bwd_kernel");
    return 0;
}
