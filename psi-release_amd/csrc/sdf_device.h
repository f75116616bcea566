// Device-side trilinear SDF interpolation shared by sdf.hip (operator) and fit.hip (fused engine).
// Semantics: torch grid_sample (5-D, bilinear, padding_mode='border') as called at fitting_proxe.py:144-151,
// including the world->[-1,1] normalisation of fitting_proxe.py:147; SURVEY.md Appendix C.
#pragma once
#include <hip/hip_runtime.h>

typedef float psi_f2u __attribute__((ext_vector_type(2), aligned(4)));     // 8-byte load at 4-byte alignment (global_load_dwordx2)

struct PsiAxis {
    int i0, i1;
    float w1, du;   // weight of the upper corner; d(u)/d(vert) (0 when clamped by the border rule)
};

__device__ __forceinline__ PsiAxis psi_axis_setup(float v, float mn, float mx, int D, int align_corners)
{
    float nrm = (v - mn) / (mx - mn) * 2.0f - 1.0f;     // fitting_proxe.py:147, in this operation order
    float u, scale;
    if (align_corners) {
        u = (nrm + 1.0f) / 2.0f * (float)(D - 1);
        scale = (float)(D - 1) / 2.0f;
    } else {
        u = ((nrm + 1.0f) * (float)D - 1.0f) / 2.0f;
        scale = (float)D / 2.0f;
    }
    float g = scale;                                    // border: clip to [0, D-1], clipped coordinate has zero gradient
    if (!(u > 0.0f)) { u = 0.0f; g = 0.0f; }
    else if (u >= (float)(D - 1)) { u = (float)(D - 1); g = 0.0f; }
    float fl = floorf(u);
    PsiAxis a;
    a.i0 = (int)fl;
    a.w1 = u - fl;
    a.i1 = min(a.i0 + 1, D - 1);                        // upper corner of the last cell has weight 0
    a.du = g * 2.0f / (mx - mn);
    return a;
}

// value at world point (x,y,z); grad[3] = d value / d (x,y,z).  vol is [D][D][D] indexed [ix][iy][iz].
__device__ __forceinline__ float psi_trilinear(const float *__restrict__ vol, const float *__restrict__ gmin,
                                               const float *__restrict__ gmax, float x, float y, float z, int D,
                                               int align_corners, float *grad)
{
    PsiAxis ax = psi_axis_setup(x, gmin[0], gmax[0], D, align_corners);
    PsiAxis ay = psi_axis_setup(y, gmin[1], gmax[1], D, align_corners);
    PsiAxis az = psi_axis_setup(z, gmin[2], gmax[2], D, align_corners);
    const size_t x0 = (size_t)ax.i0 * D, x1 = (size_t)ax.i1 * D;
    const size_t r00 = (x0 + ay.i0) * D, r01 = (x0 + ay.i1) * D, r10 = (x1 + ay.i0) * D, r11 = (x1 + ay.i1) * D;
    // the two z-neighbours of a corner pair are adjacent in memory ([ix][iy][iz] layout): FOUR 8-byte gathers instead of eight
    // 4-byte ones (the gathers are what bounds the fused skinning+SDF kernel at large batches).  At the upper border both corners
    // are the last cell (i0 == i1 == D-1): the pair is read one cell lower and its upper half used for both.
    const int zb = min(az.i0, D - 2);
    const bool top = az.i0 > zb;
    const psi_f2u p00 = *(const psi_f2u *)(vol + r00 + zb), p01 = *(const psi_f2u *)(vol + r01 + zb);
    const psi_f2u p10 = *(const psi_f2u *)(vol + r10 + zb), p11 = *(const psi_f2u *)(vol + r11 + zb);
    float c000 = top ? p00.y : p00.x, c001 = p00.y;
    float c010 = top ? p01.y : p01.x, c011 = p01.y;
    float c100 = top ? p10.y : p10.x, c101 = p10.y;
    float c110 = top ? p11.y : p11.x, c111 = p11.y;
    const float wx1 = ax.w1, wx0 = 1.0f - ax.w1;
    const float wy1 = ay.w1, wy0 = 1.0f - ay.w1;
    const float wz1 = az.w1, wz0 = 1.0f - az.w1;
    float c00 = c000 * wz0 + c001 * wz1, c01 = c010 * wz0 + c011 * wz1;
    float c10 = c100 * wz0 + c101 * wz1, c11 = c110 * wz0 + c111 * wz1;
    float c0 = c00 * wy0 + c01 * wy1, c1 = c10 * wy0 + c11 * wy1;
    if (grad) {
        float gx = c1 - c0;
        float gy = (c01 - c00) * wx0 + (c11 - c10) * wx1;
        float d00 = c001 - c000, d01 = c011 - c010, d10 = c101 - c100, d11 = c111 - c110;
        float gz = (d00 * wy0 + d01 * wy1) * wx0 + (d10 * wy0 + d11 * wy1) * wx1;
        grad[0] = gx * ax.du;
        grad[1] = gy * ay.du;
        grad[2] = gz * az.du;
    }
    return c0 * wx0 + c1 * wx1;
}


// ------------------------------------------------------------------------------------------------
// Bricked volume layout WITH A ONE-VOXEL APRON (the fused fitting engine keeps its own copy in this order): the volume is cut into
// 4 x 4 x 4-cell bricks; a brick stores the 5 x 5 x 5 voxels its cells touch (the upper faces are copies of the neighbours' lower
// faces; beyond the volume the last voxel is repeated), [lx][ly][lz] with strides 25 / 5 / 1, padded to 128 floats = 512 bytes = four
// 128-byte lines.  Every sample finds all eight corners inside ONE brick, within 31 floats of each other (one or two cache lines),
// and the two z-neighbours of a corner pair are adjacent: FOUR 8-byte gathers per sample.  (History: plain [ix][iy][iz] order = four
// cache lines per sample; apron-less 4 x 4 x 4 bricks = 8 four-byte gathers, because a z-pair straddles bricks one time in four.  The
// gathers' L1 tag lookups — one per lane and instruction — are what bounds the fused skinning + SDF kernel at large batches,
// profiles/r02_pmc_skin_fwd_sdf_b512.txt; halving the gather instructions halves them.)  D % 4 == 0.  Twice the footprint of the
// plain volume (134 MB at 256^3); a body still only touches the bricks around it.
// ------------------------------------------------------------------------------------------------
constexpr int PSI_BRICK_FLOATS = 128;

__device__ __forceinline__ float psi_trilinear_bricked(const float *__restrict__ vol, const float *__restrict__ gmin,
                                                       const float *__restrict__ gmax, float x, float y, float z, int D,
                                                       int align_corners, float *grad)
{
    PsiAxis ax = psi_axis_setup(x, gmin[0], gmax[0], D, align_corners);
    PsiAxis ay = psi_axis_setup(y, gmin[1], gmax[1], D, align_corners);
    PsiAxis az = psi_axis_setup(z, gmin[2], gmax[2], D, align_corners);
    const int nbr = D >> 2;
    // i1 = min(i0 + 1, D - 1) is voxel l + 1 of the same brick (the apron repeats the last voxel beyond the volume)
    const size_t brick = ((size_t)(ax.i0 >> 2) * nbr + (ay.i0 >> 2)) * nbr + (az.i0 >> 2);
    const float *p = vol + brick * PSI_BRICK_FLOATS + (ax.i0 & 3) * 25 + (ay.i0 & 3) * 5 + (az.i0 & 3);
    const psi_f2u p00 = *(const psi_f2u *)(p), p01 = *(const psi_f2u *)(p + 5);
    const psi_f2u p10 = *(const psi_f2u *)(p + 25), p11 = *(const psi_f2u *)(p + 30);
    const float c000 = p00.x, c001 = p00.y, c010 = p01.x, c011 = p01.y;
    const float c100 = p10.x, c101 = p10.y, c110 = p11.x, c111 = p11.y;
    const float wx1 = ax.w1, wx0 = 1.0f - ax.w1;
    const float wy1 = ay.w1, wy0 = 1.0f - ay.w1;
    const float wz1 = az.w1, wz0 = 1.0f - az.w1;
    float c00 = c000 * wz0 + c001 * wz1, c01 = c010 * wz0 + c011 * wz1;
    float c10 = c100 * wz0 + c101 * wz1, c11 = c110 * wz0 + c111 * wz1;
    float c0 = c00 * wy0 + c01 * wy1, c1 = c10 * wy0 + c11 * wy1;
    if (grad) {
        float gx = c1 - c0;
        float gy = (c01 - c00) * wx0 + (c11 - c10) * wx1;
        float d00 = c001 - c000, d01 = c011 - c010, d10 = c101 - c100, d11 = c111 - c110;
        float gz = (d00 * wy0 + d01 * wy1) * wx0 + (d10 * wy0 + d11 * wy1) * wx1;
        grad[0] = gx * ax.du;
        grad[1] = gy * ay.du;
        grad[2] = gz * az.du;
    }
    return c0 * wx0 + c1 * wx1;
}

