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


// ------------------------------------------------------------------------------------------------
// The same lookup for the fused fitting engine's skinning epilogue, written for the instruction count (the fused skinning + SDF
// kernel is vector-ALU bound at large batches: profiles/r04_pmc_skin_fwd_sdf_b512.txt).  What differs from psi_trilinear_bricked:
//   * the world -> grid map is ONE subtract and ONE multiply per axis, u = (x - o) * k, with constants prepared on the host in double
//     precision (k = (D - 1) / (max - min), o = min with align_corners; k = D / (max - min), o = min + 0.5 / k without) instead of
//     the reference's divide-by-extent chain (fitting_proxe.py:147 + grid_sample's unnormalise: two IEEE divides per axis, ~20
//     instructions each).  u differs from the reference's fp32 chain by a few ulp — the reference's own chain is as far from the
//     exact value — i.e. by 1e-5 of a voxel;
//   * clamp / floor / fraction are v_med3 / v_cvt / v_fract; the border rule's zero gradient is a lane mask that the caller
//     combines with its sdf < 0 mask in scalar registers;
//   * brick and in-brick offsets are 24-bit multiply-adds into ONE 32-bit byte offset from a wave-uniform base (no 64-bit vector
//     arithmetic); the four corner pairs are that offset + 0 / 20 / 100 / 120 bytes as instruction immediates;
//   * the interpolation works on the loaded z-pairs as packed values: lerp along y, then x (both z-ends at once), then z — the
//     differences it forms are the gradient's operands: 14 instructions for value and gradient, a + w (b - a) form.
// Returns the value; g[] = d value / d (x, y, z) BEFORE the border mask, in[] = lane strictly inside (0, D - 1) per axis.
// ------------------------------------------------------------------------------------------------
typedef float psi_f2v __attribute__((ext_vector_type(2)));
struct PsiSdfGrid {
    const float *brick;       // apron-brick volume (above)
    float o[3], ku[3];        // grid origin (see above), grid units per world unit
    float dm1;                // D - 1
    unsigned nbr;             // bricks per axis
};

static inline PsiSdfGrid psi_sdf_grid_make(const float *brick, const float *h_gmin, const float *h_gmax, int D, int align_corners)
{
    PsiSdfGrid g;
    g.brick = brick;
    for (int a = 0; a < 3; a++) {
        const double k = (double)(align_corners ? D - 1 : D) / ((double)h_gmax[a] - (double)h_gmin[a]);
        g.ku[a] = (float)k;
        g.o[a] = (float)((double)h_gmin[a] + (align_corners ? 0.0 : 0.5 / k));
    }
    g.dm1 = (float)(D - 1);
    g.nbr = (unsigned)(D >> 2);
    return g;
}

__device__ __forceinline__ unsigned psi_mad24(unsigned a, unsigned b_uniform, unsigned c)
{
    unsigned r;                                                   // full-rate 24-bit multiply-add (the compiler picks v_mul_lo_u32, a quarter-rate op)
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}

__device__ __forceinline__ float psi_sdf_sample_fast(const PsiSdfGrid &G, float x, float y, float z, float (&g)[3], bool (&in)[3])
{
    const float u[3] = {(x - G.o[0]) * G.ku[0], (y - G.o[1]) * G.ku[1], (z - G.o[2]) * G.ku[2]};
    float w[3];
    unsigned i[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        in[a] = u[a] > 0.0f && u[a] < G.dm1;                     // border: a clipped coordinate has zero gradient
        const float uc = __builtin_amdgcn_fmed3f(u[a], 0.0f, G.dm1);
        w[a] = __builtin_amdgcn_fractf(uc);                      // uc - floor(uc), exact
        i[a] = (unsigned)(int)uc;                                // uc >= 0: truncation is floor
    }
    // voxel (i, i + 1) is voxel (l, l + 1) of brick i >> 2, l = i & 3 (the apron repeats the last voxel beyond the volume)
    const unsigned brick = psi_mad24(psi_mad24(i[0] >> 2, G.nbr, i[1] >> 2), G.nbr, i[2] >> 2);
    const unsigned local = __umul24(i[0] & 3, 100) + __umul24(i[1] & 3, 20) + ((i[2] << 2) & 12);
    const unsigned off = (brick << 9) + local;                  // bytes: 512 per brick, strides 100 / 20 / 4
    const char *vb = (const char *)G.brick;                     // wave-uniform base + 32-bit lane offset (+ immediates)
    const psi_f2v q00 = *(const psi_f2u *)(vb + off), q01 = *(const psi_f2u *)(vb + (off + 20u));           // (z0, z1) pairs at (x0, y0), (x0, y1)
    const psi_f2v q10 = *(const psi_f2u *)(vb + (off + 100u)), q11 = *(const psi_f2u *)(vb + (off + 120u)); //                  (x1, y0), (x1, y1)
    const psi_f2v wy2 = {w[1], w[1]}, wx2 = {w[0], w[0]};
    const psi_f2v e0 = q01 - q00, e1 = q11 - q10;                                        // d/dy on the two x faces, at z0 and z1
    const psi_f2v r0 = __builtin_elementwise_fma(wy2, e0, q00), r1 = __builtin_elementwise_fma(wy2, e1, q10);
    const psi_f2v ex = r1 - r0;                                                          // d/dx at z0 and z1
    const psi_f2v r = __builtin_elementwise_fma(wx2, ex, r0);                            // value at z0 and z1
    const psi_f2v ey = __builtin_elementwise_fma(wx2, e1 - e0, e0);                      // d/dy at z0 and z1
    const float gz = r.y - r.x;
    g[0] = __builtin_fmaf(w[2], ex.y - ex.x, ex.x) * G.ku[0];
    g[1] = __builtin_fmaf(w[2], ey.y - ey.x, ey.x) * G.ku[1];
    g[2] = gz * G.ku[2];
    return __builtin_fmaf(w[2], gz, r.x);
}


// ------------------------------------------------------------------------------------------------
// CELL-MAJOR layout (round 4; the fused engine's default, PSI_SDF_CELLS): every cell stores its OWN eight corner values, 32 contiguous
// bytes [dx][dy][dz], cells in 4 x 4 x 4 bricks of 2 KB ([bx][by][bz][lx][ly][lz][8]); D cells per axis (the last cell repeats the
// last voxel, so the border clamp needs no special case).  A sample is TWO 16-byte gathers into one aligned 32-byte record instead of
// four 8-byte gathers spread over up to 124 bytes.  Why: counters of the fused skinning + SDF kernel at B = 512
// (profiles/r04_pmc_skin_fwd_sdf_b512*.txt) show the first-level cache at 0.48 line accesses per clock and CU for the dense AND the
// compressed-row body model — the kernel's time follows its L1 access count (45.9 M -> 158 us, 32.9 M -> 112 us), not its instruction
// count (805 -> 459 vector instructions per wave moved the dense kernel by 4 %) — and a gather costs one access per LANE and
// instruction: 4 x 64 (+ pairs that straddle a line) of a wave's ~390.  Eight times the plain volume (537 MB at 256^3; a body still
// touches only the cells around it).
// ------------------------------------------------------------------------------------------------
constexpr int PSI_CELL_BRICK_BYTES = 2048;

__device__ __forceinline__ float psi_sdf_sample_cells(const PsiSdfGrid &G, float x, float y, float z, float (&g)[3], bool (&in)[3])
{
    const float u[3] = {(x - G.o[0]) * G.ku[0], (y - G.o[1]) * G.ku[1], (z - G.o[2]) * G.ku[2]};
    float w[3];
    unsigned i[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        in[a] = u[a] > 0.0f && u[a] < G.dm1;                     // border: a clipped coordinate has zero gradient
        const float uc = __builtin_amdgcn_fmed3f(u[a], 0.0f, G.dm1);
        w[a] = __builtin_amdgcn_fractf(uc);
        i[a] = (unsigned)(int)uc;                                // cell index 0 .. D - 1 (cell D - 1: both corners the last voxel, w = 0)
    }
    const unsigned brick = psi_mad24(psi_mad24(i[0] >> 2, G.nbr, i[1] >> 2), G.nbr, i[2] >> 2);
    const unsigned local = ((i[0] & 3) << 9) | ((i[1] & 3) << 7) | ((i[2] & 3) << 5);
    const unsigned off = (brick << 11) + local;                 // bytes
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const char *vb = (const char *)G.brick;
    const f4_t lo = *(const f4_t *)(vb + off), hi = *(const f4_t *)(vb + (off + 16u));       // x0 face, x1 face: (y0z0, y0z1, y1z0, y1z1)
    const psi_f2v q00 = {lo.x, lo.y}, q01 = {lo.z, lo.w}, q10 = {hi.x, hi.y}, q11 = {hi.z, hi.w};
    const psi_f2v wy2 = {w[1], w[1]}, wx2 = {w[0], w[0]};
    const psi_f2v e0 = q01 - q00, e1 = q11 - q10;
    const psi_f2v r0 = __builtin_elementwise_fma(wy2, e0, q00), r1 = __builtin_elementwise_fma(wy2, e1, q10);
    const psi_f2v ex = r1 - r0;
    const psi_f2v r = __builtin_elementwise_fma(wx2, ex, r0);
    const psi_f2v ey = __builtin_elementwise_fma(wx2, e1 - e0, e0);
    const float gz = r.y - r.x;
    g[0] = __builtin_fmaf(w[2], ex.y - ex.x, ex.x) * G.ku[0];
    g[1] = __builtin_fmaf(w[2], ey.y - ey.x, ey.x) * G.ku[1];
    g[2] = gz * G.ku[2];
    return __builtin_fmaf(w[2], gz, r.x);
}

