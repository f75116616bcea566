// Device bodies of the joint-side half of the LBS backward — skin_bwd_A (gA = W^T (g_local (x) [v_posed; 1]), a contraction over
// vertices) and blend_bwd (g_feat = g_vposed dirs^T, a contraction over columns with the blend-shape matrix streamed again) — as
// functions of their OPERANDS, so that two kernels can share them: bwd_joint_kernel (lbs.hip: the operator's own backward, the model's
// vertices) and the fused fitting engine's fit_bwd_joint_kernel (fit.hip), which runs the same bodies over two classes of "vertices"
// in one grid: the model's 10475 (carrying the UNSCALED penetration gradient) and the engine's n_c contact slots (carrying the contact
// gradient) — everything behind dL/dverts is linear in it (lbs.py:108-116 backward), so the two parts travel side by side as extra
// slices of the same split contractions and meet, with the 1 / N of the penetration term (fitting_proxe.py:155-158), in the reduction.
#pragma once
#include "lbs_device.h"

#ifndef PSI_SKA_MARK
#define PSI_SKA_MARK(k)
#endif

// ------------------------------------------------------------------------------------------------
// skinning backward, joint part (MFMA): gA[b][j][r*4+s] = sum_v W[v][j] * g_local[b][v][r] * [v_posed;1][s]
// ------------------------------------------------------------------------------------------------
// One workgroup = one 256-vertex slice x NBODY bodies; it is ONE product  D[64 joints][12 NBODY] = W^T[64][256] P[256][12 NBODY]  with
// P[v][12 bb + 4 r + s] = g_local[bb][v][r] [v_posed[bb][v]; 1][s]: the 12 (r, s) entries of consecutive bodies are packed side by side, so
// the 16-wide MFMA tiles carry no padding (a tile per body carried four zero columns: a quarter of the instructions).  Wave w owns joint
// tile w (joints 16 w .. 16 w + 15) for the WHOLE slice and all column tiles: nothing to reduce across waves — the previous form split
// the slice's vertices over the waves and met in LDS behind two barriers per body, which behind the blend_bwd stream's MFMA bursts (the
// two share the launch and each SIMD) cost 2.5 us per body (workgroup timeline: operands staged at 5.6 us, end at 22 — later than the
// stream itself, profiles/r04_timeline_bwd_joint.txt).
//   A operand: lane (li, lk) supplies joint 16 w + li, vertex 64 lk + st in step st: 64 consecutive floats of its weight row, a quad
// (16 B) per four steps, each used for every column tile.  B operand: the same vertex, column 16 nt + li: a product of two LDS values; the
// staged operands are kept component-major ([body][component][vertex], 64-vertex runs padded by 4) so that the four steps of a quad
// are ONE 16-byte LDS read each for g_local and v_posed.
constexpr int SKA_NBODY = 8;
constexpr int SKA_ROW = 256 + 16;          // floats per staged (body, component) row: vertex v sits at v + 4 (v / 64)
constexpr int SKA_MAXT = (12 * SKA_NBODY + 15) / 16;
constexpr int SKA_SMEM_F4 = SKA_NBODY * 2 * 3 * SKA_ROW / 4;      // LDS of one skin_bwd_A workgroup, in 16-byte units

// operands of ONE 256-vertex slice
struct PsiSkaSlice {
    const float *wtt;          // the slice's four wave tiles of the tiled weights: [4][PSI_JP][64]
    const float *gl, *vp;      // g_local / posed vertices of the slice's 256 vertices (768 floats), body b at + b * row_stride
    size_t row_stride;
    float *part;               // [B][PSI_JP][16]: this slice's partial joint-transform gradients
};

// NT: column tiles of a full workgroup, (12 nbody + 15) / 16 — a compile-time count keeps the 16 quads straight-line code with all weight
// quads in registers (a workgroup with fewer bodies than nbody repeats its last column in the spare tiles)
template <int NT>
__device__ __forceinline__ void skin_bwd_A_body(const PsiSkaSlice &o, int B, int b0, int nbody, psi_f4 *smem)
{
    typedef psi_f4 f4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int nb = min(nbody, B - b0);
    const int ncol = 12 * nb;
    // my weight row: joint 16 w + li, vertices 64 lk .. 64 lk + 63 of the slice — all 16 quads requested now (behind the stream's traffic
    // a load takes ~2 us: requested two quads ahead of their use, they made every quad wait, 1 us per quad)
    f4 wa[16];
    {
        // from the wave-tiled copy ([Vpad/64][64 joints][64 vertices]: the same 256 contiguous bytes per lane)
        const f4 *wr = (const f4 *)(o.wtt + ((size_t)lk * PSI_JP + (w * 16 + li)) * 64);
#pragma unroll
        for (int q = 0; q < 16; q++) wa[q] = wr[q];
    }
    // ALL bodies' operands of this slice are requested up front, coalesced (768 consecutive floats of g_local and of v_posed per body: one
    // 16-byte load each for threads 0..191), and parked in LDS component-major
    float *sG = (float *)smem;                                  // [SKA_NBODY][3][SKA_ROW]
    float *sP = sG + SKA_NBODY * 3 * SKA_ROW;                   // [SKA_NBODY][3][SKA_ROW]
    {
        f4 og[SKA_NBODY], op[SKA_NBODY];
        const int t4 = threadIdx.x;
#pragma unroll
        for (int bb = 0; bb < SKA_NBODY; bb++)
            if (bb < nb && t4 < 192) {
                og[bb] = *(const f4 *)(o.gl + (size_t)(b0 + bb) * o.row_stride + t4 * 4);
                op[bb] = *(const f4 *)(o.vp + (size_t)(b0 + bb) * o.row_stride + t4 * 4);
            }
#pragma unroll
        for (int bb = 0; bb < SKA_NBODY; bb++)
            if (bb < nb && t4 < 192) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int i = t4 * 4 + e, v = i / 3, comp = i - 3 * v;
                    const int pos = (bb * 3 + comp) * SKA_ROW + v + 4 * (v >> 6);
                    sG[pos] = og[bb][e];
                    sP[pos] = op[bb][e];
                }
            }
    }
    __syncthreads();
    PSI_SKA_MARK(0);
    // my columns: tile nt -> column 16 nt + li = 12 bb + 4 r + s
    int goff[NT], poff[NT];                                     // float offsets of my (body, r) / (body, s) rows + my 64-vertex run; poff < 0: s == 3
    f4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int c = min(16 * nt + li, ncol - 1);             // (columns past the last body repeat the last one; they are not stored)
        const int bb = c / 12, rs = c - 12 * bb, r = rs >> 2, sx = rs & 3;
        goff[nt] = (bb * 3 + r) * SKA_ROW + lk * 68;
        poff[nt] = sx < 3 ? (bb * 3 + sx) * SKA_ROW + lk * 68 : -1;
        acc[nt] = (f4){0, 0, 0, 0};
    }
    // (requesting quad q + 1's LDS operands before quad q's MFMAs — a denser MFMA stream of this wave — measured SLOWER, 23.1 against 21.8 us:
    // the stream wave on the same SIMD then waits longer for the pipe, and the launch ends when the later of the two kinds does)
#pragma unroll
    for (int q = 0; q < 16; q++) {
        f4 bop[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const f4 g4 = *(const f4 *)(sG + goff[nt] + 4 * q);
            const f4 p4 = *(const f4 *)(sP + max(poff[nt], 0) + 4 * q);
            bop[nt] = poff[nt] >= 0 ? g4 * p4 : g4;
        }
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[q][e], bop[nt][e], acc[nt], 0, 0, 0);
        if (q == 0) PSI_SKA_MARK(1);
        __builtin_amdgcn_sched_barrier(0);                      // (the scheduler otherwise hoists every quad's LDS reads to the top and spills)
    }
    // D[row = 4 lk + e -> joint 16 w + 4 lk + e][col = li -> column 16 nt + li]
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int c = 16 * nt + li;
        if (c < ncol) {
            const int bb = c / 12, rs = c - 12 * bb;
            float *po = o.part + (((size_t)(b0 + bb)) * PSI_JP + w * 16 + lk * 4) * 16 + rs;
#pragma unroll
            for (int e = 0; e < 4; e++) po[e * 16] = acc[nt][e];
        }
    }
}

// the same, column-tile count chosen at run time (12 nbody columns)
__device__ __forceinline__ void skin_bwd_A_dispatch(const PsiSkaSlice &o, int B, int b0, int nbody, psi_f4 *smem)
{
    switch ((12 * nbody + 15) >> 4) {
    case 1: skin_bwd_A_body<1>(o, B, b0, nbody, smem); break;
    case 2: skin_bwd_A_body<2>(o, B, b0, nbody, smem); break;
    case 3: skin_bwd_A_body<3>(o, B, b0, nbody, smem); break;
    case 4: skin_bwd_A_body<4>(o, B, b0, nbody, smem); break;
    case 5: skin_bwd_A_body<5>(o, B, b0, nbody, smem); break;
    default: skin_bwd_A_body<SKA_MAXT>(o, B, b0, nbody, smem); break;
    }
}

// ------------------------------------------------------------------------------------------------
// blend backward (MFMA): g_feat[b][k] = sum_n g_vp[b][n] dirs[k][n]
// Rounds 3-6 ran this product on the fp32 MFMA (v_mfma_f32_16x16x4_f32 over a 16-column-tiled fp32 copy of the matrix; workgroup = 4 waves
// sharing a 64-row k group and an n-slice, wave w taking n-steps w, w + 4, ..., LDS reduce).  What was learnt on that form and still holds for
// the one below: ONE step of operands in flight per wave serves the launch best (with the skin_bwd_A waves sharing the SIMDs: 22.0 us at 1
// step, 22.6 / 23.9 / 25.1 at 2 / 3 / 4), and the step loop as straight-line code with a register ring does not pay
// (profiles/r06_ab_bwd_joint_pipeline.txt).
// The product on the fp16 matrix pipe at fp32-class accuracy (round 6; what lbs.hip's blend_fwd_h_kernel does for the forward product):
// the fp32 MFMA runs at 1/16 of the fp16 rate, and the stream workgroups' 608 fp32 MFMAs per SIMD were what this launch waited for beside its
// skin_bwd_A waves (with a quarter of them: 25.3 -> 21.2 us, profiles/r06_ab_blend_fp16x3.txt).  The matrix arrives as TWO fp16 parts per
// entry (hi = fp16(x), lo = fp16((x - hi) 2^11), x = value * a power of two: 22 mantissa bits in the same 4 bytes) in MFMA operand order
//   dirs_bh [n-step of 16][Kpad/32 k-tiles][part][n half][32 k][8 n]        (a wave's load of one part of one (step, k-tile) is 1 KB contiguous);
// the gradient rows stay fp32 in memory and are split by the wave that loads them, with the scale 2^s that puts the class's largest entry
// (an integer atomicMax of its bit pattern by the rows' producers: exact, order-independent) into [2^13, 2^14) — fp16 cannot overflow,
// entries down to 1e-9 of the largest keep 22 bits, smaller ones an absolute error below 1e-12 of the largest.  A product is
// hi*hi + (hi*lo + lo*hi) 2^-11 in two fp32 accumulators: 3 MFMAs of 32 cycles per 32 x 32 x 16 block instead of 16.
// Workgroup = 4 waves sharing a 64-row k group (two 32-row k-tiles) and an n-slice; wave w takes n-steps w, w + 4, ...; LDS reduce.
// ------------------------------------------------------------------------------------------------
typedef _Float16 psi_h8 __attribute__((ext_vector_type(8)));
typedef float psi_f16v __attribute__((ext_vector_type(16)));
typedef unsigned int psi_u4 __attribute__((ext_vector_type(4)));
struct PsiBlendBwdColsH {
    const float *dirs_bh;      // the class's columns, two fp16 parts per entry (layout above)
    const float *g_vp;         // [B][row_stride] fp32
    size_t row_stride;
    int Kpad, total_steps;
    float g_scale;             // 2^s for the rows' fp16 parts
    float unscale;             // 1 / (matrix scale * g_scale)
};
// scale for a class whose largest |entry| has the bit pattern `bits` (0: nothing stored this iteration)
__device__ __forceinline__ float psi_fp16_class_scale(unsigned bits)
{
    int E = (int)((bits >> 23) & 0xffu);                 // largest entry in [2^(E-127), 2^(E-126))
    if (bits == 0u) E = 140;                             // (any scale: every entry is zero)
    int se = 267 - E;                                    // biased exponent of 2^(13 - (E - 127))
    se = se < 27 ? 27 : (se > 227 ? 227 : se);           // |exponent| <= 100: the reciprocal stays a normal number
    return __uint_as_float((unsigned)se << 23);
}
template <int MTB>
constexpr int psi_blend_bwd_h_smem_f4() { return 4 * 2 * MTB * 4 * 64; }

template <int MTB>
__device__ __forceinline__ void blend_bwd_h_body(const PsiBlendBwdColsH &o, int B, int s_begin, int s_end, float *__restrict__ part, int kgroup,
                                                 int bgroup, psi_f4 *smem)
{
    typedef psi_f4 f4;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int KT = o.Kpad / 32;
    const int k0 = kgroup * 64;
    const int b0 = bgroup * 32 * MTB;
    s_end = min(s_end, o.total_steps);
    psi_f16v acc[2][MTB][2];                             // [k-tile][body tile][hi*hi | hi*lo + lo*hi]
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int t = 0; t < MTB; t++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int i = 0; i < 16; i++) acc[kt][t][c][i] = 0.0f;
    const float *grow[MTB];
#pragma unroll
    for (int t = 0; t < MTB; t++) grow[t] = o.g_vp + (size_t)min(b0 + t * 32 + li, B - 1) * o.row_stride + 8 * kh;
    const char *dbase = (const char *)o.dirs_bh + ((size_t)(2 * kgroup) * 2) * 1024 + (size_t)(kh * 32 + li) * 16;
    const float gsc = o.g_scale;
    // PF steps' operands (PF x (4 + 2 MTB) 16-byte loads per lane) are requested before the first of them is split and multiplied
#ifndef PSI_BWH_PF
#define PSI_BWH_PF 1
#endif
    constexpr int PF = PSI_BWH_PF;
    for (int st0 = s_begin + w; st0 < s_end; st0 += 4 * PF) {
        psi_u4 d[PF][2][2];
        f4 g[PF][MTB][2];
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int st = min(st0 + 4 * p, o.total_steps - 1);
#pragma unroll
            for (int kt = 0; kt < 2; kt++)
#pragma unroll
                for (int q = 0; q < 2; q++) d[p][kt][q] = *(const psi_u4 *)(dbase + ((size_t)st * KT * 2 + (size_t)(kt * 2 + q)) * 1024);
#pragma unroll
            for (int t = 0; t < MTB; t++) {
                g[p][t][0] = *(const f4 *)(grow[t] + (size_t)st * 16);
                g[p][t][1] = *(const f4 *)(grow[t] + (size_t)st * 16 + 4);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < PF; p++) {
            if (st0 + 4 * p < s_end) {
#pragma unroll
                for (int t = 0; t < MTB; t++) {
                    psi_h8 gh, gl;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const float x = g[p][t][e >> 2][e & 3] * gsc;
                        const _Float16 hi = (_Float16)x;
                        gh[e] = hi;
                        gl[e] = (_Float16)((x - (float)hi) * 2048.0f);
                    }
#pragma unroll
                    for (int kt = 0; kt < 2; kt++) {
                        const psi_h8 dh = __builtin_bit_cast(psi_h8, d[p][kt][0]), dl = __builtin_bit_cast(psi_h8, d[p][kt][1]);
                        acc[kt][t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, gh, acc[kt][t][0], 0, 0, 0);
                        acc[kt][t][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, gl, acc[kt][t][1], 0, 0, 0);
                        acc[kt][t][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, gh, acc[kt][t][1], 0, 0, 0);
                    }
                }
            }
        }
    }
    // D[row = 8 q + 4 kh + e -> k][col = li -> body]: the two accumulators combined, the four waves' sums through LDS, wave w finishes quad q = w
    f4 (*red)[2][MTB][4][64] = (f4 (*)[2][MTB][4][64])smem;      // [wave][k-tile][body tile][row quad][lane]
    const float us = o.unscale, us2 = o.unscale * (1.0f / 2048.0f);
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int t = 0; t < MTB; t++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f4 v;
#pragma unroll
                for (int e = 0; e < 4; e++) v[e] = __builtin_fmaf(acc[kt][t][1][4 * q + e], us2, acc[kt][t][0][4 * q + e] * us);
                red[w][kt][t][q][lane] = v;
            }
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
        for (int t = 0; t < MTB; t++) {
            const f4 ov = red[0][kt][t][w][lane] + red[1][kt][t][w][lane] + red[2][kt][t][w][lane] + red[3][kt][t][w][lane];
            const int b = b0 + t * 32 + li;
            if (b < B) *(f4 *)(part + (size_t)b * o.Kpad + k0 + kt * 32 + 8 * w + 4 * kh) = ov;
        }
}

// Placement of the stream workgroups: workgroup `bid` runs on XCD bid % 8 (observed dispatch order) and every XCD has its own L2, so the
// k-groups that share an n-slice — and therefore read the same g_vposed columns — are placed on ONE XCD: that slice of g_vposed is
// fetched from memory once and served to the other k-groups from L2 (it used to be fetched by all 8 XCDs).
__device__ __forceinline__ void psi_blend_bwd_place(int bid, int kgroups, int nslices, int &kg, int &slice, int &bg)
{
    if ((nslices & 7) == 0) {
        const int xcd = bid & 7, idx = bid >> 3, spx = nslices >> 3;
        kg = idx % kgroups;
        const int t = idx / kgroups;
        slice = xcd + 8 * (t % spx);
        bg = t / spx;
    } else {
        kg = bid % kgroups;
        const int rest = bid / kgroups;
        slice = rest % nslices;
        bg = rest / nslices;
    }
}
