// SMPL-X linear blend skinning (forward + hand-derived backward) for gfx950.
// Replaces the body-model call of the reference,
//   smplx.create(...)(return_verts=True, body_pose, transl, global_orient, betas, hands)   fitting_proxe.py:55-69,125-128
// whose arithmetic is `lbs` (human_body_prior/body_model/lbs.py:34-118 == smplx 0.1.13 lbs), followed by
// `+ transl` and PSI's verts_transform (cvae.py:141-149).
//
// Data layout in HBM (built once by psi_lbs_create):
//   dirs     the [Kpad][Npad] blend-shape matrix — rows 0..NB-1 = shapedirs^T, NB..NB+P-1 = posedirs (zero padded; N = 3V); shape and pose
//            blendshapes are ONE contraction: v_posed = v_template + feat @ dirs, feat[b] = [betas | (R_1..R_{J-1} - I)] (lbs.py:81 and :94-99
//            fused) — every entry as TWO fp16 parts (hi, lo: 22 mantissa bits in the fp32 value's 4 bytes) in MFMA operand order
//            [Npad/32 column tiles][Kpad/16 k-steps][part][k half][32 columns][8 k]: a wave's 32-column strip is one contiguous 64 KB run
//   dirs_bh  the same parts in the backward product's operand order [Npad/16 n-steps][Kpad/32 k-tiles][part][n half][32 k][8 n]
//   WT    [64][Vpad]     skinning weights transposed (coalesced per-vertex reads), zero padded.
//   WTt   [Vpad/64][64][64]  the same weights tiled per wave: [vertex tile][joint][vertex in tile] — the dense skinning blend of a wave
//                        walks ONE contiguous 16 KB tile, joint after joint (lbs_device.h)
//   J_t [J][3], J_s [J][3][NB]   joint regressor pre-contracted with v_template / shapedirs (fp64 on the host):
//                        J = J_t + J_s @ betas, algebraically lbs.py:85 without the V-long reduction per call.
// Kernels:
//   pose_fwd   (1 wave per body)   Rodrigues (lbs.py:165-192), joints, level-parallel kinematic chain (lbs.py:207-262)   [lbs_device.h]
//   blend_fwd  (MFMA f32 16x16x4)  v_posed = v_template + feat @ dirs       — a 64 MB stream + 6.5 us of MFMA
//   skin_fwd                       verts = cam_ext * (sum_j W_j A_j [v_posed;1] + transl)  (lbs.py:108-116, cvae.py:141-149) [lbs_device.h]
//   skin_bwd_v                     g_local = R_c^T g_verts;  g_vposed = T_R^T g_local;  partial g_transl                    [lbs_device.h]
//   bwd_joint  (MFMA)              one grid, two kinds of workgroup:
//       skin_bwd_A                 gA[b][j] = sum_v W[v][j] * g_local (x) [v_posed;1]   (contraction over V)
//       blend_bwd                  g_feat = g_vposed @ dirs^T                (contraction over N, dirs streamed again)
//   reduce_partials                sums the 41 vertex-slice and 32 column-slice partials
//   pose_bwd   (1 wave per body)   chain reverse sweep, Rodrigues derivative, joint/shape gradients                          [lbs_device.h]
// The per-body pose stages and the skinning kernels live in lbs_device.h because the fused fitting engine (fit.hip) inlines /
// re-instantiates them with its own hooks.
// The f32 MFMA (v_mfma_f32_16x16x4_f32) is bit-identical to an fmaf chain, so these are exact-f32 GEMMs.
#include "psi_internal.h"
#include "lbs_device.h"
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <vector>
#include <string.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int JP = PSI_JP;      // padded joint count (one wave)
constexpr int SKIN_BLK = PSI_SKIN_BLK;

}  // namespace

struct psi_lbs_model {
    LbsDev d;
    void *blob;   // single device allocation holding everything above
};

namespace {

// ------------------------------------------------------------------------------------------------
// workspace layout (floats)
// ------------------------------------------------------------------------------------------------
constexpr int GV_SLOTS = 1024;
struct WsLayout {
    size_t feat, R, Jl, G, A, v_posed, gl, g_vp, gA_part, gfeat_part, gt_part, gA, gfeat, gvbits, total;
    int nsv, nsn, nvb;
};

WsLayout ws_layout(const LbsDev &m, int B)
{
    WsLayout w;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) & ~(size_t)63; return r; };
    w.nsv = m.Vpad / 256;                 // v-slices of 256 vertices for skin_bwd_A
    // blend_bwd: (Kpad/64) k-groups x nsn n-slices ~= 256 workgroups = one per CU (one wave per SIMD, single round)
    w.nsn = 256 / (m.Kpad / 64) > 0 ? 256 / (m.Kpad / 64) : 1;
    if (w.nsn > m.Npad / 16) w.nsn = m.Npad / 16;
    w.nvb = m.Vpad / SKIN_BLK;
    w.feat = take((size_t)((B + 31) & ~31) * m.Kpad);        // two fp16 parts per entry, MFMA operand order (psi_feat_store, lbs_device.h)
    w.R = take((size_t)B * m.J * 12);           // rows padded to 4 floats
    w.Jl = take((size_t)B * m.J * 3);
    w.G = take((size_t)B * m.J * 12);
    w.A = take((size_t)B * m.J * 12 + PSI_A_TAIL * 12);     // + zero rows behind the last body (lbs_device.h: psi_pose_fwd_chain)
    w.v_posed = take((size_t)B * m.Npad);
    w.gl = take((size_t)B * m.Npad);
    w.g_vp = take((size_t)B * m.Npad);
    w.gA_part = take((size_t)w.nsv * B * JP * 16);
    w.gfeat_part = take((size_t)w.nsn * B * m.Kpad);
    w.gt_part = take((size_t)w.nvb * B * 4);
    w.gA = take((size_t)B * JP * 16);
    w.gfeat = take((size_t)B * m.Kpad);
    w.gvbits = take(GV_SLOTS);                  // per-workgroup maxima of |g_vposed| (bit patterns), gv_rowmax_kernel -> bwd_joint_kernel
    w.total = o;
    return w;
}

// ------------------------------------------------------------------------------------------------
// pose forward: one wave per body
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void pose_fwd_kernel(LbsDev m, const float *__restrict__ betas, const float *__restrict__ pose,
                                                      const float *__restrict__ transl, int B, float *__restrict__ feat,
                                                      float *__restrict__ Rs, float *__restrict__ Jls, float *__restrict__ Gs,
                                                      float *__restrict__ As, float *__restrict__ joints)
{
    const int b = blockIdx.x;
    psi_pose_fwd_body(m, betas + (size_t)b * m.NB, pose + (size_t)b * m.J * 3, transl, B, b, feat, Rs, Jls, Gs, As, joints);
}

// ------------------------------------------------------------------------------------------------
// blend forward: v_posed[b][n] = v_template[n] + sum_k feat[b][k] dirs[k][n]      (v_mfma_f32_32x32x16_f16, three-term split products)
// Column decomposition: a WAVE owns 32 output columns and contracts ALL of K for them, so there is no cross-wave reduction at all — no
// LDS, no barriers (the K-split kernel of rounds 1-3 ended every tile with an LDS reduction and two barriers: 3.6 TB/s).  984 waves = one
// per SIMD on 246 CUs, one wave per SIMD by design (amdgpu_waves_per_eu(1,1)); latency is hidden inside the wave by a rolling register
// ring FOUR 64-k chunks deep.
//   Arithmetic (round 6): the fp32 MFMA runs at 1/16 of the fp16 / bf16 rate on gfx950 and kept this launch's matrix pipe busy for 6.6 of
// its 15 us — with a quarter of the MFMAs the same stream ran 2.2 us faster (profiles/r06_ab_blend_fp16x3.txt).  Both operands are therefore
// stored as TWO fp16 parts per fp32 value, hi = fp16(x), lo = fp16((x - hi) 2^11) with x = value * (a power of two) — 22 mantissa bits in
// the same 4 bytes — and a product is hi*hi + (hi*lo + lo*hi) 2^-11, accumulated in fp32 in two accumulators (the dropped lo*lo term and
// the parts' rounding are 2^-22 relative per product: the accuracy class of a plain fp32 product chain, 3 MFMAs of 32 cycles per 32 x 32 x 16
// block instead of 16).  The matrix is split once at psi_lbs_create, the feature rows by their producer (psi_feat_store, lbs_device.h).
//   Layouts: both operands in MFMA operand order, [32-row tile][k-step of 16][part][k half][32 rows][8 k] fp16: a wave's load of one part
// of one k-step is 1 KB contiguous.  D[row = column n][col = body]: a lane holds, for body (lane & 31), the four 4-column runs
// n = 8 g + 4 (lane >> 5) + (0..3): 16-byte stores.  The k order inside an accumulator is ascending for every (body, column) at every batch size.
// ------------------------------------------------------------------------------------------------
// NCH = Kpad / 64 at compile time (8 for SMPL-X): the chunk loop is then STRAIGHT-LINE code.  As a loop, hipcc's wait-count insertion is
// conservative at the loop header: the first chunk of every trip waited with vmcnt(0) — for ALL outstanding loads, including the chunk
// requested a few cycles earlier, i.e. the ring was drained and a full memory latency exposed once per trip (found in the ISA, round 6;
// in straight-line code the waits are exact).  NCH = 0: any Kpad, the loop form.
typedef _Float16 psi_h8 __attribute__((ext_vector_type(8)));
typedef float psi_f16v __attribute__((ext_vector_type(16)));
typedef unsigned int psi_u4 __attribute__((ext_vector_type(4)));
template <int MTB, int NCH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void blend_fwd_h_kernel(LbsDev m, const float *__restrict__ feat, int B,
                                                                                             float *__restrict__ v_posed)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (body groups of one tile quad on ONE XCD, so that the quad comes from HBM once and from L2 for the other groups, was measured SLOWER — 199 vs 181 us
    // at B = 512, 52.1 vs 49.4 at 128: at large batches the launch reads more feature rows (every wave all of them: 1 GB at B = 512) than matrix)
    const int tq = blockIdx.x, y = blockIdx.y;
    const int gw = tq * 4 + w;                           // 32-column tile of this wave
    if (gw >= m.Npad / 32) return;
    const int bt0 = y * MTB;                             // first 32-body tile
    const int li = lane & 31, kh = lane >> 5;
    const int KS = m.Kpad / 16;
    const int nch = NCH ? NCH : m.Kpad / 64;             // 64-k chunks = 4 k-steps (Kpad % 256 == 0)
    const char *dbase = (const char *)m.dirs + (size_t)gw * m.dirs_tile * 4 + (size_t)(kh * 32 + li) * 16;
    const char *fbase = (const char *)feat + (size_t)bt0 * KS * 2048 + (size_t)(kh * 32 + li) * 16;
    auto load_chunk = [&](int ch, psi_u4 (&d)[4][2], psi_u4 (&f)[MTB][4][2]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
            for (int p = 0; p < 2; p++) {
                d[s4][p] = *(const psi_u4 *)(dbase + (size_t)(ch * 4 + s4) * 2048 + p * 1024);
#pragma unroll
                for (int t = 0; t < MTB; t++) f[t][s4][p] = *(const psi_u4 *)(fbase + ((size_t)t * KS + ch * 4 + s4) * 2048 + p * 1024);
            }
    };
    psi_f16v acc[MTB][2];                                // [body tile][hi*hi | hi*lo + lo*hi]
#pragma unroll
    for (int t = 0; t < MTB; t++)
#pragma unroll
        for (int c = 0; c < 2; c++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[t][c][i] = 0.0f;
    auto mfma_chunk = [&](const psi_u4 (&d)[4][2], const psi_u4 (&f)[MTB][4][2]) {
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
            for (int t = 0; t < MTB; t++) {
                const psi_h8 dh = __builtin_bit_cast(psi_h8, d[s4][0]), dl = __builtin_bit_cast(psi_h8, d[s4][1]);
                const psi_h8 fh = __builtin_bit_cast(psi_h8, f[t][s4][0]), fl = __builtin_bit_cast(psi_h8, f[t][s4][1]);
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, fh, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dh, fl, acc[t][1], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(dl, fh, acc[t][1], 0, 0, 0);
            }
    };
    if (NCH) {
        psi_u4 d[4][4][2], f[4][MTB][4][2];
#pragma unroll
        for (int ch = 0; ch < 3 && ch < NCH; ch++) {
            load_chunk(ch, d[ch], f[ch]);
            __builtin_amdgcn_sched_barrier(0);           // (request order = use order)
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ch++) {
            if (ch + 3 < NCH) load_chunk(ch + 3, d[(ch + 3) & 3], f[(ch + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(d[ch & 3], f[ch & 3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        psi_u4 d0[4][2], d1[4][2], f0[MTB][4][2], f1[MTB][4][2];
        load_chunk(0, d0, f0);
        for (int ch = 0; ch < nch; ch += 2) {
            if (ch + 1 < nch) load_chunk(ch + 1, d1, f1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_chunk(d0, f0);
            if (ch + 1 < nch) {
                if (ch + 2 < nch) load_chunk(ch + 2, d0, f0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_chunk(d1, f1);
            }
        }
    }
    // D[row = 8 g + 4 kh + e -> column][col = li -> body]: four 16-byte stores per body tile
    const float us = m.dirs_unscale, us2 = m.dirs_unscale * (1.0f / 2048.0f);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int n = gw * 32 + 8 * g + 4 * kh;
        const f4 vt = *(const f4 *)(m.v_template + n);
#pragma unroll
        for (int t = 0; t < MTB; t++) {
            const int b = (bt0 + t) * 32 + li;
            f4 o;
#pragma unroll
            for (int e = 0; e < 4; e++) o[e] = vt[e] + __builtin_fmaf(acc[t][1][4 * g + e], us2, acc[t][0][4 * g + e] * us);
            if (b < B) *(f4 *)(v_posed + (size_t)b * m.Npad + n) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// skinning backward, joint part + blend backward (both MFMA): the bodies live in lbs_joint_device.h (the fused fitting engine runs them
// in a grid of its own, over the model's vertices AND its contact slots)
// ------------------------------------------------------------------------------------------------
#ifdef PSI_HEAD_STOPS
__device__ unsigned long long psi_dbg_ska_mark[2 * 2048];     // dev: skin_bwd_A workgroup — operands staged, first quad done
#define PSI_SKA_MARK(k) do { if (threadIdx.x == 0 && blockIdx.x < 2048) psi_dbg_ska_mark[2 * blockIdx.x + (k)] = wall_clock64(); } while (0)
extern "C" int psi_dbg_ska_marks(unsigned long long *out, int n) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psi_dbg_ska_mark), sizeof(unsigned long long) * 2 * (size_t)(n < 2048 ? n : 2048)); }
#endif
}  // namespace
#include "lbs_joint_device.h"
namespace {

// The two joint-side halves of the LBS backward in ONE launch.  Both depend only on skin_bwd_v; blend_bwd is a 64.5 MB
// stream with 256 long-lived workgroups (one per CU, one wave per SIMD), skin_bwd_A is a short L2-latency-bound contraction.
// Block ids [0, n_blend) are the stream workgroups, so every CU picks one up at once; the skin_bwd_A workgroups behind them
// (Vpad/256 x ceil(B/8): 164 at B = 32) are all resident at the same time and finish inside the stream's shadow.
// XCD-aware mapping of the stream part: workgroup `bid` runs on XCD bid % 8 (observed dispatch order) and every XCD has its
// own L2, so the k-groups that share an n-slice — and therefore read the same g_vposed columns — are placed on ONE XCD: that
// slice of g_vposed is fetched from memory once and served to the other k-groups from L2 (it used to be fetched by all 8 XCDs).
#ifdef PSI_HEAD_STOPS
// dev: workgroup timeline of the last bwd_joint launch ({start, end} in 10 ns ticks, hardware id, kind), tools/timeline.py
__device__ unsigned long long psi_dbg_tl2[4 * 2048];
struct PsiBlockTrace2 {
    unsigned long long t0;
    int kind;
    __device__ PsiBlockTrace2() : t0(wall_clock64()), kind(0) {}
    __device__ ~PsiBlockTrace2()
    {
        if (threadIdx.x != 0 || blockIdx.x >= 2048) return;
        unsigned long long *o = psi_dbg_tl2 + 4 * (size_t)blockIdx.x;
        o[0] = t0;
        o[1] = wall_clock64();
        o[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        o[3] = (unsigned long long)kind;
    }
};
extern "C" int psi_dbg_timeline2(unsigned long long *out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(psi_dbg_tl2), sizeof(unsigned long long) * 4 * (size_t)(nblocks < 2048 ? nblocks : 2048));
}
#endif
// The largest |entry| of g_vposed [B][Npad] as GV_SLOTS per-workgroup maxima (bit patterns; plain stores: nothing to reset, nothing atomic) — the
// scale of the rows' fp16 parts in blend_bwd_h_body (lbs_joint_device.h).  The fused fitting engine gets the same number from the kernels
// that WRITE the rows (fit.hip); here the rows come from psi_skin_bwd_v kernels of several kinds, and one more pass over rows that were
// written a moment ago costs 2-15 us against the 5-90 us the fp16 matrix pipe saves on the launch behind it.
__global__ __launch_bounds__(256) void gv_rowmax_kernel(const float *__restrict__ g, size_t n4, unsigned *__restrict__ slots)
{
    __shared__ float red[4];
    float mx = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)GV_SLOTS * 256) {
        const f4 v = ((const f4 *)g)[i];
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) slots[blockIdx.x] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

template <int MT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_joint_kernel(
    LbsDev m, const float *__restrict__ g_vp, const float *__restrict__ gl, const float *__restrict__ v_posed, int B, int steps_per_slice,
    float *__restrict__ gfeat_part, float *__restrict__ gA_part, int n_blend, int kgroups, int nslices, int nsv, int nbody,
    const unsigned *__restrict__ gvbits)
{
    // blend_bwd: [4][2][MTB][4][64] f4; skin_bwd_A: SKA_NBODY bodies' staged operands (2 x 3 component rows each)
    constexpr int SMEM_B = psi_blend_bwd_h_smem_f4<(MT + 1) / 2>();
    __shared__ f4 smem[SMEM_B > SKA_SMEM_F4 ? SMEM_B : SKA_SMEM_F4];
    // the skin_bwd_A workgroups come FIRST in the grid: a CU serves its workgroups' loads in order, and behind the 72 KB each stream wave
    // requests at once the 100 KB of a skin_bwd_A workgroup arrived after 9.6 us (workgroup timeline) — in front of it they are short
    const int n_ska = (int)gridDim.x - n_blend;
    const int bid = (int)blockIdx.x >= n_ska ? (int)blockIdx.x - n_ska : (int)blockIdx.x + n_blend;
#ifdef PSI_HEAD_STOPS
    PsiBlockTrace2 trace;
    trace.kind = bid < n_blend;
#endif
    if (bid < n_blend) {
        int kg, slice, bg;
        psi_blend_bwd_place(bid, kgroups, nslices, kg, slice, bg);
        // the rows' fp16 scale: the largest |g_vposed| entry of this backward (gv_rowmax_kernel left one maximum per workgroup)
        unsigned cbits = 0u;
#pragma unroll
        for (int q = 0; q < GV_SLOTS / 64; q++) cbits = max(cbits, gvbits[q * 64 + (threadIdx.x & 63)]);
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) cbits = max(cbits, (unsigned)__shfl_xor((int)cbits, o2, 64));
        const float gsc = psi_fp16_class_scale(cbits);
        const PsiBlendBwdColsH cols = {m.dirs_bh, g_vp, (size_t)m.Npad, m.Kpad, m.Npad / 16, gsc, m.dirs_unscale * PSI_FEAT_SCALE / gsc};
        blend_bwd_h_body<(MT + 1) / 2>(cols, B, slice * steps_per_slice, (slice + 1) * steps_per_slice, gfeat_part + (size_t)slice * B * m.Kpad, kg, bg, smem);
    } else {
        const int i = bid - n_blend, vslice = i % nsv;
        // (weights from the wave-tiled copy — the copy skin_bwd_v streamed just before this launch, so most of it is still in L2 / MALL)
        const PsiSkaSlice sl = {m.WTt + (size_t)vslice * 4 * PSI_JP * 64, gl + (size_t)vslice * 768, v_posed + (size_t)vslice * 768, (size_t)m.Npad,
                                gA_part + (size_t)vslice * B * JP * 16};
        skin_bwd_A_dispatch(sl, B, (i / nsv) * nbody, nbody, smem);
    }
}

// ------------------------------------------------------------------------------------------------
// pose backward: one wave per body
// ------------------------------------------------------------------------------------------------
// sum the split-contraction partials: gA[b][j][16] over v-slices, g_feat[b][k] over n-slices, g_transl over vertex blocks
// One output per thread, ALL of its slices requested before the first add (the partials were just written by other XCDs'
// workgroups: every dependent round of loads is a ~2 us trip to memory; measured 10 us with 4 loads in flight, and 15 us
// when 32 workgroups did the whole 7.5 MB themselves).
// (sum_slices_split / PSI_RSPL: lbs_device.h — the fused tail + head kernel of the fitting engine sums the same partials the same way)
constexpr int RSPL = PSI_RSPL;
__global__ __launch_bounds__(256) void reduce_partials_kernel(int B, int Kpad, const float *__restrict__ gA_part, int nsv,
                                                              const float *__restrict__ gfeat_part, int nsn,
                                                              const float *__restrict__ gt_part, int nvb,
                                                              float *__restrict__ gA, float *__restrict__ gfeat,
                                                              float *__restrict__ g_transl)
{
    const long nA = (long)B * JP * 16, nF = (long)B * Kpad, nT = (long)B * 4;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = t / RSPL;
    const int s0 = (int)(t % RSPL);
    if (i < nA) {
        // (entries 12..15 of a joint's 16 do not exist: skin_bwd_A writes the 3 x 4 gradient only, and nothing reads them)
        const float r = (i & 15) < 12 ? psi_sum_slices_split<12>(gA_part + i, (size_t)nA, nsv, s0) : 0.0f;
        if (s0 == 0) gA[i] = r;
    } else if (i < nA + nF) {
        const long k = i - nA;
        const float r = psi_sum_slices_split<8>(gfeat_part + k, (size_t)nF, nsn, s0);
        if (s0 == 0) gfeat[k] = r;
    } else if (i < nA + nF + nT) {
        const long k = i - nA - nF;
        const int b = (int)(k >> 2), c = (int)(k & 3);
        const float r = psi_sum_slices_split<12>(gt_part + (size_t)b * 4 + (c < 3 ? c : 0), (size_t)B * 4, nvb, s0);
        if (s0 == 0 && c < 3 && g_transl) g_transl[(size_t)b * 3 + c] = r;
    }
}

__global__ __launch_bounds__(256) void pose_bwd_kernel(LbsDev m, const float *__restrict__ betas, const float *__restrict__ pose,
                                                      const float *__restrict__ Rs, const float *__restrict__ Jls,
                                                      const float *__restrict__ Gs, const float *__restrict__ gAr,
                                                      const float *__restrict__ gfeat, int B,
                                                      float *__restrict__ g_betas, float *__restrict__ g_pose, float *__restrict__ g_rot)
{
    const int b = blockIdx.x;
    psi_pose_bwd_body(m, pose + (size_t)b * m.J * 3, Rs, Jls, Gs, gAr + (size_t)b * JP * 16, gfeat + (size_t)b * m.Kpad, b,
                      g_betas ? g_betas + (size_t)b * m.NB : nullptr, g_pose ? g_pose + (size_t)b * m.J * 3 : nullptr,
                      g_rot ? g_rot + (size_t)b * m.J * 9 : nullptr);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host API
// ------------------------------------------------------------------------------------------------
extern "C" int psi_lbs_create(psi_lbs_model **out, const float *h_v_template, const float *h_shapedirs,
                              const float *h_posedirs, const float *h_J_regressor, const float *h_weights,
                              const int32_t *h_parents, int V, int J, int NB)
{
    PSI_REQUIRE(out && h_v_template && h_shapedirs && h_posedirs && h_J_regressor && h_weights && h_parents, "null pointer");
    PSI_REQUIRE(V > 0 && J > 0 && J <= JP && NB >= 0, "unsupported model size (J <= 64)");
    LbsDev d;
    memset(&d, 0, sizeof(d));
    d.V = V; d.J = J; d.NB = NB; d.P = (J - 1) * 9; d.K = NB + d.P;
    d.Kpad = (d.K + 255) / 256 * 256;
    d.N = 3 * V;
    d.Vpad = (V + 255) / 256 * 256;
    d.Npad = 3 * d.Vpad;
    std::vector<int> level(J, 0), cptr(J + 1, 0), cidx;
    d.maxlevel = 0;
    for (int j = 0; j < J; j++) {
        int p = h_parents[j];
        PSI_REQUIRE(p < j, "parents must precede children (kintree order)");
        level[j] = p < 0 ? 0 : level[p] + 1;
        if (level[j] > d.maxlevel) d.maxlevel = level[j];
    }
    for (int j = 0; j < J; j++) {
        cptr[j] = (int)cidx.size();
        for (int c = j + 1; c < J; c++)
            if (h_parents[c] == j) cidx.push_back(c);
    }
    cptr[J] = (int)cidx.size();
    if (cidx.empty()) cidx.push_back(0);
    // pointer-jumping table (2^r-th ancestors) and subtree sets (lbs_device.h: psi_pose_fwd_chain / psi_pose_bwd_body)
    std::vector<int> jump((size_t)PSI_NJUMP * JP, -1);
    std::vector<unsigned char> sub_list, sub_first(J + 1, 0);
    std::vector<unsigned int> sub_item;
    for (int j = 0; j < J; j++) jump[j] = h_parents[j] < 0 ? -1 : h_parents[j];
    d.njump = 0;
    while ((1 << d.njump) <= d.maxlevel) d.njump++;              // rounds needed: 2^njump > maxlevel
    for (int r = 1; r < PSI_NJUMP; r++)
        for (int j = 0; j < J; j++) {
            const int a = jump[(size_t)(r - 1) * JP + j];
            jump[(size_t)r * JP + j] = a < 0 ? -1 : jump[(size_t)(r - 1) * JP + a];
        }
    {
        std::vector<std::vector<int>> sub(J);
        size_t total = 0;
        for (int j = 0; j < J; j++)
            for (int a = j; a >= 0; a = h_parents[a]) { sub[a].push_back(j); total++; }          // ascending in j
        const int chunk = std::max(8, (int)((total + JP - 1) / JP));                                // => at most J + JP chunks
        for (int j = 0; j < J; j++) {
            sub_first[j] = (unsigned char)sub_item.size();
            for (size_t q = 0; q < sub[j].size(); q += chunk) {
                const size_t cnt = std::min((size_t)chunk, sub[j].size() - q);
                sub_item.push_back((unsigned int)sub_list.size() | ((unsigned int)cnt << 16));
                for (size_t i = 0; i < cnt; i++) sub_list.push_back((unsigned char)sub[j][q + i]);
            }
        }
        sub_first[J] = (unsigned char)sub_item.size();
        PSI_REQUIRE(sub_item.size() <= (size_t)PSI_ITEM_MAX && sub_list.size() <= (size_t)PSI_SUB_MAX, "kinematic tree tables out of range");
        d.n_sub = (int)sub_list.size();
        d.n_items = (int)sub_item.size();
    }
    // host staging
    // tile stride of the forward copy: a tile's 32 x Kpad floats + a skew (PSI_DIRS_SKEW floats, default 1088 = 4352 bytes), so that the
    // 984 waves that walk their tiles at the same pace do not sit on the same memory channels (tiles exactly 64 KB apart do)
    {
        const char *sk = getenv("PSI_DIRS_SKEW");
        d.dirs_tile = d.Kpad * 32 + (sk ? atoi(sk) / 4 * 4 : 1088);
    }
    std::vector<float> dirs_bh;
    std::vector<float> dense((size_t)d.Kpad * d.Npad, 0.0f), dirs((size_t)(d.Npad / 32) * d.dirs_tile, 0.0f), vt(d.Npad, 0.0f),
        WT((size_t)JP * d.Vpad, 0.0f);
    auto dirs_at = [&](int k, int n) -> float & { return dense[(size_t)k * d.Npad + n]; };
    for (int l = 0; l < NB; l++)
        for (int n = 0; n < d.N; n++) dirs_at(l, n) = h_shapedirs[(size_t)n * NB + l];                  // [V,3,NB] -> row l
    for (int p = 0; p < d.P; p++)
        for (int n = 0; n < d.N; n++) dirs_at(NB + p, n) = h_posedirs[(size_t)p * d.N + n];
    // forward copy (blend_fwd_h_kernel): every entry as TWO fp16 parts of x = v * scale — hi = fp16(x), lo = fp16((x - hi) * 2^11): 22 mantissa
    // bits in the same 4 bytes — in MFMA operand order [32-column tile][Kpad/16 k-steps][part][k half][32 columns][8 k].  scale = the power of
    // two that puts the largest entry into [2^13, 2^14): far from fp16's overflow, and entries down to 1e-9 of the largest stay normal numbers.
    {
        float amax = 0.0f;
        for (float v : dense) amax = std::max(amax, std::fabs(v));
        int e = 0;
        if (amax > 0.0f) (void)std::frexp(amax, &e);             // amax = f * 2^e, f in [0.5, 1)
        const float scale = std::ldexp(1.0f, 14 - e);
        d.dirs_unscale = 1.0f / (scale * PSI_FEAT_SCALE);
        _Float16 *dh = reinterpret_cast<_Float16 *>(dirs.data());
        for (int k = 0; k < d.Kpad; k++)
            for (int n = 0; n < d.Npad; n++) {
                const float x = dirs_at(k, n) * scale;
                const _Float16 hi = (_Float16)x;
                const _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
                const size_t o = (size_t)(n >> 5) * d.dirs_tile * 2 + ((size_t)(k >> 4) * 4 + ((k >> 3) & 1)) * 256 + (size_t)(n & 31) * 8 + (k & 7);
                dh[o] = hi;
                dh[o + 512] = lo;
            }
        // ... and the same parts in the backward product's operand order, [n-step of 16][Kpad/32][part][n half][32 k][8 n] (blend_bwd_h_body)
        dirs_bh.assign((size_t)d.Kpad * d.Npad, 0.0f);
        _Float16 *bh = reinterpret_cast<_Float16 *>(dirs_bh.data());
        const int KT = d.Kpad / 32;
        for (int k = 0; k < d.Kpad; k++)
            for (int n = 0; n < d.Npad; n++) {
                const float x = dirs_at(k, n) * scale;
                const _Float16 hi = (_Float16)x;
                const _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
                const size_t o = ((((size_t)(n >> 4) * KT + (k >> 5)) * 2) * 2 + ((n >> 3) & 1)) * 256 + (size_t)(k & 31) * 8 + (n & 7);
                bh[o] = hi;
                bh[o + 512] = lo;
            }
    }
    memcpy(vt.data(), h_v_template, sizeof(float) * d.N);
    std::vector<float> WTt((size_t)JP * d.Vpad, 0.0f);
    for (int v = 0; v < V; v++)
        for (int j = 0; j < J; j++) {
            WT[(size_t)j * d.Vpad + v] = h_weights[(size_t)v * J + j];
            WTt[((size_t)(v >> 6) * JP + j) * 64 + (v & 63)] = h_weights[(size_t)v * J + j];
        }
    std::vector<float> Jt((size_t)J * 3), Js((size_t)J * 3 * (NB > 0 ? NB : 1), 0.0f);
    {
        std::vector<double> acc((size_t)3 * (NB + 1));
        for (int j = 0; j < J; j++) {
            std::fill(acc.begin(), acc.end(), 0.0);
            const float *jr = h_J_regressor + (size_t)j * V;
            for (int v = 0; v < V; v++) {
                double wv = jr[v];
                if (wv == 0.0) continue;
                for (int c = 0; c < 3; c++) {
                    acc[c * (NB + 1) + NB] += wv * h_v_template[(size_t)v * 3 + c];
                    const float *sd = h_shapedirs + ((size_t)v * 3 + c) * NB;
                    for (int l = 0; l < NB; l++) acc[c * (NB + 1) + l] += wv * sd[l];
                }
            }
            for (int c = 0; c < 3; c++) {
                Jt[j * 3 + c] = (float)acc[c * (NB + 1) + NB];
                for (int l = 0; l < NB; l++) Js[((size_t)j * 3 + c) * NB + l] = (float)acc[c * (NB + 1) + l];
            }
        }
    }
    // compressed skinning rows: used by the skinning kernels when no vertex has more than PSI_WNZ non-zero weights
    // (PSI_LBS_DENSE=1 keeps the dense loop, for A/B tests)
    std::vector<float> Wc;
    std::vector<unsigned> Wj;
    {
        int nnz_max = 0;
        for (int v = 0; v < V; v++) {
            int c = 0;
            for (int j = 0; j < J; j++) c += h_weights[(size_t)v * J + j] != 0.0f;
            nnz_max = std::max(nnz_max, c);
        }
        const char *dense = getenv("PSI_LBS_DENSE");
        if (nnz_max <= PSI_WNZ && !(dense && dense[0] == '1')) {
            Wc.assign((size_t)PSI_WNZ * d.Vpad, 0.0f);
            Wj.assign((size_t)(PSI_WNZ / 4) * d.Vpad, 0u);
            for (int v = 0; v < V; v++) {
                int k = 0;
                for (int j = 0; j < J; j++) {
                    float w = h_weights[(size_t)v * J + j];
                    if (w != 0.0f) { Wc[(size_t)k * d.Vpad + v] = w; Wj[(size_t)(k >> 2) * d.Vpad + v] |= (unsigned)j << (8 * (k & 3)); k++; }
                }
            }
        }
    }
    // one device blob
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    size_t o_dirs = take(dirs.size() * 4), o_dirs_bh = take(dirs_bh.size() * 4), o_vt = take(vt.size() * 4), o_wt = take(WT.size() * 4), o_wtt = take(WTt.size() * 4), o_jt = take(Jt.size() * 4),
           o_js = take(Js.size() * 4), o_wc = take(Wc.size() * 4 + 4), o_wj = take(Wj.size() * 4 + 4), o_par = take(J * 4), o_lvl = take(J * 4), o_cp = take((J + 1) * 4), o_ci = take(cidx.size() * 4),
           o_jump = take(jump.size() * 4), o_sl = take(sub_list.size()), o_si = take(sub_item.size() * 4), o_sf = take(sub_first.size());
    char *blob = nullptr;
    PSI_CHECK_HIP(hipMalloc((void **)&blob, o));
    std::vector<int> par(h_parents, h_parents + J);
    struct { size_t off; const void *src; size_t bytes; } cp[] = {
        {o_dirs, dirs.data(), dirs.size() * 4}, {o_dirs_bh, dirs_bh.data(), dirs_bh.size() * 4}, {o_vt, vt.data(), vt.size() * 4}, {o_wt, WT.data(), WT.size() * 4}, {o_wtt, WTt.data(), WTt.size() * 4},
        {o_jt, Jt.data(), Jt.size() * 4}, {o_js, Js.data(), Js.size() * 4}, {o_wc, Wc.data(), Wc.size() * 4}, {o_wj, Wj.data(), Wj.size() * 4},
        {o_par, par.data(), (size_t)J * 4},
        {o_lvl, level.data(), (size_t)J * 4}, {o_cp, cptr.data(), (size_t)(J + 1) * 4}, {o_ci, cidx.data(), cidx.size() * 4},
        {o_jump, jump.data(), jump.size() * 4}, {o_sl, sub_list.data(), sub_list.size()}, {o_si, sub_item.data(), sub_item.size() * 4},
        {o_sf, sub_first.data(), sub_first.size()}};
    for (auto &c : cp) {
        if (!c.bytes) continue;
        hipError_t e = hipMemcpy(blob + c.off, c.src, c.bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(blob);
            psi_set_error("hipMemcpy failed: %s", hipGetErrorString(e));
            return (int)e;
        }
    }
    d.dirs = (const float *)(blob + o_dirs);
    d.dirs_bh = (const float *)(blob + o_dirs_bh);
    d.v_template = (const float *)(blob + o_vt);
    d.WT = (const float *)(blob + o_wt);
    d.WTt = (const float *)(blob + o_wtt);
    d.Wc = Wc.empty() ? nullptr : (const float *)(blob + o_wc);
    d.Wj = Wj.empty() ? nullptr : (const unsigned *)(blob + o_wj);
    d.J_t = (const float *)(blob + o_jt);
    d.J_s = (const float *)(blob + o_js);
    d.parents = (const int *)(blob + o_par);
    d.level = (const int *)(blob + o_lvl);
    d.child_ptr = (const int *)(blob + o_cp);
    d.child_idx = (const int *)(blob + o_ci);
    d.jump = (const int *)(blob + o_jump);
    d.sub_list = (const unsigned char *)(blob + o_sl);
    d.sub_item = (const unsigned int *)(blob + o_si);
    d.sub_first = (const unsigned char *)(blob + o_sf);
    psi_lbs_model *mdl = new psi_lbs_model;
    mdl->d = d;
    mdl->blob = blob;
    *out = mdl;
    return 0;
}

extern "C" void psi_lbs_destroy(psi_lbs_model *m)
{
    if (!m) return;
    (void)hipFree(m->blob);
    delete m;
}

extern "C" size_t psi_lbs_workspace_floats(const psi_lbs_model *m, int B)
{
    if (!m || B <= 0) return 0;
    return ws_layout(m->d, B).total;
}

static int lbs_launch_blend(const LbsDev &m, const WsLayout &L, int B, float *ws, hipStream_t st)
{
    const int bgroups = B > 32 ? psi_cdiv(B, 64) : 1;           // 32-body tiles per wave: 1, or 2 (every further group re-streams the matrix)
    const dim3 grid(psi_cdiv(m.Npad / 32, 4), bgroups);
    static const bool loop_form = getenv("PSI_BLEND_FWD_LOOP") && getenv("PSI_BLEND_FWD_LOOP")[0] == '1';      // dev A/B: the chunk loop as a loop
#define PSI_LAUNCH_BLEND(MT_, NCH_) hipLaunchKernelGGL((blend_fwd_h_kernel<MT_, NCH_>), grid, dim3(256), 0, st, m, ws + L.feat, B, ws + L.v_posed)
    const bool k8 = m.Kpad == 512 && !loop_form;              // SMPL-X: 506 feature rows -> 8 chunks
    if (B > 32) { if (k8) PSI_LAUNCH_BLEND(2, 8); else PSI_LAUNCH_BLEND(2, 0); }
    else { if (k8) PSI_LAUNCH_BLEND(1, 8); else PSI_LAUNCH_BLEND(1, 0); }
#undef PSI_LAUNCH_BLEND
    PSI_CHECK_LAUNCH("blend_fwd_kernel");
    psi_mark("blend_fwd_kernel", st);
    return 0;
}

static int lbs_launch_blend_skin(const LbsDev &m, const WsLayout &L, const float *transl, const float *cam_ext, int B, float *verts,
                                 float *ws, hipStream_t st)
{
    int rc = lbs_launch_blend(m, L, B, ws, st);
    if (rc) return rc;
    hipLaunchKernelGGL(psi_skin_fwd_kernel<PsiSkinNoEpilogue>, dim3(m.Vpad / SKIN_BLK, B), dim3(SKIN_BLK), 0, st, m, ws + L.A,
                       ws + L.v_posed, transl, cam_ext, B, verts, PsiSkinNoEpilogue());
    PSI_CHECK_LAUNCH("skin_fwd_kernel");
    psi_mark("skin_fwd_kernel", st);
    return 0;
}

extern "C" int psi_lbs_forward(const psi_lbs_model *mdl, const float *betas, const float *pose, const float *transl,
                               const float *cam_ext, int B, float *verts, float *joints, float *ws, void *stream)
{
    PSI_REQUIRE(mdl && betas && pose && verts && ws, "null pointer");
    PSI_REQUIRE(B > 0 && B <= 16384, "batch size out of range");
    const LbsDev &m = mdl->d;
    WsLayout L = ws_layout(m, B);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pose_fwd_kernel, dim3(B), dim3(64), 0, st, m, betas, pose, transl, B, ws + L.feat, ws + L.R, ws + L.Jl,
                       ws + L.G, ws + L.A, joints);
    PSI_CHECK_LAUNCH("pose_fwd_kernel");
    psi_mark("pose_fwd_kernel", st);
    return lbs_launch_blend_skin(m, L, transl, cam_ext, B, verts, ws, st);
}

// ---- entry points of the fused fitting engine (psi_internal.h): the pose stages run inside its own head / tail kernels
int psi_lbs_view(const psi_lbs_model *mdl, int B, float *ws, PsiLbsView *out)
{
    PSI_REQUIRE(mdl && ws && out && B > 0, "bad arguments");
    const LbsDev &m = mdl->d;
    WsLayout L = ws_layout(m, B);
    out->m = m;
    out->feat = ws + L.feat; out->R = ws + L.R; out->Jl = ws + L.Jl; out->G = ws + L.G; out->A = ws + L.A;
    out->v_posed = ws + L.v_posed; out->gl = ws + L.gl; out->g_vp = ws + L.g_vp; out->gt_part_w = ws + L.gt_part;
    out->gA_part = ws + L.gA_part; out->gfeat_part = ws + L.gfeat_part; out->gt_part = ws + L.gt_part;
    out->gA = ws + L.gA; out->gfeat = ws + L.gfeat;
    out->nsv = L.nsv; out->nsn = L.nsn; out->nvb = L.nvb;
    return 0;
}

int psi_lbs_blend_forward(const psi_lbs_model *mdl, int B, float *ws, hipStream_t st)
{
    const LbsDev &m = mdl->d;
    return lbs_launch_blend(m, ws_layout(m, B), B, ws, st);
}

static int lbs_launch_bwd_joint_parts(const LbsDev &m, const WsLayout &L, int B, float *ws, hipStream_t st);

static int lbs_launch_bwd_partials(const LbsDev &m, const WsLayout &L, const float *grad_verts, const float *cam_ext, int B, float *ws,
                                   hipStream_t st)
{
    if (B >= PSI_SKIN_MB_MIN_B)
        psi_launch_skin_bwd_v_mb(m, ws + L.A, PsiGradFromMemory{grad_verts, m.V}, cam_ext, B, ws + L.gl, ws + L.g_vp, ws + L.gt_part, st);
    else
        hipLaunchKernelGGL(psi_skin_bwd_v_kernel<PsiGradFromMemory>, dim3(m.Vpad / SKIN_BLK, B), dim3(SKIN_BLK), 0, st, m, ws + L.A,
                           PsiGradFromMemory{grad_verts, m.V}, cam_ext, B, ws + L.gl, ws + L.g_vp, ws + L.gt_part);
    PSI_CHECK_LAUNCH("skin_bwd_v_kernel");
    psi_mark("skin_bwd_v_kernel", st);
    return lbs_launch_bwd_joint_parts(m, L, B, ws, st);
}

static int lbs_launch_bwd_joint_parts(const LbsDev &m, const WsLayout &L, int B, float *ws, hipStream_t st)
{
    const int steps = psi_cdiv(m.Npad / 16, L.nsn);
    const int kgroups = m.Kpad / 64;
    const int mt = B > 32 ? 4 : (B > 16 ? 2 : 1);
    const int bgroups = psi_cdiv(B, 16 * mt);
    const int n_blend = kgroups * L.nsn * bgroups;
    // bodies per skin_bwd_A workgroup: about one such workgroup per CU beside its stream workgroup
    int nbody = psi_cdiv(B, 256 / L.nsv > 0 ? 256 / L.nsv : 1);
    if (nbody > SKA_NBODY) nbody = SKA_NBODY;
    if (const char *ev = getenv("PSI_SKA_NBODY")) { int v = atoi(ev); if (v >= 1 && v <= SKA_NBODY) nbody = v; }
    // (Round 4 measured the two halves as two launches of this kernel — stream workgroups, then the skin_bwd_A workgroups: 18.4 us each by
    // rocprofv3 = 36.8 against 27.2 for the heterogeneous grid, profiles/r04_ab_blend_loop.txt.  One grid it stays.)
    const int grid = n_blend + L.nsv * psi_cdiv(B, nbody);
    hipLaunchKernelGGL(gv_rowmax_kernel, dim3(GV_SLOTS), dim3(256), 0, st, ws + L.g_vp, (size_t)B * m.Npad / 4, (unsigned *)(ws + L.gvbits));
    PSI_CHECK_LAUNCH("gv_rowmax_kernel");
#define PSI_LAUNCH_JOINT(MT_)                                                                                                      \
    hipLaunchKernelGGL(bwd_joint_kernel<MT_>, dim3(grid), dim3(256), 0, st, m, ws + L.g_vp, ws + L.gl, ws + L.v_posed, B, steps,     \
                       ws + L.gfeat_part, ws + L.gA_part, n_blend, kgroups, L.nsn, L.nsv, nbody, (const unsigned *)(ws + L.gvbits))
    if (mt == 4) PSI_LAUNCH_JOINT(4);
    else if (mt == 2) PSI_LAUNCH_JOINT(2);
    else PSI_LAUNCH_JOINT(1);
#undef PSI_LAUNCH_JOINT
    PSI_CHECK_LAUNCH("bwd_joint_kernel");
    psi_mark("bwd_joint_kernel", st);
    return 0;
}

static int lbs_launch_reduce(const LbsDev &m, const WsLayout &L, int B, float *ws, float *g_transl, hipStream_t st)
{
    long nred = (long)B * JP * 16 + (long)B * m.Kpad + (long)B * 4;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(psi_cdiv(nred * RSPL, 256)), dim3(256), 0, st, B, m.Kpad, ws + L.gA_part, L.nsv,
                       ws + L.gfeat_part, L.nsn, ws + L.gt_part, L.nvb, ws + L.gA, ws + L.gfeat, g_transl);
    PSI_CHECK_LAUNCH("reduce_partials_kernel");
    psi_mark("reduce_partials_kernel", st);
    return 0;
}

int psi_lbs_backward_joint_parts(const psi_lbs_model *mdl, int B, float *ws, float *g_transl, hipStream_t st, bool reduce)
{
    const LbsDev &m = mdl->d;
    WsLayout L = ws_layout(m, B);
    int rc = lbs_launch_bwd_joint_parts(m, L, B, ws, st);
    if (rc || !reduce) return rc;                 // !reduce: the caller's next kernel sums the partials itself (fit.hip: tail_head_kernel)
    return lbs_launch_reduce(m, L, B, ws, g_transl, st);
}

int psi_lbs_backward_ex(const psi_lbs_model *mdl, const float *grad_verts, const float *betas, const float *pose,
                        const float *cam_ext, int B, float *ws, PsiLbsGradOut out, hipStream_t st)
{
    PSI_REQUIRE(mdl && grad_verts && betas && pose && ws, "null pointer");
    PSI_REQUIRE(B > 0 && B <= 16384, "batch size out of range");
    const LbsDev &m = mdl->d;
    WsLayout L = ws_layout(m, B);
    int rc = lbs_launch_bwd_partials(m, L, grad_verts, cam_ext, B, ws, st);
    if (rc) return rc;
    rc = lbs_launch_reduce(m, L, B, ws, out.g_transl, st);
    if (rc) return rc;
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(B), dim3(256), 0, st, m, betas, pose, ws + L.R, ws + L.Jl, ws + L.G, ws + L.gA,
                       ws + L.gfeat, B, out.g_betas, out.g_pose, out.g_rot);
    PSI_CHECK_LAUNCH("pose_bwd_kernel");
    psi_mark("pose_bwd_kernel", st);
    return 0;
}

void psi_lbs_dims(const psi_lbs_model *mdl, int *V, int *J, int *NB)
{
    if (V) *V = mdl->d.V;
    if (J) *J = mdl->d.J;
    if (NB) *NB = mdl->d.NB;
}

extern "C" int psi_lbs_backward(const psi_lbs_model *mdl, const float *grad_verts, const float *betas, const float *pose,
                                const float *cam_ext, int B, float *ws, float *grad_betas, float *grad_pose,
                                float *grad_transl, void *stream)
{
    PsiLbsGradOut out = {grad_betas, grad_pose, grad_transl, nullptr};
    return psi_lbs_backward_ex(mdl, grad_verts, betas, pose, cam_ext, B, ws, out, (hipStream_t)stream);
}
