// Device side of the exact NN index (nnindex.hip): node / tree descriptors, the 64-bit (distance, index) key and the search body.
// In a header because two translation units instantiate the search: nnindex.hip (the operator entry points) and fit.hip (the fused
// fitting engine runs it inside the same launch as the skinning + SDF kernel, with a query source that skins the contact vertex itself).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "psi_common.h"

namespace psikd {

constexpr int LEAF = 8;             // points per leaf; leaves are PADDED to exactly 8 records (copies of the last point)
constexpr int WIDE = 8;             // children per internal node == lanes per query
constexpr int MAXSTACK = 72;        // <= 7 pushes per level; 9 levels of fan-out 8 cover 2^24 points
constexpr int QBLK = 256;           // threads per workgroup
#ifndef PSI_KD_LPQ
#define PSI_KD_LPQ 4
#endif
constexpr int LPQ = PSI_KD_LPQ;     // lanes per query (8: one child box / leaf point per lane; 4: two).  Measured at B*n_c = 65536:
                                    // 4 lanes 28.8 us cold / 23.2 warm, 8 lanes 31.4 / 22.5; the fused iteration is 1.3 us faster with 4
constexpr int CPL = WIDE / LPQ;     // children (leaf points) per lane
constexpr int QPB = QBLK / LPQ;     // queries per workgroup
constexpr int EMPTY = (int)0x80000000;

// 8-wide node (256 bytes): per child its exact AABB and its reference, 32 bytes each, so lane c of a query's 8-lane group
// reads child c with two 16-byte loads and the group reads the 256-byte record contiguously.
// Child reference: >= 0 internal node index;  < 0 leaf number L encoded -(L) - 1 (records pts[8L .. 8L+7]);  EMPTY = none
// (its box is [+inf, -inf], i.e. infinitely far).
struct KdChild {
    float mn[3], mx[3];
    int ref;
    int pad;
};
struct KdNode {
    KdChild c[WIDE];
};
static_assert(sizeof(KdNode) == 256, "node record is 256 bytes");

struct KdDev {
    const KdNode *nodes;
    const float4 *pts;              // leaf-ordered, 8 records per leaf: {x,y,z,bitcast(orig index)}
    const float4 *opts;             // original order {x,y,z,bitcast(index)}: warm-start lookups
    int root;                       // child-reference of the root (a leaf when m <= LEAF)
    int m;
    int rows;                       // traversal stack rows this tree needs: 7 pushes per level + slack
    // Uniform grid over the same cloud, for WARM queries (ball_query below): cell (cx,cy,cz) holds gpts[cell_start[i] .. cell_start[i+1]),
    // i = (cx * gn[1] + cy) * gn[2] + cz — the cells of a z-run are one contiguous point range.  cell_start == nullptr: no grid.
    const int *cell_start;          // [gn0*gn1*gn2 + 1]
    const float4 *gpts;             // cell-ordered {x,y,z,bitcast(orig index)}
    float gorg[3], ginv;            // cell of a point: clamp(floor((p - gorg) * ginv), 0, gn - 1), evaluated in fp32 exactly like this
    int gn[3];
};

constexpr int GRID_MAX_COLS = 3 * LPQ;      // (x,y) cell columns a warm query may touch (three per lane) ...
constexpr int GRID_MAX_ZRUN = 4;            // ... and cells per column, before it takes the tree walk instead
constexpr float GRID_EPS = 1e-3f;           // slack of the cell range in cell units: covers the rounding of (p - org) * inv on either side

// lane permutations inside an 8-lane group as DPP modifiers (no LDS traffic)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
constexpr int DPP_XOR1 = 0xB1;          // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;          // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141;  // row_half_mirror: lane i <-> 7 - i inside each 8 lanes

// (d, i) packed into one 64-bit key: d >= +0 always (a sum of squares), so its IEEE-754 bit pattern orders like an
// unsigned integer and  key = bits(d) << 32 | i  orders lexicographically by (d, i) — the lowest-index-among-minima
// rule is a single unsigned 64-bit minimum, with no branches.
typedef unsigned long long kd_key;
__device__ __forceinline__ kd_key kd_pack(float d, int i) { return ((kd_key)(unsigned)__float_as_int(d) << 32) | (unsigned)i; }
__device__ __forceinline__ float kd_key_d(kd_key k) { return __int_as_float((int)(k >> 32)); }
__device__ __forceinline__ int kd_key_i(kd_key k) { return (int)(unsigned)k; }

// minimum of the key over the LPQ (4 or 8) lanes of a group, result in every lane
template <int LPQ>
__device__ __forceinline__ kd_key group_min(kd_key k)
{
#define PSI_STEP(CTRL)                                                                              \
    {                                                                                               \
        kd_key k2 = ((kd_key)(unsigned)dpp_i<CTRL>((int)(k >> 32)) << 32) | (unsigned)dpp_i<CTRL>((int)(unsigned)k); \
        k = k2 < k ? k2 : k;                                                                        \
    }
    PSI_STEP(DPP_XOR1)
    PSI_STEP(DPP_XOR2)
    if (LPQ == 8) PSI_STEP(DPP_HALF_MIRROR)
#undef PSI_STEP
    return k;
}

// CONTACT: fused contact-loss epilogue, identical to nn_resolve_kernel<true> in chamfer.hip
// MULTI: body b is searched in tab[slot[b]] (a set of scenes, one launch) instead of the single tree T0
//
// Why a lane group per query: a batch has only B*n_c = 65536 queries.  One lane per query is 1024 waves — one per SIMD, no
// latency hiding at all — each running ~8000 dependent instructions (measured: 15 cycles per instruction, 41-57 us).
// With the child boxes / leaf points of a visit spread over the group's lanes the per-wave instruction stream shrinks several
// fold in the box and leaf arithmetic, there are 4096-8192 waves to overlap the dependent node loads, and a wave diverges
// over 16 (8) queries, not 64.
// Query source: where a query point comes from.  KdQueryFromMemory reads row qidx[j] (or j) of a [rows,3] table per body; the fused
// fitting engine substitutes a source that SKINS the contact vertex on the spot (fit.hip), so the search does not have to wait for
// the skinning kernel.  prepare(b) runs once per workgroup with all threads present (it may use LDS and barriers).
struct KdQueryFromMemory {
    const float *xyz1;
    const int *qidx;
    long qstride;
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void point(int b, int j, int, float &qx, float &qy, float &qz) const
    {
        const size_t qrow = qidx ? (size_t)qidx[j] : (size_t)j;
        const float *qp = xyz1 + (size_t)b * qstride + qrow * 3;
        qx = qp[0]; qy = qp[1]; qz = qp[2];
    }
};

// sum of the per-group contact terms over the workgroup (fixed order) -> *out
__device__ __forceinline__ void kd_block_fsum(float fval, float *__restrict__ out)
{
    __shared__ float wsum[QBLK / 64];
    const int tid = threadIdx.x;
    float v = fval;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_down(v, o2, 64);
    if ((tid & 63) == 0) wsum[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < QBLK / 64; w++) t += wsum[w];
        *out = t;
    }
}

// The warm-start candidate of a query: last iteration's winner (hint[o], -1 = none) and its coordinates.  Two dependent loads that need
// nothing but the query's slot — callers issue them first and compute the query point while they are in flight.
__device__ __forceinline__ int kd_warm_candidate(const KdDev &T, const int *__restrict__ hint, bool active, size_t o, float4 &hp)
{
    hp = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (!(active && hint)) return -1;
    const int h = hint[o];
    if (h < 0 || h >= T.m) return -1;
    hp = T.opts[h];
    return h;
}

// One round of the search: the workgroup's QPB lane groups answer one query each.  (qx,qy,qz): the group's query (the same values in
// all LPQ lanes of the group), active: the group has a query, o: its output slot (dist / idx / hint / gq index).  Returns the contact
// term s / (s + c) of the query in lane 0 of its group (0 elsewhere) when CONTACT.  No barriers inside: groups are independent.
template <bool CONTACT>
__device__ __forceinline__ float kd_query_round(const KdDev &T, float qx, float qy, float qz, bool active, size_t o, float *__restrict__ dist,
                                                int *__restrict__ idx, float cconst, float gscale, float *__restrict__ gq, int *__restrict__ hint,
                                                int rows, int *smem_i, int h, const float4 &hp)
{
#pragma clang fp contract(off)    // the distance expression (PSI_SQ3) must not be re-contracted in translation units built with contraction on
    const int tid = threadIdx.x;
    const int c = tid & (LPQ - 1);                            // my first child / leaf slot (the others: c + LPQ, ...)
    const int g = tid / LPQ;                                  // query group inside the workgroup
    int *stk_n = smem_i + (size_t)g * rows * 2;               // [rows] child references
    float *stk_d = (float *)(stk_n + rows);                   // [rows] box distances
    kd_key bestk = kd_pack(INFINITY, 0x7fffffff);
    float best = INFINITY;                                    // == kd_key_d(bestk)
    if (h >= 0) {
        // warm start: the target that won for this query last time (fetched by the caller, kd_warm_candidate, BEFORE it produced the query
        // point, so the two dependent loads overlap that work) is evaluated first — an ordinary candidate, so the result is unchanged;
        // a good initial `best` prunes almost every sibling on the way down
        float x2 = hp.x - qx, y2 = hp.y - qy, z2 = hp.z - qz;
        best = PSI_SQ3(x2, y2, z2);
        bestk = kd_pack(best, h);
    }
    int sp = 0;
    int cur = T.root;
    float curd = 0.0f;
    bool have = active;
    // Warm query (a candidate from the previous iteration is known): the winner can only lie in the ball of radius sqrt(best) around
    // the query, so instead of walking the tree — ~8 DEPENDENT node / leaf loads — every point of the grid cells that ball touches is
    // evaluated: two dependent rounds of independent loads (cell ranges, then points).  Exact for the same reason the tree is: a point
    // p with computed d(p) <= best has |p - q| <= sqrt(best) (1 + 3e-7) on every axis, the cell range below covers that interval with
    // GRID_EPS cells of slack against the rounding of the cell formula, every point in range is evaluated with the identical distance
    // expression and the (d, index) key keeps the lowest index among minima.  A ball that touches too many cells (a body far from
    // the scene) takes the tree walk.  The decision is uniform over the lanes of a group (they hold the same query and bound).
    if (have && T.cell_start && best < INFINITY) {
        const float r = sqrtf(best) * 1.00001f + 1e-30f;
        int lo[3], hi[3];
        const float q3[3] = {qx, qy, qz};
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const float top = (float)(T.gn[a] - 1);          // clamped as floats: a far query must not overflow the conversion
            lo[a] = (int)fminf(fmaxf(floorf((q3[a] - r - T.gorg[a]) * T.ginv - GRID_EPS), 0.0f), top);
            hi[a] = (int)fminf(fmaxf(floorf((q3[a] + r - T.gorg[a]) * T.ginv + GRID_EPS), 0.0f), top);
        }
        const int nyc = hi[1] - lo[1] + 1, ncol = (hi[0] - lo[0] + 1) * nyc, nzc = hi[2] - lo[2] + 1;
        if (ncol >= 1 && nyc >= 1 && nzc >= 1 && ncol <= GRID_MAX_COLS && nzc <= GRID_MAX_ZRUN) {
            int cs[3], ce[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {                      // my columns: c, c + LPQ, c + 2 LPQ — all ranges requested together
                const int col = c + u * LPQ;
                cs[u] = ce[u] = 0;
                if (col < ncol) {
                    const int cx = lo[0] + col / nyc, cy = lo[1] + col % nyc;
                    const int base = (cx * T.gn[1] + cy) * T.gn[2];
                    cs[u] = T.cell_start[base + lo[2]];
                    ce[u] = T.cell_start[base + hi[2] + 1];
                }
            }
            kd_key k = ~0ull;
#pragma unroll
            for (int u = 0; u < 3; u++) {
                for (int i = cs[u]; i < ce[u]; i += 4) {        // four points per round trip (indices past the end repeat the last point)
                    float4 pp[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) pp[t] = T.gpts[min(i + t, ce[u] - 1)];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        float x2 = pp[t].x - qx, y2 = pp[t].y - qy, z2 = pp[t].z - qz;
                        const kd_key ku = kd_pack(PSI_SQ3(x2, y2, z2), __float_as_int(pp[t].w));
                        k = ku < k ? ku : k;
                    }
                }
            }
            k = group_min<LPQ>(k);
            bestk = k < bestk ? k : bestk;
            best = kd_key_d(bestk);
            have = false;                                      // done: the tree walk below is skipped for this query
        }
    }
    const int gshift = (tid & 63) & ~(LPQ - 1);               // bit position of my group inside the wave's ballot
    while (true) {
        while (!have && sp > 0) {                             // pop until something survives the current bound (group-uniform)
            --sp;
            cur = stk_n[sp];
            curd = stk_d[sp];
            have = !(curd * 0.999999f > best);
        }
        if (!have) break;
        if (cur >= 0) {
            float dc[CPL];
            int ref[CPL];
            kd_key km = ~0ull;
#pragma unroll
            for (int u = 0; u < CPL; u++) {
                const float4 *cp = (const float4 *)(T.nodes + cur) + 2 * (c + u * LPQ);
                const float4 lo = cp[0], hi = cp[1];          // {mnx,mny,mnz,mxx} {mxy,mxz,ref,-}
                float dx = fmaxf(fmaxf(lo.x - qx, qx - lo.w), 0.0f);
                float dy = fmaxf(fmaxf(lo.y - qy, qy - hi.x), 0.0f);
                float dz = fmaxf(fmaxf(lo.z - qz, qz - hi.y), 0.0f);
                dc[u] = dx * dx + dy * dy + dz * dz;          // +inf for EMPTY children
                ref[u] = __float_as_int(hi.z);
                const kd_key k = kd_pack(dc[u], c + u * LPQ);
                km = k < km ? k : km;
            }
            km = group_min<LPQ>(km);
            const float dmin = kd_key_d(km);
            const int cmin = kd_key_i(km);
            // push the other children that can still matter; descend into the nearest without a stack round trip
#pragma unroll
            for (int u = 0; u < CPL; u++) {
                const bool push = (c + u * LPQ) != cmin && dc[u] * 0.999999f <= best;
                const unsigned gm = (unsigned)(__ballot(push) >> gshift) & ((1u << LPQ) - 1u);
                if (push) {
                    const int pos = sp + __popc(gm & ((1u << c) - 1u));
                    stk_n[pos] = ref[u];
                    stk_d[pos] = dc[u];
                }
                sp += __popc(gm);
            }
            int rsel = ref[0];
#pragma unroll
            for (int u = 1; u < CPL; u++) rsel = (cmin / LPQ == u) ? ref[u] : rsel;
            cur = __shfl(rsel, (tid & 63 & ~(LPQ - 1)) | (cmin & (LPQ - 1)), 64);
            curd = dmin;
            have = dmin < INFINITY && !(dmin * 0.999999f > best);
        }
        if (have && cur < 0) {                                // leaf — possibly the one just stepped into
            kd_key k = ~0ull;
#pragma unroll
            for (int u = 0; u < CPL; u++) {
                const float4 p = T.pts[(size_t)(-cur - 1) * LEAF + c + u * LPQ];
                float x2 = p.x - qx, y2 = p.y - qy, z2 = p.z - qz;
                const float d = PSI_SQ3(x2, y2, z2);
                const kd_key ku = kd_pack(d, __float_as_int(p.w));
                k = ku < k ? ku : k;
            }
            k = group_min<LPQ>(k);
            bestk = k < bestk ? k : bestk;
            best = kd_key_d(bestk);
            have = false;
        }
    }
    float fval = 0.0f;
    if (active && c == 0) {
        const int besti = kd_key_i(bestk);
        if (dist) dist[o] = best;
        if (idx) idx[o] = besti;
        if (hint) hint[o] = besti;
        if (CONTACT) {
            const float4 w = T.opts[besti];                   // the winner's coordinates (same values the scan used)
            float sq = sqrtf(best + 1e-4f);
            float den = sq + cconst;
            fval = sq / den;
            float gg = gscale * (cconst / (2.0f * sq * den * den)) * 2.0f;
            gq[o * 3 + 0] = gg * (qx - w.x);
            gq[o * 3 + 1] = gg * (qy - w.y);
            gq[o * 3 + 2] = gg * (qz - w.z);
        }
    }
    return fval;
}

// body of the search for workgroup (bx, b) of an (nbx, B) grid; smem_i: QPB * rows * 8 bytes of LDS for the traversal stacks
template <bool CONTACT, bool MULTI, class QSrc>
__device__ __forceinline__ void kd_query_body(KdDev T0, QSrc qsrc, int n, float *__restrict__ dist, int *__restrict__ idx, float cconst,
                                              float gscale, float *__restrict__ gq, float *__restrict__ fpart, int *__restrict__ hint,
                                              int rows, const KdDev *__restrict__ tab, const int *__restrict__ slot, int bx, int b, int nbx,
                                              int *smem_i)
{
    const int tid = threadIdx.x;
    const int c = tid & (LPQ - 1), g = tid / LPQ;
    const KdDev T = MULTI ? tab[slot[b]] : T0;
    const int j = bx * QPB + g;
    const bool active = j < n;
    const size_t o = (size_t)b * n + (active ? j : 0);
    float qx = 0, qy = 0, qz = 0;
    float4 hp;
    const int h = kd_warm_candidate(T, hint, active, o, hp);   // in flight while the query point is produced
    qsrc.prepare(b);
    qsrc.point(b, active ? j : 0, c, qx, qy, qz);                // every lane of the group ends up with the same point
    const float fval = kd_query_round<CONTACT>(T, qx, qy, qz, active, o, dist, idx, cconst, gscale, gq, hint, rows, smem_i, h, hp);
    if (CONTACT) kd_block_fsum(fval, fpart + (size_t)b * nbx + bx);
}

static inline size_t kd_lds_bytes(int rows) { return (size_t)QPB * rows * 8; }

}  // namespace psikd

struct psi_nn_index;
psikd::KdDev psi_nn_index_dev(const psi_nn_index *ix);          // nnindex.hip
