// Device side of the exact NN index (nnindex.hip): node / tree descriptors, the 64-bit (distance, index) key and the search body.
// In a header because two translation units instantiate the search: nnindex.hip (the operator entry points) and fit.hip (the fused
// fitting engine runs it inside the same launch as the skinning + SDF kernel, with a query source that skins the contact vertex itself).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include "psi_common.h"

#ifndef PSI_SSTOP
#define PSI_SSTOP(k)
#endif
#ifdef PSI_HEAD_STOPS
// dev: per workgroup {queries that took the tree walk, most node + leaf visits of one query, visits summed, grid points evaluated (max)}
__device__ int psi_kd_stat[4 * 8192];
__device__ int psi_kd_reason[8];        // dev: why a warm query took the tree walk — 0 too many candidates, 1 too many surviving columns, 2 tie flag, 3 no warm candidate
__device__ unsigned long long psi_kd_mark[4 * 8192];   // wall clock (10 ns) of thread 0 at four points of the search body
#define PSI_KD_STAT(what) what
#define PSI_KD_MARK(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 8192 && psi_dbg_sstop >= 9) psi_kd_mark[4 * blockIdx.x + (k)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PSI_KD_STAT(what)
#define PSI_KD_MARK(k)
#endif

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
    const float4 *gpts;             // cell-ordered PAIR records {x0,x1,y0,y1} {z0,z1,bitcast(i0),bitcast(i1)}; NaN records past the end
    float gorg[3], ginv;            // cell of a point: clamp(floor((p - gorg) * ginv), 0, gn - 1), evaluated in fp32 exactly like this
    int gn[3];
};

constexpr int GRID_COLS_PER_LANE = 3;       // (x,y) cell columns of the ball a lane scans per pass over the list of surviving columns ...
#ifndef PSI_KD_GRID_LIST_MAX
#define PSI_KD_GRID_LIST_MAX 48
#endif
constexpr int GRID_LIST_MAX = PSI_KD_GRID_LIST_MAX;   // ... which holds at most this many (more: the tree walk) ...
#ifndef PSI_KD_GRID_PAIRS
#define PSI_KD_GRID_PAIRS 4
#endif
constexpr int GRID_PAIRS = PSI_KD_GRID_PAIRS;   // pair records a lane requests per round of its scan (nnindex.hip pads the list for the overshoot)
#ifndef PSI_KD_GRID_MAX_CAND
#define PSI_KD_GRID_MAX_CAND 64
#endif
constexpr int GRID_MAX_CAND = PSI_KD_GRID_MAX_CAND;           // ... out of at most this many columns in the ball's bounding rectangle

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
// the skinning kernel.  Three phases, so that a source's loads travel in as few DEPENDENT rounds as possible (a round trip under the load
// of the fused launch is ~2 us, and a search workgroup's life is a chain of them): issue(b, j) = loads that depend on nothing but the
// query's slot, fetch(b) = loads that need issue()'s results (called after the caller has ALSO issued its own first loads), prepare(b)
// = once per workgroup with all threads present (it may use LDS and barriers), point() = the query point.
struct KdQueryFromMemory {
    const float *xyz1;
    const int *qidx;
    long qstride;
    __device__ __forceinline__ void issue(int, int) {}
    __device__ __forceinline__ void fetch(int) {}
    __device__ __forceinline__ void prepare(int) {}
    __device__ __forceinline__ void point(int b, int j, int, float &qx, float &qy, float &qz) const
    {
        const size_t qrow = qidx ? (size_t)qidx[j] : (size_t)j;
        const float *qp = xyz1 + (size_t)b * qstride + qrow * 3;
        qx = qp[0]; qy = qp[1]; qz = qp[2];
    }
    // CONTACT searches: contact_post(o, g) sees the gradient of the query's contact term in lane 0 of its group, contact_finish(b, bx, nbx)
    // runs once per workgroup with all threads present — a source that produced the query point itself can carry that gradient further back
    // on the spot (fit.hip: ContactSkinSrc); a source that read the point from memory has nothing to add
    __device__ __forceinline__ void contact_post(size_t, float, float, float) {}
    __device__ __forceinline__ void contact_finish(int, int, int) {}
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
// nothing but the query's slot, as two calls: callers issue the first together with their own first loads and the second together with
// their own dependent loads, and compute the query point while it is in flight.
__device__ __forceinline__ int kd_warm_hint(const int *__restrict__ hint, bool active, size_t o) { return (active && hint) ? hint[o] : -1; }
__device__ __forceinline__ int kd_warm_candidate(const KdDev &T, int h, float4 &hp)
{
    hp = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (h < 0 || h >= T.m) return -1;
    hp = T.opts[h];
    return h;
}

// One round of the search: the workgroup's QPB lane groups answer one query each.  (qx,qy,qz): the group's query (the same values in
// all LPQ lanes of the group), active: the group has a query, o: its output slot (dist / idx / hint / gq index).  Returns the contact
// term s / (s + c) of the query in lane 0 of its group (0 elsewhere) when CONTACT.  No barriers inside: groups are independent.
template <bool CONTACT, class QSrc>
__device__ __forceinline__ float kd_query_round(const KdDev &T, QSrc &qsrc, float qx, float qy, float qz, bool active, size_t o, float *__restrict__ dist,
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
    PSI_KD_STAT(int nvis = 0; int ngp = 0;)
#ifndef PSI_KD_SEED_CELLS
#define PSI_KD_SEED_CELLS 1.5f
#endif
    if (have && T.cell_start && !(best * T.ginv * T.ginv <= PSI_KD_SEED_CELLS * PSI_KD_SEED_CELLS)) {
        // A FAR (or missing) warm candidate — the first iterations of a fitting loop move a body by a tenth of a metre per step, so last
        // iteration's winner is several cells away and its ball holds hundreds of points (the 30-round scans behind the launch's stragglers,
        // profiles/r06_timeline_fwd_scene.txt) — is first replaced by ANY point near the query: the two points of the pair record in the
        // middle of the query's own cell column, z cells [cz - 1, cz + 1] (two dependent loads: the range, one record).  Ordinary candidates,
        // evaluated with the same expression and compared by the (d, index) key: the result is unchanged, the ball shrinks to about a cell.
        // An empty neighbourhood changes nothing (a query with no candidate at all keeps best = +inf and walks the tree as before).
        // Uniform over the lanes of a group (they hold the same query).
        const float fx = fminf(fmaxf(floorf((qx - T.gorg[0]) * T.ginv), 0.0f), (float)(T.gn[0] - 1));
        const float fy = fminf(fmaxf(floorf((qy - T.gorg[1]) * T.ginv), 0.0f), (float)(T.gn[1] - 1));
        const float fz = fminf(fmaxf(floorf((qz - T.gorg[2]) * T.ginv), 0.0f), (float)(T.gn[2] - 1));
        const int cz0 = max((int)fz - 1, 0), cz1 = min((int)fz + 1, T.gn[2] - 1);
        const unsigned cbase = (unsigned)((int)fx * T.gn[1] + (int)fy) * (unsigned)T.gn[2];
        const int p0 = T.cell_start[cbase + cz0], p1 = T.cell_start[cbase + cz1 + 1];
        if (p1 > p0) {
            const float4 *rp = (const float4 *)((const char *)T.gpts + ((unsigned)((p0 + p1) >> 2) << 5));
            const float4 a = rp[0], b = rp[1];                  // {x0,x1,y0,y1} {z0,z1,i0,i1}; a NaN record past the end compares false
            const float xa = a.x - qx, ya = a.z - qy, za = b.x - qz, xb = a.y - qx, yb = a.w - qy, zb = b.y - qz;
            const float da = PSI_SQ3(xa, ya, za), db = PSI_SQ3(xb, yb, zb);
            const kd_key ka = kd_pack(da, __float_as_int(b.z)), kb = kd_pack(db, __float_as_int(b.w));
            if (da == da && ka < bestk) bestk = ka;
            if (db == db && kb < bestk) bestk = kb;
            best = kd_key_d(bestk);
        }
    }
    // Warm query (a candidate from the previous iteration is known): the winner can only lie in the ball of radius sqrt(best) around
    // the query, so instead of walking the tree — ~8 DEPENDENT node / leaf loads — every point of the grid cells that ball touches is
    // evaluated: two dependent rounds of independent loads (cell ranges, then points).
    //   Which cells: in cell units u = (p - org) * ginv the cell of a point is clamp(floor(u)) per axis, the query sits at uq and the
    // ball has radius ru (inflated by 1e-4 relative + 0.01 cell: far above the rounding of u, of the distance expression and of the
    // square roots below).  Column (cx, cy) can hold a point of the ball only if its (x,y) rectangle — the unit square at (cx, cy)
    // (no point has u outside [0, gn]: nnindex.hip puts the origin at the cloud's minimum) — is within ru of uq; if it is, the ball reaches
    // hz = sqrt(ru^2 - dxy^2) along z in that column, i.e. cells floor(uz - hz) .. floor(uz + hz).  The lanes of the group enumerate
    // the columns of the ball's bounding rectangle, the surviving ones are numbered with a ballot and handed out round-robin through
    // the group's (idle) stack rows, so every lane scans at most GRID_COLS_PER_LANE z-runs, each one contiguous point range.
    //   Exact for the same reason the tree is: every point with computed d <= best lies in a scanned cell, every scanned point is
    // evaluated with the identical distance expression, and the lowest index among minima wins — inside a lane by scan order (a
    // strictly smaller distance replaces; an EQUAL one raises the tie flag, and a group with a flagged lane re-does the query with
    // the tree walk and its (d, index) keys), across lanes and against the warm candidate by the key.  A ball with too many columns
    // (a body far from the scene) takes the tree walk as well.  All decisions are uniform over the lanes of a group.
    const int gshift = (tid & 63) & ~(LPQ - 1);               // bit position of my group inside the wave's ballot
    if (have && T.cell_start && best < INFINITY) {
        const float ru = __builtin_amdgcn_sqrtf(best) * T.ginv * 1.0001f + 0.01f;          // (1 ulp: inside the slack)
        const float uq[3] = {(qx - T.gorg[0]) * T.ginv, (qy - T.gorg[1]) * T.ginv, (qz - T.gorg[2]) * T.ginv};
        const float topx = (float)(T.gn[0] - 1), topy = (float)(T.gn[1] - 1), topz = (float)(T.gn[2] - 1);
        // A query OUTSIDE the grid's box by (ox, oy, oz) cells: every point of the cloud is at least that far from it along those axes, so
        // the ball reaches only rx = sqrt(ru^2 - oy^2 - oz^2) along x inside the box (ry alike) — the bounding rectangle of a body part
        // that hangs out of the scene is the small cap where its ball touches the box, not the square around the whole ball (those queries
        // used to fall to the tree walk and were the launch's tail: 12-25 dependent node visits each).
        const float ru2 = ru * ru;
        const float ox = fmaxf(fmaxf(-uq[0], uq[0] - topx - 1.0f), 0.0f), oy = fmaxf(fmaxf(-uq[1], uq[1] - topy - 1.0f), 0.0f),
                    oz = fmaxf(fmaxf(-uq[2], uq[2] - topz - 1.0f), 0.0f);
        const float oz2 = oz * oz;
        const float rx = __builtin_amdgcn_sqrtf(fmaxf(ru2 - oy * oy - oz2, 0.0f)), ry = __builtin_amdgcn_sqrtf(fmaxf(ru2 - ox * ox - oz2, 0.0f));
        // clamped as floats: a far query must not overflow the conversion
        const float flx = fminf(fmaxf(floorf(uq[0] - rx), 0.0f), topx), fhx = fminf(fmaxf(floorf(uq[0] + rx), 0.0f), topx);
        const float fly = fminf(fmaxf(floorf(uq[1] - ry), 0.0f), topy), fhy = fminf(fmaxf(floorf(uq[1] + ry), 0.0f), topy);
        const float fny = fhy - fly + 1.0f, fncand = (fhx - flx + 1.0f) * fny;
        if (fncand <= (float)GRID_MAX_CAND) {
            const int ncand = (int)fncand;
            const float rny = __builtin_amdgcn_rcpf(fny);
            const int list_max = min(GRID_LIST_MAX, 2 * rows);  // the list lives in the group's stack rows
            int nsurv = 0;
            for (int c0 = 0; c0 < ncand; c0 += LPQ) {          // (uniform over the group)
                const float fcol = (float)(c0 + c);
                const float fcx = floorf((fcol + 0.5f) * rny);   // column index / ny: the quotient's fraction is >= 0.5 / ny, far from rounding
                const float cxf = flx + fcx, cyf = fly + (fcol - fcx * fny);
                // distance from uq to the column's unit square, per axis (the grid's origin is the cloud's exact minimum and its last cell ends at
                // or beyond the maximum, nnindex.hip: no point lies outside [0, gn], so the edge cells are ordinary unit cells)
                const float dx = fmaxf(fmaxf(cxf - uq[0], uq[0] - cxf - 1.0f), 0.0f), dy = fmaxf(fmaxf(cyf - uq[1], uq[1] - cyf - 1.0f), 0.0f);
                const float h2 = ru2 - dx * dx - dy * dy;
                const bool ok = c0 + c < ncand && h2 >= oz2;         // (h2 = what is left for z; the box itself is oz away)
                const float hz = __builtin_amdgcn_sqrtf(fmaxf(h2, 0.0f));
                const int zlo = (int)fminf(fmaxf(floorf(uq[2] - hz), 0.0f), topz), zhi = (int)fminf(fmaxf(floorf(uq[2] + hz), 0.0f), topz);
                const unsigned gm = (unsigned)(__ballot(ok) >> gshift) & ((1u << LPQ) - 1u);
                const int pos = nsurv + __popc(gm & ((1u << c) - 1u));
                if (ok && pos < list_max) stk_n[pos] = (((int)cxf * T.gn[1] + (int)cyf) << 14) | (zlo << 7) | zhi;
                nsurv += __popc(gm);
            }
            if (nsurv <= list_max && nsurv > 0) {
              typedef float v2f __attribute__((ext_vector_type(2)));
              const v2f qxx = {qx, qx}, qyy = {qy, qy}, qzz = {qz, qz};
              float bd = INFINITY;
              int bi = 0x7fffffff;
              bool tie = false;
              // GRID_COLS_PER_LANE z-runs per lane at a time; a ball with more columns than that (a body part outside the scene's box touches
              // it in a wide, one-cell-deep cap) takes further passes over the list — each two dependent rounds, against the 12-25 of its tree walk
              for (int s0 = 0; s0 < nsurv; s0 += GRID_COLS_PER_LANE * LPQ) {
                // my z-runs, as ranges of PAIR records: first pair, rounds of GRID_PAIRS pairs
                int ps[GRID_COLS_PER_LANE], nr[GRID_COLS_PER_LANE];
                int cs[GRID_COLS_PER_LANE], ce[GRID_COLS_PER_LANE];
                const char *csb = (const char *)T.cell_start;
#pragma unroll
                for (int u = 0; u < GRID_COLS_PER_LANE; u++) {     // all ranges requested together; a slot past the list re-reads entry 0
                    const int e = stk_n[s0 + c + u * LPQ < nsurv ? s0 + c + u * LPQ : 0];
                    const unsigned base = (unsigned)(e >> 14) * (unsigned)T.gn[2];
                    cs[u] = *(const int *)(csb + ((base + ((unsigned)e >> 7 & 127u)) << 2));
                    ce[u] = *(const int *)(csb + ((base + ((unsigned)e & 127u) + 1u) << 2));
                }
                int rounds = 0;
                PSI_KD_MARK(1);
#pragma unroll
                for (int u = 0; u < GRID_COLS_PER_LANE; u++) {
                    ps[u] = cs[u] >> 1;
                    nr[u] = (ce[u] > cs[u] && s0 + c + u * LPQ < nsurv) ? (((ce[u] + 1) >> 1) - ps[u] + GRID_PAIRS - 1) / GRID_PAIRS : 0;
                    rounds += nr[u];
                }
                // one loop over the rounds of all my runs: round t reads pairs GRID_PAIRS t + off(t) .., off = the run's first pair minus
                // GRID_PAIRS times the rounds before it
                static_assert(GRID_COLS_PER_LANE == 3, "the round -> pair mapping below is written for three runs");
                const int r01 = nr[0] + nr[1];
                const int off1 = ps[1] - GRID_PAIRS * nr[0], off2 = ps[2] - GRID_PAIRS * r01;
                for (int t = 0; t < rounds; t++) {
                    PSI_KD_STAT(ngp += 1;)
                    const int pr = GRID_PAIRS * t + (t < nr[0] ? ps[0] : t < r01 ? off1 : off2);
                    const float4 *rp = (const float4 *)((const char *)T.gpts + ((unsigned)pr << 5));
                    float4 a[GRID_PAIRS], b[GRID_PAIRS];
#pragma unroll
                    for (int k = 0; k < GRID_PAIRS; k++) { a[k] = rp[2 * k]; b[k] = rp[2 * k + 1]; }
#pragma unroll
                    for (int k = 0; k < GRID_PAIRS; k++) {
                        const v2f x2 = (v2f){a[k].x, a[k].y} - qxx, y2 = (v2f){a[k].z, a[k].w} - qyy, z2 = (v2f){b[k].x, b[k].y} - qzz;
#ifdef PSI_CHAMFER_FMA
                        // PSI_SQ3 on two points in the fma mode (mul, fma, fma — v_pk_mul / v_pk_fma): the same rounding as the warm
                        // candidate above, the tree leaves below and chamfer.hip, which the exactness argument relies on
                        const v2f d = __builtin_elementwise_fma(z2, z2, __builtin_elementwise_fma(y2, y2, x2 * x2));
#else
                        const v2f d = x2 * x2 + y2 * y2 + z2 * z2;     // PSI_SQ3 on two points (contraction is off in this function)
#endif
                        const int i2[2] = {__float_as_int(b[k].z), __float_as_int(b[k].w)};
#pragma unroll
                        for (int e = 0; e < 2; e++) {
                            const bool lt = d[e] < bd;
                            tie = tie || d[e] == bd;
                            bi = lt ? i2[e] : bi;
                            bd = lt ? d[e] : bd;
                        }
                    }
                }
              }
                kd_key k = group_min<LPQ>(kd_pack(bd, bi));
                PSI_KD_MARK(2);
                bestk = k < bestk ? k : bestk;
                best = kd_key_d(bestk);
                have = ((__ballot(tie) >> gshift) & ((1ull << LPQ) - 1ull)) != 0;     // done, unless a lane met two points at equal distance
                PSI_KD_STAT(if (have && c == 0 && psi_dbg_sstop == 10) atomicAdd(psi_kd_reason + 2, 1);)
            } else { PSI_KD_STAT(if (c == 0 && psi_dbg_sstop == 10) { atomicAdd(psi_kd_reason + 1, 1); atomicMax(psi_kd_reason + 5, nsurv); }) }
        } else { PSI_KD_STAT(if (c == 0 && psi_dbg_sstop == 10) { atomicAdd(psi_kd_reason + 0, 1); atomicMax(psi_kd_reason + 4, (int)fminf(fncand, 1e6f)); }) }
    } else { PSI_KD_STAT(if (have && c == 0 && psi_dbg_sstop == 10) atomicAdd(psi_kd_reason + 3, 1);) }
    while (true) {
        while (!have && sp > 0) {                             // pop until something survives the current bound (group-uniform)
            --sp;
            cur = stk_n[sp];
            curd = stk_d[sp];
            have = !(curd * 0.999999f > best);
        }
        if (!have) break;
        PSI_KD_STAT(nvis += 1;)
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
    PSI_KD_STAT(if (blockIdx.x < 8192 && psi_dbg_sstop == 10) {
        int *st_ = psi_kd_stat + 4 * blockIdx.x;
        if (c == 0 && nvis > 0) { atomicAdd(st_ + 0, 1); atomicAdd(st_ + 2, nvis); }
        atomicMax(st_ + 1, nvis);
        atomicMax(st_ + 3, ngp);
    })
    float fval = 0.0f;
    PSI_KD_MARK(3);
    if (active && c == 0) {
        const int besti = kd_key_i(bestk);
        if (dist) dist[o] = best;
        if (idx) idx[o] = besti;
        if (hint) hint[o] = besti;
        if (CONTACT) {
            float4 w = hp;                                    // the winner's coordinates (same values the scan used): the warm candidate's
            if (besti != h) w = T.opts[besti];                // are here already — a load only when the winner changed since the last call
            float sq = sqrtf(best + 1e-4f);
            float den = sq + cconst;
            fval = sq / den;
            float gg = gscale * (cconst / (2.0f * sq * den * den)) * 2.0f;
            const float gx = gg * (qx - w.x), gy = gg * (qy - w.y), gz = gg * (qz - w.z);
            if (gq) {
                gq[o * 3 + 0] = gx;
                gq[o * 3 + 1] = gy;
                gq[o * 3 + 2] = gz;
            }
            qsrc.contact_post(o, gx, gy, gz);
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
    // round 1: everything that depends on the slot only (the source's own loads, last iteration's winner); round 2: what those name (the
    // winner's coordinates, the source's second-level loads) — then the workgroup phase and the arithmetic
    qsrc.issue(b, active ? j : 0);
    const int h0 = kd_warm_hint(hint, active, o);
    qsrc.fetch(b);
    const int h = kd_warm_candidate(T, h0, hp);
    qsrc.prepare(b);
    qsrc.point(b, active ? j : 0, c, qx, qy, qz);                // every lane of the group ends up with the same point
    PSI_SSTOP(4);                                                // (dev: differential timing — the query source alone)
    PSI_KD_MARK(0);
    const float fval = kd_query_round<CONTACT>(T, qsrc, qx, qy, qz, active, o, dist, idx, cconst, gscale, gq, hint, rows, smem_i, h, hp);
    if (CONTACT) {
        kd_block_fsum(fval, fpart + (size_t)b * nbx + bx);
        qsrc.contact_finish(b, bx, nbx);
    }
}

static inline size_t kd_lds_bytes(int rows) { return (size_t)QPB * rows * 8; }

}  // namespace psikd

struct psi_nn_index;
psikd::KdDev psi_nn_index_dev(const psi_nn_index *ix);          // nnindex.hip
