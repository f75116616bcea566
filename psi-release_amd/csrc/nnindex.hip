// Exact nearest-neighbour index over a STATIC target cloud (a scene's downsampled vertices), gfx950.
//
// Same contract as the brute-force Chamfer kernel (chamfer.hip; reference chamfer.cu:12-134): for each query the
// minimum of d = x2*x2 + y2*y2 + z2*z2 (target - query, fp32, left-to-right, no FMA) over ALL targets and the LOWEST
// target index attaining it.  The index only prunes: every target whose computed d could be <= the final minimum is
// still evaluated with the identical expression, so distances and indices are bit-identical to brute force.
//
//   build (host, once per scene): balanced kd-tree (median split on the widest axis) collapsed three levels at a time into
//     8-wide nodes; leaves of <= 8 points (padded to 8 records); every internal node stores the exact AABBs of its 8 children; points are re-ordered by leaf and stored as float4 {x, y, z, bitcast(original index)}.
//   query (a GROUP of 4 lanes per query — each takes two child boxes of a node / two points of a leaf; the traversal state is
//     replicated in the group's lanes, the stack is shared in LDS): depth-first, nearer child first.  A node is skipped when
//       d2box * 0.999999f > best,  d2box = squared distance from the query to the node's AABB evaluated in fp32.
//     Safety: for any point p in the box, d_hat(p) >= true(p) (1 - 3e-7) >= trueBox (1 - 3e-7) >= d2box_hat (1 - 3e-7)^2,
//     so the test implies d_hat(p) > best strictly — p is neither the minimum nor a tie.
//     Leaf points update with (d < best) || (d == best && idx < best_idx): lowest original index among equal minima.
//
// In the reference the scene cloud is static per FittingOP (fitting_proxe.py:93-96), so the tree is built once at
// construction; the brute-force op remains the general `chamfer.forward` replacement (arbitrary, per-sample clouds).
#include "psi_internal.h"
#include <algorithm>
#include <math.h>
#include <string.h>
#include <vector>

#pragma clang fp contract(off)    // the distance expression is spelled out by PSI_SQ3 (psi_common.h) in both arithmetic modes

namespace {

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
};

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
template <bool CONTACT, bool MULTI = false>
__global__ __launch_bounds__(QBLK) void kd_query_kernel(KdDev T0, const float *__restrict__ xyz1, const int *__restrict__ qidx,
                                                        long qstride, int n, float *__restrict__ dist, int *__restrict__ idx,
                                                        float cconst, float gscale, float *__restrict__ gq, float *__restrict__ fpart,
                                                        int *__restrict__ hint, int rows, const KdDev *__restrict__ tab = nullptr,
                                                        const int *__restrict__ slot = nullptr)
{
    extern __shared__ int smem_i[];
    const int tid = threadIdx.x;
    const int c = tid & (LPQ - 1);                            // my first child / leaf slot (the others: c + LPQ, ...)
    const int g = tid / LPQ;                                  // query group inside the workgroup
    int *stk_n = smem_i + (size_t)g * rows * 2;               // [rows] child references
    float *stk_d = (float *)(stk_n + rows);                   // [rows] box distances
    const int b = blockIdx.y;
    const KdDev T = MULTI ? tab[slot[b]] : T0;
    const int j = blockIdx.x * QPB + g;
    const bool active = j < n;
    const size_t o = (size_t)b * n + (active ? j : 0);
    float qx = 0, qy = 0, qz = 0;
    if (active) {
        const size_t qrow = qidx ? (size_t)qidx[j] : (size_t)j;
        const float *qp = xyz1 + (size_t)b * qstride + qrow * 3;
        qx = qp[0]; qy = qp[1]; qz = qp[2];
    }
    kd_key bestk = kd_pack(INFINITY, 0x7fffffff);
    float best = INFINITY;                                    // == kd_key_d(bestk)
    if (active && hint) {
        // warm start: the target that won for this query last time is evaluated first (an ordinary candidate, so the
        // result is unchanged); a good initial `best` prunes almost every sibling on the way down
        int h = hint[o];
        if (h >= 0 && h < T.m) {
            const float4 p = T.opts[h];
            float x2 = p.x - qx, y2 = p.y - qy, z2 = p.z - qz;
            best = PSI_SQ3(x2, y2, z2);
            bestk = kd_pack(best, h);
        }
    }
    int sp = 0;
    int cur = T.root;
    float curd = 0.0f;
    bool have = active;
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
    if (CONTACT) {
        __shared__ float wsum[QBLK / 64];
        float v = fval;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_down(v, o2, 64);
        if ((tid & 63) == 0) wsum[tid >> 6] = v;
        __syncthreads();
        if (tid == 0) {
            float t = 0.0f;
#pragma unroll
            for (int w = 0; w < QBLK / 64; w++) t += wsum[w];
            fpart[(size_t)b * gridDim.x + blockIdx.x] = t;
        }
    }
}

static inline size_t kd_lds_bytes(int rows) { return (size_t)QPB * rows * 8; }

struct Builder {
    const float *p;
    std::vector<int> order;
    std::vector<KdNode> nodes;
    std::vector<std::pair<int, int>> leaves;     // (first, count) into `order`
    int depth_max = 0;

    void bounds(int lo, int hi, float *mn, float *mx)
    {
        for (int c = 0; c < 3; c++) { mn[c] = INFINITY; mx[c] = -INFINITY; }
        for (int i = lo; i < hi; i++)
            for (int c = 0; c < 3; c++) {
                float v = p[(size_t)order[i] * 3 + c];
                mn[c] = std::min(mn[c], v);
                mx[c] = std::max(mx[c], v);
            }
    }

    int median_split(int lo, int hi)             // reorders order[lo:hi) around the median of its widest axis
    {
        float mn[3], mx[3];
        bounds(lo, hi, mn, mx);
        int ax = 0;
        if (mx[1] - mn[1] > mx[ax] - mn[ax]) ax = 1;
        if (mx[2] - mn[2] > mx[ax] - mn[ax]) ax = 2;
        int mid = (lo + hi) / 2;
        const float *pp = p;
        std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi, [pp, ax](int u, int v) {
            float a = pp[(size_t)u * 3 + ax], b = pp[(size_t)v * 3 + ax];
            return a < b || (a == b && u < v);
        });
        return mid;
    }

    // returns the child reference of the subtree over order[lo:hi)
    int build(int lo, int hi, int depth)
    {
        if (depth > depth_max) depth_max = depth;
        if (hi - lo <= LEAF) {
            leaves.push_back({lo, hi - lo});
            return -((int)leaves.size() - 1) - 1;
        }
        // up to three rounds of median splits -> up to 8 ranges (fewer when the range is small)
        std::vector<std::pair<int, int>> rg = {{lo, hi}};
        for (int round = 0; round < 3; round++) {
            std::vector<std::pair<int, int>> nx;
            for (auto &r : rg) {
                if (r.second - r.first > LEAF) {
                    int mid = median_split(r.first, r.second);
                    nx.push_back({r.first, mid});
                    nx.push_back({mid, r.second});
                } else {
                    nx.push_back(r);
                }
            }
            rg.swap(nx);
        }
        int me = (int)nodes.size();
        nodes.push_back(KdNode());
        KdNode nd;
        memset(&nd, 0, sizeof(nd));
        for (int c = 0; c < WIDE; c++) {
            nd.c[c].ref = EMPTY;
            for (int a = 0; a < 3; a++) { nd.c[c].mn[a] = INFINITY; nd.c[c].mx[a] = -INFINITY; }
        }
        for (size_t c = 0; c < rg.size(); c++) {
            float mn[3], mx[3];
            bounds(rg[c].first, rg[c].second, mn, mx);
            for (int a = 0; a < 3; a++) { nd.c[c].mn[a] = mn[a]; nd.c[c].mx[a] = mx[a]; }
            nd.c[c].ref = build(rg[c].first, rg[c].second, depth + 1);
        }
        nodes[me] = nd;
        return me;
    }
};

}  // namespace

struct psi_nn_index {
    KdDev d;
    void *blob;
};

extern "C" int psi_nn_index_create(psi_nn_index **out, const float *h_points, int m)
{
    PSI_REQUIRE(out && h_points && m > 0 && m < (1 << 24), "bad arguments (0 < m < 2^24)");
    Builder bd;
    bd.p = h_points;
    bd.order.resize(m);
    for (int i = 0; i < m; i++) bd.order[i] = i;
    bd.nodes.reserve((size_t)2 * (m / LEAF + 2));
    int root = bd.build(0, m, 0);
    const int rows = 7 * (bd.depth_max + 1) + 2;
    PSI_REQUIRE(rows <= MAXSTACK, "tree too deep for the traversal stack");
    auto rec = [&](int oi) {
        float4 r;
        r.x = h_points[(size_t)oi * 3 + 0];
        r.y = h_points[(size_t)oi * 3 + 1];
        r.z = h_points[(size_t)oi * 3 + 2];
        memcpy(&r.w, &oi, 4);
        return r;
    };
    std::vector<float4> pts(bd.leaves.size() * LEAF), opts(m);
    for (size_t L = 0; L < bd.leaves.size(); L++)
        for (int k = 0; k < LEAF; k++) {
            int kk = k < bd.leaves[L].second ? k : bd.leaves[L].second - 1;     // pad with copies of the last point
            pts[L * LEAF + k] = rec(bd.order[bd.leaves[L].first + kk]);
        }
    for (int i = 0; i < m; i++) opts[i] = rec(i);
    if (bd.nodes.empty()) bd.nodes.push_back(KdNode());
    size_t nb_nodes = bd.nodes.size() * sizeof(KdNode), nb_pts = pts.size() * sizeof(float4), nb_opts = opts.size() * sizeof(float4);
    size_t off_pts = (nb_nodes + 255) & ~(size_t)255;
    size_t off_opts = (off_pts + nb_pts + 255) & ~(size_t)255;
    char *blob = nullptr;
    PSI_CHECK_HIP(hipMalloc((void **)&blob, off_opts + nb_opts));
    hipError_t e = hipMemcpy(blob, bd.nodes.data(), nb_nodes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(blob + off_pts, pts.data(), nb_pts, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(blob + off_opts, opts.data(), nb_opts, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(blob);
        psi_set_error("psi_nn_index_create: upload failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    psi_nn_index *ix = new psi_nn_index;
    ix->blob = blob;
    ix->d.nodes = (const KdNode *)blob;
    ix->d.pts = (const float4 *)(blob + off_pts);
    ix->d.opts = (const float4 *)(blob + off_opts);
    ix->d.root = root;
    ix->d.m = m;
    ix->d.rows = rows;
    *out = ix;
    return 0;
}

extern "C" void psi_nn_index_destroy(psi_nn_index *ix)
{
    if (!ix) return;
    (void)hipFree(ix->blob);
    delete ix;
}

extern "C" int psi_nn_index_query(const psi_nn_index *ix, const float *xyz1, int B, int n, float *dist1, int32_t *idx1,
                                  int32_t *hint, void *stream)
{
    PSI_REQUIRE(ix && B >= 0 && n >= 0, "bad arguments");
    if (B == 0 || n == 0) return 0;
    PSI_REQUIRE(xyz1 && dist1 && idx1, "null pointer");
    PSI_REQUIRE(B <= 65535, "B exceeds grid.y");
    hipLaunchKernelGGL(kd_query_kernel<false>, dim3(psi_cdiv(n, QPB), B), dim3(QBLK), kd_lds_bytes(ix->d.rows), (hipStream_t)stream,
                       ix->d, xyz1, (const int *)nullptr, (long)n * 3, n, dist1, idx1, 0.0f, 0.0f, (float *)nullptr, (float *)nullptr,
                       hint, ix->d.rows);
    PSI_CHECK_LAUNCH("kd_query_kernel");
    psi_mark("kd_query_kernel", (hipStream_t)stream);
    return 0;
}

struct psi_nn_index_set {
    KdDev *tab;             // device [S]
    int S;
    int rows;               // max stack rows over the set
};

extern "C" int psi_nn_index_set_create(psi_nn_index_set **out, const psi_nn_index *const *indices, int S)
{
    PSI_REQUIRE(out && indices && S > 0, "bad arguments");
    std::vector<KdDev> h(S);
    for (int s = 0; s < S; s++) {
        PSI_REQUIRE(indices[s], "null index in set");
        h[s] = indices[s]->d;
    }
    int rows = 0;
    for (int s = 0; s < S; s++) rows = std::max(rows, h[s].rows);
    KdDev *tab = nullptr;
    PSI_CHECK_HIP(hipMalloc((void **)&tab, sizeof(KdDev) * S));
    hipError_t e = hipMemcpy(tab, h.data(), sizeof(KdDev) * S, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(tab);
        psi_set_error("psi_nn_index_set_create: upload failed: %s", hipGetErrorString(e));
        return (int)e;
    }
    psi_nn_index_set *st = new psi_nn_index_set;
    st->tab = tab;
    st->S = S;
    st->rows = rows;
    *out = st;
    return 0;
}

extern "C" void psi_nn_index_set_destroy(psi_nn_index_set *set)
{
    if (!set) return;
    (void)hipFree(set->tab);
    delete set;
}

extern "C" int psi_nn_index_set_query(const psi_nn_index_set *set, const int32_t *slot, const float *xyz1, int B, int n,
                                      float *dist1, int32_t *idx1, void *stream)
{
    PSI_REQUIRE(set && B >= 0 && n >= 0, "bad arguments");
    if (B == 0 || n == 0) return 0;
    PSI_REQUIRE(slot && xyz1 && dist1 && idx1, "null pointer");
    PSI_REQUIRE(B <= 65535, "B exceeds grid.y");
    hipLaunchKernelGGL((kd_query_kernel<false, true>), dim3(psi_cdiv(n, QPB), B), dim3(QBLK), kd_lds_bytes(set->rows),
                       (hipStream_t)stream, KdDev(), xyz1, (const int *)nullptr, (long)n * 3, n, dist1, idx1, 0.0f, 0.0f,
                       (float *)nullptr, (float *)nullptr, (int *)nullptr, set->rows, (const KdDev *)set->tab, (const int *)slot);
    PSI_CHECK_LAUNCH("kd_query_kernel<multi>");
    psi_mark("kd_query_kernel", (hipStream_t)stream);
    return 0;
}

// internal (psi_internal.h): contact-loss NN through the index, same outputs as psi_nn_contact
int psi_nn_index_contact(const psi_nn_index *ix, const float *verts, long vstride, const int *vid, int B, int n, float cconst,
                         float gscale, float *gq, float *fpart, int *hint, hipStream_t st)
{
    hipLaunchKernelGGL(kd_query_kernel<true>, dim3(psi_cdiv(n, QPB), B), dim3(QBLK), kd_lds_bytes(ix->d.rows), st, ix->d, verts, vid,
                       vstride, n, (float *)nullptr, (int *)nullptr, cconst, gscale, gq, fpart, hint, ix->d.rows);
    PSI_CHECK_LAUNCH("kd_query_kernel<contact>");
    psi_mark("kd_query_kernel", st);
    return 0;
}
int psi_nn_index_fparts(int n) { return psi_cdiv(n, QPB); }
