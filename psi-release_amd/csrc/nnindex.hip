// Exact nearest-neighbour index over a STATIC target cloud (a scene's downsampled vertices), gfx950.
//
// Same contract as the brute-force Chamfer kernel (chamfer.hip; reference chamfer.cu:12-134): for each query the
// minimum of d = x2*x2 + y2*y2 + z2*z2 (target - query, fp32, left-to-right, no FMA) over ALL targets and the LOWEST
// target index attaining it.  The index only prunes: every target whose computed d could be <= the final minimum is
// still evaluated with the identical expression, so distances and indices are bit-identical to brute force.
//
//   build (host, once per scene): balanced kd-tree, median split on the widest axis, leaves of <= 8 points (padded to 8 records); every
//     internal node stores the exact AABBs of both children (one 64-byte record per visit); points are re-ordered by leaf and stored as float4 {x, y, z, bitcast(original index)}.
//   query (one lane per query, per-lane stack in LDS): depth-first, nearer child first.  A node is skipped when
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

#ifndef PSI_CHAMFER_ALLOW_FMA
#pragma clang fp contract(off)
#endif

namespace {

constexpr int LEAF = 8;             // points per leaf; leaves are PADDED to exactly 8 records (copies of the last point)
                                    // so the scan is a fixed, fully unrolled batch of 8 independent 16-byte loads
constexpr int MAXDEPTH = 32;
constexpr int QBLK = 64;            // one wave per workgroup: per-lane stacks live in LDS [MAXDEPTH][64]

// Fat node: the AABBs of BOTH children (one 64-byte record per visit, no child re-load on pop).
// Child reference: >= 0 internal node index;  < 0 leaf number L encoded -(L) - 1; its 8 records start at pts[8*L].
struct KdNode {
    float lmin[3], lmax[3], rmin[3], rmax[3];
    int left, right, pad0, pad1;
};

struct KdDev {
    const KdNode *nodes;
    const float4 *pts;              // leaf-ordered, 8 records per leaf: {x,y,z,bitcast(orig index)}
    const float4 *opts;             // original order {x,y,z,bitcast(index)}: warm-start lookups
    int root;                       // child-reference of the root (a leaf when m <= LEAF)
    int m;
};

__device__ __forceinline__ float box_d2(const float *mn, const float *mx, float qx, float qy, float qz)
{
    float dx = fmaxf(fmaxf(mn[0] - qx, qx - mx[0]), 0.0f);
    float dy = fmaxf(fmaxf(mn[1] - qy, qy - mx[1]), 0.0f);
    float dz = fmaxf(fmaxf(mn[2] - qz, qz - mx[2]), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// CONTACT: fused contact-loss epilogue, identical to nn_resolve_kernel<true> in chamfer.hip
template <bool CONTACT>
__global__ __launch_bounds__(QBLK) void kd_query_kernel(KdDev T, const float *__restrict__ xyz1, const int *__restrict__ qidx,
                                                        long qstride, int n, float *__restrict__ dist, int *__restrict__ idx,
                                                        float cconst, float gscale, float *__restrict__ gq, float *__restrict__ fpart,
                                                        int *__restrict__ hint)
{
    __shared__ int stk_n[MAXDEPTH][QBLK];
    __shared__ float stk_d[MAXDEPTH][QBLK];
    const int lane = threadIdx.x;
    const int b = blockIdx.y;
    const int j = blockIdx.x * QBLK + lane;
    float fval = 0.0f;
    if (j < n) {
        const size_t qrow = qidx ? (size_t)qidx[j] : (size_t)j;
        const float *qp = xyz1 + (size_t)b * qstride + qrow * 3;
        const float qx = qp[0], qy = qp[1], qz = qp[2];
        float best = INFINITY;
        int besti = 0x7fffffff;
        float bx = 0, by = 0, bz = 0;
        const size_t o = (size_t)b * n + j;
        if (hint) {
            // warm start: the target that won for this query last time is evaluated first (an ordinary candidate, so
            // the result is unchanged); a good initial `best` prunes almost every far child on the way down
            int h = hint[o];
            if (h >= 0 && h < T.m) {
                const float4 p = T.opts[h];
                float x2 = p.x - qx, y2 = p.y - qy, z2 = p.z - qz;
                best = x2 * x2 + y2 * y2 + z2 * z2;
                besti = h;
                bx = p.x; by = p.y; bz = p.z;
            }
        }
        int sp = 0;
        int cur = T.root;
        float curd = 0.0f;
        bool have = true;
        while (true) {
            if (!have) {
                if (sp == 0) break;
                --sp;
                cur = stk_n[sp][lane];
                curd = stk_d[sp][lane];
            }
            have = false;
            if (curd * 0.999999f > best) continue;
            if (cur < 0) {
                const float4 *lp = T.pts + (size_t)(-cur - 1) * LEAF;
                float4 pp[LEAF];
#pragma unroll
                for (int k = 0; k < LEAF; k++) pp[k] = lp[k];
#pragma unroll
                for (int k = 0; k < LEAF; k++) {
                    const float4 p = pp[k];
                    float x2 = p.x - qx, y2 = p.y - qy, z2 = p.z - qz;
                    float d = x2 * x2 + y2 * y2 + z2 * z2;
                    int pi = __float_as_int(p.w);
                    if (d < best || (d == best && pi < besti)) {
                        best = d;
                        besti = pi;
                        bx = p.x; by = p.y; bz = p.z;
                    }
                }
            } else {
                const float4 *np = (const float4 *)(T.nodes + cur);
                const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
                const float lmin[3] = {n0.x, n0.y, n0.z}, lmax[3] = {n0.w, n1.x, n1.y};
                const float rmin[3] = {n1.z, n1.w, n2.x}, rmax[3] = {n2.y, n2.z, n2.w};
                const int left = __float_as_int(n3.x), right = __float_as_int(n3.y);
                const float dl = box_d2(lmin, lmax, qx, qy, qz), dr = box_d2(rmin, rmax, qx, qy, qz);
                const bool left_first = dl <= dr;
                const float dnear = left_first ? dl : dr, dfar = left_first ? dr : dl;
                if (dfar * 0.999999f <= best) {
                    stk_n[sp][lane] = left_first ? right : left;
                    stk_d[sp][lane] = dfar;
                    sp++;
                }
                cur = left_first ? left : right;      // descend into the nearer child without an LDS round trip
                curd = dnear;
                have = true;
            }
        }
        if (dist) dist[o] = best;
        if (idx) idx[o] = besti;
        if (hint) hint[o] = besti;
        if (CONTACT) {
            float sq = sqrtf(best + 1e-4f);
            float den = sq + cconst;
            fval = sq / den;
            float g = gscale * (cconst / (2.0f * sq * den * den)) * 2.0f;
            gq[o * 3 + 0] = g * (qx - bx);
            gq[o * 3 + 1] = g * (qy - by);
            gq[o * 3 + 2] = g * (qz - bz);
        }
    }
    if (CONTACT) {
        float v = fval;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) v += __shfl_down(v, o2, 64);
        if (lane == 0) fpart[(size_t)b * gridDim.x + blockIdx.x] = v;
    }
}

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

    // returns the child reference of the subtree over order[lo:hi)
    int build(int lo, int hi, int depth)
    {
        if (depth > depth_max) depth_max = depth;
        if (hi - lo <= LEAF) {
            leaves.push_back({lo, hi - lo});
            return -((int)leaves.size() - 1) - 1;
        }
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
        int me = (int)nodes.size();
        nodes.push_back(KdNode());
        int l = build(lo, mid, depth + 1);
        int r = build(mid, hi, depth + 1);
        KdNode nd;
        memset(&nd, 0, sizeof(nd));
        bounds(lo, mid, nd.lmin, nd.lmax);
        bounds(mid, hi, nd.rmin, nd.rmax);
        nd.left = l;
        nd.right = r;
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
    PSI_REQUIRE(bd.depth_max + 2 < MAXDEPTH, "kd-tree too deep");
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
    hipLaunchKernelGGL(kd_query_kernel<false>, dim3(psi_cdiv(n, QBLK), B), dim3(QBLK), 0, (hipStream_t)stream, ix->d, xyz1,
                       (const int *)nullptr, (long)n * 3, n, dist1, idx1, 0.0f, 0.0f, (float *)nullptr, (float *)nullptr, hint);
    PSI_CHECK_LAUNCH("kd_query_kernel");
    psi_mark("kd_query_kernel", (hipStream_t)stream);
    return 0;
}

// internal (psi_internal.h): contact-loss NN through the index, same outputs as psi_nn_contact
int psi_nn_index_contact(const psi_nn_index *ix, const float *verts, long vstride, const int *vid, int B, int n, float cconst,
                         float gscale, float *gq, float *fpart, int *hint, hipStream_t st)
{
    hipLaunchKernelGGL(kd_query_kernel<true>, dim3(psi_cdiv(n, QBLK), B), dim3(QBLK), 0, st, ix->d, verts, vid, vstride, n,
                       (float *)nullptr, (int *)nullptr, cconst, gscale, gq, fpart, hint);
    PSI_CHECK_LAUNCH("kd_query_kernel<contact>");
    psi_mark("kd_query_kernel", st);
    return 0;
}
int psi_nn_index_fparts(int n) { return psi_cdiv(n, QBLK); }
