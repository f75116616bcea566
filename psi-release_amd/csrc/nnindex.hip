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
#include "nnindex_device.h"
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <cmath>
#include <limits>
#include <vector>

#pragma clang fp contract(off)    // the distance expression is spelled out by PSI_SQ3 (psi_common.h) in both arithmetic modes

namespace {

using namespace psikd;

template <bool CONTACT, bool MULTI = false>
__global__ __launch_bounds__(QBLK) void kd_query_kernel(KdDev T0, const float *__restrict__ xyz1, const int *__restrict__ qidx,
                                                        long qstride, int n, float *__restrict__ dist, int *__restrict__ idx,
                                                        float cconst, float gscale, float *__restrict__ gq, float *__restrict__ fpart,
                                                        int *__restrict__ hint, int rows, const KdDev *__restrict__ tab = nullptr,
                                                        const int *__restrict__ slot = nullptr)
{
    extern __shared__ int smem_i[];
    kd_query_body<CONTACT, MULTI>(T0, KdQueryFromMemory{xyz1, qidx, qstride}, n, dist, idx, cconst, gscale, gq, fpart, hint, rows, tab, slot,
                                  (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, smem_i);
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
    // uniform grid for warm queries (nnindex_device.h): about 2.5 points per cell if the cloud filled its bounding box, at most 128
    // cells per axis; the cell of a point is computed in fp32 with the expression the query side uses
    float gmn[3], gmx[3];
    bd.bounds(0, m, gmn, gmx);          // `order` is a permutation of all points
    float ext[3], emax = 0.0f;
    for (int a = 0; a < 3; a++) { ext[a] = gmx[a] - gmn[a]; emax = std::max(emax, ext[a]); }
    int gn[3] = {1, 1, 1};
    float ginv = 1.0f;
    const bool use_grid = emax > 0.0f && std::isfinite(emax) && !(getenv("PSI_NN_GRID") && getenv("PSI_NN_GRID")[0] == '0');
    std::vector<int> cell_start(2, 0);
    std::vector<float4> gpts;
    if (use_grid) {
        double vol = 1.0;
        for (int a = 0; a < 3; a++) vol *= std::max((double)ext[a], 1e-3 * emax);
        double h = std::cbrt(vol * 2.5 / m);
        h = std::max(h, (double)emax / 128.0);
        ginv = (float)(1.0 / h);
        for (int a = 0; a < 3; a++) gn[a] = std::min(128, std::max(1, (int)std::floor(ext[a] * ginv) + 1));
        const size_t ncell = (size_t)gn[0] * gn[1] * gn[2];
        auto cell_of = [&](int i) {
            int cc[3];
            for (int a = 0; a < 3; a++) {
                const float u = (h_points[(size_t)i * 3 + a] - gmn[a]) * ginv;       // fp32, as in ball_query's range formula
                cc[a] = std::min(std::max((int)std::floor(u), 0), gn[a] - 1);
            }
            return ((size_t)cc[0] * gn[1] + cc[1]) * gn[2] + cc[2];
        };
        // cell-ordered points, ascending original index inside a cell.  A point with the coordinates of a lower-index point can never
        // win under the lowest-index-among-minima rule, so it is left out of the grid's list: what remains has no two entries at equal
        // coordinates, which is what lets the scan's tie flag (nnindex_device.h) stay silent on clouds with duplicated vertices.
        std::vector<size_t> cid(m);
        for (int i = 0; i < m; i++) cid[i] = cell_of(i);
        std::vector<int> ord(m);
        for (int i = 0; i < m; i++) ord[i] = i;
        // (a TOTAL order on the bit patterns — sign-flipped so that it follows the numeric order — not `<` on floats: a cloud with a NaN
        // coordinate must not hand std::sort an inconsistent comparator)
        // (-0.0 is canonicalised to +0.0 first: the duplicate test below is `==`, for which the two are equal, so the order must agree)
        auto okey = [](float f) { f += 0.0f; unsigned u; memcpy(&u, &f, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        auto coord_less = [&](int a, int b) {
            for (int k = 0; k < 3; k++) {
                const unsigned u = okey(h_points[(size_t)a * 3 + k]), v = okey(h_points[(size_t)b * 3 + k]);
                if (u != v) return u < v;
            }
            return a < b;
        };
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return cid[a] != cid[b] ? cid[a] < cid[b] : coord_less(a, b); });
        std::vector<int> keep;
        keep.reserve(m);
        for (int k = 0; k < m; k++) {
            const int i = ord[k];
            bool dup = false;
            if (k > 0 && cid[ord[k - 1]] == cid[i]) {
                const int pi = ord[k - 1];                  // equal coordinates sort next to each other, lowest index first
                dup = h_points[(size_t)pi * 3] == h_points[(size_t)i * 3] && h_points[(size_t)pi * 3 + 1] == h_points[(size_t)i * 3 + 1] &&
                      h_points[(size_t)pi * 3 + 2] == h_points[(size_t)i * 3 + 2];
                if (dup) ord[k] = pi;                       // a run of duplicates keeps comparing against its first member
            }
            if (!dup) keep.push_back(i);
        }
        std::sort(keep.begin(), keep.end(), [&](int a, int b) { return cid[a] != cid[b] ? cid[a] < cid[b] : a < b; });
        cell_start.assign(ncell + 1, 0);
        for (int i : keep) cell_start[cid[i] + 1]++;
        for (size_t c = 0; c < ncell; c++) cell_start[c + 1] += cell_start[c];
        // pair records (32 bytes): {x0,x1,y0,y1} {z0,z1,i0,i1} for list entries 2p, 2p+1 — the scan's packed arithmetic takes two points
        // per instruction.  A scan reads whole pairs, GRID_PAIRS per round, so it may run one entry before its range and a round past it: those
        // are real points of the cloud (ordinary candidates) or, past the end of the list, NaN records that no comparison accepts.
        const size_t npair = (keep.size() + 1) / 2 + GRID_PAIRS;
        const float nanf_ = std::numeric_limits<float>::quiet_NaN();
        gpts.assign(npair * 2, make_float4(nanf_, nanf_, nanf_, nanf_));
        for (size_t k = 0; k < keep.size(); k++) {
            const float4 r = rec(keep[k]);
            float4 &a = gpts[(k / 2) * 2], &b = gpts[(k / 2) * 2 + 1];
            if (k & 1) { a.y = r.x; a.w = r.y; b.y = r.z; b.w = r.w; }
            else       { a.x = r.x; a.z = r.y; b.x = r.z; b.z = r.w; }
        }
        const int none = 0x7fffffff;
        for (size_t k = keep.size(); k < npair * 2; k++) {
            float4 &b = gpts[(k / 2) * 2 + 1];
            if (k & 1) memcpy(&b.w, &none, 4); else memcpy(&b.z, &none, 4);
        }
    }
    size_t nb_nodes = bd.nodes.size() * sizeof(KdNode), nb_pts = pts.size() * sizeof(float4), nb_opts = opts.size() * sizeof(float4);
    size_t nb_cs = use_grid ? cell_start.size() * sizeof(int) : 0, nb_gp = gpts.size() * sizeof(float4);
    size_t off_pts = (nb_nodes + 255) & ~(size_t)255;
    size_t off_opts = (off_pts + nb_pts + 255) & ~(size_t)255;
    size_t off_cs = (off_opts + nb_opts + 255) & ~(size_t)255;
    size_t off_gp = (off_cs + nb_cs + 255) & ~(size_t)255;
    char *blob = nullptr;
    PSI_CHECK_HIP(hipMalloc((void **)&blob, off_gp + nb_gp + 64));
    hipError_t e = hipMemcpy(blob, bd.nodes.data(), nb_nodes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(blob + off_pts, pts.data(), nb_pts, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(blob + off_opts, opts.data(), nb_opts, hipMemcpyHostToDevice);
    if (e == hipSuccess && use_grid) e = hipMemcpy(blob + off_cs, cell_start.data(), nb_cs, hipMemcpyHostToDevice);
    if (e == hipSuccess && use_grid) e = hipMemcpy(blob + off_gp, gpts.data(), nb_gp, hipMemcpyHostToDevice);
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
    ix->d.cell_start = use_grid ? (const int *)(blob + off_cs) : nullptr;
    ix->d.gpts = (const float4 *)(blob + off_gp);
    for (int a = 0; a < 3; a++) { ix->d.gorg[a] = gmn[a]; ix->d.gn[a] = gn[a]; }
    ix->d.ginv = ginv;
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
psikd::KdDev psi_nn_index_dev(const psi_nn_index *ix) { return ix->d; }
