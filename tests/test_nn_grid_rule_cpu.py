"""CPU restatement of the RULE by which a warm nearest-neighbour query chooses the grid cells it scans
(psi-release_amd/csrc/nnindex_device.h, kd_query_round; the grid itself: nnindex.hip, psi_nn_index_create).

The HIP search is compared bit for bit with brute force on the GPU (tests/test_hip_ops_gpu.py).  What those tests cannot show
is that the rule is sound by CONSTRUCTION rather than by luck of the sampled inputs — the claim of DESIGN.md §3: every target
whose computed distance is <= the warm candidate's lies in a scanned cell, for queries inside the cloud, on cell boundaries,
and outside the cloud's box beyond a face, an edge or a corner.  This file restates the rule in numpy with the kernel's fp32
expressions (exact sqrt / reciprocal where the kernel uses the 1-ulp instructions: the rule's slack is 1e-4 relative + 0.01
cell) and checks that claim against all targets, on the cloud shapes of the GPU test.  Test infrastructure only."""
import numpy as np
import pytest

f32 = np.float32
GRID_MAX_CAND = 64          # nnindex_device.h
GRID_LIST_MAX = 48


def build_grid(y):
    """psi_nn_index_create's uniform grid: origin = the cloud's minimum, about 2.5 points per cell, at most 128 cells per axis."""
    m = len(y)
    gmn, gmx = y.min(0).astype(f32), y.max(0).astype(f32)
    ext = (gmx - gmn).astype(f32)
    emax = float(ext.max())
    vol = 1.0
    for a in range(3):
        vol *= max(float(ext[a]), 1e-3 * emax)
    h = max(np.cbrt(vol * 2.5 / m), emax / 128.0)
    ginv = f32(1.0 / h)
    gn = np.array([min(128, max(1, int(np.floor(f32(ext[a] * ginv))) + 1)) for a in range(3)])
    u = ((y.astype(f32) - gmn) * ginv).astype(f32)                                # fp32, the expression of cell_of()
    cell = np.minimum(np.maximum(np.floor(u).astype(np.int64), 0), gn - 1)
    return gmn, ginv, gn, cell


def sq3(d):
    """PSI_SQ3: x*x + y*y + z*z in fp32, left to right, no contraction."""
    d = d.astype(f32)
    return ((d[..., 0] * d[..., 0]).astype(f32) + (d[..., 1] * d[..., 1]).astype(f32)).astype(f32) + (d[..., 2] * d[..., 2]).astype(f32)


def scanned_cells(q, best, gmn, ginv, gn):
    """The rule, for one query: None (tree walk) or a list of (cx, cy, zlo, zhi)."""
    ru = f32(f32(np.sqrt(best)) * ginv * f32(1.0001) + f32(0.01))
    uq = ((q - gmn) * ginv).astype(f32)
    top = (gn - 1).astype(f32)
    ru2 = f32(ru * ru)
    o = np.maximum(np.maximum(-uq, uq - top - f32(1.0)), f32(0.0)).astype(f32)
    oz2 = f32(o[2] * o[2])
    rx = f32(np.sqrt(max(f32(ru2 - o[1] * o[1] - oz2), f32(0.0))))
    ry = f32(np.sqrt(max(f32(ru2 - o[0] * o[0] - oz2), f32(0.0))))
    clampf = lambda v, t: f32(min(max(np.floor(v), f32(0.0)), t))
    flx, fhx = clampf(uq[0] - rx, top[0]), clampf(uq[0] + rx, top[0])
    fly, fhy = clampf(uq[1] - ry, top[1]), clampf(uq[1] + ry, top[1])
    fny = f32(fhy - fly + 1.0)
    fncand = f32((fhx - flx + 1.0) * fny)
    if not fncand <= GRID_MAX_CAND:
        return None
    out = []
    for col in range(int(fncand)):
        fcx = f32(np.floor(f32((f32(col) + f32(0.5)) * f32(1.0 / fny))))
        cxf, cyf = f32(flx + fcx), f32(fly + (f32(col) - fcx * fny))
        dx = max(max(f32(cxf - uq[0]), f32(uq[0] - cxf - f32(1.0))), f32(0.0))
        dy = max(max(f32(cyf - uq[1]), f32(uq[1] - cyf - f32(1.0))), f32(0.0))
        h2 = f32(ru2 - dx * dx - dy * dy)
        if not h2 >= oz2:
            continue
        hz = f32(np.sqrt(max(h2, f32(0.0))))
        out.append((int(cxf), int(cyf), int(clampf(uq[2] - hz, top[2])), int(clampf(uq[2] + hz, top[2]))))
    return out if 0 < len(out) <= GRID_LIST_MAX else None


def clouds(kind, m, rs):
    if kind == 'uniform':
        return rs.uniform(-1.5, 1.5, (m, 3)).astype(f32)
    if kind == 'surface':
        a = rs.standard_normal((m // 2, 3))
        a = a / np.linalg.norm(a, axis=1, keepdims=True) * 1.3
        f = np.stack([rs.uniform(-2, 2, m - m // 2), rs.uniform(-2, 2, m - m // 2), np.full(m - m // 2, -1.0)], 1)
        return np.concatenate([a, f]).astype(f32)
    if kind == 'plane':
        return np.stack([rs.uniform(-2, 2, m), rs.uniform(-1, 1, m), np.full(m, 0.5)], 1).astype(f32)
    g = np.stack(np.meshgrid(*[np.arange(-4, 4)] * 3, indexing='ij'), -1).reshape(-1, 3).astype(f32) * 0.25   # lattice: ties, boundaries
    return np.concatenate([g, g[::3]])


@pytest.mark.parametrize('kind,m', [('uniform', 20000), ('surface', 12000), ('plane', 3000), ('lattice', 0), ('uniform', 9)])
def test_every_target_within_the_warm_bound_lies_in_a_scanned_cell(kind, m):
    rs = np.random.RandomState(3 + m)
    y = clouds(kind, m, rs)
    m = len(y)
    gmn, ginv, gn, cell = build_grid(y)
    lo, hi = y.min(0), y.max(0)
    ext = np.maximum(hi - lo, 0.5)
    n = 600
    # queries: near targets, on round coordinates (cell boundaries), and OUTSIDE the box beyond faces / edges / corners
    near = (y[rs.randint(0, m, n)] + rs.standard_normal((n, 3)) * 0.05 * ext).astype(f32)
    near[:60] = np.round(near[:60] * 8) / 8
    out = 10.0 ** rs.uniform(-2.5, 0.6, (n, 3)) * ext
    side = rs.randint(0, 3, (n, 3))
    inside = rs.uniform(lo, hi, (n, 3))
    outside = np.where(side == 1, lo - out, np.where(side == 2, hi + out, inside)).astype(f32)
    grid_taken = 0
    for q in np.concatenate([near, outside]):
        d = sq3(y - q)
        # warm candidates: the true winner (the fitting loop's usual case) and a stale one (a neighbour of the winner by rank)
        order = np.argsort(d, kind='stable')
        for h in (order[0], order[min(m - 1, rs.randint(1, 6))]):
            best = d[h]
            cells = scanned_cells(q, best, gmn, ginv, gn)
            if cells is None:
                continue                                                            # the tree walk answers this one
            grid_taken += 1
            must = np.nonzero(d <= best)[0]                                         # every target the exact answer may depend on
            cols = {(cx, cy): (zlo, zhi) for cx, cy, zlo, zhi in cells}
            for i in must:
                cx, cy, cz = cell[i]
                assert (cx, cy) in cols, (kind, q, y[i], 'column not scanned')
                zlo, zhi = cols[(cx, cy)]
                assert zlo <= cz <= zhi, (kind, q, y[i], 'cell outside the z-run')
    assert grid_taken > n                                                           # the rule is exercised, not bypassed


def test_a_query_outside_the_box_scans_a_cap_not_the_balls_square():
    """The point of the outside distances: a body part hanging 8 cells above a uniform cloud has a ball of radius ~8 cells — a 17 x 17
    square of columns — but touches the box in a few cells only, and the rule must keep it on the grid."""
    rs = np.random.RandomState(0)
    y = rs.uniform(-1.5, 1.5, (32768, 3)).astype(f32)
    gmn, ginv, gn, cell = build_grid(y)
    q = np.array([0.1, -0.2, 1.5 + 8.0 / float(ginv)], f32)
    d = sq3(y - q)
    cells = scanned_cells(q, d.min(), gmn, ginv, gn)
    assert cells is not None and len(cells) <= GRID_LIST_MAX
    assert all(zlo >= gn[2] - 3 for _, _, zlo, _ in cells)                          # only the top layers of the box
    side = 2 * int(np.ceil(np.sqrt(d.min()) * float(ginv))) + 1
    assert side * side > GRID_MAX_CAND                                              # the ball's square alone would have sent it to the tree
