"""GPU parity: Chamfer NN and SDF lookup through the C ABI vs the oracle (bit-exact indices / distances)."""
import numpy as np
import pytest
import torch

import psi_oracle as O
from conftest import golden, rel_err
from psi_release_amd import ops, synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=DEV)


def _chamfer_case(B, n, m, seed, dup=False):
    rs = np.random.RandomState(seed)
    x = rs.uniform(-1.5, 1.5, (B, n, 3)).astype(np.float32)
    y = rs.uniform(-1.5, 1.5, (B, m, 3)).astype(np.float32)
    if dup and m > 700:
        y[:, 700] = y[:, 5]
        y[:, m - 1] = y[:, 600]
        y[:, 6] = y[:, 5]
        x[:, 0] = y[:, 5]                     # exact zero distance, three-way tie
    return x, y


@pytest.mark.parametrize('B,n,m', [(1, 1, 1), (2, 3, 1), (2, 1, 513), (3, 17, 512), (4, 100, 100), (2, 77, 1300),
                                   (4, 512, 4096), (2, 2048, 8192), (1, 300, 64), (1, 65, 63), (3, 1000, 129)])
def test_chamfer_bit_exact(B, n, m):
    x, y = _chamfer_case(B, n, m, seed=B * 1000 + n + m, dup=True)
    d1, i1, d2, i2 = ops.chamfer_forward_raw(T(x), T(y))
    rd1, ri1, rd2, ri2 = O.chamfer_nn_np(x, y)
    assert np.array_equal(i1.cpu().numpy(), ri1) and np.array_equal(i2.cpu().numpy(), ri2)
    assert np.array_equal(d1.cpu().numpy(), rd1) and np.array_equal(d2.cpu().numpy(), rd2)


def test_chamfer_known_answer():
    """Reference's own test (chamfer_pytorch/test_chamfer.py:35-54)."""
    g = golden('chamfer_known_answer')
    d1, d2 = ops.chamferDist()(T(g['p1']), T(g['p2']))
    s = ((d1.cpu().numpy() - g['mydist1']) ** 2).sum() + ((d2.cpu().numpy() - g['mydist2']) ** 2).sum()
    assert s < 1e-8


def test_chamfer_one_sided_and_repeat_determinism():
    x, y = _chamfer_case(4, 512, 4096, 7)
    a = ops.chamfer_forward_raw(T(x), T(y), both=False)
    b = ops.chamfer_forward_raw(T(x), T(y), both=True)
    assert a[2] is None and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for _ in range(3):
        c = ops.chamfer_forward_raw(T(x), T(y), both=True)
        assert all(torch.equal(u, v) for u, v in zip(b, c))


def test_chamfer_full_size_properties():
    """BASELINE size (B=32, n_c=2048, m=32768): a strided sample of queries is checked bit-exactly against the
    oracle; every returned index reproduces its distance; no target beats the returned minimum."""
    B, n, m = 32, 2048, 32768
    x, y = _chamfer_case(B, n, m, 11)
    d1, i1, _, _ = ops.chamfer_forward_raw(T(x), T(y), both=False)
    d1, i1 = d1.cpu().numpy(), i1.cpu().numpy()
    nn = np.take_along_axis(y, i1[:, :, None].astype(np.int64).repeat(3, 2), 1)
    diff = nn - x
    rec = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert np.array_equal(rec.astype(np.float32), d1)
    sub = x[:, ::64]
    rd, ri, _, _ = O.chamfer_nn_np(sub, y, both=False)
    assert np.array_equal(ri, i1[:, ::64]) and np.array_equal(rd, d1[:, ::64])


def test_chamfer_backward_matches_oracle():
    x, y = _chamfer_case(3, 200, 900, 5)
    xt, yt = T(x).requires_grad_(), T(y).requires_grad_()
    d1, d2 = ops.chamferDist()(xt, yt)
    w1 = T(np.random.RandomState(1).standard_normal(d1.shape))
    w2 = T(np.random.RandomState(2).standard_normal(d2.shape))
    ((d1 * w1).sum() + (d2 * w2).sum()).backward()
    xo, yo = torch.tensor(x, requires_grad=True), torch.tensor(y, requires_grad=True)
    o1, o2 = O.chamfer_dist(xo, yo)
    ((o1 * w1.cpu()).sum() + (o2 * w2.cpu()).sum()).backward()
    assert rel_err(xt.grad.cpu(), xo.grad) < 1e-6 and rel_err(yt.grad.cpu(), yo.grad) < 1e-5
    # PSI usage: scene has no grad, second output discarded
    xt2 = T(x).requires_grad_()
    d1b, _ = ops.chamferDist(one_sided=True)(xt2, T(y))
    (d1b * w1).sum().backward()
    xo2 = torch.tensor(x, requires_grad=True)
    o1b, _ = O.chamfer_dist(xo2, torch.tensor(y))
    (o1b * w1.cpu()).sum().backward()
    assert rel_err(xt2.grad.cpu(), xo2.grad) < 1e-6


@pytest.mark.parametrize('ac', [True, False])
@pytest.mark.parametrize('D', [16, 32])
def test_sdf_sample_matches_grid_sample(ac, D):
    sc = synth.make_scene(0, 64, D, 14)
    rs = np.random.RandomState(4)
    B = 3
    verts = rs.uniform(-2.6, 2.6, (B, 2000, 3)).astype(np.float32)
    verts[0, 0] = [-2.0, 2.0, 0.0]
    verts[0, 1] = [2.0, 2.0, 2.0]
    vt = T(verts).requires_grad_()
    out = ops.sdf_sample(vt, T(sc.sdf), T(sc.grid_min), T(sc.grid_max), align_corners=ac)
    w = T(rs.standard_normal((B, 2000)))
    (out * w).sum().backward()
    vo = torch.tensor(verts, requires_grad=True)
    ref = O.sdf_sample(torch.tensor(sc.sdf).unsqueeze(0).expand(B, -1, -1, -1), torch.tensor(sc.grid_min)[None].expand(B, -1),
                       torch.tensor(sc.grid_max)[None].expand(B, -1), vo, align_corners=ac).view(B, -1)
    (ref * w.cpu()).sum().backward()
    assert np.abs(out.detach().cpu().numpy() - ref.detach().numpy()).max() < 5e-6
    assert np.abs(vt.grad.cpu().numpy() - vo.grad.numpy()).max() < 5e-5


def test_sdf_multi_scene_and_penetration_loss():
    a = synth.make_scene(0, 64, 16, 14, kind='room')
    b = synth.make_scene(1, 64, 16, 14, kind='sphere')
    sdf = np.stack([a.sdf, b.sdf])
    gmin = np.stack([a.grid_min, b.grid_min * 0.9])
    gmax = np.stack([a.grid_max, b.grid_max * 0.9])
    rs = np.random.RandomState(8)
    verts = rs.uniform(-1.5, 1.5, (4, 999, 3)).astype(np.float32)
    sid = np.array([0, 1, 1, 0], np.int32)
    vt = T(verts).requires_grad_()
    vals = ops.sdf_sample(vt, T(sdf), T(gmin), T(gmax), scene_id=torch.tensor(sid), align_corners=True)
    loss = ops.penetration_loss(vals)
    loss.backward()
    vo = torch.tensor(verts, requires_grad=True)
    ref = O.sdf_sample(torch.tensor(sdf)[sid.astype(np.int64)], torch.tensor(gmin)[sid.astype(np.int64)],
                       torch.tensor(gmax)[sid.astype(np.int64)], vo, True)
    rl = O.penetration_loss(ref)
    rl.backward()
    assert abs(float(loss) - float(rl)) < 1e-5 * max(1.0, abs(float(rl)))
    assert rel_err(vt.grad.cpu(), vo.grad) < 1e-4
    # no penetration at all -> exactly zero loss and zero gradient
    far = T(np.zeros((2, 10, 3), np.float32)).requires_grad_()
    l0 = ops.penetration_loss(ops.sdf_sample(far, T(a.sdf), T(a.grid_min), T(a.grid_max)))
    l0.backward()
    assert float(l0) == 0.0 and float(far.grad.abs().max()) == 0.0


def _clouds(kind, m, rs):
    if kind == 'uniform':
        return rs.uniform(-1.5, 1.5, (m, 3)).astype(np.float32)
    if kind == 'surface':                                   # points on a sphere + a floor plane: scene-like, non-uniform
        a = rs.standard_normal((m // 2, 3))
        a = a / np.linalg.norm(a, axis=1, keepdims=True) * 1.3
        f = np.stack([rs.uniform(-2, 2, m - m // 2), rs.uniform(-2, 2, m - m // 2), np.full(m - m // 2, -1.0)], 1)
        return np.concatenate([a, f]).astype(np.float32)
    if kind == 'plane':                                     # degenerate bounding box: every point on z = 0.5 exactly
        return np.stack([rs.uniform(-2, 2, m), rs.uniform(-1, 1, m), np.full(m, 0.5)], 1).astype(np.float32)
    if kind == 'lattice':                                   # exact ties everywhere
        g = np.stack(np.meshgrid(*[np.arange(-4, 4)] * 3, indexing='ij'), -1).reshape(-1, 3).astype(np.float32) * 0.25
        return np.concatenate([g, g[::3]])[:m] if m <= len(g) + len(g[::3]) else g


@pytest.mark.parametrize('kind,m', [('uniform', 1), ('uniform', 7), ('uniform', 9), ('uniform', 4096), ('surface', 5000),
                                    ('lattice', 600), ('uniform', 32768)])
def test_kdtree_index_bit_exact_vs_bruteforce(kind, m):
    """psi_nn_index_query == psi_chamfer_forward direction 1 (indices and distances bit-identical), incl. exact ties,
    queries far outside the cloud and queries that coincide with targets."""
    rs = np.random.RandomState(m)
    y = _clouds(kind, m, rs)
    m = len(y)
    B, n = 3, 700
    x = rs.uniform(-2.5, 2.5, (B, n, 3)).astype(np.float32)
    x[0, :50] = y[rs.randint(0, m, 50)]                     # zero distances
    x[1, :20] *= 40.0                                       # far outside
    if kind == 'lattice':
        x[2] = np.round(x[2] * 4) / 4 + 0.125               # equidistant to several lattice points
    index = ops.SceneNNIndex(y, DEV)
    d, i = index.query(T(x))
    rd, ri, _, _ = ops.chamfer_forward_raw(T(x), T(np.broadcast_to(y, (B, m, 3)).copy()), both=False)
    assert torch.equal(i, ri) and torch.equal(d, rd)
    # warm-start hints (good, bad, absent) never change the result and come back as the winners
    hint = torch.tensor(rs.randint(-1, m, (B, n)), dtype=torch.int32, device=DEV)
    d2, i2 = index.query(T(x), hint=hint)
    assert torch.equal(i2, i) and torch.equal(d2, d) and torch.equal(hint, i)
    d3, i3 = index.query(T(x), hint=hint)
    assert torch.equal(i3, i) and torch.equal(d3, d)
    od, oi, _, _ = O.chamfer_nn_np(x[:1, :200], y[None], both=False)
    assert np.array_equal(i[:1, :200].cpu().numpy(), oi) and np.array_equal(d[:1, :200].cpu().numpy(), od)


@pytest.mark.parametrize('kind,m,spread', [('uniform', 32768, 0.03), ('uniform', 32768, 0.3), ('surface', 20000, 0.05), ('lattice', 600, 0.125),
                                           ('plane', 3000, 0.05), ('uniform', 9, 0.1)])
def test_warm_queries_take_the_grid_and_stay_bit_exact(kind, m, spread):
    """The fitting loop's situation: every query carries last iteration's winner as its hint and has moved a little.  Warm queries are
    answered from the uniform grid (all points of the cells the ball of radius sqrt(d_hint) touches; nnindex_device.h) or, when that
    ball is large, by the tree walk — either way distances and indices must equal brute force bit for bit, over several steps, with
    queries on cell boundaries, exact ties (lattice), a flat cloud, and some hints that are stale or absent."""
    rs = np.random.RandomState(7 * m + int(spread * 1000))
    y = _clouds(kind, m, rs)
    m = len(y)
    B, n = 4, 2048
    x = (y[rs.randint(0, m, (B, n))] + rs.standard_normal((B, n, 3)) * spread).astype(np.float32)
    if kind == 'lattice':
        x[2] = np.round(x[2] * 4) / 4 + 0.125               # equidistant to several lattice points
    x[3, :64] = np.round(x[3, :64] * 8) / 8                  # coordinates on round numbers (cell boundaries of many grids)
    index = ops.SceneNNIndex(y, DEV)
    yb = T(np.broadcast_to(y, (B, m, 3)).copy())
    hint = torch.full((B, n), -1, dtype=torch.int32, device=DEV)
    for step in range(4):
        d, i = index.query(T(x), hint=hint)
        rd, ri, _, _ = ops.chamfer_forward_raw(T(x), yb, both=False)
        assert torch.equal(i, ri) and torch.equal(d, rd), (kind, step)
        assert torch.equal(hint, ri)
        x = (x + rs.standard_normal(x.shape) * spread * 0.1).astype(np.float32)
        if step == 1:                                        # some stale / absent hints in the middle of the run
            hint[0, :100] = torch.tensor(rs.randint(0, m, 100), dtype=torch.int32, device=DEV)
            hint[1, :100] = -1


@pytest.mark.parametrize('kind,m', [('uniform', 32768), ('surface', 20000), ('plane', 3000)])
def test_warm_queries_outside_the_clouds_box_stay_bit_exact(kind, m):
    """Body parts that hang out of the scene: warm queries OUTSIDE the cloud's bounding box — beyond a face, an edge, a corner, from a
    fraction of a cell to several box sizes away.  Their ball touches the box in a cap: the grid path takes the cap's rectangle
    (nnindex_device.h: the radius left after the outside distances), several passes over a long column list, or the tree walk — distances
    and indices equal brute force bit for bit in every case, over several steps of a moving query."""
    rs = np.random.RandomState(11 * m)
    y = _clouds(kind, m, rs)
    m = len(y)
    lo, hi = y.min(0), y.max(0)
    ext = np.maximum(hi - lo, 0.5)
    B, n = 4, 2048
    x = rs.uniform(lo - 0.05 * ext, hi + 0.05 * ext, (B, n, 3)).astype(np.float32)
    out = 10.0 ** rs.uniform(-2.5, 0.6, (B, n, 3)) * ext                       # 0.003 .. 4 box sizes beyond the box
    side = rs.randint(0, 3, (B, n, 3))                                         # per axis: inside / below / above
    x = np.where(side == 1, lo - out, np.where(side == 2, hi + out, x)).astype(np.float32)
    x[0, :256, 2] = hi[2] + np.float32(0.3) * ext[2]                           # a sheet of queries above the top face (one-cell-deep caps)
    index = ops.SceneNNIndex(y, DEV)
    yb = T(np.broadcast_to(y, (B, m, 3)).copy())
    hint = torch.full((B, n), -1, dtype=torch.int32, device=DEV)
    for step in range(4):
        d, i = index.query(T(x), hint=hint)
        rd, ri, _, _ = ops.chamfer_forward_raw(T(x), yb, both=False)
        assert torch.equal(i, ri) and torch.equal(d, rd), (kind, step)
        assert torch.equal(hint, ri)
        x = (x + rs.standard_normal(x.shape) * 0.01 * ext).astype(np.float32)


def test_kdtree_backward_matches_bruteforce():
    rs = np.random.RandomState(5)
    y = rs.uniform(-1, 1, (3000, 3)).astype(np.float32)
    x = rs.uniform(-1, 1, (2, 400, 3)).astype(np.float32)
    w = T(rs.standard_normal((2, 400)))
    index = ops.SceneNNIndex(y, DEV)
    a = T(x).requires_grad_()
    (ops.chamfer_to_scene(a, index) * w).sum().backward()
    b = T(x).requires_grad_()
    d1, _ = ops.chamferDist(one_sided=True)(b, T(np.broadcast_to(y, (2, 3000, 3)).copy()))
    (d1 * w).sum().backward()
    assert rel_err(a.grad.cpu(), b.grad.cpu()) < 1e-6


def test_scene_set_query_equals_chamfer_on_gathered_clouds():
    """psi_nn_index_set_query (per-body scene slot, one launch) is bit-identical to chamfer.forward on verts_table[slot]."""
    rs = np.random.RandomState(5)
    S, m, B, n = 3, 3000, 7, 333
    table = torch.tensor(rs.uniform(-1, 1, (S, m, 3)).astype(np.float32), device=DEV)
    table[1, 100] = table[1, 7]                                       # duplicate points: the lower index must win
    slot = torch.tensor(rs.randint(0, S, B).astype(np.int32), device=DEV)
    xyz1 = torch.tensor(rs.uniform(-1.2, 1.2, (B, n, 3)).astype(np.float32), device=DEV, requires_grad=True)
    ss = ops.SceneSet(table, DEV)
    d_set, i_set = ss.query(xyz1.detach(), slot)
    d_ref, i_ref, _, _ = ops.chamfer_forward_raw(xyz1.detach(), table[slot.long()].contiguous(), both=False)
    assert torch.equal(d_set, d_ref) and torch.equal(i_set, i_ref)
    g = torch.tensor(rs.randn(B, n).astype(np.float32), device=DEV)
    (ops.chamfer_to_scenes(xyz1, ss, slot) * g).sum().backward()
    ga = xyz1.grad.clone()
    xyz1.grad = None
    (ops.chamferDist(one_sided=True)(xyz1, table[slot.long()].contiguous())[0] * g).sum().backward()
    assert torch.equal(ga, xyz1.grad)
