"""The two re-formulations of the kinematic chain used by the per-body kernels (psi-release_amd/csrc/lbs_device.h), checked on the
CPU in float64 against the reference's sequential chain (lbs.py:244-250) and torch autograd through it:

* forward  — pointer jumping: every joint starts from its local transform and multiplies in its 2^r-th ancestor per round;
* backward — the gradient that reaches G_j = [GR_j | Gt_j] from its whole subtree is
                 g(G_j).t = sum_d w_d,      g(G_j).R = [ sum_d U_d - (sum_d w_d) Gt_j^T ] GR_j       (d over subtree(j))
             with  w_d = gown(G_d).t,  U_d = gown(G_d).R GR_d^T + w_d Gt_d^T  (gown: what G_d receives from its own A_d),
             from which the local gradients follow as  gR_j = GR_p^T g(G_j).R,  grel_j = GR_p^T g(G_j).t  (p = parent of j).
"""
import numpy as np
import torch

from psi_release_amd import synth


def _rodrigues(aa):
    th = torch.sqrt((aa * aa).sum(-1, keepdim=True) + 1e-16)
    k = aa / th
    K = torch.zeros(aa.shape[0], 3, 3, dtype=aa.dtype)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    s, c = torch.sin(th)[:, :, None], torch.cos(th)[:, :, None]
    return torch.eye(3, dtype=aa.dtype)[None] + s * K + (1 - c) * (K @ K)


def _chain_sequential(R, rel, parents):
    """lbs.py:244-250: G_j = G_parent(j) [R_j | rel_j], one joint after the other."""
    J = R.shape[0]
    GR, Gt = [None] * J, [None] * J
    for j in range(J):
        p = int(parents[j])
        if p < 0:
            GR[j], Gt[j] = R[j], rel[j]
        else:
            GR[j], Gt[j] = GR[p] @ R[j], GR[p] @ rel[j] + Gt[p]
    return torch.stack(GR), torch.stack(Gt)


def _problem(seed=0):
    par = np.asarray(synth.SMPLX_PARENTS).copy()
    par[0] = -1
    J = len(par)
    g = torch.Generator().manual_seed(seed)
    aa = (torch.randn(J, 3, generator=g, dtype=torch.float64) * 0.6).requires_grad_(True)
    Jl = torch.randn(J, 3, generator=g, dtype=torch.float64) * 0.3
    rel = Jl.clone()
    rel[1:] = Jl[1:] - Jl[torch.as_tensor(par[1:], dtype=torch.long)]
    rel = rel.requires_grad_(True)
    return par, aa, rel


def test_pointer_jumping_equals_the_sequential_chain():
    par, aa, rel = _problem(1)
    J = len(par)
    R = _rodrigues(aa)
    GR, Gt = _chain_sequential(R, rel, par)
    # rounds: jump[0] = parent, jump[r+1][j] = jump[r][jump[r][j]]
    depth = np.zeros(J, int)
    for j in range(1, J):
        depth[j] = depth[par[j]] + 1
    rounds = int(np.ceil(np.log2(depth.max() + 1)))
    assert rounds == 4                                            # SMPL-X: 11 levels
    PR, Pt = R.detach().clone(), rel.detach().clone()
    anc = par.copy()
    for _ in range(rounds):
        nR, nt = PR.clone(), Pt.clone()
        for j in range(J):
            a = anc[j]
            if a >= 0:
                nR[j] = PR[a] @ PR[j]
                nt[j] = PR[a] @ Pt[j] + Pt[a]
        anc = np.array([anc[a] if a >= 0 else -1 for a in anc])
        PR, Pt = nR, nt
    assert (anc < 0).all()
    assert torch.allclose(PR, GR.detach(), atol=1e-12) and torch.allclose(Pt, Gt.detach(), atol=1e-12)


def test_subtree_sum_gradient_equals_autograd_through_the_chain():
    par, aa, rel = _problem(2)
    J = len(par)
    R = _rodrigues(aa)
    R.retain_grad()
    GR, Gt = _chain_sequential(R, rel, par)
    g = torch.Generator().manual_seed(5)
    gGR, gGt = torch.randn(J, 3, 3, generator=g, dtype=torch.float64), torch.randn(J, 3, generator=g, dtype=torch.float64)
    ((GR * gGR).sum() + (Gt * gGt).sum()).backward()             # gGR / gGt play the role of gown(G_d)
    GRd, Gtd = GR.detach(), Gt.detach()
    w = gGt
    U = gGR @ GRd.transpose(1, 2) + w[:, :, None] * Gtd[:, None, :]
    sub = np.zeros((J, J), bool)                                  # sub[j, d]: d in subtree(j)
    for d in range(J):
        a = d
        while a >= 0:
            sub[a, d] = True
            a = par[a]
    S = torch.as_tensor(sub, dtype=torch.float64)
    SU, Sw = torch.einsum('jd,dab->jab', S, U), S @ w
    accR = (SU - Sw[:, :, None] * Gtd[:, None, :]) @ GRd
    acct = Sw
    for j in range(J):
        p = par[j]
        PRt = torch.eye(3, dtype=torch.float64) if p < 0 else GRd[p].T
        assert torch.allclose(PRt @ accR[j], R.grad[j], atol=1e-10), j
        assert torch.allclose(PRt @ acct[j], rel.grad[j], atol=1e-10), j
