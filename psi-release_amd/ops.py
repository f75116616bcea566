"""Autograd operators over the C ABI (include/psi_hip.h).  Same call signatures as the reference's ops.

* ``chamferFunction`` / ``chamferDist``   <- chamfer_pytorch/dist_chamfer.py:13-53 (+ idx variant, dist_chamfer_idx.py)
* ``sdf_sample``                          <- F.grid_sample call of fitting_proxe.py:144-151
* ``penetration_loss``                    <- fitting_proxe.py:155-158 (no host sync)

Inputs must live on the GPU; there is no CPU path (hip.ptr raises).
"""
from __future__ import annotations

import torch
from torch import nn
from torch.autograd import Function


from . import hip


# ------------------------------------------------------------------------------------------
# Chamfer
# ------------------------------------------------------------------------------------------
def chamfer_forward_raw(xyz1, xyz2, both=True):
    """chamfer.forward (chamfer_cuda.cpp:17-19): returns dist1, idx1, dist2, idx2 (None, None when both=False)."""
    xyz1 = xyz1.contiguous().float()
    xyz2 = xyz2.contiguous().float()
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    if xyz2.shape[0] != B or xyz1.shape[2] != 3 or xyz2.shape[2] != 3:
        raise ValueError('expected xyz1 [B,n,3] and xyz2 [B,m,3]')
    dev = xyz1.device
    dist1 = torch.zeros(B, n, device=dev)
    idx1 = torch.zeros(B, n, dtype=torch.int32, device=dev)
    dist2 = torch.zeros(B, m, device=dev) if both else None
    idx2 = torch.zeros(B, m, dtype=torch.int32, device=dev) if both else None
    L = hip.lib()
    nbytes = L.psi_chamfer_workspace_bytes(B, n, m)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=dev)
    hip.check(L.psi_chamfer_forward(hip.ptr(xyz1), hip.ptr(xyz2), B, n, m, hip.ptr(dist1), hip.ptr(idx1),
                                    hip.ptr(dist2), hip.ptr(idx2), hip.ptr(ws), hip.stream()), 'psi_chamfer_forward')
    return dist1, idx1, dist2, idx2


class chamferFunction(Function):
    """dist_chamfer.py:13-46.  ``return_idx`` selects the dist_chamfer_idx.py:28 output tuple."""

    @staticmethod
    def forward(ctx, xyz1, xyz2, return_idx=False, both=True):
        xyz1 = xyz1.contiguous()
        xyz2 = xyz2.contiguous()
        dist1, idx1, dist2, idx2 = chamfer_forward_raw(xyz1, xyz2, both)
        ctx.both = both
        if both:
            ctx.save_for_backward(xyz1, xyz2, idx1, idx2)
        else:
            ctx.save_for_backward(xyz1, xyz2, idx1)
            dist2 = torch.zeros(xyz2.shape[0], xyz2.shape[1], device=xyz1.device)
            idx2 = torch.zeros(xyz2.shape[0], xyz2.shape[1], dtype=torch.int32, device=xyz1.device)
        ctx.mark_non_differentiable(idx1, idx2)
        if return_idx:
            return dist1, dist2, idx1, idx2
        return dist1, dist2

    @staticmethod
    def backward(ctx, graddist1, graddist2, *unused):
        if ctx.both:
            xyz1, xyz2, idx1, idx2 = ctx.saved_tensors
        else:
            xyz1, xyz2, idx1 = ctx.saved_tensors
            idx2 = None
        B, n, _ = xyz1.shape
        m = xyz2.shape[1]
        graddist1 = graddist1.contiguous()
        need2 = ctx.needs_input_grad[1]
        use_dir2 = ctx.both and graddist2 is not None
        gradxyz1 = torch.zeros_like(xyz1)
        gradxyz2 = torch.zeros_like(xyz2) if (need2 or use_dir2) else None
        g2 = graddist2.contiguous() if use_dir2 else None
        hip.check(hip.lib().psi_chamfer_backward(hip.ptr(xyz1), hip.ptr(xyz2), hip.ptr(gradxyz1), hip.ptr(gradxyz2),
                                                 hip.ptr(graddist1), hip.ptr(g2), hip.ptr(idx1),
                                                 hip.ptr(idx2) if use_dir2 else None, B, n, m, hip.stream()),
                  'psi_chamfer_backward')
        return gradxyz1, (gradxyz2 if need2 else None), None, None


class chamferDist(nn.Module):
    """dist_chamfer.py:48-53: ``chamferDist()(xyz1, xyz2) -> (dist1, dist2)``.

    ``one_sided=True`` skips the scene->body direction that PSI discards (fitting_proxe.py:136);
    ``dist2`` is then returned as zeros."""

    def __init__(self, return_idx: bool = False, one_sided: bool = False):
        super().__init__()
        self.return_idx = return_idx
        self.one_sided = one_sided

    def forward(self, input1, input2):
        return chamferFunction.apply(input1, input2, self.return_idx, not self.one_sided)


# ------------------------------------------------------------------------------------------
# SDF lookup
# ------------------------------------------------------------------------------------------
class _SdfSample(Function):
    @staticmethod
    def forward(ctx, verts, sdf, scene_id, gmin, gmax, align_corners):
        verts = verts.contiguous().float()
        B, V, _ = verts.shape
        S, D = sdf.shape[0], sdf.shape[1]
        out = torch.empty(B, V, device=verts.device)
        og = torch.empty(B, V, 3, device=verts.device)
        hip.check(hip.lib().psi_sdf_sample_forward(hip.ptr(sdf), hip.ptr(scene_id), hip.ptr(gmin), hip.ptr(gmax),
                                                   hip.ptr(verts), B, V, D, S, int(bool(align_corners)), hip.ptr(out),
                                                   hip.ptr(og), hip.stream()), 'psi_sdf_sample_forward')
        ctx.save_for_backward(og)
        return out

    @staticmethod
    def backward(ctx, gout):
        (og,) = ctx.saved_tensors
        B, V, _ = og.shape
        gv = torch.zeros_like(og)
        hip.check(hip.lib().psi_sdf_sample_backward(hip.ptr(gout.contiguous()), hip.ptr(og), B, V, hip.ptr(gv),
                                                    hip.stream()), 'psi_sdf_sample_backward')
        return gv, None, None, None, None, None


def sdf_sample(verts, sdf, grid_min, grid_max, scene_id=None, align_corners=True):
    """Trilinear SDF values [B,V] of world-space ``verts`` [B,V,3].

    sdf [S,D,D,D] (or [D,D,D]) holds ONE volume per scene, ``scene_id`` [B] int32 picks it (None = scene 0);
    grid_min / grid_max [S,3] (or [3]).  Equivalent to fitting_proxe.py:144-151 with the volume not replicated."""
    if sdf.dim() == 3:
        sdf = sdf.unsqueeze(0)
    sdf = sdf.contiguous().float()
    S = sdf.shape[0]
    gmin = grid_min.reshape(-1, 3).contiguous().float()
    gmax = grid_max.reshape(-1, 3).contiguous().float()
    if gmin.shape[0] != S or gmax.shape[0] != S:
        raise ValueError('grid_min/grid_max must have one row per scene volume')
    if scene_id is not None:
        scene_id = scene_id.to(device=verts.device, dtype=torch.int32).contiguous()
    return _SdfSample.apply(verts, sdf, scene_id, gmin, gmax, align_corners)


class _PenLoss(Function):
    """mean |sdf| over the negative entries of the whole batch, 0 when there are none (fitting_proxe.py:155-158)."""

    @staticmethod
    def forward(ctx, vals):
        vals = vals.contiguous()
        stats = torch.zeros(2, device=vals.device)
        hip.check(hip.lib().psi_sdf_penetration_stats(hip.ptr(vals), vals.numel(), hip.ptr(stats), hip.stream()),
                  'psi_sdf_penetration_stats')
        ctx.save_for_backward(vals, stats)
        return torch.where(stats[1] > 0, stats[0] / stats[1].clamp(min=1.0), torch.zeros((), device=vals.device))

    @staticmethod
    def backward(ctx, g):
        vals, stats = ctx.saved_tensors
        scale = torch.where(stats[1] > 0, -g / stats[1].clamp(min=1.0), torch.zeros((), device=vals.device))
        return (vals < 0).to(vals.dtype) * scale


def penetration_loss(body_sdf):
    return _PenLoss.apply(body_sdf)


# ------------------------------------------------------------------------------------------
# Exact NN index over a static scene cloud
# ------------------------------------------------------------------------------------------
class SceneNNIndex:
    """kd-tree over one scene's points (psi_nn_index_*).  ``query`` returns exactly what ``chamfer.forward`` returns for
    the body->scene direction against that cloud (bit-identical dist1 / idx1)."""

    def __init__(self, points, device='cuda'):
        import ctypes
        import numpy as np
        pts = points.detach().cpu().numpy() if torch.is_tensor(points) else np.asarray(points)
        pts = np.ascontiguousarray(pts.reshape(-1, 3), dtype=np.float32)
        self.m = pts.shape[0]
        self.device = torch.device(device)
        self.points = torch.tensor(pts, device=self.device)           # for the backward gather
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            hip.check(hip.lib().psi_nn_index_create(ctypes.byref(h), pts.ctypes.data_as(ctypes.c_void_p), self.m),
                      'psi_nn_index_create')
        self.handle = h

    def query(self, xyz1, hint=None):
        """hint: int32 [B,n] warm-start buffer (previous winners, -1 = none), updated in place; results do not depend on it."""
        xyz1 = xyz1.contiguous().float()
        B, n, _ = xyz1.shape
        dist = torch.empty(B, n, device=xyz1.device)
        idx = torch.empty(B, n, dtype=torch.int32, device=xyz1.device)
        hip.check(hip.lib().psi_nn_index_query(self.handle, hip.ptr(xyz1), B, n, hip.ptr(dist), hip.ptr(idx), hip.ptr(hint),
                                               hip.stream()), 'psi_nn_index_query')
        return dist, idx

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                hip.lib().psi_nn_index_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _IndexedChamfer(Function):
    @staticmethod
    def forward(ctx, xyz1, index):
        dist, idx = index.query(xyz1)
        ctx.index = index
        ctx.save_for_backward(xyz1, idx)
        return dist

    @staticmethod
    def backward(ctx, gdist):
        xyz1, idx = ctx.saved_tensors
        nn = ctx.index.points[idx.long()]                                  # [B,n,3]
        return 2.0 * gdist.unsqueeze(-1) * (xyz1 - nn), None               # chamfer.cu:155-174, query side


def chamfer_to_scene(xyz1, index: SceneNNIndex):
    """dist1 [B,n] of ``chamferDist()(xyz1, scene.repeat(B))`` through the scene's exact NN index."""
    return _IndexedChamfer.apply(xyz1, index)


class SceneSet:
    """The static scenes of a training set: one exact NN index per scene slot, queried in ONE launch with the per-body
    scene slot (psi_nn_index_set_query).  ``chamfer_to_scenes`` equals ``chamferDist()(xyz1, verts_table[slot])[0]`` bit for
    bit (train_s1.py:159-169) without the O(n*m) scan or the [B,m,3] gather, and its launch shape does not depend on
    which scenes a batch mixes (so a whole training step can live in one HIP graph)."""

    def __init__(self, verts_table, device='cuda'):
        import ctypes
        self.verts_table = verts_table.contiguous()                     # [S,m,3] device (backward gather)
        self.device = torch.device(device)
        self.indices = [SceneNNIndex(self.verts_table[s], self.device) for s in range(self.verts_table.shape[0])]
        arr = (ctypes.c_void_p * len(self.indices))(*[ix.handle.value for ix in self.indices])
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            hip.check(hip.lib().psi_nn_index_set_create(ctypes.byref(h), arr, len(self.indices)), 'psi_nn_index_set_create')
        self.handle = h

    def query(self, xyz1, slot):
        xyz1 = xyz1.contiguous().float()
        B, n, _ = xyz1.shape
        dist = torch.empty(B, n, device=xyz1.device)
        idx = torch.empty(B, n, dtype=torch.int32, device=xyz1.device)
        hip.check(hip.lib().psi_nn_index_set_query(self.handle, hip.ptr(slot), hip.ptr(xyz1), B, n, hip.ptr(dist), hip.ptr(idx),
                                                   hip.stream()), 'psi_nn_index_set_query')
        return dist, idx

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                hip.lib().psi_nn_index_set_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class _SceneSetChamfer(Function):
    @staticmethod
    def forward(ctx, xyz1, scenes, slot):
        dist, idx = scenes.query(xyz1, slot)
        ctx.scenes = scenes
        ctx.save_for_backward(xyz1, idx, slot)
        return dist

    @staticmethod
    def backward(ctx, gdist):
        xyz1, idx, slot = ctx.saved_tensors
        nn = ctx.scenes.verts_table[slot.long().unsqueeze(1), idx.long()]     # [B,n,3]
        return 2.0 * gdist.unsqueeze(-1) * (xyz1 - nn), None, None            # chamfer.cu:155-174, query side


def chamfer_to_scenes(xyz1, scenes: SceneSet, slot):
    """dist1 [B,n] of every body against its own scene.  slot: int32 device [B]."""
    assert slot.dtype == torch.int32 and slot.is_contiguous()
    return _SceneSetChamfer.apply(xyz1, scenes, slot)


# ------------------------------------------------------------------------------------------
# Dense layers on the matrix cores (csrc/linear.hip): y = act(x W^T + b) (+ residual), bf16 MFMA with fp32 accumulation
# ------------------------------------------------------------------------------------------
class _LinearAct(Function):
    """nn.Linear (+ LeakyReLU, + skip connection) as ONE hand-written bf16-MFMA kernel per direction; replaces, on the bf16 path of the
    CVAEs, the cast / GEMM / bias-activation / add kernels PyTorch launches per layer (net_layers.py:28-43, cvae.py:474-492)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, slope):
        if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
            raise ValueError('linear_act: x [M,K] and weight [N,K] expected, got %s / %s' % (tuple(x.shape), tuple(weight.shape)))
        M, K = x.shape
        N = weight.shape[0]
        if K % 16 or N % 16:
            raise ValueError('linear_act: K and N must be multiples of 16 (got K=%d, N=%d)' % (K, N))
        xb = x.dtype == torch.bfloat16
        xc = x.detach().contiguous() if xb else x.detach().contiguous().float()
        w = weight.detach().contiguous().float()
        b = bias.detach().contiguous().float() if bias is not None else None
        r = residual.detach().contiguous().float() if residual is not None else None
        y = torch.empty(M, N, device=x.device)
        a_out = torch.empty(M, N, device=x.device) if (act and residual is not None) else None
        L = hip.lib()
        nws = L.psi_linear_workspace_floats(M, N, K)
        ws = torch.empty(nws, device=x.device) if nws else None
        hip.check(L.psi_linear_forward(hip.ptr(xc), int(xb), hip.ptr(w), hip.ptr(b), hip.ptr(r), M, N, K, int(act), float(slope), hip.ptr(y),
                                       hip.ptr(a_out), hip.ptr(ws), hip.stream()), 'psi_linear_forward')
        ctx.act, ctx.slope, ctx.xb = int(act), float(slope), xb
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        # without a residual the output itself carries the sign of the pre-activation
        ctx.save_for_backward(xc, w, (a_out if a_out is not None else y) if act else torch.empty(0, device=x.device))
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, w, a_out = ctx.saved_tensors
        M, K = xc.shape
        N = w.shape[0]
        gy = gy.contiguous().float()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        # (dX, dW + dbias as ONE hand-written launch pair with LDS-staged transposed operands: 26 vs 38 us for a 128 x 512 x 512 layer against the
        # library route of mask, cast, two GEMMs, column sum — tests/library_paths.py keeps that route for comparison)
        gx = torch.empty(M, K, device=gy.device, dtype=torch.bfloat16 if ctx.xb else torch.float32) if need_x else None
        gw = torch.empty(N, K, device=gy.device) if (need_w or need_b) else None         # gbias is produced by the dW kernel
        gb = torch.empty(N, device=gy.device) if need_b else None
        nws = hip.lib().psi_linear_backward_workspace_floats(M, N, K) if need_x else 0
        ws = torch.empty(nws, device=gy.device) if nws else None
        hip.check(hip.lib().psi_linear_backward(hip.ptr(gy), hip.ptr(a_out) if ctx.act else None, hip.ptr(xc), int(ctx.xb), hip.ptr(w), M, N, K,
                                                ctx.slope, hip.ptr(gx), hip.ptr(gw), hip.ptr(gb), hip.ptr(ws), hip.stream()), 'psi_linear_backward')
        return gx, gw if need_w else None, gb, gy if (ctx.has_res and ctx.needs_input_grad[3]) else None, None, None


def linear_act(x, weight, bias=None, act=None, slope=0.01, residual=None):
    """act(x @ weight.T + bias) (+ residual) on the HIP bf16-MFMA kernels; ``act`` is None or 'leaky_relu'.  x: [M,K] fp32 or bf16."""
    if act not in (None, 'leaky_relu'):
        raise ValueError('act must be None or "leaky_relu"')
    return _LinearAct.apply(x, weight, bias, residual, 1 if act else 0, slope)


# ------------------------------------------------------------------------------------------------------------------
# BatchNorm (training statistics) + ReLU + skip connection of the scene trunk: psi_bn_forward / psi_bn_backward
# ------------------------------------------------------------------------------------------------------------------
def _ptr_cl(t):
    """Device pointer of a channels_last (NHWC in memory) 4-D tensor (None -> NULL)."""
    if t is None:
        return None
    if not (t.is_cuda and t.is_contiguous(memory_format=torch.channels_last)):
        raise hip.PsiHipError('expected a channels_last GPU tensor')
    return t.data_ptr()


def _bn_mask_from_x():
    """True: a BatchNorm + ReLU layer without a skip connection takes its backward's ReLU mask from x and two per-channel numbers instead of
    reading the stored output (one map less per pass, bit-identical gradients; the tests patch this to False to check exactly that)"""
    return True


class _BNAct(Function):
    """y = act(batch_norm(x) (+ residual)) on NHWC bf16 maps, batch statistics (include/psi_hip.h: psi_bn_forward)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn, relu):
        N, C, H, W = x.shape
        M = N * H * W
        xc = x.contiguous(memory_format=torch.channels_last)
        rc = residual.contiguous(memory_format=torch.channels_last) if residual is not None else None
        y = torch.empty_like(xc, memory_format=torch.channels_last)
        mean = torch.empty(C, device=x.device, dtype=torch.float32)
        invstd = torch.empty(C, device=x.device, dtype=torch.float32)
        L = hip.lib()
        ws = torch.empty(L.psi_bn_workspace_floats(M, C), device=x.device, dtype=torch.float32)
        track = bn.track_running_stats and bn.running_mean is not None
        hip.check(L.psi_bn_forward(_ptr_cl(xc), _ptr_cl(rc), hip.ptr(weight), hip.ptr(bias),
                                   hip.ptr(bn.running_mean) if track else None, hip.ptr(bn.running_var) if track else None,
                                   hip.ptr(bn.num_batches_tracked) if track else None, M, C, int(relu),
                                   float(bn.momentum if bn.momentum is not None else 0.1), float(bn.eps), _ptr_cl(y), hip.ptr(mean),
                                   hip.ptr(invstd), hip.ptr(ws), hip.stream()), 'psi_bn_forward')
        # a ReLU layer without a skip connection: the backward recomputes the mask from x (one map less per pass) and y is not kept for it
        keep_y = relu and (residual is not None or not _bn_mask_from_x())
        ctx.save_for_backward(xc, y if keep_y else None, weight, bias, mean, invstd)
        ctx.relu, ctx.has_res, ctx.dims = bool(relu), residual is not None, (M, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, y, weight, bias, mean, invstd = ctx.saved_tensors
        M, C = ctx.dims
        dyc = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(xc, memory_format=torch.channels_last)
        dres = torch.empty_like(xc, memory_format=torch.channels_last) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        dgamma = torch.empty(C, device=xc.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=xc.device, dtype=torch.float32)
        L = hip.lib()
        ws = torch.empty(L.psi_bn_workspace_floats(M, C), device=xc.device, dtype=torch.float32)
        hip.check(L.psi_bn_backward_t(_ptr_cl(dyc), 0, _ptr_cl(xc), _ptr_cl(y), hip.ptr(weight), hip.ptr(bias.detach().float()), hip.ptr(mean),
                                      hip.ptr(invstd), M, C, int(ctx.relu), _ptr_cl(dx), _ptr_cl(dres), hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(ws),
                                      hip.stream()), 'psi_bn_backward_t')
        return dx, dgamma, dbeta, dres, None, None


def bn_act(x, bn, relu=True, residual=None):
    """``relu(bn(x) + residual)`` (each part optional) of an ``nn.BatchNorm2d`` in TRAINING mode on a bf16 channels_last map, as one
    fused HIP op (statistics pass + one apply pass; the backward likewise) instead of the library's three BN launches plus separate
    ReLU / add launches.  Updates ``bn.running_mean / running_var / num_batches_tracked`` like the module would."""
    if x.dtype != torch.bfloat16 or not x.is_cuda or x.dim() != 4:
        raise ValueError('bn_act: expected a 4-D bf16 CUDA tensor')
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape):
        raise ValueError('bn_act: residual must match x')
    return _BNAct.apply(x, bn.weight, bn.bias, residual, bn, relu)


class _MaxPool3x3s2(Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        xc = x.contiguous(memory_format=torch.channels_last)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, OH, OW), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        idx = torch.empty(N * OH * OW * C, device=x.device, dtype=torch.uint8)
        hip.check(hip.lib().psi_maxpool3x3s2_forward(_ptr_cl(xc), N, H, W, C, _ptr_cl(y), hip.ptr(idx), hip.stream()), 'psi_maxpool3x3s2_forward')
        ctx.save_for_backward(idx)
        ctx.dims = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        N, C, H, W = ctx.dims
        dyc = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=dy.dtype, memory_format=torch.channels_last)
        hip.check(hip.lib().psi_maxpool3x3s2_backward(_ptr_cl(dyc), hip.ptr(idx), N, H, W, C, _ptr_cl(dx), hip.stream()), 'psi_maxpool3x3s2_backward')
        return dx


def maxpool3x3s2(x):
    """``nn.MaxPool2d(kernel_size=3, stride=2, padding=1)`` (the trunk's stem) on a bf16 channels_last map: one gather pass each way."""
    if x.dtype != torch.bfloat16 or not x.is_cuda or x.dim() != 4 or x.shape[1] % 8:
        raise ValueError('maxpool3x3s2: expected a 4-D bf16 CUDA tensor with a multiple of 8 channels')
    return _MaxPool3x3s2.apply(x)


# ------------------------------------------------------------------------------------------------------------------
# 3x3 / stride 1 / padding 1 convolutions of the trunk on the hand-written implicit-GEMM kernel (csrc/conv.hip)
# ------------------------------------------------------------------------------------------------------------------
class _Conv3x3(Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        N, Cin, H, W = x.shape
        Cout = weight.shape[0]
        xc = x.contiguous(memory_format=torch.channels_last)
        w4 = weight.detach().permute(0, 2, 3, 1)                                           # [Cout,3,3,Cin]: a view for a channels_last weight
        wt = None
        if w4.dtype == torch.float32 and w4.is_contiguous():
            # master weight -> bf16, forward layout and (when the input needs a gradient) the rotated layout of the backward: one launch
            wb = torch.empty(Cout, 3, 3, Cin, device=x.device, dtype=torch.bfloat16)
            if ctx.needs_input_grad[0] and hip.lib().psi_conv3x3_supported(Cout, Cin, H, W):
                wt = torch.empty(Cin, 3, 3, Cout, device=x.device, dtype=torch.bfloat16)
            hip.check(hip.lib().psi_conv3x3_prepare_weight(hip.ptr(w4), Cin, Cout, hip.ptr(wb), hip.ptr(wt), hip.stream()), 'psi_conv3x3_prepare_weight')
        else:
            wb = w4.contiguous().to(torch.bfloat16)
        y = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
        b = bias.detach().float().contiguous() if bias is not None else None
        hip.check(hip.lib().psi_conv3x3_forward(_ptr_cl(xc), hip.ptr(wb), hip.ptr(b), N, H, W, Cin, Cout, _ptr_cl(y), hip.stream()),
                  'psi_conv3x3_forward')
        ctx.save_for_backward(xc, wb, wt)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, wb, wt = ctx.saved_tensors
        N, Cin, H, W = xc.shape
        Cout = wb.shape[0]
        dyc = dy.contiguous(memory_format=torch.channels_last)
        L = hip.lib()
        dx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if L.psi_conv3x3_supported(Cout, Cin, H, W):
                if wt is None:
                    wt = torch.empty(Cin, 3, 3, Cout, device=dy.device, dtype=torch.bfloat16)
                    hip.check(L.psi_conv3x3_rotate_weight(hip.ptr(wb), Cin, Cout, hip.ptr(wt), hip.stream()), 'psi_conv3x3_rotate_weight')
                dx = torch.empty((N, Cin, H, W), device=dy.device, dtype=torch.bfloat16, memory_format=torch.channels_last)
                hip.check(L.psi_conv3x3_forward(_ptr_cl(dyc), hip.ptr(wt), None, N, H, W, Cout, Cin, _ptr_cl(dx), hip.stream()),
                          'psi_conv3x3_forward (input gradient)')
            else:
                dx = torch.ops.aten.convolution_backward(dyc, xc, wb.permute(0, 3, 1, 2), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            nws = L.psi_conv3x3_wrw_workspace_floats(N, H, W, Cin, Cout)
            if nws:
                # hand-written split-K weight gradient (fp32, deterministic summation order)
                gw4 = torch.empty(Cout, 3, 3, Cin, device=dy.device, dtype=torch.float32)
                ws = torch.empty(nws, device=dy.device, dtype=torch.float32)
                hip.check(L.psi_conv3x3_weight_grad(_ptr_cl(xc), _ptr_cl(dyc), N, H, W, Cin, Cout, hip.ptr(gw4), hip.ptr(ws), hip.stream()),
                          'psi_conv3x3_weight_grad')
                gw = gw4.permute(0, 3, 1, 2)                         # [Cout,Cin,3,3] view (channels_last strides, like the parameter)
            else:
                gw = torch.ops.aten.convolution_backward(dyc, xc, wb.permute(0, 3, 1, 2), None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1].float()
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = dyc.float().sum((0, 2, 3))
        return dx, gw, gb


def conv3x3_supported(conv, x):
    """True when ``conv`` (an nn.Conv2d) applied to the bf16 CUDA map ``x`` is covered by the hand-written kernel."""
    return (x.is_cuda and x.dim() == 4 and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros'
            and bool(hip.lib().psi_conv3x3_supported(conv.in_channels, conv.out_channels, x.shape[2], x.shape[3])))


def conv3x3(x, conv):
    """``conv(x)`` for a 3x3 / stride 1 / padding 1 ``nn.Conv2d`` on a bf16 map: hand-written bf16-MFMA implicit GEMM, forward and input
    gradient (csrc/conv.hip); output bf16 channels_last."""
    if x.dtype != torch.bfloat16:
        x = x.to(torch.bfloat16)
    return _Conv3x3.apply(x, conv.weight, conv.bias)


# ------------------------------------------------------------------------------------------------------------------
# The scene encoders and dense layers at the REFERENCE'S PRECISION (fp32 models, cvae.py:427-455,474-492; net_layers.py:28-43,56-93) on
# hand-written kernels: fp32 NHWC maps, every product a three-term bf16 split with fp32 accumulation (csrc/conv_gemm.hip explains and
# quantifies it: 0.6-3.2e-5 of the reference's recorded forward passes, tests/golden/cvae.npz).  Forward: psi_conv2d_forward (ALL the
# convolutions: 7x7 stem, stride-1 / stride-2 3x3, 1x1 downsample, heads), psi_bn_forward_t (batch or running statistics, + ReLU + skip),
# psi_maxpool3x3s2_forward_t, psi_linear_forward3 (any width).  Backward: BatchNorm and max-pool on the same hand-written kernels, the
# convolutions' input and weight gradients on psi_conv2d_input_grad / psi_conv2d_weight_grad and the dense layers' on psi_linear_backward3
# (the same split products).
# ------------------------------------------------------------------------------------------------------------------
def conv2d_supported(conv):
    return (conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == 'zeros' and conv.kernel_size[0] == conv.kernel_size[1]
            and conv.stride[0] == conv.stride[1] and conv.padding[0] == conv.padding[1]
            and bool(hip.lib().psi_conv2d_supported(conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.kernel_size[1], conv.stride[0],
                                                    conv.padding[0])))


def _conv2d_prepared_ok(Cin, Cout, KH, KW, stride, pad):
    """shapes whose weight is split / re-laid out ONCE per layer and step (psi_conv2d_prepare_weight) instead of in every workgroup"""
    return bool(hip.lib().psi_conv2d_prepared_ok(Cin, Cout, KH, KW, stride, pad))


def _conv2d_dgrad_covered(Cin, Cout, KH, KW):
    """shapes whose input gradient the general kernel takes in its transposed-gather form (psi_hip.h: psi_conv2d_input_grad)"""
    return Cin % 32 == 0 and (Cout % 64 == 0 or Cout * KH * KW <= 4096)


class _Conv2dSplit(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, nterm, out_bf16):
        N, Cin, H, W = x.shape
        Cout, _, KH, KW = weight.shape
        xc = x.contiguous(memory_format=torch.channels_last)
        w4 = weight.detach().permute(0, 2, 3, 1)                     # [Cout,KH,KW,Cin]: a view of a channels_last weight, else a copy
        w4 = (w4 if w4.is_contiguous() else w4.contiguous()).float()
        OH, OW = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
        y = torch.empty((N, Cout, OH, OW), device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32, memory_format=torch.channels_last)
        b = bias.detach().float().contiguous() if bias is not None else None
        L = hip.lib()
        wt = None
        if _conv2d_prepared_ok(Cin, Cout, KH, KW, stride, pad):
            # the weight's bf16 parts once per layer and step, in the forward's layout and (when x needs a gradient) the input gradient's
            nel = Cout * KH * KW * Cin * (2 if nterm == 3 else 1)
            wf = torch.empty(nel, device=x.device, dtype=torch.bfloat16)
            if ctx.needs_input_grad[0] and _conv2d_dgrad_covered(Cin, Cout, KH, KW):
                wt = torch.empty(nel, device=x.device, dtype=torch.bfloat16)
            hip.check(L.psi_conv2d_prepare_weight(hip.ptr(w4), Cout, KH, KW, Cin, nterm, hip.ptr(wf), hip.ptr(wt), hip.stream()), 'psi_conv2d_prepare_weight')
            hip.check(L.psi_conv2d_forward_p(_ptr_cl(xc), int(xc.dtype == torch.bfloat16), hip.ptr(wf), hip.ptr(b), N, H, W, Cin, Cout, KH, KW, stride, pad,
                                             _ptr_cl(y), int(out_bf16), nterm, hip.stream()), 'psi_conv2d_forward_p')
        else:
            hip.check(L.psi_conv2d_forward(_ptr_cl(xc), int(xc.dtype == torch.bfloat16), hip.ptr(w4), hip.ptr(b), N, H, W, Cin, Cout, KH, KW,
                                           stride, pad, _ptr_cl(y), int(out_bf16), nterm, hip.stream()), 'psi_conv2d_forward')
        ctx.save_for_backward(xc, weight, wt)
        ctx.geom = (stride, pad, bias is not None, nterm)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, weight, wt_prepared = ctx.saved_tensors
        stride, pad, has_bias, nterm = ctx.geom
        N, Cin, H, W = xc.shape
        Cout, _, KH, KW = weight.shape
        dyc = dy.contiguous(memory_format=torch.channels_last)
        if dyc.dtype not in (torch.float32, torch.bfloat16):
            dyc = dyc.float()
        L = hip.lib()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if _conv2d_dgrad_covered(Cin, Cout, KH, KW):
                gx = torch.empty((N, Cin, H, W), device=dy.device, dtype=xc.dtype, memory_format=torch.channels_last)
                if wt_prepared is not None:
                    hip.check(L.psi_conv2d_input_grad_p(_ptr_cl(dyc), int(dyc.dtype == torch.bfloat16), hip.ptr(wt_prepared), N, H, W, Cin, Cout, KH, KW,
                                                        stride, pad, _ptr_cl(gx), int(gx.dtype == torch.bfloat16), nterm, hip.stream()),
                              'psi_conv2d_input_grad_p')
                else:
                    wt = weight.detach().float().permute(1, 2, 3, 0).contiguous()        # [Cin,KH,KW,Cout]
                    hip.check(L.psi_conv2d_input_grad(_ptr_cl(dyc), int(dyc.dtype == torch.bfloat16), hip.ptr(wt), N, H, W, Cin, Cout, KH, KW, stride,
                                                      pad, _ptr_cl(gx), int(gx.dtype == torch.bfloat16), nterm, hip.stream()), 'psi_conv2d_input_grad')
            else:
                gx = torch.ops.aten.convolution_backward(dyc.to(xc.dtype), xc, weight.detach().to(xc.dtype), None, (stride, stride), (pad, pad), (1, 1),
                                                         False, (0, 0), 1, (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            gw4 = torch.empty(Cout, KH, KW, Cin, device=dy.device, dtype=torch.float32)
            ws = torch.empty(L.psi_conv2d_wgrad_workspace_floats(N, H, W, Cin, Cout, KH, KW, stride, pad), device=dy.device, dtype=torch.float32)
            hip.check(L.psi_conv2d_weight_grad(_ptr_cl(xc), int(xc.dtype == torch.bfloat16), _ptr_cl(dyc), int(dyc.dtype == torch.bfloat16), N, H, W, Cin,
                                               Cout, KH, KW, stride, pad, hip.ptr(gw4), hip.ptr(ws), nterm, hip.stream()), 'psi_conv2d_weight_grad')
            gw = gw4.permute(0, 3, 1, 2)                             # [Cout,Cin,KH,KW] view with channels_last strides, like the parameter
        if has_bias and ctx.needs_input_grad[2]:
            gb = dyc.float().sum((0, 2, 3))
        return gx, gw, gb, None, None, None, None


def conv2d_split(x, conv, nterm=3, out_bf16=False):
    """``conv(x)`` for an ``nn.Conv2d`` on an fp32 (or bf16) channels_last map through the general implicit-GEMM kernel: nterm = 3 = the fp32
    model's precision (three-term split products), nterm = 1 = bf16 products."""
    return _Conv2dSplit.apply(x, conv.weight, conv.bias, conv.stride[0], conv.padding[0], nterm, out_bf16)


class _BNActT(Function):
    """y = act(batch_norm(x) (+ residual)) on NHWC maps of either type, batch statistics (training) or running statistics (eval)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, bn, relu):
        N, C, H, W = x.shape
        M = N * H * W
        f32 = x.dtype == torch.float32
        xc = x.contiguous(memory_format=torch.channels_last)
        rc = residual.contiguous(memory_format=torch.channels_last) if residual is not None else None
        y = torch.empty_like(xc, memory_format=torch.channels_last)
        evalm = not bn.training
        mean = torch.empty(C, device=x.device, dtype=torch.float32) if not evalm else None
        invstd = torch.empty(C, device=x.device, dtype=torch.float32) if not evalm else None
        L = hip.lib()
        ws = torch.empty(L.psi_bn_workspace_floats(M, C), device=x.device, dtype=torch.float32)
        track = bn.track_running_stats and bn.running_mean is not None
        hip.check(L.psi_bn_forward_t(_ptr_cl(xc), int(f32), _ptr_cl(rc), hip.ptr(weight), hip.ptr(bias),
                                     hip.ptr(bn.running_mean) if track else None, hip.ptr(bn.running_var) if track else None,
                                     hip.ptr(bn.num_batches_tracked) if (track and not evalm) else None, M, C, int(relu),
                                     float(bn.momentum if bn.momentum is not None else 0.1), float(bn.eps), _ptr_cl(y), hip.ptr(mean),
                                     hip.ptr(invstd), hip.ptr(ws), int(evalm), hip.stream()), 'psi_bn_forward_t')
        if evalm:
            # inference form: the gradient is that of an affine map with the running statistics (only needed if someone differentiates an
            # eval-mode model: kept correct through the saved statistics)
            mean, invstd = bn.running_mean.detach().float(), torch.rsqrt(bn.running_var.detach().float() + bn.eps)
        keep_y = relu and (evalm or residual is not None or not _bn_mask_from_x())
        ctx.save_for_backward(xc, y if keep_y else None, weight, bias, mean, invstd)
        ctx.relu, ctx.has_res, ctx.dims, ctx.evalm, ctx.f32 = bool(relu), residual is not None, (M, C), evalm, f32
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, y, weight, bias, mean, invstd = ctx.saved_tensors
        M, C = ctx.dims
        dyc = dy.contiguous(memory_format=torch.channels_last).to(xc.dtype)
        if ctx.evalm:
            dz = dyc if not ctx.relu else dyc * (y > 0).to(dyc.dtype)
            sc = (weight * invstd).view(1, C, 1, 1)
            xhat = (xc.float() - mean.view(1, C, 1, 1)) * invstd.view(1, C, 1, 1)
            return ((dz.float() * sc).to(xc.dtype), (dz.float() * xhat).sum((0, 2, 3)), dz.float().sum((0, 2, 3)),
                    dz if (ctx.has_res and ctx.needs_input_grad[3]) else None, None, None)
        dx = torch.empty_like(xc, memory_format=torch.channels_last)
        dres = torch.empty_like(xc, memory_format=torch.channels_last) if (ctx.has_res and ctx.needs_input_grad[3]) else None
        dgamma = torch.empty(C, device=xc.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=xc.device, dtype=torch.float32)
        L = hip.lib()
        ws = torch.empty(L.psi_bn_workspace_floats(M, C), device=xc.device, dtype=torch.float32)
        hip.check(L.psi_bn_backward_t(_ptr_cl(dyc), int(ctx.f32), _ptr_cl(xc), _ptr_cl(y), hip.ptr(weight), hip.ptr(bias.detach().float()), hip.ptr(mean),
                                      hip.ptr(invstd), M, C,
                                      int(ctx.relu), _ptr_cl(dx), _ptr_cl(dres), hip.ptr(dgamma), hip.ptr(dbeta), hip.ptr(ws), hip.stream()),
                  'psi_bn_backward_t')
        return dx, dgamma, dbeta, dres, None, None


def bn_act_t(x, bn, relu=True, residual=None):
    """``relu(bn(x) + residual)`` (each part optional) of an ``nn.BatchNorm2d`` in training OR eval mode on an fp32 or bf16 channels_last map."""
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda or x.dim() != 4:
        raise ValueError('bn_act_t: expected a 4-D fp32 / bf16 CUDA tensor')
    if residual is not None and (residual.dtype != x.dtype or residual.shape != x.shape):
        raise ValueError('bn_act_t: residual must match x')
    return _BNActT.apply(x, bn.weight, bn.bias, residual, bn, relu)


def bn_t_supported(bn):
    return (bn.affine and bn.track_running_stats and bn.running_mean is not None and (bn.momentum is not None or not bn.training)
            and bn.num_features in (8, 16, 32, 64, 128, 256))


class _MaxPoolT(Function):
    @staticmethod
    def forward(ctx, x):
        N, C, H, W = x.shape
        xc = x.contiguous(memory_format=torch.channels_last)
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, OH, OW), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
        idx = torch.empty(N * OH * OW * C, device=x.device, dtype=torch.uint8) if ctx.needs_input_grad[0] else None
        hip.check(hip.lib().psi_maxpool3x3s2_forward_t(_ptr_cl(xc), int(x.dtype == torch.float32), N, H, W, C, _ptr_cl(y), hip.ptr(idx), hip.stream()),
                  'psi_maxpool3x3s2_forward_t')
        ctx.save_for_backward(idx)
        ctx.dims = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        idx, = ctx.saved_tensors
        N, C, H, W = ctx.dims
        dyc = dy.contiguous(memory_format=torch.channels_last)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=dy.dtype, memory_format=torch.channels_last)
        hip.check(hip.lib().psi_maxpool3x3s2_backward_t(_ptr_cl(dyc), int(dy.dtype == torch.float32), hip.ptr(idx), N, H, W, C, _ptr_cl(dx), hip.stream()),
                  'psi_maxpool3x3s2_backward_t')
        return dx


def maxpool3x3s2_t(x):
    """``nn.MaxPool2d(3, 2, 1)`` on an fp32 or bf16 channels_last map."""
    if x.dtype not in (torch.float32, torch.bfloat16) or not x.is_cuda or x.dim() != 4 or x.shape[1] % 8:
        raise ValueError('maxpool3x3s2_t: expected a 4-D fp32 / bf16 CUDA tensor with a multiple of 8 channels')
    return _MaxPoolT.apply(x)


class _LinearAct3(Function):
    """nn.Linear (+ LeakyReLU, + skip connection) at the fp32 model's precision: ONE hand-written kernel forward (psi_linear_forward3: three-term
    split products, any width) and per gradient (psi_linear_backward3: dX, dW + dbias with the same products)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, act, slope):
        M, K = x.shape
        N = weight.shape[0]
        xc = x.detach().contiguous().float()
        w = weight.detach().contiguous().float()
        b = bias.detach().contiguous().float() if bias is not None else None
        r = residual.detach().contiguous().float() if residual is not None else None
        y = torch.empty(M, N, device=x.device)
        a_out = torch.empty(M, N, device=x.device) if (act and residual is not None) else None
        L = hip.lib()
        nws = L.psi_linear_workspace_floats(M, N, K)
        ws = torch.empty(nws, device=x.device) if nws else None
        hip.check(L.psi_linear_forward3(hip.ptr(xc), 0, hip.ptr(w), hip.ptr(b), hip.ptr(r), M, N, K, int(act), float(slope), hip.ptr(y), hip.ptr(a_out),
                                        hip.ptr(ws), hip.stream()), 'psi_linear_forward3')
        ctx.act, ctx.slope = int(act), float(slope)
        ctx.has_bias, ctx.has_res = bias is not None, residual is not None
        ctx.save_for_backward(xc, w, (a_out if a_out is not None else y) if act else torch.empty(0, device=x.device))
        return y

    @staticmethod
    def backward(ctx, gy):
        xc, w, a_out = ctx.saved_tensors
        gy = gy.contiguous().float()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        M, K = xc.shape
        N = w.shape[0]
        gx = torch.empty(M, K, device=gy.device) if need_x else None
        gw = torch.empty(N, K, device=gy.device) if (need_w or need_b) else None         # gbias is produced by the dW kernel
        gb = torch.empty(N, device=gy.device) if need_b else None
        L = hip.lib()
        nws = L.psi_linear_backward_workspace_floats(M, N, K) if need_x else 0
        ws = torch.empty(nws, device=gy.device) if nws else None
        hip.check(L.psi_linear_backward3(hip.ptr(gy), hip.ptr(a_out) if ctx.act else None, hip.ptr(xc), hip.ptr(w), M, N, K, ctx.slope, hip.ptr(gx),
                                         hip.ptr(gw), hip.ptr(gb), hip.ptr(ws), hip.stream()), 'psi_linear_backward3')
        return gx, gw if need_w else None, gb, gy if (ctx.has_res and ctx.needs_input_grad[3]) else None, None, None


def linear_act3(x, weight, bias=None, act=None, slope=0.01, residual=None):
    """act(x @ weight.T + bias) (+ residual) at fp32 precision on the HIP kernels (any K, N); ``act`` is None or 'leaky_relu'."""
    if act not in (None, 'leaky_relu'):
        raise ValueError('act must be None or "leaky_relu"')
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
        raise ValueError('linear_act3: x [M,K] and weight [N,K] expected, got %s / %s' % (tuple(x.shape), tuple(weight.shape)))
    return _LinearAct3.apply(x, weight, bias, residual, 1 if act else 0, slope)


# ------------------------------------------------------------------------------------------
# Body-vector glue of a CVAE training step (csrc/cvae_loss.hip): target representation, reconstruction / KL / VPoser losses
# ------------------------------------------------------------------------------------------
def _f32c(t):
    return None if t is None else t.detach().contiguous().float()


def cvae_target(xh, cam_int, max_d):
    """``GeometryTransformer.convert_to_6D_rot(GeometryTransformer.normalize_global_T(xh, cam_int, max_d))`` (cvae.py:118-127, 176-199) in
    one launch: [B,72] -> [B,75].  The training target of the CVAE; a function of the batch only (no gradient)."""
    B = xh.shape[0]
    if xh.dim() != 2 or xh.shape[1] != 72 or tuple(cam_int.shape) != (B, 3, 3) or max_d.numel() != B:
        raise ValueError('cvae_target: xh [B,72], cam_int [B,3,3], max_d [B] expected, got %s / %s / %s'
                         % (tuple(xh.shape), tuple(cam_int.shape), tuple(max_d.shape)))
    out = torch.empty(B, 75, device=xh.device)
    hip.check(hip.lib().psi_cvae_target(hip.ptr(_f32c(xh)), hip.ptr(_f32c(cam_int)), hip.ptr(_f32c(max_d)), B, hip.ptr(out), hip.stream()),
              'psi_cvae_target')
    return out


class _CvaeLosses(Function):
    @staticmethod
    def forward(ctx, rec, target, xh, cam_int, max_d, mu0, lv0, mu1, lv1, fca, w_rec, w_kl, w_vposer):
        B = rec.shape[0]
        if rec.dim() != 2 or rec.shape[1] != 75 or tuple(target.shape) != (B, 75) or tuple(xh.shape) != (B, 72):
            raise ValueError('cvae_losses: rec / target [B,75] and xh [B,72] expected, got %s / %s / %s'
                             % (tuple(rec.shape), tuple(target.shape), tuple(xh.shape)))
        t = [_f32c(v) for v in (rec, target, xh, cam_int, max_d, mu0, lv0, mu1, lv1)]
        fca_t = fca if torch.is_tensor(fca) else None
        ctx.fca = 0.0 if fca_t is not None else float(fca)
        ctx.w = (float(w_rec), float(w_kl), float(w_vposer))
        ctx.nz = (0 if mu0 is None else mu0.shape[1], 0 if mu1 is None else mu1.shape[1])
        xh_rec = torch.empty(B, 75, device=rec.device)
        losses = torch.empty(5, device=rec.device)
        ws = torch.empty(hip.lib().psi_cvae_losses_workspace_floats(), device=rec.device)
        hip.check(hip.lib().psi_cvae_losses_forward(*[hip.ptr(v) for v in t[:7]], ctx.nz[0], hip.ptr(t[7]), hip.ptr(t[8]), ctx.nz[1], B,
                                                    *ctx.w, ctx.fca, hip.ptr(_f32c(fca_t)), hip.ptr(ws), hip.ptr(xh_rec), hip.ptr(losses),
                                                    hip.stream()),
                  'psi_cvae_losses_forward')
        ctx.has = (mu0 is not None, mu1 is not None)
        ctx.dtypes = tuple(None if v is None else v.dtype for v in (rec, mu0, lv0, mu1, lv1))
        ctx.fca_t = _f32c(fca_t)
        ctx.save_for_backward(*[v for v in t if v is not None], xh_rec)
        return xh_rec, losses

    @staticmethod
    def backward(ctx, g_xh_rec, g_losses):
        saved = list(ctx.saved_tensors)
        xh_rec = saved.pop()
        rec, target, xh, cam_int, max_d = saved[:5]
        rest = saved[5:]
        mu0, lv0 = (rest[0], rest[1]) if ctx.has[0] else (None, None)
        mu1, lv1 = (rest[-2], rest[-1]) if ctx.has[1] else (None, None)
        B = rec.shape[0]
        dev = rec.device
        gl = _f32c(g_losses) if g_losses is not None else torch.zeros(5, device=dev)
        g_rec = torch.empty_like(rec)
        g = [torch.empty_like(v) if v is not None else None for v in (mu0, lv0, mu1, lv1)]
        hip.check(hip.lib().psi_cvae_losses_backward(hip.ptr(rec), hip.ptr(target), hip.ptr(xh), hip.ptr(cam_int), hip.ptr(max_d), hip.ptr(mu0),
                                                     hip.ptr(lv0), ctx.nz[0], hip.ptr(mu1), hip.ptr(lv1), ctx.nz[1], B, *ctx.w, ctx.fca,
                                                     hip.ptr(ctx.fca_t), hip.ptr(xh_rec), hip.ptr(gl), hip.ptr(_f32c(g_xh_rec)), hip.ptr(g_rec),
                                                     hip.ptr(g[0]), hip.ptr(g[1]), hip.ptr(g[2]), hip.ptr(g[3]), hip.stream()),
                  'psi_cvae_losses_backward')
        cast = lambda v, dt: v if (v is None or v.dtype == dt) else v.to(dt)            # gradients in the dtype of their inputs
        dt = ctx.dtypes
        return (cast(g_rec, dt[0]), None, None, None, None, cast(g[0], dt[1]), cast(g[1], dt[2]), cast(g[2], dt[3]), cast(g[3], dt[4]),
                None, None, None, None)


def cvae_losses(rec, target, xh, cam_int, max_d, mu0, logvar0, mu1=None, logvar1=None, fca=1.0, w_rec=1.0, w_kl=1.0, w_vposer=1.0):
    """The body-vector losses of ``cal_loss`` (train_s1.py:113-133, train_s2.py:119-139) in one launch per direction.

    rec = xhnr_rec [B,75] (the CVAE's reconstruction, 6D global rotation), target = cvae_target(xh, ...), xh [B,72] the ground truth.
    -> xh_rec [B,75] = recover_global_T(rec, cam_int, max_d) and losses [5] = (rec_t, rec_p, KL of latent 0, KL of latent 1 (0 without
    one), vposer).  ``fca``: the KL annealing factor, a float or a 0-dim device tensor (captured steps change it between replays)."""
    return _CvaeLosses.apply(rec, target, xh, cam_int, max_d, mu0, logvar0, mu1, logvar1, fca, w_rec, w_kl, w_vposer)


# ------------------------------------------------------------------------------------------
# Contact + penetration losses of a training step from the body mesh (csrc/scene_loss.hip)
# ------------------------------------------------------------------------------------------
_CHAIN_CACHE = {}


_VID32_CACHE = {}


def _vid32_of(vid):
    """int32 copy of a contact-id tensor, cached per tensor (storage, length, version): a caller that passes no ``vid32`` gets the SAME int32
    tensor on every call, so the chain below is built once for it as well (it was rebuilt on every call: every cast was a new cache key)."""
    key = (vid.data_ptr(), vid.numel(), vid._version, vid.device)
    hit = _VID32_CACHE.get(key)
    if hit is None:
        if len(_VID32_CACHE) > 64:
            _VID32_CACHE.clear()
        hit = _VID32_CACHE[key] = (vid, vid.to(torch.int32))     # keeps the int64 tensor alive: its address cannot be reused while cached
    return hit[1]


def _contact_chain(vid32):
    """psi_contact_slot_chain of a contact-id tensor, cached per tensor (the ids are a constant of the model; the cache is keyed by storage
    and version, so an id list that is modified in place gets a new chain)."""
    key = (vid32.data_ptr(), vid32.numel(), vid32._version, vid32.device)
    hit = _CHAIN_CACHE.get(key)
    if hit is None:
        chain = torch.empty(2 * vid32.numel(), dtype=torch.int32, device=vid32.device)
        hip.check(hip.lib().psi_contact_slot_chain(hip.ptr(vid32), vid32.numel(), hip.ptr(chain), hip.stream()), 'psi_contact_slot_chain')
        if len(_CHAIN_CACHE) > 64:
            _CHAIN_CACHE.clear()
        hit = _CHAIN_CACHE[key] = (vid32, chain)             # keeps the id tensor alive: its address cannot be reused while cached
    return hit[1]


class _SceneLosses(Function):
    @staticmethod
    def forward(ctx, verts, vid, scenes, slot, sdf, gmin, gmax, align_corners, w_contact, w_collision, gate, vid32):
        ctx.verts_dtype = verts.dtype
        verts = verts.detach().contiguous().float()
        B, V, _ = verts.shape
        L = hip.lib()
        xyz1 = verts.index_select(1, vid)                                   # [B,n_c,3] contact vertices (train_s1.py:161)
        n_c = xyz1.shape[1]
        dist, idx = scenes.query(xyz1, slot)                                # exact body->scene NN (chamfer dist1 / idx1)
        S, D = sdf.shape[0], sdf.shape[1]
        vals = torch.empty(B, V, device=verts.device)
        og = torch.empty(B, V, 3, device=verts.device)
        hip.check(L.psi_sdf_sample_forward(hip.ptr(sdf), hip.ptr(slot), hip.ptr(gmin), hip.ptr(gmax), hip.ptr(verts), B, V, D, S,
                                           int(bool(align_corners)), hip.ptr(vals), hip.ptr(og), hip.stream()), 'psi_sdf_sample_forward')
        ws = torch.empty(L.psi_scene_losses_workspace_floats(), device=verts.device)
        losses = torch.empty(2, device=verts.device)
        stats = torch.empty(2, device=verts.device)
        ctx.w = (float(w_contact), float(w_collision), float(gate))
        hip.check(L.psi_scene_losses_forward(hip.ptr(dist), B * n_c, hip.ptr(vals), B * V, *ctx.w, hip.ptr(ws), hip.ptr(losses), hip.ptr(stats),
                                             hip.stream()), 'psi_scene_losses_forward')
        ctx.scenes = scenes
        ctx.save_for_backward(dist, xyz1, idx, slot, vid32 if vid32 is not None else _vid32_of(vid), vals, og, stats)
        return losses

    @staticmethod
    def backward(ctx, g):
        dist, xyz1, idx, slot, vid32, vals, og, stats = ctx.saved_tensors
        B, V = vals.shape
        table = ctx.scenes.verts_table
        g_verts = torch.empty(B, V, 3, device=vals.device)
        chain = _contact_chain(vid32)                       # slots of a repeated contact vertex, chained in slot order (built once per id list)
        hip.check(hip.lib().psi_scene_losses_backward(hip.ptr(g.contiguous().float()), hip.ptr(stats), hip.ptr(dist), hip.ptr(xyz1), hip.ptr(idx),
                                                      hip.ptr(slot), hip.ptr(table), table.shape[1], hip.ptr(vid32), hip.ptr(vals), hip.ptr(og), B,
                                                      V, xyz1.shape[1], *ctx.w, hip.ptr(chain), hip.ptr(g_verts), hip.stream()),
                  'psi_scene_losses_backward')
        return (g_verts if ctx.verts_dtype == g_verts.dtype else g_verts.to(ctx.verts_dtype),) + (None,) * 11


def scene_losses(body_verts, vid, scenes: SceneSet, slot, sdf, grid_min, grid_max, align_corners, w_contact, w_collision, gate=1.0, vid32=None):
    """(loss_contact, loss_sdf_pene) of ``cal_loss`` (train_s1.py:156-204) from the camera-frame body mesh [B,V,3]: the contact rows ``vid``
    against each body's scene (``scenes`` / ``slot`` as in chamfer_to_scenes), every vertex against its scene's SDF volume (sdf [S,D,D,D],
    grid_min / grid_max [S,3], the same ``slot``).  Two exact-NN / SDF kernels and two small reduction kernels forward, two kernels
    backward — the operator form of the same expressions is chamfer_to_scenes + sdf_sample + penetration_loss plus ~40 elementwise launches.
    ``vid``: int64 [n_c] device; ``vid32``: the same as int32 when the caller keeps one (saves a cast launch per step)."""
    assert slot.dtype == torch.int32 and slot.is_contiguous()
    out = _SceneLosses.apply(body_verts, vid, scenes, slot, sdf.contiguous().float(), grid_min.reshape(-1, 3).contiguous().float(),
                             grid_max.reshape(-1, 3).contiguous().float(), align_corners, w_contact, w_collision, gate, vid32)
    return out[0], out[1]
