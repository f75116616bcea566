"""Conditional VAEs of PSI with the reference's checkpoint (state_dict) layout.

* ``HumanCVAES1``  <- source/cvae.py:411-534   (one-stage: global translation + local pose sampled together)
* ``HumanCVAES2``  <- source/cvae.py:341-400 = ``BodyGlobalPoseVAE`` + ``BodyLocalPoseVAE`` (source/net_layers.py:47-234)
* ``ResBlock``     <- source/net_layers.py:28-43
* scene encoder: ``Conv2d(2,64,7,2,3,bias=False)`` + ``children()[1:6]`` of torchvision 0.4.0 ``resnet18`` (bn1, relu,
  maxpool, layer1, layer2; cvae.py:427-435).  torchvision is a third-party package that is neither in the reference
  tree nor installed, so the BasicBlock stack is restated here with the SAME attribute names — the state_dict keys
  (``resnet.0.weight``, ``resnet.4.0.conv1.weight``, ``resnet.5.0.downsample.0.weight`` ...) and shapes are those of
  SURVEY.md Appendix B, so reference checkpoints (``epoch-*.ckp`` -> ``model_h_state_dict``) load with strict=True.

These are genuine contractions (conv / linear): on a GPU they run on the matrix cores through the hand-written kernels of libpsi_hip.so
(ops.conv2d_split / conv3x3 / bn_act(_t) / maxpool3x3s2(_t) / linear_act(3): csrc/conv_gemm.hip, conv_stem.hip, conv.hip, bnorm.hip,
linear.hip) in both precisions — the fp32 model (the reference's, cvae.py:427-455) with three-term split products, ``autocast_bf16=True``
with bf16 products (losses stay fp32).  The ``nn.Module`` calls that remain in the forward methods are what CPU tensors take (the CPU
tests of the checkpoint layout and the host logic); there is no environment switch back to the vendor libraries — the tests that compare
against them patch these predicates (tests/library_paths.py).  Differences from the reference that do not
change arithmetic: the reparameterisation noise is drawn on the model's device (the reference draws it with the CPU
generator and copies it, net_layers.py:88-92) and can be injected (``eps=``) for reproducible tests.
"""
from __future__ import annotations

import torch
from torch import nn


def _precise(x):
    """The fp32 models (``autocast_bf16=False``: the reference's precision, cvae.py:427-455) on a GPU run their convolutions, BatchNorms,
    max-pool and dense layers on the hand-written kernels of libpsi_hip.so as well: fp32 NHWC maps, three-term split products on the
    bf16 matrix cores with fp32 accumulation (ops.conv2d_split / bn_act_t / maxpool3x3s2_t / linear_act3; csrc/conv_gemm.hip quantifies the
    arithmetic: 0.6-3.2e-5 of the reference's recorded forward passes)."""
    return x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()


def _lin(owner, layer, x, act=None, slope=0.01, residual=None):
    """``act(layer(x)) (+ residual)`` for an nn.Linear: the fp32 model on the GPU -> ops.linear_act3 (any width); the bf16 mode -> ``_linear``."""
    if _precise(x) and not getattr(owner, 'hip_linear', False):
        from .ops import linear_act3
        if x.dim() != 2:                                      # nn.Linear semantics: any number of leading dimensions
            r2 = residual.reshape(-1, residual.shape[-1]) if residual is not None else None
            return linear_act3(x.reshape(-1, x.shape[-1]), layer.weight, layer.bias, act, slope, r2).reshape(*x.shape[:-1], layer.out_features)
        return linear_act3(x, layer.weight, layer.bias, act, slope, residual)
    if _use_hip_linear(owner, x):
        # the bf16 mode: bias, LeakyReLU and the skip connection are part of the same launch; the 3-, 72-, 75-wide layers (no 16-wide operand
        # tiles) take the any-width kernel, whose three-term products are at least the library's fp32 precision
        if layer.in_features % 16 == 0 and layer.out_features % 16 == 0:
            from .ops import linear_act
            return linear_act(x, layer.weight, layer.bias, 'leaky_relu' if act else None, slope, residual=residual)
        if x.dim() == 2:
            from .ops import linear_act3
            return linear_act3(x.float(), layer.weight, layer.bias, act, slope, residual)
    y = _linear(owner, layer, x)
    if act:
        y = torch.nn.functional.leaky_relu(y, slope)
    return y if residual is None else y + residual


class ResBlock(nn.Module):
    def __init__(self, n_dim):
        super().__init__()
        self.n_dim = n_dim
        self.fc1 = nn.Linear(n_dim, n_dim)
        self.fc2 = nn.Linear(n_dim, n_dim)
        self.acfun = nn.LeakyReLU()

    hip_linear = False          # set by the owning model: dense layers on the hand-written bf16 MFMA kernels (ops.linear_act)

    def forward(self, x0):
        if _use_hip_linear(self, x0) and self.n_dim % 16 == 0:
            from .ops import linear_act
            slope = self.acfun.negative_slope
            x = linear_act(x0, self.fc1.weight, self.fc1.bias, 'leaky_relu', slope)
            return linear_act(x, self.fc2.weight, self.fc2.bias, 'leaky_relu', slope, residual=x0)      # fc2 + LeakyReLU + skip: one kernel
        if _precise(x0) and not self.hip_linear:
            slope = self.acfun.negative_slope
            x = _lin(self, self.fc1, x0, 'leaky_relu', slope)
            return _lin(self, self.fc2, x, 'leaky_relu', slope, residual=x0)
        x = self.acfun(self.fc1(x0))
        x = self.acfun(self.fc2(x))
        return x + x0


class _BasicBlock(nn.Module):
    """ResNet BasicBlock with torchvision's attribute names (conv1, bn1, relu, conv2, bn2, downsample)."""

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        if _precise_trunk(self.bn1, x):
            from .ops import bn_act_t
            identity = x if self.downsample is None else bn_act_t(_conv(self.downsample[0], x), self.downsample[1], relu=False)
            out = bn_act_t(_conv(self.conv1, x), self.bn1, relu=True)
            return bn_act_t(_conv(self.conv2, out), self.bn2, relu=True, residual=identity)
        if _use_hip_bn(self.bn1, x):
            # training statistics on bf16 NHWC maps: BN + ReLU (+ the skip connection) as ONE fused HIP op per BN (ops.bn_act) instead
            # of the library's three launches per BN and separate ReLU / add launches — same arithmetic, same running-statistics update
            from .ops import bn_act
            identity = x if self.downsample is None else bn_act(_conv(self.downsample[0], x), self.downsample[1], relu=False)
            out = bn_act(_conv(self.conv1, x), self.bn1, relu=True)
            return bn_act(_conv(self.conv2, out), self.bn2, relu=True, residual=identity)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(_conv(self.conv1, x)))
        out = self.bn2(_conv(self.conv2, out))
        return self.relu(out + identity)


def _precise_trunk(bn, x):
    from . import ops
    return _precise(x) and x.dim() == 4 and ops.bn_t_supported(bn)


def _conv(conv, x):
    """A trunk convolution.  fp32 model on the GPU: the general implicit-GEMM kernel with three-term split products (ops.conv2d_split).
    Under bf16 autocast: the hand-written stride-1 3x3 kernel where it applies (ops.conv3x3), the general kernel with one-term bf16
    products for the rest (7x7 stem, strided 3x3, 1x1 downsample, 128 -> 32 head)."""
    if _precise(x):
        from . import ops
        if ops.conv2d_supported(conv):
            return ops.conv2d_split(x, conv, nterm=3)
        return conv(x)
    if x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16:
        from . import ops
        if ops.conv3x3_supported(conv, x):
            return ops.conv3x3(x, conv)
        if ops.conv2d_supported(conv):
            return ops.conv2d_split(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), conv, nterm=1, out_bf16=True)
    return conv(x)


def _use_hip_bn(bn, x):
    """The fused BN kernels cover the training-mode trunk under bf16 autocast."""
    return (bn.training and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16
            and bn.affine and bn.track_running_stats and bn.momentum is not None          # what the kernels implement (nn.BatchNorm2d defaults)
            and bn.num_features in (8, 16, 32, 64, 128, 256))


def run_trunk(trunk, x):
    """forward of ``scene_trunk`` (an nn.Sequential, so that the state_dict keys stay ``resnet.0.weight``, ``resnet.1.*`` ...) with the
    stem's BN + ReLU fused when the fused kernels apply."""
    if _precise_trunk(trunk[1], x):
        from .ops import bn_act_t, maxpool3x3s2_t
        x = bn_act_t(_conv(trunk[0], x), trunk[1], relu=True)
        mp = trunk[3]
        x = maxpool3x3s2_t(x) if (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False) else mp(x)
        for i in range(4, len(trunk)):
            x = trunk[i](x)
        return x
    if _use_hip_bn(trunk[1], x):
        from .ops import bn_act, maxpool3x3s2
        x = bn_act(_conv(trunk[0], x), trunk[1], relu=True)
        mp = trunk[3]
        x = maxpool3x3s2(x) if (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False) else mp(x)
        for i in range(4, len(trunk)):
            x = trunk[i](x)
        return x
    return trunk(x)


def scene_trunk(in_dim=2):
    """``nn.Sequential(Conv2d(in_dim,64,7,2,3,bias=False), bn1, relu, maxpool, layer1, layer2)`` -> [B,128,16,16] for 128x128 input."""
    return nn.Sequential(
        nn.Conv2d(in_dim, 64, kernel_size=7, stride=2, padding=3, bias=False),
        nn.BatchNorm2d(64),
        nn.ReLU(inplace=True),
        nn.MaxPool2d(kernel_size=3, stride=2, padding=1),
        nn.Sequential(_BasicBlock(64, 64), _BasicBlock(64, 64)),
        nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128)))


def load_pretrained_resnet18(trunk, ckpt_path):
    """``data/resnet18.pth`` (a missing blob in the reference tree): torchvision resnet18 state_dict -> trunk children 1..5."""
    sd = torch.load(ckpt_path, map_location='cpu')
    remap = {'bn1.': '1.', 'layer1.': '4.', 'layer2.': '5.'}
    own = trunk.state_dict()
    for k, v in sd.items():
        for src, dst in remap.items():
            if k.startswith(src) and dst + k[len(src):] in own:
                own[dst + k[len(src):]] = v
    trunk.load_state_dict(own)


def set_hip_linear(model, on=True):
    """Route the dense layers of ``model`` (ResBlock stacks and the scene-feature ``fc``) through the hand-written bf16 MFMA kernels of
    libpsi_hip.so (ops.linear_act: Linear + bias + LeakyReLU + skip in one launch, fp32 master weights rounded to bf16 on load).
    Enabled together with ``autocast_bf16``; see ``_use_hip_linear`` for what that means for precision.
    Measured on MI355X at batch 128 (train_s2 step): 7.68 -> 7.59 ms with the layers on these kernels; the backward pair (dX, dW + dbias)
    takes 26 us for a 512 x 512 layer against 38 us for the library route (tools/time_linear_bwd.py)."""
    for m in model.modules():
        if isinstance(m, (ResBlock, _SceneCond)):
            m.hip_linear = bool(on)


def _use_hip_linear(module, x):
    """Dense layers of a model built with ``autocast_bf16=True`` run on the hand-written bf16 MFMA kernels (forward AND backward:
    ops.linear_act) — in training and in no_grad mode alike, so a model gives the same numbers in both.  This is a precision choice of
    the bf16 mode: the ResBlock stacks, which sit outside the autocast region of the library path and ran in fp32 there, round their
    operands to bf16 (fp32 accumulation, fp32 outputs), like the trunk."""
    return bool(getattr(module, 'hip_linear', False)) and x.is_cuda


def _linear(owner, layer, x):
    """``layer(x)`` for an nn.Linear of a model in the bf16 mode: the hand-written MFMA kernels when the layer's shape is covered (both
    widths multiples of 16 — the latent heads and the decoders' input layers), the library otherwise (3-, 72-, 75-wide layers)."""
    if _use_hip_linear(owner, x) and layer.in_features % 16 == 0 and layer.out_features % 16 == 0:
        from .ops import linear_act
        return linear_act(x, layer.weight, layer.bias)
    return layer(x)


def _decode(owner, seq, x):
    """``seq(x)`` for the decoders ``nn.Sequential(Linear, ResBlock, ResBlock, Linear)`` (net_layers.py:88-93, 181-186): the first layer through
    ``_linear`` (the state_dict keys stay ``decode.0.weight`` ...)."""
    x = _lin(owner, seq[0], x)
    for i in range(1, len(seq)):
        x = _lin(owner, seq[i], x) if isinstance(seq[i], nn.Linear) else seq[i](x)
    return x


def _reparam(mu, logvar, eps=None):
    std = torch.exp(0.5 * logvar)
    if eps is None:
        eps = torch.randn_like(std)
    return eps * std + mu


class _SceneCond(nn.Module):
    """Shared shape of the three scene-conditioned VAEs: trunk -> conv -> fc."""

    def _scene_feature(self, scene, rows=None):
        """Scene feature z_s [b, num_hidden].  ``rows``: the generation drivers condition n samples on ONE view and feed the trunk n copies
        of it (test_habitat_s2.py:192-195: ``xs.repeat(self.n_samples, 1, 1, 1)``); in eval mode every row of the trunk's output depends on
        its own input only (BatchNorm on running statistics), so the single view is encoded once and the feature row repeated."""
        if rows is not None and rows != scene.size(0):
            if scene.size(0) != 1 or self.training:
                raise ValueError('rows=%d needs ONE scene view and a model in eval mode (got %d views, training=%s)' % (rows, scene.size(0), self.training))
            return self._scene_feature(scene).expand(rows, -1)
        b = scene.size(0)
        if getattr(self, 'autocast_bf16', False) and scene.is_cuda:
            if not getattr(self, '_nhwc', False):           # MIOpen's bf16 implicit-GEMM convs are NHWC: keep weights and
                self.resnet.to(memory_format=torch.channels_last)   # activations in that layout instead of transposing around every conv
                self.conv.to(memory_format=torch.channels_last)
                self._nhwc = True
            scene = scene.contiguous(memory_format=torch.channels_last)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                f = _conv(self.conv, run_trunk(self.resnet, scene))
                if _use_hip_linear(self, scene):
                    # the 8192 / 32768 -> num_hidden layer: the bf16 feature map goes straight into the MFMA kernel, the fp32 master
                    # weight (up to 33.5 MB) is read once and rounded on load instead of being cast by a separate kernel every step
                    from .ops import linear_act
                    return linear_act(f.reshape(b, -1), self.fc.weight, self.fc.bias)
                return self.fc(f.reshape(b, -1)).float()
        if _precise(scene):
            if not getattr(self, '_nhwc', False):             # NHWC maps and [Cout,KH,KW,Cin] weights: what the kernels read
                self.resnet.to(memory_format=torch.channels_last)
                self.conv.to(memory_format=torch.channels_last)
                self._nhwc = True
            f = _conv(self.conv, run_trunk(self.resnet, scene.contiguous(memory_format=torch.channels_last)))
            return _lin(self, self.fc, f.reshape(b, -1))
        f = self.conv(self.resnet(scene))
        return self.fc(f.reshape(b, -1))


class BodyGlobalPoseVAE(_SceneCond):
    """net_layers.py:47-134: global translation (3-D) conditioned on the scene."""

    def __init__(self, zdim, num_hidden=512, f_dim=32, test=False, in_dim=3, pretrained_resnet=None):
        super().__init__()
        self.test, self.zdim = test, zdim
        self.resnet = scene_trunk(in_dim)
        if pretrained_resnet is not None:
            load_pretrained_resnet18(self.resnet, pretrained_resnet)
        self.conv = nn.Conv2d(128, f_dim, 3, 1, 1)
        self.fc = nn.Linear(f_dim * 16 * 16, num_hidden)
        self.torso_linear = nn.Linear(3, num_hidden)
        self.encode = nn.Sequential(ResBlock(2 * num_hidden), ResBlock(2 * num_hidden))
        self.mean_linear = nn.Linear(2 * num_hidden, zdim)
        self.log_var_linear = nn.Linear(2 * num_hidden, zdim)
        self.decode = nn.Sequential(nn.Linear(num_hidden + zdim, f_dim), ResBlock(f_dim), ResBlock(f_dim), nn.Linear(f_dim, 3))

    def forward(self, scene, torso=None, eps=None, rows=None, z_s=None):
        if z_s is None:                                       # (HumanCVAES2 may have computed it on a second stream)
            z_s = self._scene_feature(scene, rows)
        if self.test:
            z = torch.randn(z_s.size(0), self.zdim, device=scene.device) if eps is None else eps
            return _decode(self, self.decode, torch.cat([z, z_s], dim=1))
        feature = self.encode(torch.cat((z_s, _lin(self, self.torso_linear, torso)), dim=1))            # net_layers.py:118
        mean, log_var = _lin(self, self.mean_linear, feature), _lin(self, self.log_var_linear, feature)
        z = _reparam(mean, log_var, eps)
        return _decode(self, self.decode, torch.cat([z, z_s], dim=1)), mean, log_var        # net_layers.py:131


class BodyLocalPoseVAE(_SceneCond):
    """net_layers.py:144-234: the remaining 72 dims conditioned on scene and (reconstructed) translation."""

    def __init__(self, zdim, num_hidden=512, f_dim=128, test=False, in_dim=3, pretrained_resnet=None):
        super().__init__()
        self.test, self.zdim = test, zdim
        self.resnet = scene_trunk(in_dim)
        if pretrained_resnet is not None:
            load_pretrained_resnet18(self.resnet, pretrained_resnet)
        self.conv = nn.Conv2d(128, f_dim, 3, 1, 1)
        self.fc = nn.Linear(f_dim * 16 * 16, num_hidden)
        self.torso_linear = nn.Linear(3, num_hidden)
        self.pose_linear = nn.Linear(72, num_hidden)
        self.encode = nn.Sequential(ResBlock(3 * num_hidden), ResBlock(3 * num_hidden))
        self.mean_linear = nn.Linear(3 * num_hidden, zdim)
        self.log_var_linear = nn.Linear(3 * num_hidden, zdim)
        self.decode = nn.Sequential(nn.Linear(2 * num_hidden + zdim, f_dim), ResBlock(f_dim), ResBlock(f_dim), nn.Linear(f_dim, 72))

    def forward(self, scene, torso=None, pose=None, eps=None, rows=None, z_s=None):
        if z_s is None:
            z_s = self._scene_feature(scene, rows)
        z_g = _lin(self, self.torso_linear, torso)
        if self.test:
            z = torch.randn(z_s.size(0), self.zdim, device=scene.device) if eps is None else eps
            return _decode(self, self.decode, torch.cat([z, z_g, z_s], dim=1))
        feature = self.encode(torch.cat([_lin(self, self.pose_linear, pose), z_g, z_s], dim=1))         # net_layers.py:220
        mean, log_var = _lin(self, self.mean_linear, feature), _lin(self, self.log_var_linear, feature)
        z = _reparam(mean, log_var, eps)
        return _decode(self, self.decode, torch.cat([z, z_g, z_s], dim=1)), mean, log_var   # net_layers.py:231


_SIDE_STREAMS = {}            # device index -> second stream of HumanCVAES2.forward


class HumanCVAES2(nn.Module):
    """cvae.py:341-400.  ``eps_g`` / ``eps_l`` of the reference signature are unused there (cvae.py:369-385); here they are
    honoured only when ``use_eps=True`` (tests), otherwise noise is drawn internally like the reference does."""

    def __init__(self, latentD_g=512, latentD_l=512, scene_model_ckpt=None, n_dim_body=72, n_dim_scene=128, test=False,
                 autocast_bf16=False):
        super().__init__()
        self.latentD_g, self.latentD_l = latentD_g, latentD_l
        self.n_dim_g, self.n_dim_l = 3, n_dim_body - 3
        self.trans_vae = BodyGlobalPoseVAE(zdim=32, in_dim=2, num_hidden=latentD_g, pretrained_resnet=scene_model_ckpt, test=test)
        self.pose_vae = BodyLocalPoseVAE(zdim=32, in_dim=2, num_hidden=latentD_g, pretrained_resnet=scene_model_ckpt, test=test)
        self.trans_vae.autocast_bf16 = self.pose_vae.autocast_bf16 = autocast_bf16
        set_hip_linear(self, autocast_bf16)

    def forward(self, x_body, eps_g, eps_l, x_s, use_eps=False):
        x_g, x_l = x_body[:, :3], x_body[:, 3:]
        z_s_l = None
        if x_s.is_cuda and self.training:
            # The two scene trunks read the same view and meet only after their scene features: the local VAE's trunk runs on a second
            # stream beside the global VAE's (autograd replays each backward on its forward's stream, so the two backward halves overlap
            # too; inside a captured step the two become parallel branches of the graph).  Their many small launches — BN finalize, weight
            # re-layout, split-K reductions — no longer queue behind each other.
            cur = torch.cuda.current_stream(x_s.device)
            side = _SIDE_STREAMS.get(x_s.device.index)          # (kept outside the module: streams do not pickle / deep-copy)
            if side is None:
                side = _SIDE_STREAMS[x_s.device.index] = torch.cuda.Stream(x_s.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                z_s_l = self.pose_vae._scene_feature(x_s)
            x_g_rec, mu_g, lv_g = self.trans_vae(x_s, x_g, eps=eps_g if use_eps else None)
            cur.wait_stream(side)
            z_s_l.record_stream(cur)
        else:
            x_g_rec, mu_g, lv_g = self.trans_vae(x_s, x_g, eps=eps_g if use_eps else None)
        x_l_rec, mu_l, lv_l = self.pose_vae(x_s, x_g_rec, x_l, eps=eps_l if use_eps else None, z_s=z_s_l)
        return torch.cat([x_g_rec, x_l_rec], dim=1), mu_g, lv_g, mu_l, lv_l

    def sample(self, x_s, eps_g=None, eps_l=None, use_eps=False, rows=None):
        """``rows``: number of samples to draw for ONE view ``x_s [1,2,128,128]`` (the view is encoded once, see _scene_feature)."""
        x_g = self.trans_vae(x_s, eps=eps_g if use_eps else None, rows=rows)
        x_l = self.pose_vae(x_s, x_g, eps=eps_l if use_eps else None, rows=rows)
        return torch.cat([x_g, x_l], dim=1)


class HumanCVAES1(_SceneCond):
    """cvae.py:411-534."""

    def __init__(self, latentD=512, n_dim_body=75, scene_model_ckpt=None, test=False, autocast_bf16=False):
        super().__init__()
        self.test, self.eps_d, self.autocast_bf16 = test, 32, autocast_bf16
        self.resnet = scene_trunk(2)
        if scene_model_ckpt is not None:
            print('[INFO][SceneNet] Using pretrained resnet18 weights.')
            load_pretrained_resnet18(self.resnet, scene_model_ckpt)
        self.conv = nn.Conv2d(128, 32, 3, 1, 1)
        self.fc = nn.Linear(32 * 16 * 16, latentD)
        self.linear_in = nn.Linear(n_dim_body, latentD)
        self.human_encoder = nn.Sequential(ResBlock(2 * latentD), ResBlock(2 * latentD))
        self.mu_enc = nn.Linear(2 * latentD, self.eps_d)
        self.logvar_enc = nn.Linear(2 * latentD, self.eps_d)
        self.linear_latent = nn.Linear(self.eps_d, latentD)
        self.human_decoder = nn.Sequential(ResBlock(2 * latentD), ResBlock(2 * latentD))
        self.linear_out = nn.Linear(2 * latentD, n_dim_body)
        set_hip_linear(self, autocast_bf16)

    def forward(self, x_body, x_s, eps=None):
        z_s = self._scene_feature(x_s)
        z_hs = self.human_encoder(torch.cat([_lin(self, self.linear_in, x_body), z_s], dim=1))         # cvae.py:480
        mu, logvar = _lin(self, self.mu_enc, z_hs), _lin(self, self.logvar_enc, z_hs)
        z_h = _lin(self, self.linear_latent, _reparam(mu, logvar, eps))
        return _lin(self, self.linear_out, self.human_decoder(torch.cat([z_h, z_s], dim=1))), mu, logvar   # cvae.py:488

    def _decode_latent(self, x_s, eps, rows=None):
        z_s = self._scene_feature(x_s, rows)
        return _lin(self, self.linear_out, self.human_decoder(torch.cat([_lin(self, self.linear_latent, eps), z_s], dim=1)))

    def sample(self, x_s, eps=None, rows=None, **kwargs):
        if eps is None:
            eps = torch.randn(rows or x_s.shape[0], self.eps_d, dtype=torch.float32, device=x_s.device)
        return self._decode_latent(x_s, eps, rows)

    def sample_line(self, x_s, **kwargs):
        """cvae.py:516-534: latent swept along the diagonal from -3 to 3."""
        b_ = x_s.shape[0]
        eps = torch.arange(-3, 3, 6.0 / b_, dtype=torch.float32, device=x_s.device)[:b_].unsqueeze(1).repeat(1, self.eps_d)
        return self._decode_latent(x_s, eps), eps
