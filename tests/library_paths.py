"""Comparison arms of the CVAE tests: the vendor-library (torch / MIOpen / hipBLASLt) routes of the operators the package runs on its own
hand-written kernels.  The package has no switch back to the libraries (psi_release_amd/models.py, ops.py, optim.py); a test that wants the
library arm patches the package's routing predicates or an autograd Function's backward with the implementations below, for the duration
of the test (pytest's ``monkeypatch``).  Test infrastructure only: nothing under psi-release_amd/ imports this module."""
import torch

from psi_release_amd import models, ops


# ---- model routing ---------------------------------------------------------------------------------------------------------
def fp32_models_on_the_library(monkeypatch):
    """fp32 CUDA models: nn.Conv2d / nn.BatchNorm2d / nn.Linear as PyTorch runs them (MIOpen / hipBLASLt fp32)."""
    monkeypatch.setattr(models, '_precise', lambda x: False)


def bf16_batchnorm_on_the_library(monkeypatch):
    monkeypatch.setattr(models, '_use_hip_bn', lambda bn, x: False)


def bf16_dense_layers_on_the_library(monkeypatch):
    monkeypatch.setattr(models, '_use_hip_linear', lambda module, x: False)


def all_models_on_the_library(monkeypatch):
    """Every layer of both precisions through PyTorch's own operators (what torch.utils.flop_counter can see)."""
    fp32_models_on_the_library(monkeypatch)
    bf16_batchnorm_on_the_library(monkeypatch)
    bf16_dense_layers_on_the_library(monkeypatch)
    monkeypatch.setattr(models, '_conv', lambda conv, x: conv(x))


# ---- backward passes through the libraries, behind the hand-written forward -------------------------------------------------
def _linear_act_backward_library(ctx, gy):
    """ops._LinearAct.backward through hipBLASLt on the same bf16-rounded operands: mask, cast, two GEMMs, column sum."""
    xc, w, a_out = ctx.saved_tensors
    gy = gy.contiguous().float()
    need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
    g = gy if not ctx.act else torch.where(a_out > 0, gy, gy * ctx.slope)
    gb16 = g.to(torch.bfloat16)
    gx = (gb16 @ w.to(torch.bfloat16)).to(xc.dtype) if need_x else None
    gw = (gb16.t() @ xc.to(torch.bfloat16)).float() if need_w else None
    gb = g.sum(0) if need_b else None
    return gx, gw, gb, gy if (ctx.has_res and ctx.needs_input_grad[3]) else None, None, None


def _linear_act3_backward_library(ctx, gy):
    """ops._LinearAct3.backward: the two gradient GEMMs through hipBLASLt (fp32) on the saved operands."""
    xc, w, a_out = ctx.saved_tensors
    gy = gy.contiguous().float()
    need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
    g = gy if not ctx.act else torch.where(a_out > 0, gy, gy * ctx.slope)
    return (g @ w if need_x else None, g.t() @ xc if need_w else None, g.sum(0) if need_b else None,
            gy if (ctx.has_res and ctx.needs_input_grad[3]) else None, None, None)


def _conv2d_split_backward_library(ctx, dy):
    """ops._Conv2dSplit.backward: both gradients through aten.convolution_backward (MIOpen) on the saved operands."""
    xc, weight, _ = ctx.saved_tensors
    stride, pad, has_bias, nterm = ctx.geom
    Cout = weight.shape[0]
    dyc = dy.contiguous(memory_format=torch.channels_last)
    if dyc.dtype not in (torch.float32, torch.bfloat16):
        dyc = dyc.float()
    mask = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], has_bias and ctx.needs_input_grad[2])
    gx, gw, gb = torch.ops.aten.convolution_backward(dyc.to(xc.dtype), xc, weight.detach().to(xc.dtype), [Cout] if has_bias else None,
                                                     (stride, stride), (pad, pad), (1, 1), False, (0, 0), 1, mask)
    return gx, gw.float() if gw is not None else None, gb.float() if gb is not None else None, None, None, None, None


def bf16_dense_backward_on_the_library(monkeypatch):
    monkeypatch.setattr(ops._LinearAct, 'backward', staticmethod(_linear_act_backward_library))


def fp32_backward_on_the_library(monkeypatch):
    """the fp32 model's convolution and dense BACKWARD passes through the libraries, behind the same hand-written forward"""
    monkeypatch.setattr(ops._Conv2dSplit, 'backward', staticmethod(_conv2d_split_backward_library))
    monkeypatch.setattr(ops._LinearAct3, 'backward', staticmethod(_linear_act3_backward_library))


# ---- variants of the hand-written path itself --------------------------------------------------------------------------------
def conv_weights_split_in_every_workgroup(monkeypatch):
    """the general convolution without psi_conv2d_prepare_weight (what shapes outside psi_conv2d_prepared_ok always take)"""
    monkeypatch.setattr(ops, '_conv2d_prepared_ok', lambda *a: False)


def bn_relu_mask_from_the_stored_output(monkeypatch):
    monkeypatch.setattr(ops, '_bn_mask_from_x', lambda: False)
