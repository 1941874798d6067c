"""Eval-mode / deploy path: conv+BN folding and RepConv re-parameterisation for the Hip modules.

  fuse_conv_and_bn : src/utils/fuse.py:32-54 (same arithmetic: W' = diag(gamma / sqrt(var + eps)) W, b' = gamma (b - mean) / sqrt(var+eps) + beta)
  fuse_model       : src/utils/fuse.py:56-64 generalised to the module shapes used on the hot path: HipConvModule (conv, bn, act),
                     HipConvBN / nn.Sequential(conv, bn) pairs, torchvision-style sibling `convN` / `bnN` attributes (ResNet Bottleneck)
  reparam_repconv  : src/models/modules/yolov7_modules.py:215-300 (3x3 + padded 1x1 + identity-BN -> one 3x3 conv with bias)
After folding, an eval forward of a ConvModule is ONE MFMA conv with bias + ONE activation pass (no BN scale/shift pass at all).
The folding arithmetic is plain fp32 tensor math on the parameters (device-agnostic, unit-tested on CPU)."""
import torch
import torch.nn as nn

from . import lib as L
from .bricks import HipBN, HipConv2d, HipConvBN, HipConvModule


@torch.no_grad()
def fuse_conv_and_bn(conv, bn):
    """-> HipConv2d with bias carrying conv∘bn (eval statistics). groups / dilation are preserved."""
    fused = HipConv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding, conv.dilation, conv.groups, True)
    fused = fused.to(conv.weight.device)
    scale = bn.weight.div(torch.sqrt(bn.eps + bn.running_var)) if bn.weight is not None else 1.0 / torch.sqrt(bn.eps + bn.running_var)
    w = conv.weight * scale.reshape(-1, 1, 1, 1)
    fused.weight.data = w.contiguous(memory_format=torch.channels_last)
    b_conv = torch.zeros(conv.out_channels, device=conv.weight.device) if conv.bias is None else conv.bias
    beta = bn.bias if bn.bias is not None else torch.zeros_like(bn.running_mean)
    fused.bias.copy_((b_conv - bn.running_mean) * scale + beta)
    return fused


def _is_bn(m):
    return isinstance(m, nn.BatchNorm2d)


@torch.no_grad()
def fuse_model(model):
    """Fold every (conv, BatchNorm) pair of an eval-mode model in place; returns the model."""
    if model.training:
        raise L.CvhipError("fuse_model: put the model in eval() mode first (folding uses the running statistics)")
    for m in list(model.modules()):
        if isinstance(m, HipConvModule) and m.with_norm and _is_bn(m.norm) and m.order[:2] == ("conv", "norm"):
            m.conv = fuse_conv_and_bn(m.conv, m.norm)
            delattr(m, m.norm_name)
            m.norm_name = None
            m.with_norm = False
            m.with_bias = True
        elif isinstance(m, HipConvBN):
            fused = fuse_conv_and_bn(m[0], m[1])
            m[0] = fused
            m[1] = nn.Identity()
            m.forward = _fused_seq_forward.__get__(m)
        elif isinstance(m, nn.Sequential) and not isinstance(m, HipConvBN):
            # (conv, bn) neighbours of a Sequential: the BN slot becomes nn.Identity so that index-based forwards keep their
            # positions (deeplab.ResNet.stem / Bottleneck.downsample, stdc._DwPwSkip read [i], [i+1] and run the folded pair as
            # conv + bias when they find the Identity)
            i = 0
            while i + 1 < len(m):
                if isinstance(m[i], nn.Conv2d) and _is_bn(m[i + 1]):
                    m[i] = fuse_conv_and_bn(m[i], m[i + 1])
                    m[i + 1] = nn.Identity()
                    i += 2
                else:
                    i += 1
        else:
            # torchvision-style sibling attributes convN / bnN (ResNet Bottleneck: conv1/bn1 .. conv3/bn3)
            for name, child in list(m._modules.items()):
                if name.startswith("conv") and name[4:].isdigit() and isinstance(child, nn.Conv2d):
                    bn = m._modules.get("bn" + name[4:])
                    if _is_bn(bn):
                        setattr(m, name, fuse_conv_and_bn(child, bn))
                        setattr(m, "bn" + name[4:], nn.Identity())
    return model


def _fused_seq_forward(self, x, residual=None):
    from . import ops
    conv = self[0]
    xx, w = conv._effective(x)
    return ops.conv_bn_act(xx, w, conv.bias, None, None, None, None, residual, conv.make_cfg(self._act, 0.0, None))


@torch.no_grad()
def repconv_equivalent_kernel_bias(rep):
    """(kernel, bias) of the single 3x3 conv equivalent to an eval-mode RepConv (yolov7_modules.py:215-262)."""
    def fold(conv_w, bn):
        std = (bn.running_var + bn.eps).sqrt()
        t = (bn.weight / std).reshape(-1, 1, 1, 1)
        return conv_w * t, bn.bias - bn.running_mean * bn.weight / std

    k3, b3 = fold(rep.rbr_dense[0].weight, rep.rbr_dense[1])
    k1, b1 = fold(rep.rbr_1x1[0].weight, rep.rbr_1x1[1])
    k = k3 + torch.nn.functional.pad(k1, [1, 1, 1, 1])
    b = b3 + b1
    if rep.rbr_identity is not None:
        c = rep.in_channels
        idk = torch.zeros(c, c, 3, 3, device=k.device, dtype=k.dtype)
        idk[torch.arange(c), torch.arange(c), 1, 1] = 1.0
        ki, bi = fold(idk, rep.rbr_identity)
        k, b = k + ki, b + bi
    return k, b


@torch.no_grad()
def reparam_repconv(model):
    """Replace the three training branches of every yolov7.RepConv by `rbr_reparam` (one 3x3 conv + bias), in place."""
    from .yolov7 import RepConv
    for m in model.modules():
        if isinstance(m, RepConv) and not hasattr(m, "rbr_reparam"):
            k, b = repconv_equivalent_kernel_bias(m)
            conv = HipConv2d(m.in_channels, m.out_channels, 3, m.rbr_dense[0].stride, 1, bias=True).to(k.device)
            conv.weight.data = k.contiguous(memory_format=torch.channels_last)
            conv.bias.data = b.clone()
            m.rbr_reparam = conv
            for name in ("rbr_dense", "rbr_1x1", "rbr_identity"):
                if name in m._modules:
                    del m._modules[name]
            m.rbr_dense = m.rbr_1x1 = m.rbr_identity = None
    return model
