"""YOLO building blocks on the HIP engine — same classes, constructor arguments and sub-module names
as the reference's src/models/modules/yolo_modules.py, so reference `state_dict`s load unchanged.

What differs from the reference is only *how* the glue ops execute:
  * DarknetBottleneck's shortcut add rides in the second ConvModule's BN+act pass (yolo_modules.py:102)
  * UpsamplingModule's nearest-x2 + cat is one kernel (yolo_modules.py:147,152)
  * CSP / SPPF / Downsampling concats are slice copies into one NHWC buffer whose backward hands out
    channel-slice views (yolo_modules.py:139,162,190)
"""
import os

import torch
import torch.nn as nn

from . import ops
from .bricks import HipConvModule as ConvModule
from .bricks import HipDepthwiseSeparableConvModule as DepthwiseSeparableConvModule
from .bricks import HipMaxPool2d, HipUpsampleNearest2x, bn_tick, sync_of

_PAIR_ENABLED = os.environ.get("CVHIP_PAIR", "1") != "0"
_CAT_INPLACE = os.environ.get("CVHIP_CAT_INPLACE", "1") != "0"
_LAZY_CAT = os.environ.get("CVHIP_LAZY_CAT", "1") != "0"   # CSP conv2's concat slice stays raw (split store + on-load transform in conv3)
_SPPF_CHAIN = os.environ.get("CVHIP_SPPF_CHAIN", "1") != "0"   # 0: separate pools + fan-out adds + concat copy (A/B switch)
_GRAD_LINK = os.environ.get("CVHIP_GRAD_LINK", "1") != "0"


def sibling_pair_forward(m1, m2, x, owner, out2=None, out=None, lazy1=False, lazy2=False):
    """Run two 1x1 Conv-BN-act modules that share the input `x` as ONE fused convolution (ops.ConvBnActPair) when their
    tensors are adjacent in the flat training arenas; None -> the caller runs them one by one (eval mode, no arena, SyncBN,
    odd channel counts, CVHIP_PAIR=0)."""
    if not _PAIR_ENABLED:
        return None
    f1, f2 = m1._fusable(True, True), m2._fusable(True, True)
    if f1 is None or f2 is None or f1[0] is None or f2[0] is None or f1[1] != f2[1]:
        return None
    c1, c2 = m1.conv, m2.conv
    if (c1.stride, c1.padding, c1.dilation, c1.groups) != ((1, 1), (0, 0), (1, 1), 1) or c1.bias is not None or c2.bias is not None:
        return None
    if (c2.stride, c2.padding, c2.dilation, c2.groups) != ((1, 1), (0, 0), (1, 1), 1):
        return None
    bn1, bn2 = f1[0], f2[0]
    if sync_of(bn1) is not None or sync_of(bn2) is not None:
        return None
    operands = ops.pair_operands(c1, bn1, c2, bn2)
    if operands is None:
        return None
    state = owner.__dict__.setdefault("_hip_pair_state", ops.ConvState())
    aid, ap = f1[1]
    cfg = ops.ConvCfg(c1.stride, c1.padding, c1.dilation, 1, aid, ap, has_bn=True, bn_training=True, momentum=bn1.momentum,
                      eps=bn1.eps, state=state, track=True)
    cfg.vkey = (id(c1.weight), c1.weight._version, id(c2.weight), c2.weight._version)
    bn_tick(bn1)
    bn_tick(bn2)
    cfg.out = out   # both halves side by side into one slice of a concat buffer (or None)
    cfg.lazy_half1 = bool(lazy1)   # the first sibling's consumers are Hip conv modules: its result may stay raw (ops.LazyAct)
    # the second sibling's concat slice may hold its RAW output (split store): the concat's only consumer is a Hip 1x1 conv module
    cfg.lazy_half2 = bool(lazy2) and out2 is not None
    cfg.acc_owner, cfg.acc_attr = bn1, "_hip_acc_pair"   # the pair's statistic accumulators (K1 + K2 channels) hang on the first layer
    return ops.conv_bn_act_pair(x, operands, cfg, out2)


class Focus(nn.Module):
    """yolo_modules.py:19-37. Space-to-depth (TL, BL, TR, BR) then ConvModule. For fp32 NCHW image input the
    gather is fused with the bf16/NHWC relayout (cvhip_focus_nchw_f32_to_nhwc_bf16)."""

    def __init__(self, in_channels, out_channels, kernel_sizes=1, stride=1, conv_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="Swish")):
        super().__init__()
        self.conv = ConvModule(in_channels * 4, out_channels, kernel_sizes, stride, padding=(kernel_sizes - 1) // 2,
                               conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x):
        c4 = x.shape[1] * 4
        if x.dtype.is_floating_point and x.dtype != ops.ACT_DTYPE and not x.requires_grad:
            xs = ops.images_to_nhwc(x, cpad=(c4 + 7) // 8 * 8, focus=True)
            return self.conv(xs)
        tl, tr = x[..., ::2, ::2], x[..., ::2, 1::2]
        bl, br = x[..., 1::2, ::2], x[..., 1::2, 1::2]
        return self.conv(ops.cat([tl, bl, tr, br]))


class DarknetBottleneck(nn.Module):
    """yolo_modules.py:40-104."""

    def __init__(self, in_channels, out_channels, expansion=0.5, shortcut=True, depthwise=False, conv_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="Swish"), init_cfg=None):
        super().__init__()
        hidden_channels = int(out_channels * expansion)
        conv = DepthwiseSeparableConvModule if depthwise else ConvModule
        self.conv1 = ConvModule(in_channels, hidden_channels, 1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv2 = conv(hidden_channels, out_channels, 3, stride=1, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                          act_cfg=act_cfg)
        self.shortcut = shortcut and in_channels == out_channels
        self.depthwise = depthwise

    def forward(self, x, out=None):
        """`out`: optional channel slice of a concat buffer for the block's result (see HipConvModule.forward)"""
        if self.shortcut and not self.depthwise:
            # the shortcut's gradient rides into conv1's dgrad epilogue (ops.GradLink) instead of an autograd accumulation add
            link = ops.GradLink() if (_GRAD_LINK and x.requires_grad and torch.is_grad_enabled()) else None
            h = self.conv1(x, dx_link=link)
            return self.conv2(h, residual=x, out=out, res_link=link)  # add fused into conv2's BN+act pass
        h = self.conv1(x)
        if self.depthwise or self.shortcut:
            h = self.conv2(h)
            return ops.add(h, x) if self.shortcut else h
        return self.conv2(h, out=out)


class CSPLayer(nn.Module):
    """yolo_modules.py:107-140 (C3)."""

    def __init__(self, in_channels, out_channels, n=1, expansion=0.5, shortcut=True, depthwise=False, conv_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="Swish")):
        super().__init__()
        hidden_channels = int(out_channels * expansion)
        self.conv1 = ConvModule(in_channels, hidden_channels, 1, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv2 = ConvModule(in_channels, hidden_channels, 1, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv3 = ConvModule(2 * hidden_channels, out_channels, 1, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.m = nn.Sequential(*[
            DarknetBottleneck(hidden_channels, hidden_channels, 1.0, shortcut, depthwise, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                              act_cfg=act_cfg) for _ in range(n)])

    def hip_sibling_pairs(self):
        """conv1 / conv2 read the same tensor: arena.FlatTrainState lays their parameters out back to back so that training
        can run them as one convolution (ops.ConvBnActPair)."""
        return [(self.conv1, self.conv2)]

    def forward(self, x):
        # concat elimination: conv2's result and the last bottleneck's result are written straight into the two halves of the
        # buffer conv3 reads (ops.cat then finds adjacent slices and copies nothing)
        N, _, H, W = x.shape
        hid = self.conv1.out_channels
        buf = None
        if _CAT_INPLACE and x.is_cuda and len(self.m) > 0 and not self.m[-1].depthwise and hid % 8 == 0:
            buf = ops.empty_nhwc(N, 2 * hid, H, W, x.device)
        out2 = buf[:, hid:] if buf is not None else None
        # conv1's result only feeds the first bottleneck (its 1x1 conv1 and, with a shortcut, conv2's residual operand): both read a
        # lazy activation on load, so the branch may stay raw (ops.LazyAct)
        lazy1 = (self.training and len(self.m) > 0 and isinstance(self.m[0], DarknetBottleneck) and not self.m[0].depthwise and x.is_cuda
                 and ops.lazy_edge_ok(N, hid, H, W, self.m[0].conv1.out_channels, ops.act_id_of(self.conv1)))
        # conv2's result only feeds the concat that conv3 reads: its slice may stay raw too (conv3 transforms that channel range on load)
        lazy2 = (_LAZY_CAT and self.training and buf is not None and x.is_cuda
                 and ops.lazy_edge_ok(N, 2 * hid, H, W, self.conv3.out_channels, ops.act_id_of(self.conv2)))
        pair = sibling_pair_forward(self.conv1, self.conv2, x, self, out2, lazy1=lazy1, lazy2=lazy2) if self.training else None
        if pair is not None:
            x_1, x_2 = pair
        else:
            x_1 = self.conv1(x, lazy=lazy1)
            x_2 = self.conv2(x, out=out2)
        for i, blk in enumerate(self.m):
            x_1 = blk(x_1, out=buf[:, :hid]) if (buf is not None and i == len(self.m) - 1) else blk(x_1)
        return self.conv3(ops.cat([x_1, x_2]))


class UpsamplingModule(nn.Module):
    """yolo_modules.py:143-152."""

    def __init__(self, c1, c2, layer=3, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="SiLU")):
        super().__init__()
        self.conv = ConvModule(c1, c2, 1, 1, 0, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.up = HipUpsampleNearest2x(scale_factor=2)
        self.fuse = CSPLayer(c2 * 2, c2, n=layer, shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x, y):
        x_conv, side = ops.fanout(self.conv(x), 2)   # -> the upsample + concat here and, as side output, a later DownsamplingModule
        return self.fuse(ops.upsample2x_cat(x_conv, y)), side


class DownsamplingModule(nn.Module):
    """yolo_modules.py:155-162."""

    def __init__(self, c1, c2, layer=3, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="SiLU")):
        super().__init__()
        self.down = ConvModule(c1, c1, 3, 2, 1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.fuse = CSPLayer(c1 * 2, c2, n=layer, shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x, y, dx_link=None):
        """`dx_link`: x is the main alias of ops.fanout_linked — the side consumer's gradient rides into the stride-2 conv's dgrad"""
        if _CAT_INPLACE and x.is_cuda and y.dim() == 4:
            c1 = self.down.out_channels
            buf = ops.empty_nhwc(y.shape[0], c1 + y.shape[1], y.shape[2], y.shape[3], y.device)
            return self.fuse(ops.cat([self.down(x, out=buf[:, :c1], dx_link=dx_link), y], into=buf))   # only y is copied
        return self.fuse(ops.cat([self.down(x, dx_link=dx_link), y]))


class SPPF(nn.Module):
    """yolo_modules.py:165-194: int kernel => chained k (SPPF), tuple => parallel pools (SPP)."""

    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), conv_cfg=None,
                 norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="Swish"), init_cfg=None):
        super().__init__()
        self.kernel_sizes = kernel_sizes
        hidden_channels = in_channels // 2
        self.conv1 = ConvModule(in_channels, hidden_channels, 1, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        if isinstance(kernel_sizes, int):
            self.m = HipMaxPool2d(kernel_size=kernel_sizes, stride=1, padding=kernel_sizes // 2)
        else:
            self.m = nn.ModuleList([HipMaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.conv2 = ConvModule(hidden_channels * 4, out_channels, 1, stride=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x):
        if (_SPPF_CHAIN and isinstance(self.kernel_sizes, int) and x.is_cuda and x.dim() == 4 and isinstance(self.conv1, ConvModule)
                and self.conv1.conv.out_channels % 8 == 0 and ops.nhwc_ld(x) is not None):
            # round 4: conv1 writes the first slice of the concat buffer, the three pools the other slices (ops.SppfChain)
            c = self.conv1.conv.out_channels
            buf = ops.empty_nhwc(x.shape[0], 4 * c, x.shape[2], x.shape[3], x.device)
            x0 = self.conv1(x, out=buf[:, :c])
            if x0.data_ptr() == buf.data_ptr():   # (the unfused fallback path of ConvModule ignores `out`)
                return self.conv2(ops.sppf_chain(x0, self.kernel_sizes))
            x = x0
        elif (_SPPF_CHAIN and not isinstance(self.kernel_sizes, int) and x.is_cuda and x.dim() == 4 and isinstance(self.conv1, ConvModule)
                and self.conv1.conv.out_channels % 8 == 0 and ops.nhwc_ld(x) is not None
                and all(isinstance(m, HipMaxPool2d) for m in self.m)):
            # round 6: the parallel form (SPP) on one buffer too (ops.SppParallel)
            c = self.conv1.conv.out_channels
            buf = ops.empty_nhwc(x.shape[0], (1 + len(self.kernel_sizes)) * c, x.shape[2], x.shape[3], x.device)
            x0 = self.conv1(x, out=buf[:, :c])
            if x0.data_ptr() == buf.data_ptr():
                return self.conv2(ops.spp_parallel(x0, self.kernel_sizes))
            x = x0
        else:
            x = self.conv1(x)
        if isinstance(self.kernel_sizes, int):
            x, xa = ops.fanout(x, 2)          # every tensor of the chain feeds the next pool AND the concat
            y1, y1a = ops.fanout(self.m(xa), 2)
            y2, y2a = ops.fanout(self.m(y1a), 2)
            x = ops.cat([x, y1, y2, self.m(y2a)])
        else:
            xs = ops.fanout(x, 1 + len(self.m))
            x = ops.cat([xs[0]] + [m(xi) for m, xi in zip(self.m, xs[1:])])
        return self.conv2(x)
