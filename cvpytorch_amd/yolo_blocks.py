"""YOLO building blocks on the HIP engine — same classes, constructor arguments and sub-module names
as the reference's src/models/modules/yolo_modules.py, so reference `state_dict`s load unchanged.

What differs from the reference is only *how* the glue ops execute:
  * DarknetBottleneck's shortcut add rides in the second ConvModule's BN+act pass (yolo_modules.py:102)
  * UpsamplingModule's nearest-x2 + cat is one kernel (yolo_modules.py:147,152)
  * CSP / SPPF / Downsampling concats are slice copies into one NHWC buffer whose backward hands out
    channel-slice views (yolo_modules.py:139,162,190)
"""
import torch.nn as nn

from . import ops
from .bricks import HipConvModule as ConvModule
from .bricks import HipDepthwiseSeparableConvModule as DepthwiseSeparableConvModule
from .bricks import HipMaxPool2d, HipUpsampleNearest2x


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
        if x.dtype.is_floating_point and x.dtype != ops.BF16 and not x.requires_grad:
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

    def forward(self, x):
        out = self.conv1(x)
        if self.shortcut and not self.depthwise:
            return self.conv2(out, residual=x)  # add fused into conv2's BN+act pass
        out = self.conv2(out)
        return ops.add(out, x) if self.shortcut else out


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

    def forward(self, x):
        x_1 = self.conv1(x)
        x_2 = self.conv2(x)
        x_1 = self.m(x_1)
        return self.conv3(ops.cat([x_1, x_2]))


class UpsamplingModule(nn.Module):
    """yolo_modules.py:143-152."""

    def __init__(self, c1, c2, layer=3, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="SiLU")):
        super().__init__()
        self.conv = ConvModule(c1, c2, 1, 1, 0, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.up = HipUpsampleNearest2x(scale_factor=2)
        self.fuse = CSPLayer(c2 * 2, c2, n=layer, shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x, y):
        x_conv = self.conv(x)
        return self.fuse(ops.upsample2x_cat(x_conv, y)), x_conv


class DownsamplingModule(nn.Module):
    """yolo_modules.py:155-162."""

    def __init__(self, c1, c2, layer=3, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="SiLU")):
        super().__init__()
        self.down = ConvModule(c1, c1, 3, 2, 1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.fuse = CSPLayer(c1 * 2, c2, n=layer, shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x, y):
        return self.fuse(ops.cat([self.down(x), y]))


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
        x = self.conv1(x)
        if isinstance(self.kernel_sizes, int):
            y1 = self.m(x)
            y2 = self.m(y1)
            x = ops.cat([x, y1, y2, self.m(y2)])
        else:
            x = ops.cat([x] + [m(x) for m in self.m])
        return self.conv2(x)
