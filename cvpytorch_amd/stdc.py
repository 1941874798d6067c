"""STDC backbone + neck on the HIP engine with the reference's module tree and state_dict keys.

  backbone : src/models/backbones/seg/stdcnet.py:18-27 (ConvX), :30-77 (AddBottleneck), :80-127 (CatBottleneck), :130-192 (STDCNet)
  neck     : src/models/necks/seg/stdc_neck.py:16-58 (ARM), :61-114 (FFM), :117-145 (STDCNeck)

Execution differences only: conv+BN(+ReLU) are single fused ops (also for the `nn.Sequential(conv, bn)` avd / skip layers),
the 3x3/s2 average pool runs on the depthwise-conv kernels, concats are slice copies into one NHWC buffer, the attention
gating is one scale pass with gradients to both operands.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lib as L
from . import ops
from .bricks import bn_tick  # noqa: E402
from .bricks import (HipAdaptiveAvgPool1x1, HipAvgPool2d, HipBN, HipConv2d, HipConvBN, HipSigmoid)
from .bricks import HipConvModule as ConvModule


class ConvX(ConvModule):
    """conv (bias=False, pad k//2) + BN + ReLU; sub-modules `conv`, `bn` as in stdcnet.py:18-27."""

    def __init__(self, in_planes, out_planes, kernel=3, stride=1):
        super().__init__(in_planes, out_planes, kernel, stride=stride, padding=kernel // 2, norm_cfg=dict(type="BN"), act_cfg=dict(type="ReLU"))


def _conv_list(in_planes, out_planes, block_num):
    convs = nn.ModuleList()
    for idx in range(block_num):
        if idx == 0:
            convs.append(ConvX(in_planes, out_planes // 2, kernel=1))
        elif idx == 1 and block_num == 2:
            convs.append(ConvX(out_planes // 2, out_planes // 2))
        elif idx == 1 and block_num > 2:
            convs.append(ConvX(out_planes // 2, out_planes // 4))
        elif idx < block_num - 1:
            convs.append(ConvX(out_planes // int(math.pow(2, idx)), out_planes // int(math.pow(2, idx + 1))))
        else:
            convs.append(ConvX(out_planes // int(math.pow(2, idx)), out_planes // int(math.pow(2, idx))))
    return convs


class _DwPwSkip(nn.Sequential):
    """AddBottleneck.skip: Sequential(dw3x3 s2, BN, 1x1, BN) (keys 0..3) as two fused conv+BN ops (stdcnet.py:43-48)."""

    def __init__(self, in_planes, out_planes):
        super().__init__(HipConv2d(in_planes, in_planes, 3, 2, 1, groups=in_planes, bias=False), HipBN(in_planes),
                         HipConv2d(in_planes, out_planes, 1, bias=False), HipBN(out_planes))

    def forward(self, x):
        for conv, bn in ((self[0], self[1]), (self[2], self[3])):
            xx, w = conv._effective(x)
            if not isinstance(bn, nn.BatchNorm2d):   # deploy.fuse_model folded the BN into the conv (bias) and left nn.Identity
                x = ops.conv_bn_act(xx, w, conv.bias, None, None, None, None, None, conv.make_cfg(L.ACT_NONE, 0.0, None))
                continue
            bn_tick(bn)
            x = ops.conv_bn_act(xx, w, None, bn.weight, bn.bias, bn.running_mean, bn.running_var, None, conv.make_cfg(L.ACT_NONE, 0.0, bn))
        return x


class AddBottleneck(nn.Module):
    def __init__(self, in_planes, out_planes, block_num=3, stride=1):
        super().__init__()
        assert block_num > 1
        self.stride = stride
        if stride == 2:
            self.avd_layer = HipConvBN(out_planes // 2, out_planes // 2, 3, 2, 1, groups=out_planes // 2)
            self.skip = _DwPwSkip(in_planes, out_planes)
        self.conv_list = _conv_list(in_planes, out_planes, block_num)

    def forward(self, x):
        outs, out = [], x
        for idx, conv in enumerate(self.conv_list):
            out = self.avd_layer(conv(out)) if (idx == 0 and self.stride == 2) else conv(out)
            outs.append(out)
        if self.stride == 2:
            x = self.skip(x)
        return ops.add(ops.cat(outs), x)


class CatBottleneck(nn.Module):
    def __init__(self, in_planes, out_planes, block_num=3, stride=1):
        super().__init__()
        assert block_num > 1
        self.stride = stride
        if stride == 2:
            self.avd_layer = HipConvBN(out_planes // 2, out_planes // 2, 3, 2, 1, groups=out_planes // 2)
            self.skip = HipAvgPool2d(kernel_size=3, stride=2, padding=1)
        self.conv_list = _conv_list(in_planes, out_planes, block_num)

    def forward(self, x):
        convs = list(self.conv_list)
        widths = [c.conv.out_channels for c in convs]
        # Round 6: the ConvX outputs go straight into their channel slices of the concat buffer (`out=`: the BatchNorm + ReLU pass of each
        # ConvModule writes there) — four copy kernels per block before; the stride-2 block's pooled first slice is still copied in
        buf = None
        if (x.is_cuda and x.dim() == 4 and ops.nhwc_ld(x) is not None and all(w % 8 == 0 for w in widths) and torch.is_grad_enabled()):
            N, _, H, W = x.shape
            Ho, Wo = ((H - 1) // 2 + 1, (W - 1) // 2 + 1) if self.stride == 2 else (H, W)
            buf = ops.empty_nhwc(N, sum(widths), Ho, Wo, x.device)
        offs = [sum(widths[:i]) for i in range(len(widths))]
        sl = (lambda i: buf[:, offs[i]:offs[i] + widths[i]]) if buf is not None else (lambda i: None)
        out1 = convs[0](x, out=sl(0) if self.stride == 1 else None)
        # every ConvX output but the last feeds the next ConvX AND the concat: the concat's share of its gradient rides into the next
        # convolution's dgrad epilogue (ops.fanout_linked) instead of an autograd add pass per tensor
        cat_in, cur = [], out1
        for idx, conv in enumerate(convs[1:]):
            if idx == 0 and self.stride == 2:
                nxt = conv(self.avd_layer(cur), out=sl(1))
                cat_in.append(self.skip(cur))
            else:
                xr, xs, lk = ops.fanout_linked(cur)
                nxt = conv(xr, out=sl(idx + 1), dx_link=lk)
                cat_in.append(ops.fanout_side(cur, xs, lk))
            cur = nxt
        cat_in.append(cur)
        return ops.cat(cat_in, into=buf)


class STDCNet(nn.Module):
    def __init__(self, subtype="stdc1", out_channels=(32, 64, 256, 512, 1024), layers=(2, 2, 2), block_num=4, out_stages=(2, 3, 4),
                 output_stride=32, classifier=False, num_classes=1000, backbone_path=None, pretrained=False):
        super().__init__()
        if classifier:
            raise L.CvhipError("STDCNet(classifier=True) is not built (segmentation feature extractor only)")
        oc = list(out_channels)
        self.subtype, self.out_stages = subtype, list(out_stages)
        self.stem = ConvX(3, oc[0], 3, 2)
        self.layer1 = ConvX(oc[0], oc[1], 3, 2)
        self.layer2 = self._make_layers(oc[1], oc[2], layers[0], block_num)
        self.layer3 = self._make_layers(oc[2], oc[3], layers[1], block_num)
        self.layer4 = self._make_layers(oc[3], oc[4], layers[2], block_num)
        self.out_channels = [oc[i] for i in self.out_stages]
        self.init_weights()

    @staticmethod
    def _make_layers(inplanes, planes, layer, block_num, block=CatBottleneck):
        feats = [block(inplanes, planes, block_num, 2)]
        feats += [block(planes, planes, block_num, 1) for _ in range(layer - 1)]
        return nn.Sequential(*feats)

    def init_weights(self):
        """stdcnet.py:211-223."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        out = []
        x = self.stem(x)
        for i in range(1, 5):
            x = getattr(self, "layer%d" % i)(x)
            if i in self.out_stages:
                out.append(x)
        return out if len(self.out_stages) > 1 else out[0]


class AttentionRefinementModule(nn.Module):
    def __init__(self, in_channels, out_channel, conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="ReLU"),
                 init_cfg=None):
        super().__init__()
        self.conv_layer = ConvModule(in_channels, out_channel, 3, stride=1, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.atten_conv_layer = nn.Sequential(HipAdaptiveAvgPool1x1(),
                                              ConvModule(out_channel, out_channel, 1, bias=False, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=None),
                                              HipSigmoid())

    def forward(self, x):
        x = self.conv_layer(x)
        return ops.channel_scale(x, self.atten_conv_layer(x))


class FeatureFusionModule(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor=4, norm_cfg=dict(type="BN"), act_cfg=dict(type="ReLU"), init_cfg=None):
        super().__init__()
        ch = out_channels // scale_factor
        self.conv0 = ConvModule(in_channels, out_channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.attention = nn.Sequential(HipAdaptiveAvgPool1x1(),
                                       ConvModule(out_channels, ch, 1, norm_cfg=None, bias=False, act_cfg=act_cfg),
                                       ConvModule(ch, out_channels, 1, norm_cfg=None, bias=False, act_cfg=None), HipSigmoid())

    def forward(self, spatial_inputs, context_inputs):
        x = self.conv0(ops.cat([spatial_inputs, context_inputs]))
        return ops.add(ops.channel_scale(x, self.attention(x)), x)


def _nearest_to(x, size):
    """F.interpolate(x, size, mode='nearest'): 1x1 -> HxW broadcast and exact x2 take their dedicated paths, every other size the
    general nearest-resize kernel (ops.resize_nearest) — nothing leaves the engine."""
    h, w = int(size[0]), int(size[1])
    if x.shape[2] == h and x.shape[3] == w:
        return x
    if x.shape[2] * 2 == h and x.shape[3] * 2 == w:
        return ops.upsample2x_cat(x, None)
    if x.shape[2] == 1 and x.shape[3] == 1:
        return x.expand(x.shape[0], x.shape[1], h, w)
    return ops.resize_nearest(x, (h, w))


class STDCNeck(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024), out_channels=256, aux_out_channels=128, norm_cfg=dict(type="BN"), **kwargs):
        super().__init__()
        self.in_channels, self.out_channels, self.aux_out_channels = list(in_channels), out_channels, aux_out_channels
        self.arms, self.convs = nn.ModuleList(), nn.ModuleList()
        for c in self.in_channels[1:]:
            self.arms.append(AttentionRefinementModule(c, aux_out_channels))
            self.convs.append(ConvModule(aux_out_channels, aux_out_channels, 3, padding=1, norm_cfg=norm_cfg))
        self.conv_avg = ConvModule(self.in_channels[-1], aux_out_channels, 1, norm_cfg=norm_cfg)
        self.ffm = FeatureFusionModule(in_channels=self.in_channels[0] + aux_out_channels, out_channels=out_channels)

    def forward(self, x):
        avg_feat = self.conv_avg(ops.global_avg_pool(x[-1]))
        feature_up = _nearest_to(avg_feat, x[-1].shape[2:])
        arms_out = []
        for i in range(len(self.arms) - 1, -1, -1):
            x_arm = ops.add(self.arms[i](x[i + 1]), feature_up)
            feature_up = self.convs[i](_nearest_to(x_arm, x[i].shape[2:]))
            arms_out.append(feature_up)
        return self.ffm(x[0], arms_out[1]), [x[0]] + arms_out
