"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — CPU fp32 restatement of the reference's STDC path.

Pinned by tests/test_oracle_golden.py against fixtures captured from the reference's own classes (tools/gen_golden_more.py):
CatBottleneck (stride 1 and 2), AddBottleneck, a reduced-width STDCNet, AttentionRefinementModule, FeatureFusionModule,
STDCNeck, plus the full STDC1 parameter-name / count / output-shape contract.

  backbone : src/models/backbones/seg/stdcnet.py:18-27 (ConvX), :30-77 (AddBottleneck), :80-127 (CatBottleneck), :130-192 (STDCNet)
  neck     : src/models/necks/seg/stdc_neck.py:16-58 (ARM), :61-114 (FFM), :117-145 (STDCNeck)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .torch_ref import ConvModule


class ConvX(nn.Module):
    def __init__(self, in_planes, out_planes, kernel=3, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel, stride=stride, padding=kernel // 2, bias=False)
        self.bn = nn.BatchNorm2d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.relu(self.bn(self.conv(x)))


def _conv_list(in_planes, out_planes, block_num, stride):
    """stdcnet.py:50-60 / :95-105: channel schedule out/2, out/4, ..., last two equal."""
    convs = nn.ModuleList()
    for idx in range(block_num):
        if idx == 0:
            convs.append(ConvX(in_planes, out_planes // 2, kernel=1))
        elif idx == 1 and block_num == 2:
            convs.append(ConvX(out_planes // 2, out_planes // 2, stride=stride))
        elif idx == 1 and block_num > 2:
            convs.append(ConvX(out_planes // 2, out_planes // 4, stride=stride))
        elif idx < block_num - 1:
            convs.append(ConvX(out_planes // int(math.pow(2, idx)), out_planes // int(math.pow(2, idx + 1))))
        else:
            convs.append(ConvX(out_planes // int(math.pow(2, idx)), out_planes // int(math.pow(2, idx))))
    return convs


class AddBottleneck(nn.Module):
    def __init__(self, in_planes, out_planes, block_num=3, stride=1):
        super().__init__()
        self.stride = stride
        if stride == 2:
            self.avd_layer = nn.Sequential(nn.Conv2d(out_planes // 2, out_planes // 2, 3, 2, 1, groups=out_planes // 2, bias=False),
                                           nn.BatchNorm2d(out_planes // 2))
            self.skip = nn.Sequential(nn.Conv2d(in_planes, in_planes, 3, 2, 1, groups=in_planes, bias=False), nn.BatchNorm2d(in_planes),
                                      nn.Conv2d(in_planes, out_planes, 1, bias=False), nn.BatchNorm2d(out_planes))
        self.conv_list = _conv_list(in_planes, out_planes, block_num, 1)

    def forward(self, x):
        outs, out = [], x
        for idx, conv in enumerate(self.conv_list):
            out = self.avd_layer(conv(out)) if (idx == 0 and self.stride == 2) else conv(out)
            outs.append(out)
        if self.stride == 2:
            x = self.skip(x)
        return torch.cat(outs, dim=1) + x


class CatBottleneck(nn.Module):
    def __init__(self, in_planes, out_planes, block_num=3, stride=1):
        super().__init__()
        self.stride = stride
        if stride == 2:
            self.avd_layer = nn.Sequential(nn.Conv2d(out_planes // 2, out_planes // 2, 3, 2, 1, groups=out_planes // 2, bias=False),
                                           nn.BatchNorm2d(out_planes // 2))
            self.skip = nn.AvgPool2d(kernel_size=3, stride=2, padding=1)
        self.conv_list = _conv_list(in_planes, out_planes, block_num, 1)

    def forward(self, x):
        out1 = self.conv_list[0](x)
        outs = []
        out = None
        for idx, conv in enumerate(self.conv_list[1:]):
            if idx == 0:
                out = conv(self.avd_layer(out1)) if self.stride == 2 else conv(out1)
            else:
                out = conv(out)
            outs.append(out)
        if self.stride == 2:
            out1 = self.skip(out1)
        return torch.cat([out1] + outs, dim=1)


class STDCNet(nn.Module):
    def __init__(self, subtype="stdc1", out_channels=(32, 64, 256, 512, 1024), layers=(2, 2, 2), block_num=4, out_stages=(2, 3, 4)):
        super().__init__()
        oc = list(out_channels)
        self.out_stages = list(out_stages)
        self.stem = ConvX(3, oc[0], 3, 2)
        self.layer1 = ConvX(oc[0], oc[1], 3, 2)
        self.layer2 = self._make_layers(oc[1], oc[2], layers[0], block_num)
        self.layer3 = self._make_layers(oc[2], oc[3], layers[1], block_num)
        self.layer4 = self._make_layers(oc[3], oc[4], layers[2], block_num)
        self.out_channels = [oc[i] for i in self.out_stages]
        for m in self.modules():  # stdcnet.py init_weights: kaiming_normal(fan_out) convs, BN 1/0
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    @staticmethod
    def _make_layers(inplanes, planes, layer, block_num):
        feats = [CatBottleneck(inplanes, planes, block_num, 2)]
        feats += [CatBottleneck(planes, planes, block_num, 1) for _ in range(layer - 1)]
        return nn.Sequential(*feats)

    def forward(self, x):
        out = []
        x = self.stem(x)
        for i in range(1, 5):
            x = getattr(self, "layer%d" % i)(x)
            if i in self.out_stages:
                out.append(x)
        return out


class AttentionRefinementModule(nn.Module):
    def __init__(self, in_channels, out_channel, norm_cfg=dict(type="BN"), act_cfg=dict(type="ReLU")):
        super().__init__()
        self.conv_layer = ConvModule(in_channels, out_channel, 3, stride=1, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.atten_conv_layer = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)),
                                              ConvModule(out_channel, out_channel, 1, bias=False, norm_cfg=norm_cfg, act_cfg=None),
                                              nn.Sigmoid())

    def forward(self, x):
        x = self.conv_layer(x)
        return x * self.atten_conv_layer(x)


class FeatureFusionModule(nn.Module):
    def __init__(self, in_channels, out_channels, scale_factor=4, norm_cfg=dict(type="BN"), act_cfg=dict(type="ReLU")):
        super().__init__()
        ch = out_channels // scale_factor
        self.conv0 = ConvModule(in_channels, out_channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.attention = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)),
                                       ConvModule(out_channels, ch, 1, norm_cfg=None, bias=False, act_cfg=act_cfg),
                                       ConvModule(ch, out_channels, 1, norm_cfg=None, bias=False, act_cfg=None), nn.Sigmoid())

    def forward(self, spatial_inputs, context_inputs):
        x = self.conv0(torch.cat([spatial_inputs, context_inputs], dim=1))
        return x * self.attention(x) + x


class STDCNeck(nn.Module):
    def __init__(self, in_channels=(256, 512, 1024), out_channels=256, aux_out_channels=128, norm_cfg=dict(type="BN")):
        super().__init__()
        self.arms, self.convs = nn.ModuleList(), nn.ModuleList()
        for c in in_channels[1:]:
            self.arms.append(AttentionRefinementModule(c, aux_out_channels))
            self.convs.append(ConvModule(aux_out_channels, aux_out_channels, 3, padding=1, norm_cfg=norm_cfg))
        self.conv_avg = ConvModule(in_channels[-1], aux_out_channels, 1, norm_cfg=norm_cfg)
        self.ffm = FeatureFusionModule(in_channels[0] + aux_out_channels, out_channels)

    def forward(self, x):
        avg_feat = self.conv_avg(F.adaptive_avg_pool2d(x[-1], 1))
        feature_up = F.interpolate(avg_feat, x[-1].shape[2:], mode="nearest")
        arms_out = []
        for i in range(len(self.arms) - 1, -1, -1):
            x_arm = self.arms[i](x[i + 1]) + feature_up
            feature_up = self.convs[i](F.interpolate(x_arm, x[i].shape[2:], mode="nearest"))
            arms_out.append(feature_up)
        return self.ffm(x[0], arms_out[1]), [x[0]] + arms_out
