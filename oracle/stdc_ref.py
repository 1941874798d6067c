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


# ---- STDC train path (round 6): heads, losses, the EncoderDecoder with auxiliary heads ---------------------------------------------------
# Pinned by tests/test_oracle_stdc_train.py against fixtures captured from the reference's own classes (tools/gen_golden_stdc_train.py).
#   heads     : src/models/heads/seg/fcn_head.py:14-63 (FCNHead), stdc_head.py:16-18 (STDCHead), base_seg_head.py:12-41 (dropout + cls_seg)
#   losses    : src/losses/seg/cross_entropy_loss.py:51-69 (OhemCrossEntropyLoss2d), src/losses/seg/detail_loss.py:11-88 (DetailAggregateLoss)
#   segmentor : src/models/segmentors/encoder_decoder.py:89-150 (loss_forward, forward with the auxiliary-head branch)
class FCNHead(nn.Module):
    def __init__(self, num_classes, in_channels, channels, num_convs=2, kernel_size=3, is_concat=True, dilation=1, dropout_ratio=0.1,
                 norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="ReLU")):
        super().__init__()
        self.is_concat = is_concat
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.cls_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        if num_convs == 0:
            self.convs = nn.Identity()
        else:
            pad = (kernel_size // 2) * dilation
            convs = [ConvModule(in_channels, channels, kernel_size, padding=pad, dilation=dilation, norm_cfg=norm_cfg, act_cfg=act_cfg)]
            for _ in range(num_convs - 1):
                convs.append(ConvModule(channels, channels, kernel_size, padding=pad, dilation=dilation, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.convs = nn.Sequential(*convs)
        if is_concat:
            self.conv_cat = ConvModule(in_channels + channels, channels, kernel_size, padding=kernel_size // 2, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x):
        feats = self.convs(x)
        if self.is_concat:
            feats = self.conv_cat(torch.cat([x, feats], dim=1))
        if self.dropout is not None:
            feats = self.dropout(feats)
        return self.cls_seg(feats)


class OhemCrossEntropyLoss2d(nn.Module):
    """the reference's control flow, literally: sort, branch on loss[min_kept], boolean mask / slice, mean"""
    loss_name = "ohem_ce_loss"

    def __init__(self, thresh=0.7, min_kept=100000, ignore_index=255, loss_weight=1.0):
        super().__init__()
        self.thresh = -torch.log(torch.tensor(thresh, dtype=torch.float))
        self.min_kept, self.ignore_index, self.loss_weight = min_kept, ignore_index, loss_weight

    def forward(self, pred, target):
        loss = self.loss_weight * F.cross_entropy(pred, target.long(), ignore_index=self.ignore_index, reduction="none").view(-1)
        loss, _ = torch.sort(loss, descending=True)
        if loss[self.min_kept] > self.thresh:
            loss = loss[loss > self.thresh]
        else:
            loss = loss[:self.min_kept]
        return torch.mean(loss)


class DetailAggregateLoss(nn.Module):
    loss_name = "detail_agg_loss"

    def __init__(self, loss_weight=1.0, bce_loss_weight=1.0, dice_loss_weight=1.0, boundary_threshold=0.1):
        super().__init__()
        self.loss_weight, self.bce_loss_weight, self.dice_loss_weight, self.thr = loss_weight, bce_loss_weight, dice_loss_weight, boundary_threshold
        self.lap = torch.tensor([-1, -1, -1, -1, 8, -1, -1, -1, -1], dtype=torch.float32).reshape(1, 1, 3, 3)
        self.fuse = torch.tensor([[6.0 / 10], [3.0 / 10], [1.0 / 10]], dtype=torch.float32).reshape(1, 3, 1, 1)

    def forward(self, boundary_logits, gtmasks):
        g = gtmasks.unsqueeze(1).float()

        def level(stride):
            t = F.conv2d(g, self.lap, stride=stride, padding=1).clamp(min=0)
            return t

        b1 = level(1)
        b1[b1 > self.thr] = 1
        b1[b1 <= self.thr] = 0
        ups = []
        for s in (2, 4):
            t = F.interpolate(level(s), b1.shape[2:], mode="nearest")
            t[t > self.thr] = 1
            t[t <= self.thr] = 0
            ups.append(t)
        pyr = F.conv2d(torch.stack((b1, ups[0], ups[1]), dim=1).squeeze(2), self.fuse)
        pyr[pyr > self.thr] = 1
        pyr[pyr <= self.thr] = 0
        if boundary_logits.shape[-1] != b1.shape[-1]:
            boundary_logits = F.interpolate(boundary_logits, b1.shape[2:], mode="bilinear", align_corners=True)
        bce = F.binary_cross_entropy_with_logits(boundary_logits, pyr)
        p = torch.sigmoid(boundary_logits)
        n = p.size(0)
        pf, tf = p.view(n, -1), pyr.view(n, -1)
        dice = (1 - ((2.0 * (pf * tf).sum(1) + 1.0) / (pf.sum(1) + tf.sum(1) + 1.0))).mean()
        return self.loss_weight * (self.bce_loss_weight * bce + self.dice_loss_weight * dice)


class STDCEncoderDecoder(nn.Module):
    """STDCNet -> STDCNeck -> head (+ auxiliary heads on the neck's auxiliary maps); the small-width configuration of the fixture
    (tools/gen_golden_stdc_train.py gen_encoder_decoder) or the full conf/seg/stdc/cityscapes_stdc1.yml:55-68"""

    def __init__(self, out_channels=(32, 64, 256, 512, 1024), neck_out=256, aux_out=128, head_channels=256, aux_channels=(64, 64, 64), num_classes=19,
                 min_kept=100000, dropout_ratio=0.1):
        super().__init__()
        oc = list(out_channels)
        self.backbone = STDCNet("stdc1", out_channels=oc, layers=[2, 2, 2], block_num=4, out_stages=[2, 3, 4])
        self.neck = STDCNeck(in_channels=oc[2:], out_channels=neck_out, aux_out_channels=aux_out)
        self.head = FCNHead(num_classes, neck_out, head_channels, num_convs=1, is_concat=False, dropout_ratio=dropout_ratio)
        self.auxiliary_head = nn.ModuleList([
            FCNHead(1, oc[2], aux_channels[0], num_convs=1, is_concat=False, dropout_ratio=dropout_ratio),
            FCNHead(num_classes, aux_out, aux_channels[1], num_convs=1, is_concat=False, dropout_ratio=dropout_ratio),
            FCNHead(num_classes, aux_out, aux_channels[2], num_convs=1, is_concat=False, dropout_ratio=dropout_ratio)])
        self.loss = [OhemCrossEntropyLoss2d(min_kept=min_kept)]
        self.auxiliary_loss = [DetailAggregateLoss(), OhemCrossEntropyLoss2d(min_kept=min_kept), OhemCrossEntropyLoss2d(min_kept=min_kept)]

    @staticmethod
    def loss_forward(preds, targets, loss):
        preds = F.interpolate(preds, size=targets.shape[-2:], mode="bilinear", align_corners=False)
        out = {}
        for l in (loss if isinstance(loss, (list, tuple)) else [loss]):
            out[l.loss_name] = out.get(l.loss_name, 0) + l(preds, targets)
        return out

    def forward(self, imgs, targets=None, mode="infer"):
        feats, aux_feats = self.neck(self.backbone(imgs))
        preds = self.head(feats)
        if mode != "train":
            return torch.argmax(F.interpolate(preds, size=targets.shape[-2:], mode="bilinear", align_corners=False), dim=1)
        losses = self.loss_forward(preds, targets, self.loss)
        for i, (h, f, l) in enumerate(zip(self.auxiliary_head, aux_feats, self.auxiliary_loss)):
            for k, v in self.loss_forward(h(f), targets, l).items():
                losses["aux%d_%s" % (i, k)] = v
        losses["loss"] = sum(losses.values())
        return losses
