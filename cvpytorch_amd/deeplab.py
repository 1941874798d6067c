"""DeepLabv3+ / ResNet-50-v1c on the HIP engine (BASELINE.json config 3), with the reference's module tree so
its `state_dict` keys line up (`backbone.stem.0.weight`, `backbone.layer1.0.conv1.weight`, `head.aspp.1.depthwise_conv.conv.weight` ...).

Reference files restated:
  backbone : src/models/backbones/seg/resnet.py:27-154 (deep 'v1c' stem :67-80, max-pool :80, torchvision resnet50
             layers :52-54,91-94 — torchvision's Bottleneck (stride on conv2, expansion 4, layers [3,4,6,3]) is a third-party
             dependency, restated from its public definition). NOTE the reference's output_stride rewrite only matches
             resnet18/34 (:102-118), so ResNet-50 stays OS-32 "as written"; `output_stride=8/16` here applies the INTENDED
             dilation (stride->1, dilation 2/4 on layer3/4 conv2) and is reported separately.
  head     : src/models/heads/seg/deeplabv3plus_head.py:14-68, deeplabv3_head.py:15-74, base_seg_head.py:13-37
  model    : src/models/segmentors/encoder_decoder.py:93-150 (bilinear resize of the logits to label size + CE, ignore 255)
"""
import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .bricks import bn_tick  # noqa: E402
from .bricks import HipBN, HipConv2d, HipMaxPool2d
from .bricks import HipConvModule as ConvModule
from .bricks import HipDepthwiseSeparableConvModule as DepthwiseSeparableConvModule


_GRAD_LINK = __import__("os").environ.get("CVHIP_GRAD_LINK", "1") != "0"
_FUSE_TAIL = __import__("os").environ.get("CVHIP_FUSE_TAIL", "1") != "0"


def _cba(x, conv, bn, act, residual=None, dx_link=None, res_pre=False, res_link=None):
    """conv -> bn -> act as ONE fused op (conv and bn are sibling modules, torchvision style); `res_pre`: the residual joins
    before the activation (bottleneck tail)."""
    folded = not isinstance(bn, nn.BatchNorm2d)   # deploy.fuse_model folded the BN into conv.weight / conv.bias (nn.Identity left)
    cfg = conv.make_cfg(act, 0.0, None if folded else bn)
    cfg.dx_link = dx_link
    cfg.res_pre = res_pre
    cfg.res_link = res_link if residual is not None else None
    xx, w = conv._effective(x)
    if folded:
        return ops.conv_bn_act(xx, w, conv.bias, None, None, None, None, residual, cfg)
    bn_tick(bn)
    return ops.conv_bn_act(xx, w, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, cfg)


class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5): 1x1 -> 3x3 (stride) -> 1x1 (x4), + identity, ReLU."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = HipConv2d(inplanes, planes, 1, bias=False)
        self.bn1 = HipBN(planes)
        self.conv2 = HipConv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = HipBN(planes)
        self.conv3 = HipConv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = HipBN(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, in_link=None):
        """`in_link`: x is the main alias of an ops.fanout_linked in the caller (the stage output that also feeds the head): the side
        consumer's gradient is folded into this block's projection-shortcut dgrad (projection blocks only; else the caller keeps it)"""
        # identity blocks: the skip connection's gradient is added in conv1's dgrad epilogue (ops.GradLink), not by autograd
        link = ops.GradLink() if (_GRAD_LINK and self.downsample is None and x.requires_grad and torch.is_grad_enabled()) else None
        xd, x_full = x, x
        if self.downsample is not None:
            # conv1 and the projection shortcut both read x: explicit fan-out; round 5: the shortcut's gradient rides into conv1's dgrad
            # epilogue (ops.fanout_linked) — the projection's backward runs right after the tail's, long before conv1's
            x, xd, link = ops.fanout_linked(x_full)
        out = _cba(x, self.conv1, self.bn1, L.ACT_RELU, dx_link=link)
        if self.downsample is not None:
            xd = ops.fanout_side(x_full, xd, link)
            link = None   # (not a skip-connection link: the tail below must not park the identity gradient in it)
        out = _cba(out, self.conv2, self.bn2, L.ACT_RELU)
        identity = x if self.downsample is None else _cba(xd, self.downsample[0], self.downsample[1], L.ACT_NONE, dx_link=in_link)
        if _FUSE_TAIL:   # relu(bn3(conv3(out)) + identity) in conv3's own BN pass (one pass over the 4x-wide tensor less)
            return _cba(out, self.conv3, self.bn3, L.ACT_RELU, residual=identity, res_pre=True, res_link=link)
        out = _cba(out, self.conv3, self.bn3, L.ACT_NONE)
        return ops.add_act(out, identity, L.ACT_RELU, link=link)


class ResNet(nn.Module):
    def __init__(self, subtype="resnet50v1c", out_stages=(1, 4), output_stride=32, classifier=False, num_classes=1000):
        super().__init__()
        if subtype not in ("resnet50", "resnet50v1c"):
            raise NotImplementedError(subtype)
        self.out_stages = list(out_stages)
        self.classifier = classifier
        self.deep_stem = subtype.endswith("c")
        if self.deep_stem:  # resnet.py:67-80
            self.stem = nn.Sequential(HipConv2d(3, 32, 3, stride=2, padding=1, bias=False), HipBN(32), nn.ReLU(inplace=True),
                                      HipConv2d(32, 32, 3, stride=1, padding=1, bias=False), HipBN(32), nn.ReLU(inplace=True),
                                      HipConv2d(32, 64, 3, stride=1, padding=1, bias=False), HipBN(64), nn.ReLU(inplace=True))
        else:
            self.stem = nn.Sequential(HipConv2d(3, 64, 7, stride=2, padding=3, bias=False), HipBN(64), nn.ReLU(inplace=True))
        self.maxpool = HipMaxPool2d(kernel_size=3, stride=2, padding=1)
        dilate3, dilate4 = {32: (False, False), 16: (False, True), 8: (True, True)}[output_stride]
        self.inplanes, self._dilation = 64, 1
        self.layer1 = self._make_layer(64, 3, 1, False)
        self.layer2 = self._make_layer(128, 4, 2, False)
        self.layer3 = self._make_layer(256, 6, 2, dilate3)
        self.layer4 = self._make_layer(512, 3, 2, dilate4)
        if classifier:
            self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():  # torchvision init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride, dilate):
        """torchvision ResNet._make_layer incl. replace_stride_with_dilation (first block keeps the previous dilation)."""
        previous_dilation = self._dilation
        if dilate:
            self._dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(HipConv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), HipBN(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, dilation=previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self._dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        for i in range(0, len(self.stem), 3):
            x = _cba(x, self.stem[i], self.stem[i + 1], L.ACT_RELU)
        x = self.maxpool(x)
        output = []
        link, pending = None, None
        for i in range(1, 5):
            blocks = list(getattr(self, "layer%d" % i))
            # a stage output that also feeds the head: its head-side gradient folds into the next stage's first (projection) block
            take = link is not None and blocks[0].downsample is not None
            x_in = x
            x = blocks[0](x_in, in_link=link) if take else blocks[0](x_in)
            if pending is not None:
                if take:
                    output.append(ops.fanout_side(pending[0], pending[1], link))
                else:   # (no projection block to fold into: the plain fan-out's summing alias)
                    output.append(pending[1] if pending[1] is not None else pending[0])
                pending = None
            link = None
            for blk in blocks[1:]:
                x = blk(x)
            if i in self.out_stages and not self.classifier:
                if i < 4:
                    nxt = getattr(self, "layer%d" % (i + 1))[0]
                    if getattr(nxt, "downsample", None) is not None:
                        x_full = x
                        x, keep, link = ops.fanout_linked(x_full)   # feeds the next layer and the head
                        pending = (x_full, keep)
                    else:
                        x, keep = ops.fanout(x, 2)
                        output.append(keep)
                else:
                    output.append(x)
        if self.classifier:
            # torchvision's avgpool + fc (seg/resnet.py:149-153): the fc is a 1x1 convolution on the pooled (N, 2048, 1, 1) map —
            # the engine's conv kernels (bias in the epilogue, fp32 master weight = the nn.Linear parameter viewed as K x C x 1 x 1)
            return self._fc(ops.global_avg_pool(x))
        return output if len(self.out_stages) > 1 else output[0]

    def _fc(self, pooled):
        """(N, C, 1, 1) -> logits (N, classes, 1, 1), NHWC storage dtype"""
        fc = self.fc
        st = self.__dict__.setdefault("_hip_fc_state", ops.ConvState())
        w = fc.weight.view(fc.out_features, fc.in_features, 1, 1)
        cfg = ops.ConvCfg((1, 1), (0, 0), (1, 1), state=st)
        cfg.vkey = (id(fc.weight), fc.weight._version)
        ar = getattr(fc.weight, "_hip_arena", None)
        if ar is not None and torch.is_grad_enabled():
            cfg.arena, cfg.gw, cfg.idx_w = ar[0], fc.weight._hip_grad.view_as(w), ar[1]
            if fc.bias is not None and getattr(fc.bias, "_hip_arena", None) is not None:
                cfg.gb, cfg.idx_b = fc.bias._hip_grad, fc.bias._hip_arena[1]
        return ops.conv_bn_act(pooled, w, fc.bias, None, None, None, None, None, cfg)


_CAT_INPLACE = __import__("os").environ.get("CVHIP_DEEPLAB_CAT_INPLACE", "1") != "0"   # A/B switch of the decoder concat elimination


class ASPP(nn.ModuleList):
    """deeplabv3_head.py:15-48 with the depthwise-separable replacement of deeplabv3plus_head.py:14-30."""

    def __init__(self, dilations, in_channels, channels, norm_cfg, act_cfg, depthwise=True):
        super().__init__()
        self.dilations = dilations
        for d in dilations:
            if d > 1 and depthwise:
                self.append(DepthwiseSeparableConvModule(in_channels, channels, 3, dilation=d, padding=d, norm_cfg=norm_cfg, act_cfg=act_cfg))
            else:
                self.append(ConvModule(in_channels, channels, 1 if d == 1 else 3, dilation=d, padding=0 if d == 1 else d,
                                       norm_cfg=norm_cfg, act_cfg=act_cfg))

    def forward(self, x):
        xs = x if isinstance(x, (tuple, list)) else ops.fanout(x, len(self))   # one alias per branch (ops.Fanout sums their gradients)
        return [m(xi) for m, xi in zip(self, xs)]


class _GAP(nn.Module):
    def forward(self, x):
        return ops.global_avg_pool(x)


class Deeplabv3PlusHead(nn.Module):
    def __init__(self, num_classes, in_channels=2048, channels=512, dilations=(1, 12, 24, 36), low_in_channels=256, low_channels=48,
                 dropout_ratio=0.1, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="ReLU")):
        super().__init__()
        self.num_classes, self.in_channels, self.channels, self.dilations = num_classes, in_channels, channels, dilations
        self.dropout_ratio = dropout_ratio
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None  # parameter-free; applied by ops.dropout2d
        self.cls_seg = HipConv2d(channels, num_classes, kernel_size=1)
        self.proj = nn.Sequential(_GAP(), ConvModule(in_channels, channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg))
        self.aspp = ASPP(dilations, in_channels, channels, norm_cfg, act_cfg, depthwise=True)
        self.reduce = ConvModule((len(dilations) + 1) * channels, channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.low_proj = ConvModule(low_in_channels, low_channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg) if low_in_channels > 0 else None
        self.fuse = nn.Sequential(
            DepthwiseSeparableConvModule(channels + low_channels, channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg),
            DepthwiseSeparableConvModule(channels, channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg))

    def classify(self, feat):
        if self.dropout is not None:
            feat = ops.dropout2d(feat, self.dropout_ratio, self.training)
        return self.cls_seg(feat)

    def forward(self, x):
        his = ops.fanout(x[1], 1 + len(self.aspp))   # the image-pooling branch and the ASPP branches all read the same map
        hi = his[0]
        outs = [ops.resize_bilinear(self.proj(hi), hi.shape[2:], False)]
        outs.extend(self.aspp(his[1:]))
        outs = self.reduce(ops.cat(outs))
        if self.low_proj is not None:
            lo = x[0]
            ch, lc = outs.shape[1], self.low_proj.out_channels
            if lo.is_cuda and ch % 8 == 0 and lc % 8 == 0 and _CAT_INPLACE:
                # concat elimination: the upsampled map and the projected low-level features are produced straight into their channel
                # slices of the buffer the fuse convs read (the x8 upsample is 537 MB at batch 16: the copy was 0.25 ms per step)
                buf = ops.empty_nhwc(lo.shape[0], ch + lc, lo.shape[2], lo.shape[3], lo.device)
                low = self.low_proj(lo, out=buf[:, ch:])
                up = ops.resize_bilinear(outs, lo.shape[2:], False, out=buf[:, :ch])
                outs = ops.cat([up, low])
            else:
                low = self.low_proj(lo)
                outs = ops.cat([ops.resize_bilinear(outs, low.shape[2:], False), low])
        return self.classify(self.fuse(outs))


class EncoderDecoder(nn.Module):
    """encoder_decoder.py:109-150: forward(imgs, targets, mode) -> {'ce_loss', 'loss'} | argmax map."""

    def __init__(self, num_classes=19, output_stride=32, dropout_ratio=0.1, ignore_index=255):
        super().__init__()
        self.num_classes, self.ignore_index = num_classes, ignore_index
        self.loss_capturable = True  # resize + cross-entropy are libcvhip ops: arena.FlatTrainStep captures ONE graph
        self.backbone = ResNet("resnet50v1c", out_stages=(1, 4), output_stride=output_stride)
        self.head = Deeplabv3PlusHead(num_classes, in_channels=2048, channels=512, dilations=(1, 12, 24, 36), low_in_channels=256,
                                      low_channels=48, dropout_ratio=dropout_ratio)

    def forward_features(self, imgs):
        return None, [self.head(self.backbone(imgs))]

    def loss_from_features(self, feats, targets):
        # resize to label size + CE as one fused pass (ops.seg_cross_entropy_resized): the label-resolution logits stay in registers
        ce = ops.seg_cross_entropy_resized(feats[0], targets, self.ignore_index, False)
        return {"ce_loss": ce, "loss": ce}

    def forward(self, imgs, targets=None, mode="infer", **kwargs):
        _, feats = self.forward_features(imgs)
        if mode == "train":
            return self.loss_from_features(feats, targets)
        size = targets.shape[-2:] if targets is not None else imgs.shape[-2:]
        return torch.argmax(ops.to_nchw_f32(ops.resize_bilinear(feats[0], size, False)), dim=1)
