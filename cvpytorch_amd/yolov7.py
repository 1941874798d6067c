"""YOLOv7 on the HIP engine (BASELINE config 5): E-ELAN / DownA-B / SPPCSPC / UpSampling / FeatureFusion /
RepConv blocks, neck, head, detect and the model wiring, with the reference's module tree and state_dict keys.

  blocks  : src/models/modules/yolov7_modules.py:20-33 (Conv), :36-61 (DownA/DownB), :64-82 (EELAN), :85-95 (UpSampling),
            :98-120 (FeatureFusion — conv4 applied three times, conv5/conv6 never called), :122-140 (SPPCSPC), :168-213 (RepConv)
  neck    : src/models/necks/yolov7_neck.py:13-55          head : src/models/heads/yolov7_head.py:12-40
  detect  : src/models/detects/yolov7_detect.py:71-122     model: src/models/yolov7.py:150-256
The reference's backbone for this config is a stub (backbones/det/yolov7_csp_vovnet.py:46-56 builds empty stages; the yml
names a class that does not exist): `YOLOv7Backbone` is the public YOLOv7-l layout built from the reference's blocks, whose
outputs (512/1024/1024 @ /8,/16,/32) are what the reference neck's in_channels require (conf/coco_yolov7.yml:67).
Loss for config 5: YOLOv5-style dense loss with YOLOv7 gains/anchors; the OTA loss is a "next" row.

What differs from the reference is only how the glue executes: concats are slice copies into one NHWC buffer (ops.cat),
UpSampling's nearest-x2 + cat is one kernel, RepConv's three-branch sum + SiLU is two fused passes.
"""
import math

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .bricks import bn_tick, sync_of  # noqa: E402
from .bricks import HipBN, HipConv2d, HipConvBN, HipConvModule, HipMaxPool2d, HipSiLU
from .yolo_blocks import sibling_pair_forward
from .yolov5 import YOLOv5Loss, YOLOv5LossFused, targets_to_tensor, non_max_suppression

ANCHORS = [[[1.50000, 2.00000], [2.37500, 4.50000], [5.00000, 3.50000]],
           [[2.25000, 4.68750], [4.75000, 3.43750], [4.50000, 9.12500]],
           [[4.43750, 3.43750], [6.00000, 7.59375], [14.34375, 12.53125]]]


def _torch_default_conv_init(conv):
    conv.reset_parameters()
    conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)


def _siblings(m1, m2, x, owner, out=None):
    """two 1x1 Conv modules on the same input: one fused convolution while training on the flat arenas (yolo_blocks.
    sibling_pair_forward), the two modules one after the other otherwise. `out`: channel slice (of a concat buffer) that
    receives both results side by side."""
    pair = sibling_pair_forward(m1, m2, x, owner, out=out) if owner.training else None
    if pair is not None:
        return pair
    k1 = m1.out_channels
    return (m1(x), m2(x)) if out is None else (m1(x, out=out[:, :k1]), m2(x, out=out[:, k1:]))


class Conv(HipConvModule):
    """yolov7_modules.py:20-33: Conv2d(bias=False, autopad) + BatchNorm2d + SiLU, sub-modules `conv`, `bn`, (`act`)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__(c1, c2, k, stride=s, padding=(k // 2 if p is None else p), groups=g, norm_cfg=dict(type="BN"),
                         act_cfg=dict(type="SiLU") if act is True else None)
        _torch_default_conv_init(self.conv)


class DownA(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.branch1 = nn.Sequential(HipMaxPool2d(kernel_size=2, stride=2), Conv(c1, c2, 1, 1))
        self.branch2 = nn.Sequential(Conv(c1, c2, 1, 1), Conv(c2, c2, 3, 2))

    def _branches(self, x, extra=0):
        """both branch results written straight into one concat buffer (plus `extra` trailing channels for DownB's y)"""
        c2 = self.branch1[1].out_channels
        # the pool branch and the conv branch read x: explicit fan-out. Round 5: the conv branch runs FIRST and is the main consumer
        # (ops.fanout_linked) — the pool branch's gradient rides into its 1x1 conv's dgrad instead of an add pass
        x_full = x
        xb, xs, link = ops.fanout_linked(x_full)
        h = self.branch2[0](xb, dx_link=link)
        x = ops.fanout_side(x_full, xs, link)
        a = self.branch1[0](x)
        if not (x.is_cuda and c2 % 8 == 0 and extra % 8 == 0):
            return None, self.branch1[1](a), self.branch2[1](h)
        buf = ops.empty_nhwc(a.shape[0], 2 * c2 + extra, a.shape[2], a.shape[3], x.device)
        b2 = self.branch2[1](h, out=buf[:, c2:2 * c2])
        b1 = self.branch1[1](a, out=buf[:, :c2])
        return buf, b1, b2

    def forward(self, x):
        _, b1, b2 = self._branches(x)
        return ops.cat([b1, b2])


class DownB(DownA):
    def forward(self, x, y):
        buf, b1, b2 = self._branches(x, extra=y.shape[1])
        return ops.cat([b1, b2, y], into=buf)   # only y is copied


class EELAN(nn.Module):
    def __init__(self, c1, c2, c3):
        super().__init__()
        self.conv1 = Conv(c1, c2, 1, 1)
        self.conv2 = Conv(c1, c2, 1, 1)
        self.conv3 = nn.Sequential(Conv(c2, c2, 3, 1), Conv(c2, c2, 3, 1))
        self.conv4 = nn.Sequential(Conv(c2, c2, 3, 1), Conv(c2, c2, 3, 1))
        self.conv5 = Conv(c2 * 4, c3, 1, 1)

    def hip_sibling_pairs(self):
        return [(self.conv1, self.conv2)]

    def forward(self, x):
        # concat elimination: all four producers write into their slice of the buffer conv5 reads
        c2 = self.conv1.out_channels
        buf = ops.empty_nhwc(x.shape[0], 4 * c2, x.shape[2], x.shape[3], x.device) if (x.is_cuda and c2 % 8 == 0) else None
        x1, x2 = _siblings(self.conv1, self.conv2, x, self, out=None if buf is None else buf[:, :2 * c2])
        # x2 and x3 feed the concat AND the next pair of convs: the concat-side gradient (a slice of conv5's input gradient, ready before
        # anything upstream runs) rides into the next conv's dgrad (ops.fanout_linked) instead of an add pass
        x2n, x2s, l2 = ops.fanout_linked(x2)
        h3 = self.conv3[0](x2n, dx_link=l2)
        x2 = ops.fanout_side(x2, x2s, l2)
        t3 = self.conv3[1](h3) if buf is None else self.conv3[1](h3, out=buf[:, 2 * c2:3 * c2])
        x3n, x3s, l3 = ops.fanout_linked(t3)
        h4 = self.conv4[0](x3n, dx_link=l3)
        x3 = ops.fanout_side(t3, x3s, l3)
        x4 = self.conv4[1](h4) if buf is None else self.conv4[1](h4, out=buf[:, 3 * c2:])
        return self.conv5(ops.cat([x1, x2, x3, x4]))


class UpSampling(nn.Module):
    def __init__(self, c1, c2, c3):
        super().__init__()
        self.conv1 = Conv(c1, c3, 1, 1)
        self.upsampling = nn.UpsamplingNearest2d(scale_factor=2)  # parameter-free; executed by cvhip_upsample2x_cat
        self.conv2 = Conv(c2, c3, 1, 1)

    def forward(self, x, y):
        return ops.upsample2x_cat(self.conv1(x), self.conv2(y))


class FeatureFusion(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        mid = c2 // 2
        self.conv1 = Conv(c1, c2, 1, 1)
        self.conv2 = Conv(c1, c2, 1, 1)
        self.conv3 = Conv(c2, mid, 3, 1)
        self.conv4 = Conv(mid, mid, 3, 1)
        self.conv5 = Conv(mid, mid, 3, 1)  # in the state_dict, never called (yolov7_modules.py:113-120)
        self.conv6 = Conv(mid, mid, 3, 1)
        self.conv7 = Conv(c2 * 4, c2, 1, 1)

    def hip_sibling_pairs(self):
        return [(self.conv1, self.conv2)]

    def forward(self, x):
        c2, mid = self.conv1.out_channels, self.conv3.out_channels
        if x.is_cuda and c2 % 8 == 0 and mid % 8 == 0:   # all six producers write into their slice of the buffer conv7 reads
            buf = ops.empty_nhwc(x.shape[0], 2 * c2 + 4 * mid, x.shape[2], x.shape[3], x.device)
            o = 2 * c2
            x1, x2 = _siblings(self.conv1, self.conv2, x, self, out=buf[:, :o])
            outs = [buf[:, o:o + mid], buf[:, o + mid:o + 2 * mid], buf[:, o + 2 * mid:o + 3 * mid], buf[:, o + 3 * mid:]]
        else:
            x1, x2 = _siblings(self.conv1, self.conv2, x, self)
            outs = [None] * 4
        # every tensor of the chain feeds the concat AND the next conv: the concat-side gradient rides into that conv's dgrad
        # (ops.fanout_linked; round 5) instead of an add pass per link of the chain
        chain, t = [], x2
        for conv, dst in zip((self.conv3, self.conv4, self.conv4, self.conv4), outs):
            tn, ts, lk = ops.fanout_linked(t)
            nxt = conv(tn, out=dst, dx_link=lk) if dst is not None else conv(tn, dx_link=lk)
            chain.append(ops.fanout_side(t, ts, lk))
            t = nxt
        x2, x3, x4, x5 = chain
        x6 = t
        return self.conv7(ops.cat([x1, x2, x3, x4, x5, x6]))


class SPPCSPC(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5, k=(5, 9, 13)):
        super().__init__()
        c_ = int(2 * c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(c_, c_, 3, 1)
        self.cv4 = Conv(c_, c_, 1, 1)
        self.m = nn.ModuleList([HipMaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])
        self.cv5 = Conv(4 * c_, c_, 1, 1)
        self.cv6 = Conv(c_, c_, 3, 1)
        self.cv7 = Conv(2 * c_, c2, 1, 1)

    def hip_sibling_pairs(self):
        return [(self.cv1, self.cv2)]

    def forward(self, x):
        c_ = self.cv1.out_channels
        inplace = x.is_cuda and c_ % 8 == 0
        a, b = _siblings(self.cv1, self.cv2, x, self)
        if inplace:
            pool_buf = ops.empty_nhwc(x.shape[0], 4 * c_, x.shape[2], x.shape[3], x.device)
            xs = ops.fanout(self.cv4(self.cv3(a), out=pool_buf[:, :c_]), 1 + len(self.m))   # the concat and the three pools read x1
            pooled = ops.cat([xs[0]] + [m(xi) for m, xi in zip(self.m, xs[1:])], into=pool_buf)      # the pools are copied, x1 is in place
            out_buf = ops.empty_nhwc(x.shape[0], 2 * c_, x.shape[2], x.shape[3], x.device)
            y1 = self.cv6(self.cv5(pooled), out=out_buf[:, :c_])
            return self.cv7(ops.cat([y1, b], into=out_buf))
        xs = ops.fanout(self.cv4(self.cv3(a)), 1 + len(self.m))
        y1 = self.cv6(self.cv5(ops.cat([xs[0]] + [m(xi) for m, xi in zip(self.m, xs[1:])])))
        return self.cv7(ops.cat([y1, b]))


class RepConv(nn.Module):
    """Training-time RepConv: SiLU(BN(conv3x3(x)) + BN(conv1x1(x)) [+ BN(x)]) — yolov7_modules.py:168-213."""

    def __init__(self, c1, c2, k=3, s=1, p=None, g=1, act=True, deploy=False):
        super().__init__()
        if deploy or k != 3 or g != 1:
            raise L.CvhipError("RepConv: only the training-time k=3 form is built (re-parameterised deploy form: next row)")
        self.in_channels, self.out_channels = c1, c2
        self.act = HipSiLU() if act is True else nn.Identity()
        self.rbr_identity = HipBN(c1) if c2 == c1 and s == 1 else None
        self.rbr_dense = HipConvBN(c1, c2, k, s, 1)
        self.rbr_1x1 = HipConvBN(c1, c2, 1, s, 0)
        for m in (self.rbr_dense[0], self.rbr_1x1[0]):
            _torch_default_conv_init(m)

    def forward(self, x):
        if hasattr(self, "rbr_reparam"):  # deploy form (cvpytorch_amd.deploy.reparam_repconv): one 3x3 conv + bias, then the activation
            return self.act(self.rbr_reparam(x))
        # every branch sum rides in the next branch's BN-apply pass (the `residual` operand of cvhip_bn_act_fwd)
        a = self.rbr_dense(x)
        bn = self.rbr_identity
        if bn is not None:
            bn_tick(bn)
            a = ops.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, a, True, bn.training or bn.running_mean is None,
                           bn.momentum, bn.eps, L.ACT_NONE, 0.0, bn.track_running_stats and bn.training, sync_of(bn))
        return self.act(self.rbr_1x1(x, residual=a))


def _bn_fix(module):
    for m in module.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03


class YOLOv7Neck(nn.Module):
    def __init__(self, in_channels=(512, 1024, 1024), out_channels=(128, 256, 512), depth_mul=1.0, width_mul=1.0):
        super().__init__()
        ic = [max(round(x * width_mul), 1) for x in in_channels]
        oc = [max(round(x * width_mul), 1) for x in out_channels]
        self.spp = SPPCSPC(ic[2], ic[0])
        self.up1_1 = UpSampling(ic[0], ic[1], oc[1])
        self.featurefusion1_1 = FeatureFusion(oc[1] * 2, oc[1])
        self.up1_2 = UpSampling(oc[1], ic[0], oc[0])
        self.featurefusion1_2 = FeatureFusion(oc[0] * 2, oc[0])
        self.down2_1 = DownB(oc[0], oc[0])
        self.featurefusion2_1 = FeatureFusion(oc[1] * 2, oc[1])
        self.down2_2 = DownB(oc[1], oc[1])
        self.featurefusion2_2 = FeatureFusion(oc[2] * 2, oc[2])
        _bn_fix(self)

    def forward(self, x):
        x3, x4, x5 = x
        x5, x5b = ops.fanout(self.spp(x5), 2)                                        # -> up1_1 and down2_2
        x4_up, x4_upb = ops.fanout(self.featurefusion1_1(self.up1_1(x5, x4)), 2)     # -> up1_2 and down2_1
        x3_up, x3_out = ops.fanout(self.featurefusion1_2(self.up1_2(x4_up, x3)), 2)  # -> down2_1 and the head
        x4_down, x4_out = ops.fanout(self.featurefusion2_1(self.down2_1(x3_up, x4_upb)), 2)   # -> down2_2 and the head
        x5_down = self.featurefusion2_2(self.down2_2(x4_down, x5b))
        return [x3_out, x4_out, x5_down]


class YOLOv7Head(nn.Module):
    def __init__(self, in_channels=(128, 256, 512), out_channels=(256, 512, 1024), depth_mul=1.0, width_mul=1.0):
        super().__init__()
        ic = [int(x * width_mul) for x in in_channels]
        oc = [int(x * width_mul) for x in out_channels]
        self.conv1, self.conv2, self.conv3 = RepConv(ic[0], oc[0]), RepConv(ic[1], oc[1]), RepConv(ic[2], oc[2])
        _bn_fix(self)

    def forward(self, x):
        return [self.conv1(x[0]), self.conv2(x[1]), self.conv3(x[2])]


class YOLOv7Detect(nn.Module):
    """yolov7_detect.py:71-122: same per-pixel 1x1 conv + (N,3,H,W,85) permute + sigmoid/grid/anchor decode as YOLOv5's."""

    def __init__(self, num_classes=80, in_channels=(256, 512, 1024), stride=(8., 16., 32.), anchors=ANCHORS, depth_mul=1.0, width_mul=1.0):
        super().__init__()
        ic = [int(x * width_mul) for x in in_channels]
        self.num_classes, self.num_outputs = num_classes, num_classes + 5
        self.num_layers, self.num_anchors = len(anchors), len(anchors[0])
        self.stride = list(stride)
        a = torch.tensor(anchors).float().view(self.num_layers, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", (a.clone() * torch.tensor(stride).view(-1, 1, 1)).view(self.num_layers, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(HipConv2d(x, self.num_outputs * self.num_anchors, 1) for x in ic)
        for mi, s in zip(self.m, self.stride):
            b = mi.bias.view(self.num_anchors, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (self.num_classes - 0.99))
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def forward_raw(self, x):
        return [self.m[i](x[i]) for i in range(self.num_layers)]

    def decode(self, raw):
        anchors_px = [self.anchors[i] * self.stride[i] for i in range(self.num_layers)]
        return ops.yolov5_decode(raw, self.stride, anchors_px, self.num_anchors, self.num_outputs)

    def forward(self, x):
        raw = self.forward_raw(x)
        train_out = [ops.head_permute(r, self.num_anchors, self.num_outputs) for r in raw]
        if self.training:
            return None, train_out
        return self.decode(raw), train_out


class YOLOv7Backbone(nn.Module):
    def __init__(self, width_mul=1.0):
        super().__init__()
        w = lambda c: max(int(c * width_mul), 1)  # noqa: E731
        self.stem = nn.Sequential(Conv(3, w(32), 3, 1), Conv(w(32), w(64), 3, 2), Conv(w(64), w(64), 3, 1))
        self.stage1 = nn.Sequential(Conv(w(64), w(128), 3, 2), EELAN(w(128), w(64), w(256)))
        self.stage2 = nn.Sequential(DownA(w(256), w(128)), EELAN(w(256), w(128), w(512)))
        self.stage3 = nn.Sequential(DownA(w(512), w(256)), EELAN(w(512), w(256), w(1024)))
        self.stage4 = nn.Sequential(DownA(w(1024), w(512)), EELAN(w(1024), w(256), w(1024)))
        _bn_fix(self)

    def forward(self, x):
        x = self.stage1(self.stem(x))
        p3, p3n = ops.fanout(self.stage2(x), 2)    # each stage output feeds the next stage and the neck
        p4, p4n = ops.fanout(self.stage3(p3n), 2)
        p5 = self.stage4(p4n)
        return [p3, p4, p5]


class YOLOv7(nn.Module):
    """src/models/yolov7.py:150-256. forward(imgs, targets, mode): 'train' -> losses dict; 'val' -> (losses, outputs)."""
    anchors = ANCHORS

    def __init__(self, num_classes=80, width_mul=1.0, max_targets=None, fused_loss=False, loss="v5", max_per_image=32):
        """loss="v5": the YOLOv5-style assignment (fused_loss=True -> libcvhip kernels, graph-capturable);
        loss="ota": the reference's YOLOv7Loss (find_3_positive + OTA matching): fused_loss=True -> libcvhip kernels on the raw head maps
        (cvhip_ota_assign + cvhip_yolov5_loss_level_fwd_assigned: the whole step is ONE hipGraph), fused_loss=False -> the fixed-shape
        torch-op restatement (YOLOv7OTALoss)."""
        super().__init__()
        assert loss in ("v5", "ota")
        self.num_classes = num_classes
        self.fused_loss = fused_loss
        self.loss_kind = loss
        self.loss_capturable = fused_loss
        self.backbone = YOLOv7Backbone(width_mul)
        self.neck = YOLOv7Neck(width_mul=width_mul)
        self.head = YOLOv7Head(width_mul=width_mul)
        self.detect = YOLOv7Detect(num_classes, width_mul=width_mul)
        if loss == "ota" and fused_loss:
            self.loss = YOLOv7OTALossFused(num_classes, anchors=ANCHORS, max_per_image=max_per_image)
        elif loss == "ota":
            self.loss = YOLOv7OTALoss(num_classes, anchors=ANCHORS, max_per_image=max_per_image)
        else:
            self.loss = (YOLOv5LossFused if fused_loss else YOLOv5Loss)(num_classes, anchors=ANCHORS, hyp_box=0.05, hyp_obj=0.7, hyp_cls=0.3)
        self.conf_thres, self.iou_thres = 0.001, 0.65
        self.max_targets = max_targets
        _bn_fix(self)

    def forward_features(self, imgs):
        self._img_h = int(imgs.shape[2])                      # the reference's OTA matching scales boxes by imgs[b].shape[1]
        x = self.head(self.neck(self.backbone(imgs)))
        if self.fused_loss:
            raw = self.detect.forward_raw(x)
            return (None if self.training else self.detect.decode(raw)), raw
        return self.detect(x)

    def loss_from_features(self, train_out, gts):
        losses = {}
        if self.loss_kind == "ota" and self.fused_loss:
            losses["loss"], st = self.loss(train_out, gts, self._img_h)
        elif self.loss_kind == "ota":
            losses["loss"], st = self.loss([t.float() for t in train_out], gts, self._img_h)
        else:
            losses["loss"], st = self.loss(train_out, gts)
        losses["box_loss"], losses["obj_loss"], losses["cls_loss"] = st[0], st[1], st[2]
        return losses

    def forward(self, imgs, targets=None, mode="infer", **kwargs):
        if mode == "infer":
            return
        gts = targets if torch.is_tensor(targets) else targets_to_tensor(targets, self.max_targets, imgs.device)
        out, train_out = self.forward_features(imgs)
        losses = self.loss_from_features(train_out, gts)
        if mode == "val":
            outputs = []
            if out is not None:
                for pred in non_max_suppression(out, self.conf_thres, self.iou_thres, multi_label=True):
                    outputs.append({"boxes": pred[:, :4], "labels": pred[:, 5], "scores": pred[:, 4]})
            return losses, outputs
        return losses


# ------------------------------------------------------------------------------------------------------
# OTA loss (the reference's YOLOv7Loss) in fixed-shape form
# ------------------------------------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402

from .yolov5 import bbox_ciou_xywh  # noqa: E402


def flat_to_padded(targets, batch, max_per_image):
    """(T, 6) [img, cls, cx, cy, w, h] in image order, img < 0 = padding  ->  (B, G, 6) per-image padded + valid mask (B, G).
    No host sync: the rank of a target inside its image comes from a bincount / cumsum."""
    T = targets.shape[0]
    dev = targets.device
    img = targets[:, 0].long()
    valid = img >= 0
    imc = img.clamp(0, batch - 1)
    counts = torch.zeros(batch + 1, dtype=torch.long, device=dev).index_add_(
        0, torch.where(valid, imc, torch.full_like(imc, batch)), torch.ones_like(imc))[:batch]       # bincount() would sync
    first = torch.cumsum(counts, 0) - counts
    rank = torch.arange(T, device=dev) - first[imc]
    ok = valid & (rank < max_per_image) & (rank >= 0)
    slot = torch.where(ok, imc * max_per_image + rank, torch.full_like(imc, batch * max_per_image))
    out = torch.zeros((batch * max_per_image + 1, 6), device=dev, dtype=targets.dtype)
    out[:, 0] = -1
    out = out.index_copy(0, slot, torch.where(ok[:, None], targets, out[-1:].expand(T, 6)))
    out = out[:-1].view(batch, max_per_image, 6)
    return out, out[..., 0] >= 0


class YOLOv7OTALossFused(nn.Module):
    """The reference's YOLOv7Loss (src/losses/yolov7_loss.py:129-420) on libcvhip kernels, straight on the raw 16-bit NHWC head maps
    [(N, A*NO, H, W)]: cvhip_ota_assign (find_3_positive candidates pooled per image, IoU + cost per gt, dynamic-k, conflict
    resolution — one wave per gt) decides the positives, the YOLOv5-form loss kernels evaluate CIoU / class / objectness and their
    gradients on that assignment. No torch autograd ops and no host sync: YOLOv7(loss="ota", fused_loss=True) captures as one graph.
    targets: (T, 6) flat rows [img, cls, cx, cy, w, h], rows of an image contiguous, img < 0 = padding."""

    def __init__(self, num_classes, stride=(8., 16., 32.), anchors=ANCHORS, max_per_image=32):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers, self.num_anchors = len(anchors), len(anchors[0])
        self.anchors = [[list(map(float, a)) for a in lvl] for lvl in anchors]
        self.stride = [float(s) for s in stride]
        self.anchor_t, self.hyp_box, self.hyp_obj, self.hyp_cls = 4.0, 0.05, 0.7, 0.3
        self.balance = [4.0, 1.0, 0.4]
        self.max_per_image = int(max_per_image)
        self._const_cache = {}
        self.ota = None
        self.last_assign = None

    def forward(self, raws, targets, img_size):
        self.ota = dict(G=self.max_per_image, img_size=float(img_size), stride=self.stride)
        total, stats3 = ops.yolov5_loss_fused(list(raws), targets, self)
        return total, torch.cat((stats3, stats3.sum().reshape(1)))


class YOLOv7OTALoss(nn.Module):
    """src/losses/yolov7_loss.py:129-420 (find_3_positive + per-image OTA matching + CIoU/obj/cls) with FIXED shapes: the
    per-image python loop, boolean-mask compaction and per-gt `.item()` top-k become dense (B, G, Nc = L*5*na*G) tensors with
    masks — no host sync. p: list of (B, na, H, W, 5+nc) fp32; targets (T, 6) flat (image order, img < 0 padding);
    img_size = the reference's `imgs[b].shape[1]`. Equal to the reference on its golden vectors (tests/test_yolov7_ota.py)."""

    def __init__(self, num_classes, stride=(8., 16., 32.), anchors=ANCHORS, max_per_image=32):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers, self.num_anchors = len(anchors), len(anchors[0])
        self.stride = [float(s) for s in stride]
        self.register_buffer("anchors", torch.tensor(anchors).float())
        self.register_buffer("off", torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]).float() * 0.5)
        self.hyp_anchor_t, self.hyp_box, self.hyp_obj, self.hyp_cls = 4.0, 0.05, 0.7, 0.3
        self.balance = [4.0, 1.0, 0.4]
        self.max_per_image = max_per_image

    def _candidates(self, tg, valid, i, ny, nx):
        """level i: (B, Ncl = 5*na*G) candidate attributes in the reference's (offset, anchor, target) order."""
        B, G, _ = tg.shape
        na = self.num_anchors
        anchors = self.anchors[i]
        gx, gy, gw, gh = tg[..., 2] * nx, tg[..., 3] * ny, tg[..., 4] * nx, tg[..., 5] * ny          # (B,G)
        r = torch.stack((gw, gh), -1)[:, None] / anchors[None, :, None, :]                              # (B,na,G,2)
        jm = (torch.max(r, 1. / r).max(-1)[0] < self.hyp_anchor_t) & valid[:, None, :]                   # (B,na,G)
        gxy = torch.stack((gx, gy), -1)                                                                  # (B,G,2)
        gxi = torch.stack((nx - gx, ny - gy), -1)
        jk = (gxy % 1. < 0.5) & (gxy > 1.)
        lm = (gxi % 1. < 0.5) & (gxi > 1.)
        rule = torch.stack((torch.ones_like(jk[..., 0]), jk[..., 0], jk[..., 1], lm[..., 0], lm[..., 1]), 1)  # (B,5,G)
        sel = rule[:, :, None, :] & jm[:, None]                                                          # (B,5,na,G)
        gij = (gxy[:, None] - self.off[None, :, None, :]).long()                                         # (B,5,G,2)
        gi = gij[..., 0].clamp(0, nx - 1)[:, :, None, :].expand(B, 5, na, G)
        gj = gij[..., 1].clamp(0, ny - 1)[:, :, None, :].expand(B, 5, na, G)
        a = torch.arange(na, device=tg.device)[None, None, :, None].expand(B, 5, na, G)
        g = torch.arange(G, device=tg.device)[None, None, None, :].expand(B, 5, na, G)
        flat = lambda t: t.reshape(B, -1)  # noqa: E731
        return flat(sel), flat(a), flat(gj), flat(gi), flat(g)

    def forward(self, p, targets, img_size):
        dev = targets.device
        B = p[0].shape[0]
        nc, na = self.num_classes, self.num_anchors
        tg, valid = flat_to_padded(targets, B, self.max_per_image)
        G = tg.shape[1]
        tcls = tg[..., 1].long().clamp(0, nc - 1)
        txy = tg[..., 2:6] * float(img_size)
        txyxy = torch.cat((txy[..., :2] - txy[..., 2:] / 2, txy[..., :2] + txy[..., 2:] / 2), -1)       # (B,G,4)
        lv = []
        with torch.no_grad():
            boxes, objs, clss, sels = [], [], [], []
            for i, pi in enumerate(p):
                _, _, ny, nx, no = pi.shape
                sel, a, gj, gi, g = self._candidates(tg, valid, i, ny, nx)
                cell = (a * ny + gj) * nx + gi
                ps = torch.gather(pi.detach().reshape(B, na * ny * nx, no), 1, cell[..., None].expand(-1, -1, no))
                anch = self.anchors[i][a]
                grid = torch.stack((gi, gj), -1).float()
                pxy = (ps[..., :2].sigmoid() * 2. - 0.5 + grid) * self.stride[i]
                pwh = (ps[..., 2:4].sigmoid() * 2) ** 2 * anch * self.stride[i]
                boxes.append(torch.cat((pxy - pwh / 2, pxy + pwh / 2), -1))
                objs.append(ps[..., 4])
                clss.append(ps[..., 5:])
                sels.append(sel)
                lv.append((sel, a, gj, gi, g, cell))
            box = torch.cat(boxes, 1)                                                                    # (B,Nc,4)
            obj, cls, csel = torch.cat(objs, 1), torch.cat(clss, 1), torch.cat(sels, 1)
            Nc = box.shape[1]
            usable = csel[:, None, :] & valid[..., None]                                                 # (B,G,Nc)
            a1 = ((txyxy[..., 2] - txyxy[..., 0]) * (txyxy[..., 3] - txyxy[..., 1]))[..., None]
            a2 = ((box[..., 2] - box[..., 0]) * (box[..., 3] - box[..., 1]))[:, None, :]
            inter = (torch.min(txyxy[:, :, None, 2:], box[:, None, :, 2:]) - torch.max(txyxy[:, :, None, :2], box[:, None, :, :2])).clamp(0).prod(-1)
            iou = inter / (a1 + a2 - inter)
            iou = torch.where(usable, iou, torch.zeros((), device=dev))
            kk = min(20, Nc)
            dyn_k = torch.clamp(torch.topk(iou, kk, dim=2)[0].sum(2).int(), min=1)
            y = (cls.float().sigmoid() * obj.float().sigmoid()[..., None]).sqrt()
            z = torch.log(y / (1 - y))
            base = F.binary_cross_entropy_with_logits(z, torch.zeros_like(z), reduction="none").sum(-1)  # (B,Nc)
            corr = -z.transpose(1, 2)                                                                    # bce(z,1) - bce(z,0) = -z
            cls_cost = base[:, None, :] + torch.gather(corr, 1, tcls[..., None].expand(B, G, Nc))
            cost = cls_cost + 3.0 * (-torch.log(iou + 1e-8))
            cost = torch.where(usable, cost, torch.full((), float("inf"), device=dev))
            cvals, cidx = torch.topk(cost, kk, dim=2, largest=False)
            pick = (torch.arange(kk, device=dev)[None, None, :] < dyn_k[..., None]) & torch.isfinite(cvals)
            matching = torch.zeros(B, G, Nc, device=dev).scatter_(2, cidx, pick.float())
            multi = matching.sum(1) > 1
            onehot = F.one_hot(torch.argmin(cost, dim=1), G).permute(0, 2, 1).to(matching.dtype)
            matching = torch.where(multi[:, None, :], onehot, matching)
            fg = matching.sum(1) > 0                                                                     # (B,Nc)
            mgt = matching.argmax(1)
        lcls = torch.zeros(1, device=dev)
        lbox = torch.zeros(1, device=dev)
        lobj = torch.zeros(1, device=dev)
        o = 0
        for i, pi in enumerate(p):
            _, _, ny, nx, no = pi.shape
            sel, a, gj, gi, g, cell = lv[i]
            Ncl = sel.shape[1]
            f = fg[:, o:o + Ncl]
            m = mgt[:, o:o + Ncl]
            o += Ncl
            ff = f.float()
            n = ff.sum()
            denom = n.clamp(min=1.0)
            ps = torch.gather(pi.reshape(B, na * ny * nx, no), 1, cell[..., None].expand(-1, -1, no))
            pxy = ps[..., :2].sigmoid() * 2. - 0.5
            pwh = (ps[..., 2:4].sigmoid() * 2) ** 2 * self.anchors[i][a]
            t = torch.gather(tg, 1, m[..., None].expand(-1, -1, 6))
            tbox = torch.stack((t[..., 2] * nx - gi, t[..., 3] * ny - gj, t[..., 4] * nx, t[..., 5] * ny), -1)
            iou = bbox_ciou_xywh(torch.cat((pxy, pwh), -1), tbox)
            lbox = lbox + (torch.where(f, 1.0 - iou, torch.zeros((), device=dev)).sum() / denom) * (n > 0)
            # objectness target: last fg candidate of a cell wins (image-major, candidate order) -> largest candidate index
            score = iou.detach().clamp(0).to(pi.dtype)
            ncell = B * na * ny * nx
            flatcell = torch.arange(B, device=dev)[:, None] * (na * ny * nx) + cell
            ordinal = torch.arange(B * Ncl, device=dev).view(B, Ncl)
            flatcell = torch.where(f, flatcell, ncell + ordinal)
            winner = torch.full((ncell + B * Ncl,), -1, dtype=torch.long, device=dev)
            winner = winner.scatter_reduce(0, flatcell.reshape(-1), ordinal.reshape(-1), reduce="amax", include_self=True)[:ncell]
            tobj = torch.where(winner >= 0, score.reshape(-1)[winner.clamp(min=0)], torch.zeros((), device=dev, dtype=pi.dtype))
            if nc > 1:
                tc = F.one_hot(t[..., 1].long().clamp(0, nc - 1), nc).to(ps.dtype)
                bce = F.binary_cross_entropy_with_logits(ps[..., 5:], tc, reduction="none")
                lcls = lcls + (torch.where(f[..., None], bce, torch.zeros((), device=dev)).sum() / (denom * nc)) * (n > 0)
            lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4].reshape(-1), tobj) * self.balance[i]
        lbox, lobj, lcls = lbox * self.hyp_box, lobj * self.hyp_obj, lcls * self.hyp_cls
        loss = lbox + lobj + lcls
        return loss * B, torch.cat((lbox, lobj, lcls, loss)).detach()
