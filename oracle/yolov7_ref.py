"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — CPU fp32 restatement of the reference's YOLOv7 path.

Pinned by tests/test_oracle_golden.py against fixtures captured from the reference's own classes (tools/gen_golden_more.py):
EELAN, DownA, DownB, SPPCSPC, UpSampling, FeatureFusion, RepConv, YOLOv7Neck, YOLOv7Head, YOLOv7Detect.

  blocks  : src/models/modules/yolov7_modules.py:20-33 (Conv), :36-61 (DownA/DownB), :64-82 (EELAN), :85-95 (UpSampling),
            :98-120 (FeatureFusion: conv4 applied three times, conv5/conv6 never used), :122-140 (SPPCSPC), :168-213 (RepConv)
  neck    : src/models/necks/yolov7_neck.py:13-55
  head    : src/models/heads/yolov7_head.py:12-40
  detect  : src/models/detects/yolov7_detect.py:71-122
  model   : src/models/yolov7.py:150-256
The reference's YOLOv7 backbone is a stub whose stages are empty (backbones/det/yolov7_csp_vovnet.py:46-56) and the yml names
a class that does not exist ('YOLOv7Backbone', conf/coco_yolov7.yml:66): PARITY UNPINNED for the backbone topology — it is
restated from the public YOLOv7-l layout (stem 32-64-64, [Conv s2 | DownA] + E-ELAN x4 -> 512/1024/1024 at /8,/16,/32, which
is what the neck's in_channels=[512,1024,1024] (coco_yolov7.yml:67) requires) out of the reference's own blocks.
The loss used for config 5 is the YOLOv5-style loss with YOLOv7's gains/anchors; the OTA loss
(losses/yolov7_loss.py:217-365) is a "next" row.
"""
import math

import torch
import torch.nn as nn

from .torch_ref import YOLOv5Loss, targets_to_gts

ANCHORS = [[[1.50000, 2.00000], [2.37500, 4.50000], [5.00000, 3.50000]],
           [[2.25000, 4.68750], [4.75000, 3.43750], [4.50000, 9.12500]],
           [[4.43750, 3.43750], [6.00000, 7.59375], [14.34375, 12.53125]]]


class Conv(nn.Module):
    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2 if p is None else p, groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else nn.Identity()

    def forward(self, x):
        return self.act(self.bn(self.conv(x)))


class DownA(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        self.branch1 = nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=2), Conv(c1, c2, 1, 1))
        self.branch2 = nn.Sequential(Conv(c1, c2, 1, 1), Conv(c2, c2, 3, 2))

    def forward(self, x):
        return torch.cat([self.branch1(x), self.branch2(x)], dim=1)


class DownB(DownA):
    def forward(self, x, y):
        return torch.cat([self.branch1(x), self.branch2(x), y], dim=1)


class EELAN(nn.Module):
    def __init__(self, c1, c2, c3):
        super().__init__()
        self.conv1 = Conv(c1, c2, 1, 1)
        self.conv2 = Conv(c1, c2, 1, 1)
        self.conv3 = nn.Sequential(Conv(c2, c2, 3, 1), Conv(c2, c2, 3, 1))
        self.conv4 = nn.Sequential(Conv(c2, c2, 3, 1), Conv(c2, c2, 3, 1))
        self.conv5 = Conv(c2 * 4, c3, 1, 1)

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = self.conv2(x)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        return self.conv5(torch.cat([x1, x2, x3, x4], dim=1))


class UpSampling(nn.Module):
    def __init__(self, c1, c2, c3):
        super().__init__()
        self.conv1 = Conv(c1, c3, 1, 1)
        self.upsampling = nn.UpsamplingNearest2d(scale_factor=2)
        self.conv2 = Conv(c2, c3, 1, 1)

    def forward(self, x, y):
        return torch.cat([self.upsampling(self.conv1(x)), self.conv2(y)], dim=1)


class FeatureFusion(nn.Module):
    def __init__(self, c1, c2):
        super().__init__()
        mid = c2 // 2
        self.conv1 = Conv(c1, c2, 1, 1)
        self.conv2 = Conv(c1, c2, 1, 1)
        self.conv3 = Conv(c2, mid, 3, 1)
        self.conv4 = Conv(mid, mid, 3, 1)
        self.conv5 = Conv(mid, mid, 3, 1)  # present in the state_dict, never called (yolov7_modules.py:113-120)
        self.conv6 = Conv(mid, mid, 3, 1)
        self.conv7 = Conv(c2 * 4, c2, 1, 1)

    def forward(self, x):
        x1 = self.conv1(x)
        x2 = self.conv2(x)
        x3 = self.conv3(x2)
        x4 = self.conv4(x3)
        x5 = self.conv4(x4)
        x6 = self.conv4(x5)
        return self.conv7(torch.cat([x1, x2, x3, x4, x5, x6], dim=1))


class SPPCSPC(nn.Module):
    def __init__(self, c1, c2, e=0.5, k=(5, 9, 13)):
        super().__init__()
        c_ = int(2 * c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(c_, c_, 3, 1)
        self.cv4 = Conv(c_, c_, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])
        self.cv5 = Conv(4 * c_, c_, 1, 1)
        self.cv6 = Conv(c_, c_, 3, 1)
        self.cv7 = Conv(2 * c_, c2, 1, 1)

    def forward(self, x):
        x1 = self.cv4(self.cv3(self.cv1(x)))
        y1 = self.cv6(self.cv5(torch.cat([x1] + [m(x1) for m in self.m], 1)))
        return self.cv7(torch.cat((y1, self.cv2(x)), dim=1))


class RepConv(nn.Module):
    """Training-time three-branch block: act(BN(conv3x3) + BN(conv1x1) + BN(identity if c1==c2 and s==1))."""

    def __init__(self, c1, c2, k=3, s=1):
        super().__init__()
        self.act = nn.SiLU()
        self.rbr_identity = nn.BatchNorm2d(c1) if c2 == c1 and s == 1 else None
        self.rbr_dense = nn.Sequential(nn.Conv2d(c1, c2, k, s, 1, bias=False), nn.BatchNorm2d(c2))
        self.rbr_1x1 = nn.Sequential(nn.Conv2d(c1, c2, 1, s, 0, bias=False), nn.BatchNorm2d(c2))

    def forward(self, x):
        id_out = 0 if self.rbr_identity is None else self.rbr_identity(x)
        return self.act(self.rbr_dense(x) + self.rbr_1x1(x) + id_out)


def _bn_fix(module):
    """yolov7_neck.py:36-45 / yolov7.py:187-196: BN eps 1e-3, momentum 0.03 (conv init left at torch default)."""
    for m in module.modules():
        if type(m) is nn.BatchNorm2d:
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
        x5 = self.spp(x5)
        x4_up = self.featurefusion1_1(self.up1_1(x5, x4))
        x3_up = self.featurefusion1_2(self.up1_2(x4_up, x3))
        x4_down = self.featurefusion2_1(self.down2_1(x3_up, x4_up))
        x5_down = self.featurefusion2_2(self.down2_2(x4_down, x5))
        return [x3_up, x4_down, x5_down]


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
    def __init__(self, num_classes=80, in_channels=(256, 512, 1024), stride=(8., 16., 32.), anchors=ANCHORS, depth_mul=1.0, width_mul=1.0):
        super().__init__()
        ic = [int(x * width_mul) for x in in_channels]
        self.num_classes, self.num_outputs = num_classes, num_classes + 5
        self.num_layers, self.num_anchors = len(anchors), len(anchors[0])
        self.stride = list(stride)
        a = torch.tensor(anchors).float().view(self.num_layers, -1, 2)
        self.register_buffer("anchors", a)
        self.register_buffer("anchor_grid", (a.clone() * torch.tensor(stride).view(-1, 1, 1)).view(self.num_layers, 1, -1, 1, 1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.num_outputs * self.num_anchors, 1) for x in ic)
        for mi, s in zip(self.m, self.stride):  # yolov7_detect.py:91-98
            b = mi.bias.view(self.num_anchors, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (self.num_classes - 0.99))
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def forward(self, x):
        z, out = [], []
        for i in range(self.num_layers):
            y = self.m[i](x[i])
            bs, _, ny, nx = y.shape
            y = y.view(bs, self.num_anchors, self.num_outputs, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            out.append(y)
            if not self.training:
                yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
                grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
                s = y.sigmoid()
                xy = (s[..., 0:2] * 2. - 0.5 + grid) * self.stride[i]
                wh = (s[..., 2:4] * 2) ** 2 * self.anchor_grid[i]
                z.append(torch.cat((xy, wh, s[..., 4:]), -1).view(bs, -1, self.num_outputs))
        return (None, out) if self.training else (torch.cat(z, 1), out)


class YOLOv7Backbone(nn.Module):
    """Public YOLOv7-l backbone assembled from the reference's blocks (see module docstring: parity unpinned)."""

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
        p3 = self.stage2(x)
        p4 = self.stage3(p3)
        p5 = self.stage4(p4)
        return [p3, p4, p5]


class YOLOv7(nn.Module):
    """src/models/yolov7.py:150-256 wiring: detect(head(neck(backbone(x)))) -> loss."""

    def __init__(self, num_classes=80, width_mul=1.0):
        super().__init__()
        self.num_classes = num_classes
        self.backbone = YOLOv7Backbone(width_mul)
        self.neck = YOLOv7Neck(width_mul=width_mul)
        self.head = YOLOv7Head(width_mul=width_mul)
        self.detect = YOLOv7Detect(num_classes, width_mul=width_mul)
        self.loss = YOLOv5Loss(num_classes, anchors=ANCHORS, hyp_box=0.05, hyp_obj=0.7, hyp_cls=0.3)
        _bn_fix(self)

    def forward(self, imgs, targets=None, mode="train"):
        _, train_out = self.detect(self.head(self.neck(self.backbone(imgs))))
        gts = targets if torch.is_tensor(targets) else targets_to_gts(targets)
        losses = {}
        losses["loss"], st = self.loss(train_out, gts)
        losses["box_loss"], losses["obj_loss"], losses["cls_loss"] = st[0], st[1], st[2]
        return losses


# ------------------------------------------------------------------------------------------------------
# OTA loss (src/losses/yolov7_loss.py:129-420) — pinned by tests/golden/v7_ota_loss_*.npz (tools/gen_golden_more.py)
# ------------------------------------------------------------------------------------------------------
import torch.nn.functional as F  # noqa: E402

from .torch_ref import bbox_iou  # noqa: E402


def _xywh2xyxy(x):
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def _box_iou_xyxy(b1, b2):
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    inter = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(0).prod(2)
    return inter / (a1[:, None] + a2 - inter)


class YOLOv7OTALoss:
    """YOLOv7Loss (:129-215) with build_targets = find_3_positive (:365-420) + per-image SimOTA-style matching (:217-363)."""

    def __init__(self, num_classes, stride=(8., 16., 32.), anchors=ANCHORS):
        self.num_classes = num_classes
        self.num_layers, self.num_anchors = len(anchors), len(anchors[0])
        self.stride = [float(s) for s in stride]
        self.anchors = torch.tensor(anchors).float()
        self.hyp_anchor_t, self.hyp_box, self.hyp_obj, self.hyp_cls = 4.0, 0.05, 0.7, 0.3
        self.balance = [4.0, 1.0, 0.4]
        self.gr = 1.0

    def find_3_positive(self, p, targets):
        na, nt = self.num_anchors, targets.shape[0]
        indices, anch = [], []
        gain = torch.ones(7)
        ai = torch.arange(na).float().view(na, 1).repeat(1, nt)
        targets = torch.cat((targets.repeat(na, 1, 1), ai[:, :, None]), 2)
        g = 0.5
        off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]).float() * g
        for i in range(self.num_layers):
            anchors = self.anchors[i]
            gain[2:6] = torch.tensor(p[i].shape)[[3, 2, 3, 2]]
            t = targets * gain
            if nt:
                r = t[:, :, 4:6] / anchors[:, None]
                j = torch.max(r, 1. / r).max(2)[0] < self.hyp_anchor_t
                t = t[j]
                gxy = t[:, 2:4]
                gxi = gain[[2, 3]] - gxy
                j, k = ((gxy % 1. < g) & (gxy > 1.)).T
                l, m = ((gxi % 1. < g) & (gxi > 1.)).T
                j = torch.stack((torch.ones_like(j), j, k, l, m))
                t = t.repeat((5, 1, 1))[j]
                offsets = (torch.zeros_like(gxy)[None] + off[:, None])[j]
            else:
                t = targets[0]
                offsets = 0
            b, c = t[:, :2].long().T
            gxy = t[:, 2:4]
            gij = (gxy - offsets).long()
            gi, gj = gij.T
            a = t[:, 6].long()
            indices.append((b, a, gj.clamp_(0, int(gain[3]) - 1), gi.clamp_(0, int(gain[2]) - 1)))
            anch.append(anchors[a])
        return indices, anch

    def build_targets(self, p, targets, imgs):
        indices, anch = self.find_3_positive(p, targets)
        nl = len(p)
        out = [[[] for _ in range(nl)] for _ in range(6)]
        for bi in range(p[0].shape[0]):
            this_target = targets[targets[:, 0] == bi]
            if this_target.shape[0] == 0:
                continue
            txyxy = _xywh2xyxy(this_target[:, 2:6] * imgs[bi].shape[1])
            pxyxys, p_cls, p_obj, fwl, ab, aa, agj, agi, aan = [], [], [], [], [], [], [], [], []
            for i, pi in enumerate(p):
                b, a, gj, gi = indices[i]
                idx = b == bi
                b, a, gj, gi = b[idx], a[idx], gj[idx], gi[idx]
                ab.append(b); aa.append(a); agj.append(gj); agi.append(gi); aan.append(anch[i][idx])
                fwl.append(torch.ones(len(b)) * i)
                fg = pi[b, a, gj, gi]
                p_obj.append(fg[:, 4:5])
                p_cls.append(fg[:, 5:])
                grid = torch.stack([gi, gj], dim=1)
                pxy = (fg[:, :2].sigmoid() * 2. - 0.5 + grid) * self.stride[i]
                pwh = (fg[:, 2:4].sigmoid() * 2) ** 2 * anch[i][idx] * self.stride[i]
                pxyxys.append(_xywh2xyxy(torch.cat([pxy, pwh], dim=-1)))
            pxyxys = torch.cat(pxyxys, 0)
            if pxyxys.shape[0] == 0:
                continue
            p_obj, p_cls, fwl = torch.cat(p_obj, 0), torch.cat(p_cls, 0), torch.cat(fwl, 0)
            ab, aa, agj, agi, aan = torch.cat(ab), torch.cat(aa), torch.cat(agj), torch.cat(agi), torch.cat(aan)
            iou = _box_iou_xyxy(txyxy, pxyxys)
            iou_loss = -torch.log(iou + 1e-8)
            top_k, _ = torch.topk(iou, min(20, iou.shape[1]), dim=1)
            dynamic_ks = torch.clamp(top_k.sum(1).int(), min=1)
            num_gt = this_target.shape[0]
            onehot = F.one_hot(this_target[:, 1].to(torch.int64), self.num_classes).float().unsqueeze(1).repeat(1, pxyxys.shape[0], 1)
            y = (p_cls.float().unsqueeze(0).repeat(num_gt, 1, 1).sigmoid() * p_obj.unsqueeze(0).repeat(num_gt, 1, 1).sigmoid()).sqrt()
            cls_loss = F.binary_cross_entropy_with_logits(torch.log(y / (1 - y)), onehot, reduction="none").sum(-1)
            cost = cls_loss + 3.0 * iou_loss
            matching = torch.zeros_like(cost)
            for g in range(num_gt):
                _, pos = torch.topk(cost[g], k=int(dynamic_ks[g]), largest=False)
                matching[g][pos] = 1.0
            multi = matching.sum(0) > 1
            if multi.sum() > 0:
                _, amin = torch.min(cost[:, multi], dim=0)
                matching[:, multi] *= 0.0
                matching[amin, multi] = 1.0
            fgm = matching.sum(0) > 0.0
            mgt = matching[:, fgm].argmax(0)
            fwl, ab, aa, agj, agi, aan = fwl[fgm], ab[fgm], aa[fgm], agj[fgm], agi[fgm], aan[fgm]
            tt = this_target[mgt]
            for i in range(nl):
                li = fwl == i
                for k, v in enumerate((ab[li], aa[li], agj[li], agi[li], tt[li], aan[li])):
                    out[k][i].append(v)
        res = []
        for k in range(6):
            res.append([torch.cat(out[k][i], 0) if out[k][i] else (torch.zeros(0, 6) if k == 4 else (torch.zeros(0, 2) if k == 5 else torch.zeros(0, dtype=torch.long)))
                        for i in range(nl)])
        return res

    def __call__(self, p, targets, imgs, return_assign=False):
        lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
        bs, as_, gjs, gis, tg, anchors = self.build_targets([q.detach() for q in p], targets, imgs)
        for i, pi in enumerate(p):
            b, a, gj, gi = bs[i], as_[i], gjs[i], gis[i]
            tobj = torch.zeros_like(pi[..., 0])
            n = b.shape[0]
            if n:
                ps = pi[b, a, gj, gi]
                grid = torch.stack([gi, gj], dim=1)
                pxy = ps[:, :2].sigmoid() * 2. - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
                pbox = torch.cat((pxy, pwh), 1)
                gain = torch.tensor(pi.shape)[[3, 2, 3, 2]]
                tbox = tg[i][:, 2:6] * gain
                tbox = torch.cat([tbox[:, :2] - grid, tbox[:, 2:]], 1)
                iou = bbox_iou(pbox.T, tbox, x1y1x2y2=False, CIoU=True)
                lbox = lbox + (1.0 - iou).mean()
                tobj[b, a, gj, gi] = (1.0 - self.gr) + self.gr * iou.detach().clamp(0).type(tobj.dtype)
                if self.num_classes > 1:
                    t = torch.zeros_like(ps[:, 5:])
                    t[range(n), tg[i][:, 1].long()] = 1.0
                    lcls = lcls + F.binary_cross_entropy_with_logits(ps[:, 5:], t)
            lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 4], tobj) * self.balance[i]
        lbox, lobj, lcls = lbox * self.hyp_box, lobj * self.hyp_obj, lcls * self.hyp_cls
        bsz = p[0].shape[0]
        loss = lbox + lobj + lcls
        out = (loss * bsz, torch.cat((lbox, lobj, lcls, loss)).detach())
        return (out, (bs, as_, gjs, gis, tg)) if return_assign else out
