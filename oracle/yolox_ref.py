"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — CPU fp32 restatement of the reference's YOLOX path.

Pinned by tests/test_oracle_golden.py against fixtures captured from the reference's own classes
(tools/gen_golden.py): YOLOXCSPDarknet, YOLOXHead, YOLOXLoss (assignments + loss values + gradients).
YOLOXNeck cannot be constructed in the reference at HEAD (`super().__init__(**kwargs)` never
receives subtype/cfg), so it is restated from the file as specification and pinned only through shapes.

  backbone : src/models/backbones/det/yolox_csp_darknet.py:17-100 (Focus stem :37-44, SPP in stage 4 before CSP :66-73)
  neck     : src/models/necks/det/yolox_neck.py:16-105
  head     : src/models/heads/det/yolox_head.py:16-98
  loss     : src/losses/det/yolox_loss.py:14-435
  model    : src/models/yolox.py:18-68 (post-process), :112-157 (target format + forward)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .torch_ref import ConvModule, CSPLayer, DepthwiseSeparableConvModule, Focus, SPPF, nms

SCALES = {"n": (0.33, 0.25), "nano": (0.33, 0.25), "t": (0.33, 0.375), "tiny": (0.33, 0.375), "s": (0.33, 0.5),
          "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}
BN = dict(type="BN", momentum=0.03, eps=0.001)


def _init(module):
    """yolox_csp_darknet.py:92-100 / yolox_neck.py:69-78 / yolox_head.py:65-72."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03


class YOLOXCSPDarknet(nn.Module):
    def __init__(self, subtype="cspdark_s", in_channels=3, out_channels=(64, 128, 256, 512, 1024), num_blocks=(3, 9, 9, 3),
                 spp_ksizes=(5, 9, 13), norm_cfg=BN, act_cfg=dict(type="SiLU"), out_stages=(2, 3, 4)):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        ch = [int(x * width_mul) for x in out_channels]            # base_yolo_backbone.py:42
        nb = [max(round(x * depth_mul), 1) for x in num_blocks]    # base_yolo_backbone.py:43
        self.out_stages = list(out_stages)
        self.stem = Focus(in_channels, ch[0], kernel_sizes=3, norm_cfg=norm_cfg, act_cfg=act_cfg)
        for idx in range(4):
            stage = [ConvModule(ch[idx], ch[idx + 1], 3, stride=2, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg)]
            if idx == 3:
                stage.append(SPPF(ch[idx + 1], ch[idx + 1], kernel_sizes=spp_ksizes, norm_cfg=norm_cfg, act_cfg=act_cfg))
            stage.append(CSPLayer(ch[idx + 1], ch[idx + 1], n=nb[idx], shortcut=(idx != 3), norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.add_module("stage%d" % (idx + 1), nn.Sequential(*stage))
        _init(self)

    def forward(self, x):
        x = self.stem(x)
        out = []
        for i in range(1, 5):
            x = getattr(self, "stage%d" % i)(x)
            if i in self.out_stages:
                out.append(x)
        return out


class CSPDarknet(nn.Module):
    """src/models/backbones/det/csp_darknet.py:25-103 — the generic CSPDarknet (YOLOv4 / YOLOX / AIRDet family): Focus stem, four
    stride-2 stages of conv (or depthwise-separable conv) + CSPLayer, SPPF before the last CSPLayer, `depthwise` threaded into
    the bottlenecks; widths int(c * width_mul), depths max(round(n * depth_mul), 1) (:50-52); kaiming-uniform init (:96-103)."""
    cfg = {"n": [0.33, 0.25], "t": [0.33, 0.375], "s": [0.33, 0.5], "m": [0.67, 0.75], "l": [1.0, 1.0], "x": [1.33, 1.25]}

    def __init__(self, subtype="cspdark_s", out_channels=(64, 128, 256, 512, 1024), layers=(3, 9, 9, 3), spp_ksizes=(5, 9, 13),
                 depthwise=False, norm_cfg=BN, act_cfg=dict(type="Swish"), out_stages=(2, 3, 4)):
        super().__init__()
        depth_mul, width_mul = self.cfg[subtype.split("_")[1]]
        ch = [int(x * width_mul) for x in out_channels]
        nb = [max(round(x * depth_mul), 1) for x in layers]
        self.out_stages = list(out_stages)
        conv = DepthwiseSeparableConvModule if depthwise else ConvModule
        self.stem = Focus(3, ch[0], kernel_sizes=3, norm_cfg=norm_cfg, act_cfg=act_cfg)
        for idx in range(4):
            stage = [conv(ch[idx], ch[idx + 1], 3, 2, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg)]
            if idx == 3:
                stage.append(SPPF(ch[idx + 1], ch[idx + 1], kernel_sizes=spp_ksizes, norm_cfg=norm_cfg, act_cfg=act_cfg))
            stage.append(CSPLayer(ch[idx + 1], ch[idx + 1], n=nb[idx], shortcut=(idx != 3), depthwise=depthwise, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.add_module("stage%d" % (idx + 1), nn.Sequential(*stage))
        self.out_channels = ch[self.out_stages[0]:self.out_stages[-1] + 1]
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.stem(x)
        out = []
        for i in range(1, 5):
            x = getattr(self, "stage%d" % i)(x)
            if i in self.out_stages:
                out.append(x)
        return out if len(self.out_stages) > 1 else out[0]


class YOLOXNeck(nn.Module):
    def __init__(self, subtype="yolox_s", in_channels=(256, 512, 1024), out_channels=256, num_blocks=(3, 3, 3, 3),
                 norm_cfg=dict(type="BN"), act_cfg=dict(type="Swish")):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        c = [max(round(x * width_mul), 1) for x in in_channels]
        oc = max(round(out_channels * width_mul), 1)
        nb = [max(round(x * depth_mul), 1) for x in num_blocks]
        self.in_channels, self.out_channels = c, oc
        self.upsample = nn.Upsample(scale_factor=2, mode="nearest")
        self.reduce_layers, self.top_down_blocks = nn.ModuleList(), nn.ModuleList()
        for idx in range(len(c) - 1, 0, -1):
            self.reduce_layers.append(ConvModule(c[idx], c[idx - 1], 1, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.top_down_blocks.append(CSPLayer(c[idx - 1] * 2, c[idx - 1], n=nb[idx], shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg))
        self.downsamples, self.bottom_up_blocks = nn.ModuleList(), nn.ModuleList()
        for idx in range(len(c) - 1):
            self.downsamples.append(ConvModule(c[idx], c[idx], 3, stride=2, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.bottom_up_blocks.append(CSPLayer(c[idx] * 2, c[idx + 1], n=nb[idx], shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg))
        self.out_convs = nn.ModuleList(ConvModule(ci, oc, 1, norm_cfg=norm_cfg, act_cfg=act_cfg) for ci in c)
        _init(self)

    def forward(self, x):
        n = len(self.in_channels)
        inner = [x[-1]]
        for idx in range(n - 1, 0, -1):
            hi = self.reduce_layers[n - 1 - idx](inner[0])
            inner[0] = hi
            inner.insert(0, self.top_down_blocks[n - 1 - idx](torch.cat([self.upsample(hi), x[idx - 1]], 1)))
        outs = [inner[0]]
        for idx in range(n - 1):
            outs.append(self.bottom_up_blocks[idx](torch.cat([self.downsamples[idx](outs[-1]), inner[idx + 1]], 1)))
        return [conv(o) for conv, o in zip(self.out_convs, outs)]


class YOLOXHead(nn.Module):
    def __init__(self, subtype="yolox_s", num_classes=80, in_channels=256, channels=256, stacked_convs=2, strides=(8, 16, 32),
                 norm_cfg=dict(type="BN"), act_cfg=dict(type="Swish")):
        super().__init__()
        _, width_mul = SCALES[subtype.split("_")[1]]
        cin = max(round(in_channels * width_mul), 1)
        ch = max(round(channels * width_mul), 1)
        self.num_classes, self.strides = num_classes, list(strides)
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        self.cls_preds, self.reg_preds, self.obj_preds = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for _ in self.strides:
            self.cls_convs.append(nn.Sequential(*[ConvModule(cin if i == 0 else ch, ch, 3, 1, 1, norm_cfg=norm_cfg, act_cfg=act_cfg)
                                                  for i in range(stacked_convs)]))
            self.reg_convs.append(nn.Sequential(*[ConvModule(cin if i == 0 else ch, ch, 3, 1, 1, norm_cfg=norm_cfg, act_cfg=act_cfg)
                                                  for i in range(stacked_convs)]))
            self.cls_preds.append(nn.Conv2d(ch, num_classes, 1, 1, 0))
            self.reg_preds.append(nn.Conv2d(ch, 4, 1, 1, 0))
            self.obj_preds.append(nn.Conv2d(ch, 1, 1, 1, 0))
        _init(self)
        bias_init = float(-math.log((1 - 1e-2) / 1e-2))  # yolox_head.py:74-80
        for conv in list(self.cls_preds) + list(self.obj_preds):
            conv.bias.data.fill_(bias_init)

    def forward(self, x):
        outs = []
        for k, xx in enumerate(x):
            cls_feat = self.cls_convs[k](xx)
            reg_feat = self.reg_convs[k](xx)
            outs.append(torch.cat([self.reg_preds[k](reg_feat), self.obj_preds[k](reg_feat), self.cls_preds[k](cls_feat)], 1))
        return outs


# ------------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------------
def bboxes_iou_cxcywh(a, b):
    """yolox_loss.py:14-31 with xyxy=False: (Ga,4) x (Gb,4) -> (Ga,Gb)."""
    tl = torch.max(a[:, None, :2] - a[:, None, 2:] / 2, b[:, :2] - b[:, 2:] / 2)
    br = torch.min(a[:, None, :2] + a[:, None, 2:] / 2, b[:, :2] + b[:, 2:] / 2)
    area_a = torch.prod(a[:, 2:], 1)
    area_b = torch.prod(b[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=2)
    area_i = torch.prod(br - tl, 2) * en
    return area_i / (area_a[:, None] + area_b - area_i)


def iou_loss(pred, target):
    """yolox_loss.py:34-69, loss_type 'iou', reduction none."""
    tl = torch.max(pred[:, :2] - pred[:, 2:] / 2, target[:, :2] - target[:, 2:] / 2)
    br = torch.min(pred[:, :2] + pred[:, 2:] / 2, target[:, :2] + target[:, 2:] / 2)
    area_p = torch.prod(pred[:, 2:], 1)
    area_g = torch.prod(target[:, 2:], 1)
    en = (tl < br).type(tl.type()).prod(dim=1)
    area_i = torch.prod(br - tl, 1) * en
    iou = area_i / (area_p + area_g - area_i + 1e-16)
    return 1 - iou ** 2


class YOLOXLoss:
    """SimOTA assignment + 5*IoU + obj + cls (yolox_loss.py:73-435, use_l1=False, reid_dim=0)."""

    def __init__(self, num_classes, strides=(8, 16, 32)):
        self.num_classes = num_classes
        self.strides = list(strides)

    def decode(self, preds):
        """get_output_and_grid :138-153 for every level -> outputs (B,A,5+nc), x_shift, y_shift, stride (A,)."""
        outs, xs, ys, ss = [], [], [], []
        for stride, p in zip(self.strides, preds):
            b, c, h, w = p.shape
            yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
            grid = torch.stack((xv, yv), 2).view(1, h * w, 2).type(p.dtype)
            q = p.permute(0, 2, 3, 1).reshape(b, h * w, c)
            q = torch.cat([(q[..., :2] + grid) * stride, torch.exp(q[..., 2:4]) * stride, q[..., 4:]], -1)
            outs.append(q)
            xs.append(grid[0, :, 0])
            ys.append(grid[0, :, 1])
            ss.append(torch.full((h * w,), float(stride), dtype=p.dtype))
        return torch.cat(outs, 1), torch.cat(xs), torch.cat(ys), torch.cat(ss)

    @staticmethod
    def in_boxes_info(gt, strides, xs, ys):
        """get_in_boxes_info :350-403. Returns fg candidates (A,) and in-box-and-centre (G, n_cand)."""
        xc = ((xs * strides) + 0.5 * strides)[None]
        yc = ((ys * strides) + 0.5 * strides)[None]
        l, r = (gt[:, 0] - 0.5 * gt[:, 2])[:, None], (gt[:, 0] + 0.5 * gt[:, 2])[:, None]
        t, b = (gt[:, 1] - 0.5 * gt[:, 3])[:, None], (gt[:, 1] + 0.5 * gt[:, 3])[:, None]
        in_boxes = torch.stack([xc - l, yc - t, r - xc, b - yc], 2).min(dim=-1).values > 0.0
        rad = 2.5 * strides[None]
        cl, cr = gt[:, 0:1] - rad, gt[:, 0:1] + rad
        ct, cb = gt[:, 1:2] - rad, gt[:, 1:2] + rad
        in_centers = torch.stack([xc - cl, yc - ct, cr - xc, cb - yc], 2).min(dim=-1).values > 0.0
        cand = (in_boxes.sum(0) > 0) | (in_centers.sum(0) > 0)
        return cand, (in_boxes[:, cand] & in_centers[:, cand])

    @staticmethod
    def dynamic_k_matching(cost, ious, gt_classes, fg_mask):
        """:405-435. Mutates fg_mask like the reference. Returns num_fg, matched classes, matched ious, matched gt index."""
        num_gt = cost.shape[0]
        matching = torch.zeros_like(cost)
        topk_ious, _ = torch.topk(ious, min(10, ious.size(1)), dim=1)
        dynamic_ks = torch.clamp(topk_ious.sum(1).int(), min=1)
        for g in range(num_gt):
            _, pos = torch.topk(cost[g], k=int(dynamic_ks[g]), largest=False)
            matching[g][pos] = 1.0
        multi = matching.sum(0) > 1
        if multi.sum() > 0:
            _, amin = torch.min(cost[:, multi], dim=0)
            matching[:, multi] *= 0.0
            matching[amin, multi] = 1.0
        fg_in = matching.sum(0) > 0.0
        num_fg = int(fg_in.sum())
        fg_mask[fg_mask.clone()] = fg_in
        matched = matching[:, fg_in].argmax(0)
        return num_fg, gt_classes[matched], (matching * ious).sum(0)[fg_in], matched

    @torch.no_grad()
    def assign(self, gt_boxes, gt_classes, boxes, cls_logits, obj_logits, strides, xs, ys):
        """get_assignments :291-348 for one image."""
        fg_mask, in_both = self.in_boxes_info(gt_boxes, strides, xs, ys)
        b_ = boxes[fg_mask]
        ious = bboxes_iou_cxcywh(gt_boxes, b_)
        onehot = F.one_hot(gt_classes.to(torch.int64), self.num_classes).float()[:, None, :].repeat(1, b_.shape[0], 1)
        iou_cost = -torch.log(ious + 1e-8)
        p = cls_logits[fg_mask].float().sigmoid()[None].repeat(len(gt_boxes), 1, 1) * obj_logits[fg_mask].float().sigmoid()[None].repeat(len(gt_boxes), 1, 1)
        cls_cost = F.binary_cross_entropy(p.sqrt(), onehot, reduction="none").sum(-1)
        cost = cls_cost + 3.0 * iou_cost + 100000.0 * (~in_both)
        num_fg, m_cls, m_iou, m_gt = self.dynamic_k_matching(cost, ious, gt_classes, fg_mask)
        return m_cls, fg_mask, m_iou, m_gt, num_fg, cost, ious

    def __call__(self, preds, targets, return_assign=False):
        """preds: list of (B, 5+nc, H, W) raw head maps; targets (B, max_gt, 5) = [cls, cx, cy, w, h] in pixels, zero rows pad."""
        outputs, xs, ys, ss = self.decode(preds)
        nc = self.num_classes
        boxes, obj, cls = outputs[:, :, :4], outputs[:, :, 4:5], outputs[:, :, 5:5 + nc]
        nlabel = (targets.sum(dim=2) > 0).sum(dim=1)
        A = outputs.shape[1]
        cls_t, reg_t, obj_t, fg_masks, assigns = [], [], [], [], []
        num_fg, num_gts = 0.0, 0.0
        for b in range(outputs.shape[0]):
            g = int(nlabel[b])
            num_gts += g
            if g == 0:
                cls_t.append(outputs.new_zeros((0, nc)))
                reg_t.append(outputs.new_zeros((0, 4)))
                obj_t.append(outputs.new_zeros((A, 1)))
                fg_masks.append(outputs.new_zeros(A).bool())
                assigns.append(None)
                continue
            gcls, gbox = targets[b, :g, 0], targets[b, :g, 1:5]
            m_cls, fg, m_iou, m_gt, nf, cost, ious = self.assign(gbox, gcls, boxes[b].detach(), cls[b].detach(), obj[b].detach(), ss, xs, ys)
            num_fg += nf
            cls_t.append(F.one_hot(m_cls.to(torch.int64), nc) * m_iou.unsqueeze(-1))
            obj_t.append(fg.unsqueeze(-1).to(outputs.dtype))
            reg_t.append(gbox[m_gt])
            fg_masks.append(fg)
            assigns.append((fg, m_gt, m_iou))
        cls_t, reg_t, obj_t, fg_masks = torch.cat(cls_t, 0), torch.cat(reg_t, 0), torch.cat(obj_t, 0), torch.cat(fg_masks, 0)
        num_fg = max(num_fg, 1)
        bce = nn.BCEWithLogitsLoss(reduction="none")
        loss_iou = iou_loss(boxes.reshape(-1, 4)[fg_masks], reg_t).sum() / num_fg
        loss_obj = bce(obj.reshape(-1, 1), obj_t).sum() / num_fg
        loss_cls = bce(cls.reshape(-1, nc)[fg_masks], cls_t).sum() / num_fg
        loss = 5.0 * loss_iou + loss_obj + loss_cls
        out = {"loss": loss, "conf_loss": loss_obj, "cls_loss": loss_cls, "iou_loss": 5.0 * loss_iou,
               "num_fg": torch.tensor(num_fg / max(num_gts, 1), dtype=outputs.dtype)}
        return (out, assigns) if return_assign else out


def targets_to_padded(targets, size=None):
    """src/models/yolox.py:112-139: list of {'labels','boxes'} -> (B, max_labels, 5) [cls, box...] zero padded."""
    mx = max(int(t["labels"].shape[0]) for t in targets)
    out = torch.zeros(len(targets), mx, 5)
    for i, t in enumerate(targets):
        n = t["labels"].shape[0]
        if n:
            out[i, :n] = torch.cat([t["labels"].float().unsqueeze(1), t["boxes"].float()], 1)
    return out


def yolox_post_process(outputs, strides, num_classes, conf_thre, nms_thre):
    """src/models/yolox.py:18-68 with torchvision.batched_nms restated as class-offset NMS (torchvision's own strategy)."""
    grids, ss = [], []
    for o, s in zip(outputs, strides):
        h, w = o.shape[-2:]
        yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
        grids.append(torch.stack((xv, yv), 2).view(1, -1, 2))
        ss.append(torch.full((1, h * w, 1), float(s)))
    out = torch.cat([x.flatten(start_dim=2) for x in outputs], dim=2).permute(0, 2, 1).clone()
    grids, ss = torch.cat(grids, 1).type(out.dtype), torch.cat(ss, 1).type(out.dtype)
    xy = (out[..., 0:2] + grids) * ss
    wh = torch.exp(out[..., 2:4]) * ss
    sc = torch.sigmoid(out[..., 4:5 + num_classes])
    res = []
    for i in range(out.shape[0]):
        box = torch.cat([xy[i] - wh[i] / 2, xy[i] + wh[i] / 2], 1)
        conf, pred = torch.max(sc[i, :, 1:], 1, keepdim=True)
        keep = (sc[i, :, 0] * conf.squeeze(1) >= conf_thre)
        det = torch.cat((box, sc[i, :, 0:1], conf, pred.float()), 1)[keep]
        if not det.size(0):
            res.append(None)
            continue
        offs = det[:, 6:7] * (det[:, :4].max() + 1)
        idx = nms(det[:, :4] + offs, det[:, 4] * det[:, 5], nms_thre)
        res.append(det[idx])
    return res


class YOLOX(nn.Module):
    """src/models/yolox.py:71-188 wiring: backbone -> neck -> head -> loss(out, gt)."""

    def __init__(self, num_classes=80, subtype="s"):
        super().__init__()
        self.num_classes = num_classes
        self.backbone = YOLOXCSPDarknet("cspdark_" + subtype)
        self.neck = YOLOXNeck("yolox_" + subtype)
        self.head = YOLOXHead("yolox_" + subtype, num_classes=num_classes, norm_cfg=BN)
        self.loss = YOLOXLoss(num_classes)
        self.stride = [8, 16, 32]
        self.conf_thr, self.nms_thr = 0.01, 0.65

    def forward(self, imgs, targets=None, mode="train"):
        out = self.head(self.neck(self.backbone(imgs)))
        losses = self.loss(out, targets_to_padded(targets) if isinstance(targets, (list, tuple)) else targets)
        if mode == "val":
            return losses, yolox_post_process([o.detach() for o in out], self.stride, self.num_classes, self.conf_thr, self.nms_thr)
        return losses


def synthetic_batch(batch, size=640, num_classes=80, seed=1029, max_boxes=20):
    """pixel-unit cxcywh targets, 1..max_boxes per image."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(batch, 3, size, size, generator=g)
    targets = []
    for _ in range(batch):
        n = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        cxy = (torch.rand(n, 2, generator=g) * 0.8 + 0.1) * size
        wh = (torch.rand(n, 2, generator=g) * 0.48 + 0.02) * size
        targets.append({"labels": torch.randint(0, num_classes, (n,), generator=g), "boxes": torch.cat([cxy, wh], 1)})
    return imgs, targets
