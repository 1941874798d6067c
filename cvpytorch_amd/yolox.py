"""YOLOX on the HIP engine (BASELINE config 4): backbone / PAFPN neck / decoupled head / SimOTA loss / post-process,
with the reference's module tree (reference checkpoints load) and call contract
`model(imgs, targets, mode) -> {'loss': ...}`.

Reference files restated (the reference's neck/head constructors are broken at HEAD — SURVEY.md §0.2 — so the
assembly follows the files as specification):
  backbone : src/models/backbones/det/yolox_csp_darknet.py:17-100   (Focus stem, SPP(5,9,13) ahead of the last CSP)
  neck     : src/models/necks/det/yolox_neck.py:16-105
  head     : src/models/heads/det/yolox_head.py:16-98
  loss     : src/losses/det/yolox_loss.py:73-435 — re-formulated with FIXED shapes: the per-image python loop, the
             boolean-mask compaction and the per-gt `.item()` top-k loop become dense (B, G, A) tensors with masks, so
             there is no host sync and the whole step stays capturable. The assignment is the reference's SimOTA
             (tests compare fg masks / matched gts with the oracle's restatement of the reference loop).
  model    : src/models/yolox.py:18-68 (post-process), :112-157
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .bricks import HipConv2d
from .bricks import HipConvModule as ConvModule
from .yolo_blocks import CSPLayer, DepthwiseSeparableConvModule, Focus, SPPF
from .yolov5 import SCALES

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
                 spp_ksizes=(5, 9, 13), norm_cfg=BN, act_cfg=dict(type="SiLU", inplace=True), out_stages=(2, 3, 4)):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        ch = [int(x * width_mul) for x in out_channels]
        nb = [max(round(x * depth_mul), 1) for x in num_blocks]
        self.out_channels = ch
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
        link, pending = None, None
        for i in range(1, 5):
            stage = list(getattr(self, "stage%d" % i))
            # (round 5) a stage output that also feeds the neck: the neck-side gradient rides into the next stage's stride-2 dgrad
            x_in = x
            x = stage[0](x_in, dx_link=link)
            if pending is not None:
                out.append(ops.fanout_side(pending[0], pending[1], link))
                pending = None
            link = None
            for m in stage[1:]:
                x = m(x)
            if i in self.out_stages:
                if i < 4:
                    x_full = x
                    x, keep, link = ops.fanout_linked(x_full)   # feeds the next stage AND the neck
                    pending = (x_full, keep)
                else:
                    out.append(x)
        return out if len(self.out_stages) > 1 else out[0]


class CSPDarknet(nn.Module):
    """The generic CSPDarknet backbone of src/models/backbones/det/csp_darknet.py:25-103 (YOLOv4 / YOLOX / AIRDet family) on the HIP
    engine: same constructor arguments, sub-module names (stem, stage1..stage4), `out_channels` attribute and state_dict keys, so
    reference checkpoints load. `depthwise=True` swaps the stride-2 stage convs and the bottlenecks' 3x3 for
    DepthwiseSeparableConvModule (dwconv.hip kernels)."""
    cfg = {"n": [0.33, 0.25], "t": [0.33, 0.375], "s": [0.33, 0.5], "m": [0.67, 0.75], "l": [1.0, 1.0], "x": [1.33, 1.25]}

    def __init__(self, subtype="cspdark_s", out_channels=(64, 128, 256, 512, 1024), layers=(3, 9, 9, 3), spp_ksizes=(5, 9, 13),
                 depthwise=False, conv_cfg=None, norm_cfg=dict(type="BN", momentum=0.03, eps=0.001), act_cfg=dict(type="Swish"),
                 out_stages=(2, 3, 4), output_stride=32, backbone_path=None, pretrained=False, frozen_stages=-1, norm_eval=False):
        super().__init__()
        self.subtype, self.out_stages, self.output_stride = subtype, list(out_stages), output_stride
        self.backbone_path, self.pretrained, self.frozen_stages, self.norm_eval = backbone_path, pretrained, frozen_stages, norm_eval
        conv = DepthwiseSeparableConvModule if depthwise else ConvModule
        depth_mul, width_mul = self.cfg[subtype.split("_")[1]]
        ch = [int(x * width_mul) for x in out_channels]
        nb = [max(round(x * depth_mul), 1) for x in layers]
        self.layers = nb
        self.stem = Focus(3, ch[0], kernel_sizes=3, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)
        for idx in range(4):
            stage = [conv(ch[idx], ch[idx + 1], 3, 2, padding=1, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)]
            if idx == 3:
                stage.append(SPPF(ch[idx + 1], ch[idx + 1], kernel_sizes=spp_ksizes, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg))
            stage.append(CSPLayer(ch[idx + 1], ch[idx + 1], n=nb[idx], shortcut=(idx != 3), depthwise=depthwise, conv_cfg=conv_cfg,
                                  norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.add_module("stage%d" % (idx + 1), nn.Sequential(*stage))
        self.out_channels = ch[self.out_stages[0]:self.out_stages[-1] + 1]
        self.init_weights()

    def init_weights(self):
        for m in self.modules():   # csp_darknet.py:96-103
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.stem(x)
        output = []
        for i in range(1, 5):
            x = getattr(self, "stage%d" % i)(x)
            if i in self.out_stages:
                if i < 4:
                    x, keep = ops.fanout(x, 2)
                    output.append(keep)
                else:
                    output.append(x)
        return output if len(self.out_stages) > 1 else output[0]


class YOLOXNeck(nn.Module):
    def __init__(self, subtype="yolox_s", in_channels=(256, 512, 1024), out_channels=256, num_blocks=(3, 3, 3, 3),
                 norm_cfg=dict(type="BN"), act_cfg=dict(type="Swish")):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        c = [max(round(x * width_mul), 1) for x in in_channels]
        oc = max(round(out_channels * width_mul), 1)
        nb = [max(round(x * depth_mul), 1) for x in num_blocks]
        self.in_channels, self.out_channels, self.num_blocks = c, oc, nb
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
            hi, hi_keep = ops.fanout(self.reduce_layers[n - 1 - idx](inner[0]), 2)   # -> the upsample here and a bottom-up concat below
            inner[0] = hi_keep
            # nn.Upsample(2, 'nearest') + cat in one kernel (yolox_neck.py:90-92)
            inner.insert(0, self.top_down_blocks[n - 1 - idx](ops.upsample2x_cat(hi, x[idx - 1])))
        outs = [inner[0]]
        heads = []
        for idx in range(n - 1):
            # -> the downsample (main consumer: folds the out_conv's gradient into its stride-2 dgrad) and its out_conv
            o_full = outs[-1]
            o, o_head, lk = ops.fanout_linked(o_full)
            d = self.downsamples[idx](o, dx_link=lk)
            heads.append(ops.fanout_side(o_full, o_head, lk))
            outs.append(self.bottom_up_blocks[idx](ops.cat([d, inner[idx + 1]])))
        heads.append(outs[-1])
        return [conv(o) for conv, o in zip(self.out_convs, heads)]


class YOLOXHead(nn.Module):
    """Decoupled head. Output per level is the raw (N, 4+1+nc, H, W) map, channel order [reg, obj, cls] (yolox_head.py:94)."""

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
            self.cls_preds.append(HipConv2d(ch, num_classes, 1, 1, 0))
            self.reg_preds.append(HipConv2d(ch, 4, 1, 1, 0))
            self.obj_preds.append(HipConv2d(ch, 1, 1, 1, 0))
        _init(self)
        bias_init = float(-math.log((1 - 1e-2) / 1e-2))
        for conv in list(self.cls_preds) + list(self.obj_preds):
            conv.bias.data.fill_(bias_init)

    def forward(self, x):
        outs = []
        for k, xx in enumerate(x):
            # class tower and box tower: the box tower runs first and is the main consumer (the class tower's gradient rides into its
            # first 3x3 dgrad: ops.fanout_linked)
            xr, xs, lk = ops.fanout_linked(xx)
            reg = self.reg_convs[k][1:](self.reg_convs[k][0](xr, dx_link=lk))
            xc = ops.fanout_side(xx, xs, lk)
            cls_feat = self.cls_convs[k](xc)
            rf, of = ops.fanout(reg, 2)      # box and objectness predictors
            outs.append(ops.cat([self.reg_preds[k](rf), self.obj_preds[k](of), self.cls_preds[k](cls_feat)]))
        return outs


# ------------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------------
def _pair_iou_cxcywh(g, p):
    """g (B,G,4), p (B,A,4) -> (B,G,A); yolox_loss.py:14-31 (xyxy=False)."""
    tl = torch.max((g[:, :, None, :2] - g[:, :, None, 2:] / 2), (p[:, None, :, :2] - p[:, None, :, 2:] / 2))
    br = torch.min((g[:, :, None, :2] + g[:, :, None, 2:] / 2), (p[:, None, :, :2] + p[:, None, :, 2:] / 2))
    area_g = (g[..., 2] * g[..., 3])[:, :, None]
    area_p = (p[..., 2] * p[..., 3])[:, None, :]
    en = ((tl < br).to(tl.dtype)).prod(dim=3)
    area_i = (br - tl).prod(dim=3) * en
    return area_i / (area_g + area_p - area_i)


class YOLOXLoss(nn.Module):
    """Dense SimOTA + 5*IoU + obj + cls. `preds`: list of (B, HW_l, 5+nc) fp32 raw maps (level order = strides);
    `targets`: (B, G, 5) [cls, cx, cy, w, h] in pixels, all-zero rows = padding (src/models/yolox.py:112-139)."""

    def __init__(self, num_classes, strides=(8, 16, 32)):
        super().__init__()
        self.num_classes = num_classes
        self.strides = list(strides)
        self._grid_cache = {}

    def _grids(self, hw, dev):
        key = (tuple(hw), str(dev))
        g = self._grid_cache.get(key)
        if g is None:
            xs, ys, ss = [], [], []
            for (h, w), s in zip(hw, self.strides):
                yv, xv = torch.meshgrid([torch.arange(h), torch.arange(w)], indexing="ij")
                xs.append(xv.reshape(-1).float())
                ys.append(yv.reshape(-1).float())
                ss.append(torch.full((h * w,), float(s)))
            g = (torch.cat(xs).to(dev), torch.cat(ys).to(dev), torch.cat(ss).to(dev))
            self._grid_cache[key] = g
        return g

    def decode(self, preds, hw):
        """get_output_and_grid :138-153 on the concatenated (B, A, C) map."""
        q = torch.cat(preds, 1)
        xs, ys, ss = self._grids(hw, q.device)
        grid = torch.stack((xs, ys), 1)
        out = torch.cat([(q[..., :2] + grid) * ss[:, None], torch.exp(q[..., 2:4]) * ss[:, None], q[..., 4:]], -1)
        return out, xs, ys, ss

    @torch.no_grad()
    def assign(self, boxes, obj, cls, targets, xs, ys, ss):
        """SimOTA (get_assignments :291-348, get_in_boxes_info :350-403, dynamic_k_matching :405-435), all images at once.
        Returns fg (B,A) bool, matched gt index (B,A) long, matched IoU (B,A), valid-gt mask (B,G)."""
        B, A, _ = boxes.shape
        G = targets.shape[1]
        nc = self.num_classes
        nlabel = (targets.sum(dim=2) > 0).sum(dim=1)                               # :165
        vg = torch.arange(G, device=boxes.device)[None, :] < nlabel[:, None]     # first nlabel rows are the labels (:190-191)
        gt = targets[..., 1:5]
        gcls = targets[..., 0].long().clamp(0, nc - 1)
        xc = (xs * ss + 0.5 * ss)[None, None, :]
        yc = (ys * ss + 0.5 * ss)[None, None, :]
        gl, gr = (gt[..., 0] - 0.5 * gt[..., 2])[..., None], (gt[..., 0] + 0.5 * gt[..., 2])[..., None]
        gt_, gb = (gt[..., 1] - 0.5 * gt[..., 3])[..., None], (gt[..., 1] + 0.5 * gt[..., 3])[..., None]
        in_boxes = (torch.minimum(torch.minimum(xc - gl, yc - gt_), torch.minimum(gr - xc, gb - yc)) > 0.0) & vg[..., None]
        rad = 2.5 * ss[None, None, :]
        cx, cy = gt[..., 0:1], gt[..., 1:2]
        in_centers = (torch.minimum(torch.minimum(xc - (cx - rad), yc - (cy - rad)), torch.minimum((cx + rad) - xc, (cy + rad) - yc)) > 0.0) & vg[..., None]
        cand = in_boxes.any(1) | in_centers.any(1)                                # (B,A) = is_in_boxes_anchor
        in_both = in_boxes & in_centers
        usable = cand[:, None, :] & vg[..., None]                                # (B,G,A)

        ious = _pair_iou_cxcywh(gt, boxes)
        ious = torch.where(usable, ious, torch.zeros((), device=ious.device))
        iou_cost = -torch.log(ious + 1e-8)
        # cls cost = BCE(sqrt(sigmoid(cls)*sigmoid(obj)), onehot(gt cls)).sum(classes) (:328-332), split into a per-anchor base
        # (all-negative labels) plus the correction of the one positive class -> (B,A,nc) work instead of (B,G,A,nc)
        p = (cls.float().sigmoid() * obj.float().sigmoid()).sqrt()
        log_p = torch.log(p).clamp(min=-100.0)
        log_1p = torch.log(1.0 - p).clamp(min=-100.0)
        base = -log_1p.sum(-1)                                                    # (B,A)
        corr = (log_1p - log_p).transpose(1, 2)                                   # (B,nc,A)
        cls_cost = base[:, None, :] + torch.gather(corr, 1, gcls[..., None].expand(B, G, A))
        cost = cls_cost + 3.0 * iou_cost + 100000.0 * (~in_both).float()
        big = torch.full((), float("inf"), device=cost.device)
        cost = torch.where(usable, cost, big)

        kk = min(10, A)
        topk_ious, _ = torch.topk(ious, kk, dim=2)
        dyn_k = torch.clamp(topk_ious.sum(2).int(), min=1)                        # (B,G)
        cvals, cidx = torch.topk(cost, kk, dim=2, largest=False)                  # sorted ascending
        sel = (torch.arange(kk, device=cost.device)[None, None, :] < dyn_k[..., None]) & torch.isfinite(cvals)
        matching = torch.zeros(B, G, A, device=cost.device).scatter_(2, cidx, sel.float())
        multi = matching.sum(1) > 1                                               # (B,A)
        amin = torch.argmin(cost, dim=1)                                          # (B,A)
        onehot = F.one_hot(amin, G).permute(0, 2, 1).to(matching.dtype)           # (B,G,A)
        matching = torch.where(multi[:, None, :], onehot, matching)
        fg = matching.sum(1) > 0
        matched = matching.argmax(1)
        m_iou = (matching * ious).sum(1)
        return fg, matched, m_iou, vg

    def forward(self, preds, targets, hw=None, return_assign=False):
        if hw is None:
            raise ValueError("hw (list of (H, W) per level) is required")
        nc = self.num_classes
        out, xs, ys, ss = self.decode(preds, hw)
        boxes, obj, cls = out[..., :4], out[..., 4], out[..., 5:5 + nc]
        fg, matched, m_iou, vg = self.assign(boxes.detach(), obj.detach()[..., None], cls.detach(), targets, xs, ys, ss)
        fgf = fg.to(out.dtype)
        num_fg = fgf.sum().clamp(min=1.0)
        num_gts = vg.sum().clamp(min=1).to(out.dtype)
        tbox = torch.gather(targets[..., 1:5], 1, matched[..., None].expand(-1, -1, 4))     # (B,A,4)
        tcls = torch.gather(targets[..., 0].long().clamp(0, nc - 1), 1, matched)             # (B,A)
        # IOUloss (:34-69), 'iou' type
        tl = torch.max(boxes[..., :2] - boxes[..., 2:] / 2, tbox[..., :2] - tbox[..., 2:] / 2)
        br = torch.min(boxes[..., :2] + boxes[..., 2:] / 2, tbox[..., :2] + tbox[..., 2:] / 2)
        area_p = boxes[..., 2] * boxes[..., 3]
        area_g = tbox[..., 2] * tbox[..., 3]
        en = (tl < br).to(out.dtype).prod(dim=-1)
        area_i = (br - tl).prod(dim=-1) * en
        iou = area_i / (area_p + area_g - area_i + 1e-16)
        zero = torch.zeros((), device=out.device, dtype=out.dtype)
        loss_iou = torch.where(fg, 1 - iou ** 2, zero).sum() / num_fg
        loss_obj = F.binary_cross_entropy_with_logits(obj, fgf, reduction="sum") / num_fg
        cls_t = F.one_hot(tcls, nc).to(out.dtype) * m_iou[..., None]
        bce = F.binary_cross_entropy_with_logits(cls, cls_t, reduction="none")
        loss_cls = torch.where(fg[..., None], bce, zero).sum() / num_fg
        loss = 5.0 * loss_iou + loss_obj + loss_cls
        res = {"loss": loss, "conf_loss": loss_obj.detach(), "cls_loss": loss_cls.detach(), "iou_loss": (5.0 * loss_iou).detach(),
               "num_fg": (fgf.sum().clamp(min=1.0) / num_gts).detach()}
        return (res, (fg, matched, m_iou)) if return_assign else res


class YOLOXLossFused(nn.Module):
    """Same loss computed by libcvhip's SimOTA kernels (cvhip_simota_loss_*) directly on the raw bf16 head maps
    [(B, 5+nc, H, W)]: no (B, G, A) tensors, ~9 launches, no torch autograd ops => the whole YOLOX step is ONE hipGraph."""

    def __init__(self, num_classes, strides=(8, 16, 32)):
        super().__init__()
        self.num_classes = num_classes
        self.strides = list(strides)

    def forward(self, raws, targets, hw=None, return_assign=False):
        out5 = ops.simota_loss_fused(list(raws), targets, self)
        res = {"loss": out5[0], "conf_loss": out5[1].detach(), "cls_loss": out5[2].detach(), "iou_loss": out5[3].detach(),
               "num_fg": out5[4].detach()}
        if return_assign:
            fn = out5.grad_fn
            m, u = ops.simota_read_assignment(fn.desc, fn.ws)
            return res, (m >= 0, m.clamp(min=0).long(), u)
        return res


def targets_to_padded(targets, max_labels=None, device=None):
    """src/models/yolox.py:112-139: list of {'labels','boxes' (pixel cxcywh)} -> (B, max_labels, 5). A fixed `max_labels`
    keeps the shape static (hipGraph replay)."""
    mx = max([int(t["labels"].shape[0]) for t in targets] + [1])
    if max_labels is not None:
        if mx > max_labels:
            raise ValueError("an image has %d labels, more than max_labels=%d" % (mx, max_labels))
        mx = max_labels
    out = torch.zeros(len(targets), mx, 5)
    for i, t in enumerate(targets):
        n = t["labels"].shape[0]
        if n:
            out[i, :n] = torch.cat([t["labels"].float().unsqueeze(1).cpu(), t["boxes"].float().cpu()], 1)
    return out.to(device) if device is not None else out


class YOLOX(nn.Module):
    """src/models/yolox.py:71-188."""

    def __init__(self, num_classes=80, subtype="s", max_labels=None, fused_loss=False):
        super().__init__()
        self.num_classes = num_classes
        self.fused_loss = fused_loss
        self.loss_capturable = fused_loss
        self.depth_mul, self.width_mul = SCALES[subtype]
        self.backbone = YOLOXCSPDarknet("cspdark_" + subtype)
        self.neck = YOLOXNeck("yolox_" + subtype)
        self.head = YOLOXHead("yolox_" + subtype, num_classes=num_classes, norm_cfg=BN)
        self.loss = (YOLOXLossFused if fused_loss else YOLOXLoss)(num_classes)
        self.stride = [8, 16, 32]
        self.conf_thr, self.nms_thr = 0.01, 0.65
        self.max_labels = max_labels
        self._hw = None

    def forward_features(self, imgs):
        """-> (None, [(B, HW_l, 5+nc) fp32 per level]); the NHWC head maps already are the (B, HW, C) layout the loss reads."""
        raw = self.head(self.neck(self.backbone(imgs)))
        self._hw = [(int(r.shape[2]), int(r.shape[3])) for r in raw]
        if self.fused_loss:
            return None, list(raw)  # the fused loss reads the bf16 NHWC maps as they are
        c = self.num_classes + 5
        return None, [ops.head_permute(r, 1, c).view(r.shape[0], -1, c) for r in raw]

    def loss_from_features(self, feats, gts):
        return self.loss(feats, gts, hw=self._hw)

    def forward(self, imgs, targets=None, mode="infer", **kwargs):
        if mode == "infer":
            return
        gts = targets if torch.is_tensor(targets) else targets_to_padded(targets, self.max_labels, imgs.device)
        _, feats = self.forward_features(imgs)
        losses = self.loss_from_features(feats, gts)
        if mode == "val":
            if self.fused_loss:
                c = self.num_classes + 5
                feats = [ops.head_permute(r.detach(), 1, c).view(r.shape[0], -1, c) for r in feats]
            return losses, decode_and_nms([f.detach() for f in feats], self._hw, self.stride, self.num_classes, self.conf_thr, self.nms_thr)
        return losses


def decode_and_nms(feats, hw, strides, num_classes, conf_thre, nms_thre):
    """yolox_post_process (src/models/yolox.py:18-68) on (B, HW_l, 5+nc) maps; per image (x1,y1,x2,y2,obj,cls_conf,cls) or None.
    batched_nms = NMS on boxes shifted by class * (max coordinate + 1) (torchvision's own strategy), run by cvhip_nms_sorted."""
    out = torch.cat(feats, 1).float()
    dev = out.device
    xs, ys, ss = [], [], []
    for (h, w), s in zip(hw, strides):
        yv, xv = torch.meshgrid([torch.arange(h, device=dev), torch.arange(w, device=dev)], indexing="ij")
        xs.append(xv.reshape(-1).float())
        ys.append(yv.reshape(-1).float())
        ss.append(torch.full((h * w,), float(s), device=dev))
    grid = torch.stack((torch.cat(xs), torch.cat(ys)), 1)
    ss = torch.cat(ss)[:, None]
    xy = (out[..., 0:2] + grid) * ss
    wh = torch.exp(out[..., 2:4]) * ss
    sc = torch.sigmoid(out[..., 4:5 + num_classes])
    # every image at once on the device (nms.yolox_post_process -> cvhip_detect_postprocess mode 1): confidence filter, device
    # sort, class offsets of (max coordinate + 1), greedy NMS
    from . import nms as NMS
    return NMS.yolox_post_process(torch.cat((xy, wh, sc), -1), num_classes, conf_thre, nms_thre)
