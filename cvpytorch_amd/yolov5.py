"""YOLOv5 on the HIP engine: backbone / neck / detect / loss / post-process with the reference's module
tree (so reference checkpoints load) and call contract `model(imgs, targets, mode) -> {'loss': ...}`.

Reference files restated (the reference's own wiring is partly broken at HEAD — SURVEY.md §0.2 — so
the assembly follows the files as specification):
  backbone : src/models/backbones/det/yolov5_csp_darknet.py:17-102, base_yolo_backbone.py:16-112
  neck     : src/models/necks/det/yolov5_neck.py:15-61 (+ base_det_neck.py:29-36 scaling)
  detect   : src/models/detects/yolov5_detect.py:12-65
  model    : src/models/yolov5.py:156-287 (anchors :157-159, width/depth :160-165, NMS thresholds :189-190)
  loss     : src/losses/yolov5_loss.py:135-278 — re-formulated with FIXED shapes (no boolean-mask
             indexing, no host syncs, hipGraph-capturable); results equal the reference's on the same
             inputs (tests/test_yolov5_loss.py checks it against the oracle restatement).
  NMS      : src/models/yolov5.py:62-153
"""
import math

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .bricks import HipConv2d
from .bricks import HipConvModule as ConvModule
from .yolo_blocks import CSPLayer, DownsamplingModule, SPPF, UpsamplingModule

ANCHORS = [[[1.25000, 1.62500], [2.00000, 3.75000], [4.12500, 2.87500]],
           [[1.87500, 3.81250], [3.87500, 2.81250], [3.68750, 7.43750]],
           [[3.62500, 2.81250], [4.87500, 6.18750], [11.65625, 10.18750]]]
SCALES = {"n": (0.33, 0.25), "nano": (0.33, 0.25), "t": (0.33, 0.375), "tiny": (0.33, 0.375), "s": (0.33, 0.5),
          "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}


def _yolo_init(module):
    """yolov5_csp_darknet.py:94-102 / yolov5_neck.py:42-50."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            m.eps = 1e-3
            m.momentum = 0.03


class YOLOv5CSPDarknet(nn.Module):
    def __init__(self, subtype="cspdark_s", in_channels=3, out_channels=(64, 128, 256, 512, 1024), num_blocks=(3, 6, 9, 3),
                 spp_ksizes=5, norm_cfg=dict(type="BN", momentum=0.03, eps=0.001), act_cfg=dict(type="SiLU", inplace=True),
                 out_stages=(2, 3, 4)):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        self.out_channels = [int(x * width_mul) for x in out_channels]
        self.num_blocks = [max(round(x * depth_mul), 1) for x in num_blocks]
        self.out_stages = list(out_stages)
        self.norm_cfg, self.act_cfg = norm_cfg, act_cfg
        self.stem = ConvModule(in_channels, self.out_channels[0], kernel_size=6, stride=2, padding=2, norm_cfg=norm_cfg, act_cfg=act_cfg)
        for idx, (cin, cout, nb) in enumerate(zip(self.out_channels[:-1], self.out_channels[1:], self.num_blocks)):
            stage = [ConvModule(cin, cout, kernel_size=3, stride=2, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg),
                     CSPLayer(cout, cout, n=nb, shortcut=(idx != 3), norm_cfg=norm_cfg, act_cfg=act_cfg)]
            if idx == 3:
                stage.append(SPPF(cout, cout, kernel_sizes=spp_ksizes, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.add_module("stage%d" % (idx + 1), nn.Sequential(*stage))
        _yolo_init(self)

    def forward(self, x):
        x = self.stem(x)
        output = []
        link, pending = None, None   # fan-out link of the previous stage's output (its neck-side gradient folds into this stage's stride-2 dgrad)
        for i in range(1, 5):
            stage = getattr(self, "stage%d" % i)
            # the stride-2 conv's result only feeds the CSP layer's two 1x1 siblings: it may stay lazy (ops.LazyAct)
            csp = stage[1]
            lazy = False
            if self.training and isinstance(csp, CSPLayer) and x.is_cuda:
                s0 = stage[0]
                ph, pw = (x.shape[2] + 2 * 1 - 3) // 2 + 1, (x.shape[3] + 2 * 1 - 3) // 2 + 1
                lazy = ops.lazy_edge_ok(x.shape[0], s0.out_channels, ph, pw, csp.conv1.out_channels + csp.conv2.out_channels, ops.act_id_of(s0))
            x_in = x
            x = stage[0](x_in, lazy=lazy, dx_link=link)
            if pending is not None:   # the previous stage's output also feeds the neck: its side alias, made after the main consumer ran
                output.append(ops.fanout_side(pending[0], pending[1], link))
                pending = None
            link = None
            for m in list(stage)[1:]:
                x = m(x)
            if i in self.out_stages:
                if i < 4:   # feeds the next stage AND the neck: explicit fan-out; the neck's gradient rides into the next stage's dgrad
                    x_full = x
                    x, keep, link = ops.fanout_linked(x_full)
                    pending = (x_full, keep)
                else:
                    output.append(x)
        return output if len(self.out_stages) > 1 else output[0]


class YOLOv5Neck(nn.Module):
    def __init__(self, subtype="yolov5_s", in_channels=(256, 512, 1024), out_channels=(256, 512, 1024), num_blocks=(3, 3, 3, 3),
                 norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="SiLU", inplace=True)):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        self.in_channels = [max(round(x * width_mul), 1) for x in in_channels]
        self.out_channels = [max(round(x * width_mul), 1) for x in out_channels]
        self.num_blocks = [max(round(x * depth_mul), 1) for x in num_blocks]
        c = self.in_channels
        self.up_1 = UpsamplingModule(c[2], c[1], self.num_blocks[0], norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.up_2 = UpsamplingModule(c[1], self.out_channels[0], self.num_blocks[1], norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.down_1 = DownsamplingModule(c[0], c[1], self.num_blocks[2], norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.down_2 = DownsamplingModule(c[1], c[2], self.num_blocks[3], norm_cfg=norm_cfg, act_cfg=act_cfg)
        _yolo_init(self)

    def forward(self, x):
        x3, x4, x5 = x
        x4_up, x4_t = self.up_1(x5, x4)
        x3_up, x3_t = self.up_2(x4_up, x3)
        # x3_up -> down_1 (main consumer: folds the detect head's gradient into its stride-2 dgrad) and the detect head; same for x4_down
        x3_full = x3_up
        x3_up, x3_out, l3 = ops.fanout_linked(x3_full)
        x4_full = self.down_1(x3_up, x3_t, dx_link=l3)
        x3_out = ops.fanout_side(x3_full, x3_out, l3)
        x4_down, x4_out, l4 = ops.fanout_linked(x4_full)
        x5_down = self.down_2(x4_down, x4_t, dx_link=l4)
        x4_out = ops.fanout_side(x4_full, x4_out, l4)
        return [x3_out, x4_out, x5_down]


class YOLOv5Detect(nn.Module):
    """Per-pixel 1x1 prediction conv (+bias) and, in eval mode, the sigmoid / grid / anchor decode."""

    def __init__(self, num_classes=80, in_channels=(256, 512, 1024), stride=(8., 16., 32.), anchors=ANCHORS, depth_mul=1.0, width_mul=1.0):
        super().__init__()
        in_channels = [int(x * width_mul) for x in in_channels]
        self.num_classes = num_classes
        self.num_outputs = num_classes + 5
        self.num_layers = len(anchors)
        self.num_anchors = len(anchors[0])
        self.stride = list(stride)
        self.register_buffer("anchors", torch.tensor(anchors).float())
        self.m = nn.ModuleList(HipConv2d(x, self.num_outputs * self.num_anchors, 1) for x in in_channels)
        self.init_weight()

    def init_weight(self):
        """yolov5_detect.py:29-36 bias prior."""
        for mi, s in zip(self.m, self.stride):
            b = mi.bias.view(self.num_anchors, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (self.num_classes - 0.999999))
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def forward_raw(self, x):
        """the per-level 1x1 prediction convs only: [(N, na*no, H, W)] bf16 NHWC (input of the fused loss)"""
        return [self.m[i](x[i]) for i in range(self.num_layers)]

    def forward(self, x):
        raw = self.forward_raw(x)  # (N, na*no, H, W) NHWC bf16
        train_out = [ops.head_permute(r, self.num_anchors, self.num_outputs) for r in raw]  # (N, na, H, W, no) fp32
        if self.training:
            return None, train_out
        return self.decode(raw), train_out

    def decode(self, raw):
        """eval-mode sigmoid + grid/anchor decode of the raw maps -> (N, sum(na*H*W), no) (yolov5_detect.py:48-57)"""
        anchors_px = [self.anchors[i] * self.stride[i] for i in range(self.num_layers)]
        return ops.yolov5_decode(raw, self.stride, anchors_px, self.num_anchors, self.num_outputs)


# ------------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------------
def bbox_ciou_xywh(box1, box2, eps=1e-7):
    """CIoU of xywh boxes, both (..., 4). src/losses/yolov5_loss.py:12-54 with x1y1x2y2=False, CIoU=True."""
    b1_x1, b1_x2 = box1[..., 0] - box1[..., 2] / 2, box1[..., 0] + box1[..., 2] / 2
    b1_y1, b1_y2 = box1[..., 1] - box1[..., 3] / 2, box1[..., 1] + box1[..., 3] / 2
    b2_x1, b2_x2 = box2[..., 0] - box2[..., 2] / 2, box2[..., 0] + box2[..., 2] / 2
    b2_y1, b2_y2 = box2[..., 1] - box2[..., 3] / 2, box2[..., 1] + box2[..., 3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
    ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
    v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


class YOLOv5Loss(nn.Module):
    """Fixed-shape YOLOv5 loss. `targets` is (T, 6) [img, cls, cx, cy, w, h] (normalised); rows whose
    image index is < 0 are padding. Every intermediate has a static shape (3 anchors x T targets x 5
    offsets candidates per level, masked) so there is no data-dependent indexing and no host sync.

    Candidate ordinal = (offset*3 + anchor)*T + target reproduces the reference's row order after its
    boolean-mask filters (yolov5_loss.py:247-259), which matters for the objectness scatter: on
    duplicate cells the reference's (CPU) index_put keeps the LAST writer.
    """

    def __init__(self, num_classes, stride=(8., 16., 32.), anchors=ANCHORS, hyp_box=0.05, hyp_obj=1.0, hyp_cls=0.5):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = len(anchors)
        self.num_anchors = len(anchors[0])
        self.hyp_anchor_t = 4.0
        self.hyp_box, self.hyp_obj, self.hyp_cls = hyp_box, hyp_obj, hyp_cls
        self.register_buffer("anchors", torch.tensor(anchors).float())
        self.register_buffer("off", torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]).float() * 0.5)
        self.balance = {3: [4.0, 1.0, 0.4]}.get(self.num_layers, [4.0, 1.0, 0.25, 0.06, .02])
        self.cp, self.cn = 1.0, 0.0
        self.gr = 1.0
        self._gain_cache = {}

    def build_targets(self, shapes, targets):
        """Returns per level (b, a, gj, gi, tbox, anch, tcls, valid) with leading shape (5, na, T)."""
        na, T = self.num_anchors, targets.shape[0]
        dev = targets.device
        tvalid = targets[:, 0] >= 0
        ai = torch.arange(na, device=dev).float().view(na, 1).expand(na, T)
        t7 = torch.cat((targets.unsqueeze(0).expand(na, T, 6), ai[:, :, None]), 2)  # (na, T, 7)
        out = []
        for i in range(self.num_layers):
            anchors = self.anchors[i]
            ny, nx = shapes[i]
            key = (ny, nx, str(dev))
            gain = self._gain_cache.get(key)
            if gain is None:  # built once per (grid, device): a host->device copy here would break hipGraph capture
                gain = torch.tensor([1, 1, nx, ny, nx, ny, 1], device=dev, dtype=torch.float32)
                self._gain_cache[key] = gain
            t = t7 * gain
            r = t[:, :, 4:6] / anchors[:, None]
            jm = (torch.max(r, 1. / r).max(2)[0] < self.hyp_anchor_t) & tvalid[None]  # (na, T)
            gxy = t[:, :, 2:4]
            gxi = gain[2:4] - gxy  # (a python-list index would upload an index tensor: not capturable)
            g = 0.5
            jk = ((gxy % 1. < g) & (gxy > 1.))  # (na,T,2): j (x), k (y)
            lm = ((gxi % 1. < g) & (gxi > 1.))
            sel = torch.stack((torch.ones_like(jm), jk[..., 0], jk[..., 1], lm[..., 0], lm[..., 1])) & jm[None]  # (5, na, T)
            offsets = self.off[:, None, None, :]  # (5,1,1,2)
            gij = (gxy[None] - offsets).long()  # trunc toward zero == .long() in the reference (values >= -0.5 -> 0)
            gi = gij[..., 0].clamp(0, nx - 1)
            gj = gij[..., 1].clamp(0, ny - 1)
            b = t[None, :, :, 0].long().expand(5, na, T).clamp(min=0)
            c = t[None, :, :, 1].long().expand(5, na, T).clamp(0, self.num_classes - 1)
            a = t[None, :, :, 6].long().expand(5, na, T)
            # the reference clamps gi/gj IN PLACE through views of gij before building tbox (yolov5_loss.py:268-274)
            gij_c = torch.stack((gi, gj), -1).float()
            tbox = torch.cat(((gxy[None] - gij_c), t[None, :, :, 4:6].expand(5, na, T, 2)), -1)  # (5,na,T,4)
            anch = anchors[a]
            out.append((b, a, gj, gi, tbox, anch, c, sel))
        return out

    def forward(self, p, targets):
        dev = targets.device
        lcls = torch.zeros(1, device=dev)
        lbox = torch.zeros(1, device=dev)
        lobj = torch.zeros(1, device=dev)
        shapes = [(pi.shape[2], pi.shape[3]) for pi in p]
        tgt = self.build_targets(shapes, targets)
        for i, pi in enumerate(p):
            b, a, gj, gi, tbox, anch, tcls, sel = tgt[i]
            bs, na, ny, nx, no = pi.shape
            self_f = sel.float()
            n = self_f.sum()
            cell = ((b * na + a) * ny + gj) * nx + gi  # flat (image, anchor, row, col) index of every candidate
            # dense gather of all 5*na*T candidates; index_select's backward is an atomic index_add (the
            # advanced-indexing form pi[b,a,gj,gi] sorts its indices first: 1.2 ms per level on MI355X)
            ps = pi.reshape(-1, no).index_select(0, cell.reshape(-1)).view(5, na, -1, no)
            pxy = ps[..., :2].sigmoid() * 2. - 0.5
            pwh = (ps[..., 2:4].sigmoid() * 2) ** 2 * anch
            pbox = torch.cat((pxy, pwh), -1)
            iou = bbox_ciou_xywh(pbox, tbox)  # (5, na, T)
            denom = n.clamp(min=1.0)
            lbox = lbox + ((1.0 - iou) * self_f).sum() / denom * (n > 0)
            # objectness target: scatter iou into (bs, na, ny, nx); last candidate (largest ordinal) wins duplicates
            score = iou.detach().clamp(0).to(pi.dtype)
            ncell = bs * na * ny * nx
            ordinal = torch.arange(cell.numel(), device=dev).view_as(cell)
            # invalid candidates go to PRIVATE dump slots behind the grid (a single shared dump slot would
            # serialise ~16k atomic-max operations on one address: 5 ms per level on MI355X)
            cell = torch.where(sel, cell, ncell + ordinal)
            winner = torch.full((ncell + cell.numel(),), -1, dtype=torch.long, device=dev)
            winner = winner.scatter_reduce(0, cell.reshape(-1), ordinal.reshape(-1), reduce="amax", include_self=True)
            w = winner[:ncell]
            tobj = torch.where(w >= 0, score.reshape(-1)[w.clamp(min=0)] * self.gr + (1.0 - self.gr), torch.zeros((), device=dev, dtype=pi.dtype))
            tobj = tobj.view(bs, na, ny, nx)
            if self.num_classes > 1:
                tc = torch.full_like(ps[..., 5:], self.cn)
                tc.scatter_(-1, tcls.unsqueeze(-1), self.cp)
                bce = nn.functional.binary_cross_entropy_with_logits(ps[..., 5:], tc, reduction="none")
                lcls = lcls + (bce * self_f.unsqueeze(-1)).sum() / (denom * self.num_classes) * (n > 0)
            obji = nn.functional.binary_cross_entropy_with_logits(pi[..., 4], tobj)
            lobj = lobj + obji * self.balance[i]
        lbox = lbox * self.hyp_box
        lobj = lobj * self.hyp_obj
        lcls = lcls * self.hyp_cls
        bs = p[0].shape[0]
        return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()


class YOLOv5LossFused(nn.Module):
    """Same loss, same constructor, computed by libcvhip's fused kernels (cvhip_yolov5_loss_*) directly on the raw bf16 head
    maps [(N, A*NO, H, W)] instead of the permuted fp32 (N, A, H, W, NO) copies: build_targets, CIoU, BCE and all gradients
    in 8 launches per level, no torch autograd ops => the whole train step can be captured in ONE hipGraph."""

    def __init__(self, num_classes, stride=(8., 16., 32.), anchors=ANCHORS, hyp_box=0.05, hyp_obj=1.0, hyp_cls=0.5):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = len(anchors)
        self.num_anchors = len(anchors[0])
        self.anchors = [[list(map(float, a)) for a in lvl] for lvl in anchors]
        self.anchor_t = 4.0
        self.hyp_box, self.hyp_obj, self.hyp_cls = hyp_box, hyp_obj, hyp_cls
        self.balance = {3: [4.0, 1.0, 0.4]}.get(self.num_layers, [4.0, 1.0, 0.25, 0.06, .02])
        self._const_cache = {}

    def forward(self, raws, targets):
        total, stats = ops.yolov5_loss_fused(list(raws), targets, self)
        return total, stats


def targets_to_tensor(targets, max_targets=None, device=None):
    """list[dict(labels (n,), boxes (n,4) cxcywh)] -> (T,6) [img, cls, cx, cy, w, h], padded with img=-1 rows.
    (trans_specific_format, src/models/yolov5.py:218-244; padding makes the shape static.)"""
    rows = []
    for i, t in enumerate(targets):
        n = t["labels"].shape[0]
        g = torch.zeros((n, 6), device=t["labels"].device)
        g[:, 0] = i
        g[:, 1] = t["labels"].float()
        g[:, 2:] = t["boxes"].float()
        rows.append(g)
    out = torch.cat(rows, 0) if rows else torch.zeros((0, 6))
    if max_targets is not None:
        if out.shape[0] > max_targets:
            raise ValueError("more targets (%d) than max_targets (%d)" % (out.shape[0], max_targets))
        pad = torch.zeros((max_targets - out.shape[0], 6), device=out.device)
        pad[:, 0] = -1
        pad[:, 2:] = 0.5
        out = torch.cat([out, pad], 0)
    return out.to(device) if device is not None else out


# ------------------------------------------------------------------------------------------------------
# post-process
# ------------------------------------------------------------------------------------------------------
def unletterbox_boxes(boxes, pad, scale, width, height):
    """src/models/yolov5.py:269-284 without the device -> host -> numpy -> device round trip per image: remove the letterbox
    padding (pad = (pad_h, pad_w)), undo the resize (scale = (scale_h, scale_w)), clip to the original (width, height). Same fp32
    operations in the same order, so the result equals the reference's numpy code bit for bit. boxes (n, 4) xyxy."""
    dev, dt = boxes.device, boxes.dtype
    as_t = lambda v: torch.as_tensor(v, device=dev).to(dt).reshape(-1)  # noqa: E731
    pad, scale, width, height = as_t(pad), as_t(scale), as_t(width), as_t(height)
    sub = torch.stack((pad[1], pad[0], pad[1], pad[0]))
    div = torch.stack((scale[1], scale[0], scale[1], scale[0]))
    hi = torch.stack((width[0], height[0], width[0], height[0]))
    out = (boxes - sub) / div
    return torch.minimum(torch.maximum(out, torch.zeros((), device=dev, dtype=dt)), hi)


def xywh2xyxy(x):
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300):
    """src/models/yolov5.py:62-153 on the device: every image of the batch in ONE launch set (nms.non_max_suppression: confidence
    filter, device sort, class offsets, ballot / scan NMS, fixed-capacity outputs, one host read). `classes=` is a mask on the sort
    keys' input, images with more candidates than the batched kernels hold are redone without a capacity
    (nms.nms_one_image_unbounded). There is no host path: a CPU tensor raises."""
    if not prediction.is_cuda:
        raise L.CvhipError("non_max_suppression needs a device tensor (the HIP engine has no CPU fallback)")
    from . import nms as NMS
    # (validation thresholds — conf 0.001, multi_label — produce thousands of candidates per image: largest capacity)
    return NMS.non_max_suppression(prediction, conf_thres, iou_thres, classes, agnostic, multi_label, max_det, cap=8192)


class YOLOv5(nn.Module):
    """src/models/yolov5.py:156-287. forward(imgs, targets, mode): 'train' -> losses dict; 'val' -> (losses, outputs)."""
    anchors = ANCHORS

    def __init__(self, num_classes=80, subtype="s", max_targets=None, fused_loss=False):
        super().__init__()
        self.num_classes = num_classes
        self.fused_loss = fused_loss
        self.loss_capturable = fused_loss  # no torch autograd ops in the loss => arena.FlatTrainStep captures one graph
        self.depth_mul, self.width_mul = SCALES[subtype]
        self.backbone = YOLOv5CSPDarknet(subtype="cspdark_" + subtype, out_stages=(2, 3, 4))
        self.neck = YOLOv5Neck(subtype="yolov5_" + subtype, in_channels=(256, 512, 1024), out_channels=(256, 512, 1024), num_blocks=(3, 3, 3, 3))
        self.detect = YOLOv5Detect(num_classes=num_classes, in_channels=(256, 512, 1024), anchors=ANCHORS, depth_mul=self.depth_mul,
                                   width_mul=self.width_mul)
        self.loss = (YOLOv5LossFused if fused_loss else YOLOv5Loss)(num_classes, anchors=ANCHORS)
        self.conf_thres, self.iou_thres = 0.001, 0.6
        self.max_targets = max_targets
        for m in self.modules():  # yolov5.py:194-203
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    def forward_features(self, imgs):
        x = self.neck(self.backbone(imgs))
        if self.fused_loss:
            raw = self.detect.forward_raw(x)
            return (None if self.training else self.detect.decode(raw)), raw
        return self.detect(x)

    def loss_from_features(self, train_out, gts):
        losses = {}
        losses["loss"], st = self.loss(train_out, gts)
        losses["box_loss"], losses["obj_loss"], losses["cls_loss"] = st[0], st[1], st[2]
        return losses

    def forward(self, imgs, targets=None, mode="infer", meta=None, **kwargs):
        """`meta` (val mode, optional): dict with per-image 'width', 'height', 'scales' (scale_h, scale_w), 'pads' (pad_h, pad_w) as
        the reference's collate delivers them in `targets` (models/yolov5.py:266-267): boxes are then mapped back to the original
        images ON THE DEVICE (unletterbox_boxes) instead of through numpy on the host."""
        if mode == "infer":
            return
        gts = targets if torch.is_tensor(targets) else targets_to_tensor(targets, self.max_targets, imgs.device)
        out, train_out = self.forward_features(imgs)
        losses = self.loss_from_features(train_out, gts)
        if mode == "val":
            outputs = []
            if out is not None:
                preds = non_max_suppression(out, self.conf_thres, self.iou_thres, multi_label=True)
                for i, pred in enumerate(preds):
                    boxes = pred[:, :4]
                    if meta is not None:
                        boxes = unletterbox_boxes(boxes, meta["pads"][i], meta["scales"][i], meta["width"][i], meta["height"][i])
                    outputs.append({"boxes": boxes, "labels": pred[:, 5], "scores": pred[:, 4]})
            return losses, outputs
        return losses
