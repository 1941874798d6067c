"""oracle/torch_ref.py — TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Plain-PyTorch (CPU, fp32) restatement of the reference's hot-path modules. Every class cites the
reference file:line it follows. The restatement is pinned against the reference itself: the fixtures
in tests/golden/ were produced by tools/gen_golden.py, which imports the reference's own modules from
/root/reference (in the build container only) and records inputs / parameters / outputs / gradients;
tests/test_oracle_golden.py replays them through this file.

Nothing under cvpytorch_amd/ imports this module.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- bricks ------------------------------------------------------------------------------------------
class Swish(nn.Module):
    """src/models/bricks/swish.py:8-25."""

    def forward(self, x):
        return x * torch.sigmoid(x)


def build_act(act_cfg):
    """src/models/bricks/activation.py:13-30,86-98 (subset used by the hot path)."""
    if act_cfg is None:
        return None
    t = act_cfg["type"]
    if t == "SiLU":
        return nn.SiLU()
    if t == "Swish":
        return Swish()
    if t == "ReLU":
        return nn.ReLU()
    if t == "LeakyReLU":
        return nn.LeakyReLU(act_cfg.get("negative_slope", 0.01))
    if t == "Sigmoid":
        return nn.Sigmoid()
    raise KeyError(t)


class ConvModule(nn.Module):
    """src/models/bricks/conv_module.py:20-214: conv -> bn -> act; conv bias only without norm (:108-110);
    kaiming-normal(fan_out, relu) conv init, BN weight 1 / bias 0 (:180-199); norm default eps 1e-5
    (bricks/norm.py:111)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, order=("conv", "norm", "act")):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_act = act_cfg is not None
        self.order = order
        if bias == "auto":
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                              groups=groups, bias=bias)
        if self.with_norm:
            cfg = dict(norm_cfg)
            cfg.pop("type")
            requires_grad = cfg.pop("requires_grad", True)
            cfg.setdefault("eps", 1e-5)
            self.bn = nn.BatchNorm2d(out_channels if order.index("norm") > order.index("conv") else in_channels, **cfg)
            for p in self.bn.parameters():
                p.requires_grad = requires_grad
        if self.with_act:
            self.act = build_act(act_cfg)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        if self.with_norm:
            nn.init.constant_(self.bn.weight, 1)
            nn.init.constant_(self.bn.bias, 0)

    def forward(self, x, activate=True, norm=True):
        for layer in self.order:
            if layer == "conv":
                x = self.conv(x)
            elif layer == "norm" and norm and self.with_norm:
                x = self.bn(x)
            elif layer == "act" and activate and self.with_act:
                x = self.act(x)
        return x


class DepthwiseSeparableConvModule(nn.Module):
    """src/models/bricks/depthwise_separable_conv_module.py:10-99."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, norm_cfg=None,
                 act_cfg=dict(type="ReLU"), **kwargs):
        super().__init__()
        self.depthwise_conv = ConvModule(in_channels, in_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                                         groups=in_channels, norm_cfg=norm_cfg, act_cfg=act_cfg, **kwargs)
        self.pointwise_conv = ConvModule(in_channels, out_channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg, **kwargs)

    def forward(self, x):
        return self.pointwise_conv(self.depthwise_conv(x))


# ---- yolo blocks (src/models/modules/yolo_modules.py) -----------------------------------------------------
class Focus(nn.Module):
    """:19-37 — concat order TL, BL, TR, BR."""

    def __init__(self, in_channels, out_channels, kernel_sizes=1, stride=1, norm_cfg=dict(type="BN"), act_cfg=dict(type="Swish")):
        super().__init__()
        self.conv = ConvModule(in_channels * 4, out_channels, kernel_sizes, stride, padding=(kernel_sizes - 1) // 2, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x):
        tl, tr = x[..., ::2, ::2], x[..., ::2, 1::2]
        bl, br = x[..., 1::2, ::2], x[..., 1::2, 1::2]
        return self.conv(torch.cat((tl, bl, tr, br), dim=1))


class DarknetBottleneck(nn.Module):
    """:40-104."""

    def __init__(self, in_channels, out_channels, expansion=0.5, shortcut=True, depthwise=False, norm_cfg=dict(type="BN"),
                 act_cfg=dict(type="Swish")):
        super().__init__()
        hidden = int(out_channels * expansion)
        conv = DepthwiseSeparableConvModule if depthwise else ConvModule
        self.conv1 = ConvModule(in_channels, hidden, 1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv2 = conv(hidden, out_channels, 3, stride=1, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.shortcut = shortcut and in_channels == out_channels

    def forward(self, x):
        out = self.conv2(self.conv1(x))
        return out + x if self.shortcut else out


class CSPLayer(nn.Module):
    """:107-140."""

    def __init__(self, in_channels, out_channels, n=1, expansion=0.5, shortcut=True, depthwise=False, norm_cfg=dict(type="BN"),
                 act_cfg=dict(type="Swish")):
        super().__init__()
        hidden = int(out_channels * expansion)
        self.conv1 = ConvModule(in_channels, hidden, 1, stride=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv2 = ConvModule(in_channels, hidden, 1, stride=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.conv3 = ConvModule(2 * hidden, out_channels, 1, stride=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.m = nn.Sequential(*[DarknetBottleneck(hidden, hidden, 1.0, shortcut, depthwise, norm_cfg=norm_cfg, act_cfg=act_cfg)
                                 for _ in range(n)])

    def forward(self, x):
        x_1 = self.m(self.conv1(x))
        x_2 = self.conv2(x)
        return self.conv3(torch.cat((x_1, x_2), dim=1))


class UpsamplingModule(nn.Module):
    """:143-152."""

    def __init__(self, c1, c2, layer=3, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")):
        super().__init__()
        self.conv = ConvModule(c1, c2, 1, 1, 0, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.up = nn.UpsamplingNearest2d(scale_factor=2)
        self.fuse = CSPLayer(c2 * 2, c2, n=layer, shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x, y):
        x_conv = self.conv(x)
        return self.fuse(torch.cat([self.up(x_conv), y], dim=1)), x_conv


class DownsamplingModule(nn.Module):
    """:155-162."""

    def __init__(self, c1, c2, layer=3, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")):
        super().__init__()
        self.down = ConvModule(c1, c1, 3, 2, 1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.fuse = CSPLayer(c1 * 2, c2, n=layer, shortcut=False, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x, y):
        return self.fuse(torch.cat([self.down(x), y], dim=1))


class SPPF(nn.Module):
    """:165-194."""

    def __init__(self, in_channels, out_channels, kernel_sizes=(5, 9, 13), norm_cfg=dict(type="BN"), act_cfg=dict(type="Swish")):
        super().__init__()
        self.kernel_sizes = kernel_sizes
        hidden = in_channels // 2
        self.conv1 = ConvModule(in_channels, hidden, 1, stride=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        if isinstance(kernel_sizes, int):
            self.m = nn.MaxPool2d(kernel_size=kernel_sizes, stride=1, padding=kernel_sizes // 2)
        else:
            self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=ks, stride=1, padding=ks // 2) for ks in kernel_sizes])
        self.conv2 = ConvModule(hidden * 4, out_channels, 1, stride=1, norm_cfg=norm_cfg, act_cfg=act_cfg)

    def forward(self, x):
        x = self.conv1(x)
        if isinstance(self.kernel_sizes, int):
            y1 = self.m(x)
            y2 = self.m(y1)
            x = torch.cat([x, y1, y2, self.m(y2)], dim=1)
        else:
            x = torch.cat([x] + [m(x) for m in self.m], dim=1)
        return self.conv2(x)


# ---- YOLOv5 assembly ------------------------------------------------------------------------------------
ANCHORS = [[[1.25000, 1.62500], [2.00000, 3.75000], [4.12500, 2.87500]],
           [[1.87500, 3.81250], [3.87500, 2.81250], [3.68750, 7.43750]],
           [[3.62500, 2.81250], [4.87500, 6.18750], [11.65625, 10.18750]]]  # src/models/yolov5.py:157-159
SCALES = {"n": (0.33, 0.25), "t": (0.33, 0.375), "s": (0.33, 0.5), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}


def _yolo_init(module):
    """src/models/backbones/det/yolov5_csp_darknet.py:94-102."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_uniform_(m.weight, a=math.sqrt(5))
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            m.eps = 1e-3
            m.momentum = 0.03


class YOLOv5CSPDarknet(nn.Module):
    """src/models/backbones/det/yolov5_csp_darknet.py:17-91 + base_yolo_backbone.py:42-51."""

    def __init__(self, subtype="cspdark_s", in_channels=3, out_channels=(64, 128, 256, 512, 1024), num_blocks=(3, 6, 9, 3), spp_ksizes=5,
                 norm_cfg=dict(type="BN", momentum=0.03, eps=0.001), act_cfg=dict(type="SiLU"), out_stages=(2, 3, 4)):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        oc = [int(x * width_mul) for x in out_channels]
        nb = [max(round(x * depth_mul), 1) for x in num_blocks]
        self.out_stages = list(out_stages)
        self.stem = ConvModule(in_channels, oc[0], kernel_size=6, stride=2, padding=2, norm_cfg=norm_cfg, act_cfg=act_cfg)
        for idx, (cin, cout, n) in enumerate(zip(oc[:-1], oc[1:], nb)):
            stage = [ConvModule(cin, cout, kernel_size=3, stride=2, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg),
                     CSPLayer(cout, cout, n=n, shortcut=(idx != 3), norm_cfg=norm_cfg, act_cfg=act_cfg)]
            if idx == 3:
                stage.append(SPPF(cout, cout, kernel_sizes=spp_ksizes, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.add_module("stage%d" % (idx + 1), nn.Sequential(*stage))
        _yolo_init(self)

    def forward(self, x):
        x = self.stem(x)
        out = []
        for i in range(1, 5):
            x = getattr(self, "stage%d" % i)(x)
            if i in self.out_stages:
                out.append(x)
        return out


class YOLOv5Neck(nn.Module):
    """src/models/necks/det/yolov5_neck.py:22-61 (scaling: base_det_neck.py:29-36)."""

    def __init__(self, subtype="yolov5_s", in_channels=(256, 512, 1024), out_channels=(256, 512, 1024), num_blocks=(3, 3, 3, 3),
                 norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype.split("_")[1]]
        c = [max(round(x * width_mul), 1) for x in in_channels]
        oc = [max(round(x * width_mul), 1) for x in out_channels]
        nb = [max(round(x * depth_mul), 1) for x in num_blocks]
        self.up_1 = UpsamplingModule(c[2], c[1], nb[0], norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.up_2 = UpsamplingModule(c[1], oc[0], nb[1], norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.down_1 = DownsamplingModule(c[0], c[1], nb[2], norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.down_2 = DownsamplingModule(c[1], c[2], nb[3], norm_cfg=norm_cfg, act_cfg=act_cfg)
        _yolo_init(self)

    def forward(self, x):
        x3, x4, x5 = x
        x4_up, x4_t = self.up_1(x5, x4)
        x3_up, x3_t = self.up_2(x4_up, x3)
        x4_down = self.down_1(x3_up, x3_t)
        x5_down = self.down_2(x4_down, x4_t)
        return [x3_up, x4_down, x5_down]


class YOLOv5Detect(nn.Module):
    """src/models/detects/yolov5_detect.py:12-65."""

    def __init__(self, num_classes=80, in_channels=(256, 512, 1024), stride=(8., 16., 32.), anchors=ANCHORS, depth_mul=1.0, width_mul=1.0):
        super().__init__()
        in_channels = [int(x * width_mul) for x in in_channels]
        self.num_classes = num_classes
        self.num_outputs = num_classes + 5
        self.num_layers = len(anchors)
        self.num_anchors = len(anchors[0])
        self.stride = list(stride)
        self.register_buffer("anchors", torch.tensor(anchors).float())
        self.m = nn.ModuleList(nn.Conv2d(x, self.num_outputs * self.num_anchors, 1) for x in in_channels)
        for mi, s in zip(self.m, self.stride):
            b = mi.bias.view(self.num_anchors, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (self.num_classes - 0.999999))
            mi.bias = torch.nn.Parameter(b.view(-1), requires_grad=True)

    def forward(self, x):
        x = list(x)
        z = []
        for i in range(self.num_layers):
            x[i] = self.m[i](x[i])
            bs, _, ny, nx = x[i].shape
            x[i] = x[i].view(bs, self.num_anchors, self.num_outputs, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
            if not self.training:
                yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
                grid = torch.stack((xv, yv), 2).expand((1, self.num_anchors, ny, nx, 2)).float()
                anchor_grid = (self.anchors[i].clone() * self.stride[i]).view((1, self.num_anchors, 1, 1, 2)).expand(
                    (1, self.num_anchors, ny, nx, 2)).float()
                y = x[i].sigmoid()
                y[..., 0:2] = (y[..., 0:2] * 2 - 0.5 + grid) * self.stride[i]
                y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * anchor_grid
                z.append(y.view(bs, -1, self.num_outputs))
        return (None, x) if self.training else (torch.cat(z, 1), x)


def bbox_iou(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False, eps=1e-7):
    """src/losses/yolov5_loss.py:12-54. box1 is 4 (x n), box2 is n x 4."""
    box2 = box2.T
    if x1y1x2y2:
        b1_x1, b1_y1, b1_x2, b1_y2 = box1[0], box1[1], box1[2], box1[3]
        b2_x1, b2_y1, b2_x2, b2_y2 = box2[0], box2[1], box2[2], box2[3]
    else:
        b1_x1, b1_x2 = box1[0] - box1[2] / 2, box1[0] + box1[2] / 2
        b1_y1, b1_y2 = box1[1] - box1[3] / 2, box1[1] + box1[3] / 2
        b2_x1, b2_x2 = box2[0] - box2[2] / 2, box2[0] + box2[2] / 2
        b2_y1, b2_y2 = box2[1] - box2[3] / 2, box2[1] + box2[3] / 2
    inter = (torch.min(b1_x2, b2_x2) - torch.max(b1_x1, b2_x1)).clamp(0) * (torch.min(b1_y2, b2_y2) - torch.max(b1_y1, b2_y1)).clamp(0)
    w1, h1 = b1_x2 - b1_x1, b1_y2 - b1_y1 + eps
    w2, h2 = b2_x2 - b2_x1, b2_y2 - b2_y1 + eps
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    if GIoU or DIoU or CIoU:
        cw = torch.max(b1_x2, b2_x2) - torch.min(b1_x1, b2_x1)
        ch = torch.max(b1_y2, b2_y2) - torch.min(b1_y1, b2_y1)
        if CIoU or DIoU:
            c2 = cw ** 2 + ch ** 2 + eps
            rho2 = ((b2_x1 + b2_x2 - b1_x1 - b1_x2) ** 2 + (b2_y1 + b2_y2 - b1_y1 - b1_y2) ** 2) / 4
            if DIoU:
                return iou - rho2 / c2
            v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
            with torch.no_grad():
                alpha = v / (v - iou + (1 + eps))
            return iou - (rho2 / c2 + v * alpha)
        c_area = cw * ch + eps
        return iou - (c_area - union) / c_area
    return iou


class YOLOv5Loss:
    """src/losses/yolov5_loss.py:135-278 (the data-dependent, boolean-mask formulation). The only change
    is the integer clamp of :273 (`gj.clamp_(0, gain[3]-1)` with a float-tensor bound is rejected by
    torch >= 1.10): bounds are converted with int(), which is what torch <= 1.9 did implicitly."""

    def __init__(self, num_classes, stride=(8., 16., 32.), anchors=ANCHORS, hyp_box=0.05, hyp_obj=1.0, hyp_cls=0.5):
        self.num_classes = num_classes
        self.num_layers = len(anchors)
        self.num_anchors = len(anchors[0])
        self.hyp_anchor_t = 4.0
        self.hyp_box, self.hyp_obj, self.hyp_cls = hyp_box, hyp_obj, hyp_cls
        self.anchors = torch.tensor(anchors)
        self.BCEcls = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([1.0]))
        self.BCEobj = nn.BCEWithLogitsLoss(pos_weight=torch.tensor([1.0]))
        self.cp, self.cn = 1.0, 0.0
        self.balance = {3: [4.0, 1.0, 0.4]}.get(self.num_layers, [4.0, 1.0, 0.25, 0.06, .02])
        self.gr = 1.0

    def __call__(self, p, targets):
        lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
        tcls, tbox, indices, anchors = self.build_targets(p, targets)
        for i, pi in enumerate(p):
            b, a, gj, gi = indices[i]
            tobj = torch.zeros_like(pi[..., 0])
            n = b.shape[0]
            if n:
                ps = pi[b, a, gj, gi]
                pxy = ps[:, :2].sigmoid() * 2. - 0.5
                pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anchors[i]
                pbox = torch.cat((pxy, pwh), 1)
                iou = bbox_iou(pbox.T, tbox[i], x1y1x2y2=False, CIoU=True)
                lbox += (1.0 - iou).mean()
                score_iou = iou.detach().clamp(0).type(tobj.dtype)
                tobj[b, a, gj, gi] = (1.0 - self.gr) + self.gr * score_iou
                if self.num_classes > 1:
                    t = torch.full_like(ps[:, 5:], self.cn)
                    t[range(n), tcls[i]] = self.cp
                    lcls += self.BCEcls(ps[:, 5:], t)
            obji = self.BCEobj(pi[..., 4], tobj)
            lobj += obji * self.balance[i]
        lbox *= self.hyp_box
        lobj *= self.hyp_obj
        lcls *= self.hyp_cls
        bs = tobj.shape[0]
        return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach()

    def build_targets(self, p, targets):
        num_anchors, nt = self.num_anchors, targets.shape[0]
        tcls, tbox, indices, anch = [], [], [], []
        gain = torch.ones(7)
        ai = torch.arange(num_anchors).float().view(num_anchors, 1).repeat(1, nt)
        targets = torch.cat((targets.repeat(num_anchors, 1, 1), ai[:, :, None]), 2)
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
            gwh = t[:, 4:6]
            gij = (gxy - offsets).long()
            gi, gj = gij.T  # views: the in-place clamps below also clamp gij (as in the reference)
            a = t[:, 6].long()
            indices.append((b, a, gj.clamp_(0, int(gain[3]) - 1), gi.clamp_(0, int(gain[2]) - 1)))
            tbox.append(torch.cat((gxy - gij, gwh), 1))
            anch.append(anchors[a])
            tcls.append(c)
        return tcls, tbox, indices, anch


def targets_to_gts(targets):
    """src/models/yolov5.py:218-244 (the 'gts' part)."""
    rows = []
    for i, t in enumerate(targets):
        g = torch.zeros((t["labels"].shape[0], 6))
        g[:, 0] = i
        g[:, 1:] = torch.cat([t["labels"].unsqueeze(1).float(), t["boxes"].float()], 1)
        rows.append(g)
    return torch.cat(rows, 0)


def xywh2xyxy(x):
    """src/models/yolov5.py:52-59."""
    y = x.clone()
    y[:, 0] = x[:, 0] - x[:, 2] / 2
    y[:, 1] = x[:, 1] - x[:, 3] / 2
    y[:, 2] = x[:, 0] + x[:, 2] / 2
    y[:, 3] = x[:, 1] + x[:, 3] / 2
    return y


def box_iou(box1, box2):
    """src/models/yolov5.py:27-49."""
    def box_area(box):
        return (box[2] - box[0]) * (box[3] - box[1])
    area1, area2 = box_area(box1.T), box_area(box2.T)
    inter = (torch.min(box1[:, None, 2:], box2[:, 2:]) - torch.max(box1[:, None, :2], box2[:, :2])).clamp(0).prod(2)
    return inter / (area1[:, None] + area2 - inter)


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms contract (third-party; pinned torchvision 0.7 per reference README.md:54-55; not
    vendored in the reference, so PARITY UNPINNED against a torchvision binary): stable descending score
    sort; keep box i, suppress every later j with IoU(i,j) > thr; IoU = inter/(area_i+area_j-inter),
    area=(x2-x1)*(y2-y1), all in fp32; returns int64 indices in decreasing-score order."""
    import numpy as np
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    b = boxes.detach().float().cpu().numpy().astype(np.float32)
    s = scores.detach().float().cpu()
    order = torch.sort(s, descending=True, stable=True)[1].numpy()
    x1, y1, x2, y2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    n = len(order)
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    for _i in range(n):
        if suppressed[_i]:
            continue
        i = order[_i]
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest])
        yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest])
        yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1)
        h = np.maximum(np.float32(0), yy2 - yy1)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[_i + 1:] |= ovr > thr
    return torch.as_tensor(np.asarray(keep, dtype=np.int64))


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300):
    """src/models/yolov5.py:62-153 (time limit and merge-NMS branches are dead at the reference's settings)."""
    nc = prediction.shape[2] - 5
    xc = prediction[..., 4] > conf_thres
    max_wh, max_nms = 4096, 30000
    multi_label &= nc > 1
    output = [torch.zeros((0, 6))] * prediction.shape[0]
    for xi, x in enumerate(prediction):
        x = x[xc[xi]]
        if not x.shape[0]:
            continue
        x[:, 5:] *= x[:, 4:5]
        box = xywh2xyxy(x[:, :4])
        if multi_label:
            i, j = (x[:, 5:] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, j + 5, None], j[:, None].float()), 1)
        else:
            conf, j = x[:, 5:].max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes)).any(1)]
        n = x.shape[0]
        if not n:
            continue
        elif n > max_nms:
            x = x[x[:, 4].argsort(descending=True)[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        boxes, scores = x[:, :4] + c, x[:, 4]
        i = nms(boxes, scores, iou_thres)
        if i.shape[0] > max_det:
            i = i[:max_det]
        output[xi] = x[i]
    return output


class YOLOv5(nn.Module):
    """src/models/yolov5.py:156-261: backbone -> neck -> detect -> loss."""

    def __init__(self, num_classes=80, subtype="s"):
        super().__init__()
        depth_mul, width_mul = SCALES[subtype]
        self.backbone = YOLOv5CSPDarknet("cspdark_" + subtype)
        self.neck = YOLOv5Neck("yolov5_" + subtype)
        self.detect = YOLOv5Detect(num_classes, anchors=ANCHORS, depth_mul=depth_mul, width_mul=width_mul)
        self.loss = YOLOv5Loss(num_classes, anchors=ANCHORS)
        self.conf_thres, self.iou_thres = 0.001, 0.6
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    def forward(self, imgs, targets=None, mode="train"):
        gts = targets if torch.is_tensor(targets) else targets_to_gts(targets)
        losses = {}
        out, train_out = self.detect(self.neck(self.backbone(imgs)))
        losses["loss"], st = self.loss(train_out, gts)
        losses["box_loss"], losses["obj_loss"], losses["cls_loss"] = st[0], st[1], st[2]
        if mode == "val":
            preds = non_max_suppression(out, self.conf_thres, self.iou_thres, multi_label=True)
            return losses, [{"boxes": p[:, :4], "labels": p[:, 5], "scores": p[:, 4]} for p in preds]
        return losses


def unpad_scale_clip_numpy(pred, pad, scale, width, height):
    """src/models/yolov5.py:269-284, verbatim order of operations in numpy: letterbox padding removed, resize scale undone,
    boxes clipped to the original image. pred (n, 6) [x1, y1, x2, y2, score, label]; pad = (pad_h, pad_w), scale = (scale_h, scale_w)."""
    import numpy as np
    scale = np.asarray(scale.cpu().numpy() if torch.is_tensor(scale) else scale)
    pad = np.asarray(pad.cpu().numpy() if torch.is_tensor(pad) else pad)
    width = np.asarray(width.cpu().numpy() if torch.is_tensor(width) else width)
    height = np.asarray(height.cpu().numpy() if torch.is_tensor(height) else height)
    b = pred.clone()[:, :4].cpu().numpy()
    b[:, [0, 2]] -= pad[1]
    b[:, [1, 3]] -= pad[0]
    b[:, [0, 2]] /= scale[1]
    b[:, [1, 3]] /= scale[0]
    b[:, [0, 2]] = b[:, [0, 2]].clip(0, width)
    b[:, [1, 3]] = b[:, [1, 3]].clip(0, height)
    return {"boxes": torch.tensor(b), "labels": pred[:, 5], "scores": pred[:, 4]}


def synthetic_batch(batch, size=640, num_classes=80, seed=1029, max_boxes=20):
    """randn images; per image U{1..max_boxes} boxes, labels U{0..nc-1},
    cx,cy ~ U(.1,.9), w,h ~ U(.02,.5) clipped to the image; seed 1029 (trainer.py:55)."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(batch, 3, size, size, generator=g)
    targets = []
    for _ in range(batch):
        n = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        labels = torch.randint(0, num_classes, (n,), generator=g)
        cxy = torch.rand(n, 2, generator=g) * 0.8 + 0.1
        wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
        wh = torch.min(wh, 2 * torch.min(cxy, 1 - cxy))
        targets.append({"labels": labels, "boxes": torch.cat([cxy, wh], 1)})
    return imgs, targets


# ---- DeepLabv3+ / ResNet-50-v1c (BASELINE config 3) ------------------------------------------------------
class Bottleneck(nn.Module):
    """torchvision.models.resnet.Bottleneck (v1.5). THIRD-PARTY: torchvision is neither vendored in the reference nor
    installed here (README.md:54-55 names 0.7.0), so this block is restated from torchvision's public definition and is
    PARITY-UNPINNED against a torchvision binary."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        return self.relu(out + identity)


class ResNet50(nn.Module):
    """src/models/backbones/seg/resnet.py:27-154 for subtype resnet50 / resnet50v1c (deep stem :67-80). As written the
    reference never dilates ResNet-50 (:102-118) => output_stride 32; 8/16 give the intended torchvision dilation."""

    def __init__(self, subtype="resnet50v1c", out_stages=(1, 4), output_stride=32, classifier=False, num_classes=1000):
        super().__init__()
        self.out_stages, self.classifier = list(out_stages), classifier
        if subtype.endswith("c"):
            self.stem = nn.Sequential(nn.Conv2d(3, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                                      nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                                      nn.Conv2d(32, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        else:
            self.stem = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        dilate3, dilate4 = {32: (False, False), 16: (False, True), 8: (True, True)}[output_stride]
        self.inplanes, self._dilation = 64, 1
        self.layer1 = self._make_layer(64, 3, 1, False)
        self.layer2 = self._make_layer(128, 4, 2, False)
        self.layer3 = self._make_layer(256, 6, 2, dilate3)
        self.layer4 = self._make_layer(512, 3, 2, dilate4)
        if classifier:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, planes, blocks, stride, dilate):
        previous_dilation = self._dilation
        if dilate:
            self._dilation *= stride
            stride = 1
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample, dilation=previous_dilation)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes, dilation=self._dilation))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.stem(x))
        output = []
        for i in range(1, 5):
            x = getattr(self, "layer%d" % i)(x)
            if i in self.out_stages and not self.classifier:
                output.append(x)
        if self.classifier:
            return self.fc(torch.flatten(self.avgpool(x), 1))
        return output if len(self.out_stages) > 1 else output[0]


class ASPP(nn.ModuleList):
    """src/models/heads/seg/deeplabv3_head.py:15-48; depthwise-separable branches per deeplabv3plus_head.py:14-30."""

    def __init__(self, dilations, in_channels, channels, norm_cfg, act_cfg, depthwise=True):
        super().__init__()
        for d in dilations:
            if d > 1 and depthwise:
                self.append(DepthwiseSeparableConvModule(in_channels, channels, 3, dilation=d, padding=d, norm_cfg=norm_cfg, act_cfg=act_cfg))
            else:
                self.append(ConvModule(in_channels, channels, 1 if d == 1 else 3, dilation=d, padding=0 if d == 1 else d, norm_cfg=norm_cfg,
                                       act_cfg=act_cfg))

    def forward(self, x):
        return [m(x) for m in self]


class Deeplabv3PlusHead(nn.Module):
    """src/models/heads/seg/deeplabv3plus_head.py:33-68 (+ deeplabv3_head.py:50-74, base_seg_head.py:13-37)."""

    def __init__(self, num_classes, in_channels=2048, channels=512, dilations=(1, 12, 24, 36), low_in_channels=256, low_channels=48,
                 dropout_ratio=0.1, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="ReLU")):
        super().__init__()
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.cls_seg = nn.Conv2d(channels, num_classes, kernel_size=1)
        self.proj = nn.Sequential(nn.AdaptiveAvgPool2d(1), ConvModule(in_channels, channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg))
        self.aspp = ASPP(dilations, in_channels, channels, norm_cfg, act_cfg, depthwise=True)
        self.reduce = ConvModule((len(dilations) + 1) * channels, channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg)
        self.low_proj = ConvModule(low_in_channels, low_channels, 1, norm_cfg=norm_cfg, act_cfg=act_cfg) if low_in_channels > 0 else None
        self.fuse = nn.Sequential(
            DepthwiseSeparableConvModule(channels + low_channels, channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg),
            DepthwiseSeparableConvModule(channels, channels, 3, padding=1, norm_cfg=norm_cfg, act_cfg=act_cfg))

    def forward(self, x):
        outs = [F.interpolate(self.proj(x[1]), size=x[1].size()[2:], mode="bilinear", align_corners=False)]
        outs.extend(self.aspp(x[1]))
        outs = self.reduce(torch.cat(outs, dim=1))
        if self.low_proj is not None:
            low = self.low_proj(x[0])
            outs = torch.cat([F.interpolate(outs, size=low.size()[2:], mode="bilinear", align_corners=False), low], dim=1)
        outs = self.fuse(outs)
        if self.dropout is not None:
            outs = self.dropout(outs)
        return self.cls_seg(outs)


class EncoderDecoder(nn.Module):
    """src/models/segmentors/encoder_decoder.py:93-150 with CrossEntropyLoss2d (losses/seg/cross_entropy_loss.py:32-40)."""

    def __init__(self, num_classes=19, output_stride=32, dropout_ratio=0.1, ignore_index=255):
        super().__init__()
        self.backbone = ResNet50("resnet50v1c", (1, 4), output_stride)
        self.head = Deeplabv3PlusHead(num_classes, dropout_ratio=dropout_ratio)
        self.criterion = nn.CrossEntropyLoss(ignore_index=ignore_index)

    def forward(self, imgs, targets=None, mode="train"):
        preds = self.head(self.backbone(imgs))
        if mode == "train":
            preds = F.interpolate(preds, size=targets.shape[-2:], mode="bilinear", align_corners=False)
            ce = self.criterion(preds, targets.long())
            return {"ce_loss": ce, "loss": ce}
        return torch.argmax(F.interpolate(preds, size=targets.shape[-2:], mode="bilinear", align_corners=False), dim=1)


def synthetic_seg_batch(batch, size=(512, 1024), num_classes=19, seed=1029, ignore_frac=0.05):
    """SURVEY.md §8(d) config 3."""
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(batch, 3, size[0], size[1], generator=g)
    tgt = torch.randint(0, num_classes, (batch, size[0], size[1]), generator=g)
    tgt[torch.rand(batch, size[0], size[1], generator=g) < ignore_frac] = 255
    return imgs, tgt
