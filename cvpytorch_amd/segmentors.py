"""STDC segmentation TRAIN path on the engine (SURVEY.md §8(a) row 10 widened to what conf/seg/stdc/cityscapes_stdc1.yml:55-68 wires):

  FCNHead / STDCHead            reference src/models/heads/seg/fcn_head.py:14-63, stdc_head.py:16-18, base_seg_head.py:12-41
  OhemCrossEntropyLoss2d        reference src/losses/seg/cross_entropy_loss.py:51-69
  DetailAggregateLoss           reference src/losses/seg/detail_loss.py:23-88
  EncoderDecoder (with the auxiliary-head branch)   reference src/models/segmentors/encoder_decoder.py:21-150

The heads' convolutions are Hip ConvModules (the engine's kernels); every head's logits are resized to label size by the engine's
bilinear kernel. The two STDC losses are FIXED-SHAPE restatements in torch ops on those label-resolution logits — no boolean-mask
indexing and no host read, so a step keeps static shapes:
  * OHEM: the reference sorts all per-pixel losses, then branches on `loss[min_kept] > thresh` and averages either the losses above
    the threshold or the `min_kept` largest. Both averages are functions of the (min_kept + 1)-th largest value v alone:
    mean{l > thresh} = sum(l * [l > thresh]) / count, and the mean of the min_kept largest = (sum(l * [l > v]) + (min_kept -
    count(l > v)) * v) / min_kept (ties at v contribute v each) — one torch.topk, two masked sums, one torch.where.
  * Detail loss: the Laplacian pyramid of the label map is label-only arithmetic (no gradient); BCE-with-logits + dice on the boundary
    logits. (The reference's `fuse_kernel` is an nn.Parameter that receives no gradient — the targets are thresholded — and is kept as
    a buffer here.)
Round 6: the label-resolution logits no longer exist for the OHEM heads. The per-pixel losses come from the fused resize +
cross-entropy kernel on the LOW-resolution logits (cvhip_seg_ce_bilinear_fwd_px), the selection above runs on that fp32 vector, and
the backward is the fused kernel with the selection as per-pixel weights (ops.OhemCrossEntropyBilinear; x8 / x16 up-sampling through
smaller backward tiles); the detail loss's boundary targets are one kernel (cvhip_detail_boundary_targets). The torch-op forms above
remain the fallback (CPU, geometries the fused kernels refuse) and the statement the tests compare against.
Later in round 6 the selection moved to the device as well (cvhip_ohem_select: a three-pass radix select of the cut value and one
masked-sum reduction; the backward kernel derives each pixel's weight from its forward loss): nothing in the loss sorts or reads back,
`loss_capturable = True`, and arena.FlatTrainStep captures the whole step as one hipGraph."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .bricks import ConvModule, HipConv2d
from .stdc import STDCNet, STDCNeck


class FCNHead(nn.Module):
    """fcn_head.py:14-63 on BaseSegHead (base_seg_head.py:12-41): `num_convs` ConvModules, optional concat + conv_cat, Dropout2d, 1x1 cls_seg"""

    def __init__(self, num_classes, in_channels=None, channels=None, num_convs=2, kernel_size=3, is_concat=True, dilation=1, dropout_ratio=0.1,
                 conv_cfg=None, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="ReLU")):
        super().__init__()
        self.num_classes, self.in_channels, self.channels = num_classes, in_channels, channels
        self.num_convs, self.is_concat, self.kernel_size, self.dropout_ratio = num_convs, is_concat, kernel_size, dropout_ratio
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None   # parameter-free; applied by ops.dropout2d
        self.cls_seg = HipConv2d(channels, num_classes, kernel_size=1)
        if num_convs == 0:
            assert in_channels == channels
            self.convs = nn.Identity()
        else:
            pad = (kernel_size // 2) * dilation
            convs = [ConvModule(in_channels, channels, kernel_size, padding=pad, dilation=dilation, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg)]
            for _ in range(num_convs - 1):
                convs.append(ConvModule(channels, channels, kernel_size, padding=pad, dilation=dilation, conv_cfg=conv_cfg, norm_cfg=norm_cfg, act_cfg=act_cfg))
            self.convs = nn.Sequential(*convs)
        if is_concat:
            self.conv_cat = ConvModule(in_channels + channels, channels, kernel_size, padding=kernel_size // 2, conv_cfg=conv_cfg, norm_cfg=norm_cfg,
                                       act_cfg=act_cfg)
        self._init_weight()

    def _init_weight(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, (nn.BatchNorm2d, nn.Linear)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def classify(self, feat):
        if self.dropout is not None:
            feat = ops.dropout2d(feat, self.dropout_ratio, self.training)
        return self.cls_seg(feat)

    def forward(self, x):
        if self.is_concat:
            xa, xb = ops.fanout(x, 2)
            feats = self.conv_cat(ops.cat([xa, self.convs(xb)]))
        else:
            feats = self.convs(x)
        return self.classify(feats)


class STDCHead(FCNHead):
    """stdc_head.py:16-18: an FCNHead under another name (the detail head of STDC-Seg: num_classes = 1)"""


class CrossEntropyLoss2d(nn.Module):
    """cross_entropy_loss.py:33-48 on label-resolution logits (float32 NCHW)"""

    def __init__(self, ignore_index=255, loss_weight=1.0, loss_name="ce_loss"):
        super().__init__()
        self.ignore_index, self.loss_weight, self.loss_name = ignore_index, loss_weight, loss_name

    def forward(self, pred, target):
        return self.loss_weight * F.cross_entropy(pred, target.long(), ignore_index=self.ignore_index)


class OhemCrossEntropyLoss2d(nn.Module):
    """cross_entropy_loss.py:51-69 without the sort / data-dependent branch / boolean indexing (see the module docstring)"""

    def __init__(self, thresh=0.7, min_kept=100000, ignore_index=255, loss_weight=1.0, loss_name="ohem_ce_loss"):
        super().__init__()
        self.min_kept, self.ignore_index, self.loss_weight, self.loss_name = int(min_kept), ignore_index, loss_weight, loss_name
        self.register_buffer("thresh", -torch.log(torch.tensor(thresh, dtype=torch.float)), persistent=False)
        self.thresh_nlog = float(-torch.log(torch.tensor(thresh, dtype=torch.float)))   # the same fp32 value as a host scalar

    def forward_lowres(self, pred_lowres, target):
        """the same loss of `pred_lowres` resized (bilinear, half-pixel) to the label size, without the label-resolution logits: per-pixel
        losses and the weighted backward from the fused resize + cross-entropy kernels (ops.OhemCrossEntropyBilinear)"""
        if ops._OHEM_SELECT:   # the selection on the device as well (cvhip_ohem_select)
            return ops.OhemCrossEntropyBilinearFused.apply(pred_lowres, target, self.thresh_nlog, self.min_kept, self.ignore_index, self.loss_weight)
        return ops.OhemCrossEntropyBilinear.apply(pred_lowres, target, self.thresh, self.min_kept, self.ignore_index, self.loss_weight)

    def forward(self, pred, target):
        loss = self.loss_weight * F.cross_entropy(pred, target.long(), ignore_index=self.ignore_index, reduction="none").view(-1)
        if loss.numel() <= self.min_kept:
            raise IndexError("OhemCrossEntropyLoss2d: %d pixels, min_kept %d (the reference indexes loss[min_kept])" % (loss.numel(), self.min_kept))
        v = torch.topk(loss.detach(), self.min_kept + 1, sorted=True).values[self.min_kept]   # = sorted(loss, descending)[min_kept]
        thr = self.thresh.to(loss.dtype)
        above = (loss > thr).to(loss.dtype)
        mean_above = (loss * above).sum() / above.sum().clamp_min(1.0)
        gt = (loss > v).to(loss.dtype)
        mean_top = ((loss * gt).sum() + (self.min_kept - gt.sum()) * v) / float(self.min_kept)
        # the gradient of the tie term: the (min_kept - count) tied pixels each carry 1 / min_kept in the reference's loss[:min_kept];
        # WHICH of the tied pixels its sort keeps is implementation-defined — here their common gradient is spread over all ties
        ties = ((loss == v).to(loss.dtype))
        tie_term = (loss * ties).sum() / ties.sum().clamp_min(1.0)
        mean_top = mean_top + (self.min_kept - gt.sum()) / float(self.min_kept) * (tie_term - tie_term.detach())
        return torch.where(v > thr, mean_above, mean_top)


def dice_loss_func(inp, target):
    """detail_loss.py:11-20"""
    n = inp.size(0)
    iflat, tflat = inp.reshape(n, -1), target.reshape(n, -1)
    inter = (iflat * tflat).sum(1)
    return (1 - ((2.0 * inter + 1.0) / (iflat.sum(1) + tflat.sum(1) + 1.0))).mean()


class DetailAggregateLoss(nn.Module):
    """detail_loss.py:23-88: boundary targets from a Laplacian pyramid of the label map (strides 1 / 2 / 4, nearest up-sampling,
    thresholding, the 0.6 / 0.3 / 0.1 fuse and a last threshold), BCE-with-logits + dice against the detail head's logits"""

    def __init__(self, loss_weight=1.0, bce_loss_weight=1.0, dice_loss_weight=1.0, boundary_threshold=0.1, loss_name="detail_agg_loss"):
        super().__init__()
        self.loss_weight, self.bce_loss_weight, self.dice_loss_weight = loss_weight, bce_loss_weight, dice_loss_weight
        self.boundary_threshold, self.loss_name = boundary_threshold, loss_name
        self.register_buffer("laplacian_kernel", torch.tensor([-1, -1, -1, -1, 8, -1, -1, -1, -1], dtype=torch.float32).reshape(1, 1, 3, 3), persistent=False)
        self.register_buffer("fuse_kernel", torch.tensor([[6.0 / 10], [3.0 / 10], [1.0 / 10]], dtype=torch.float32).reshape(1, 3, 1, 1))

    @torch.no_grad()
    def boundary_targets(self, gtmasks):
        if gtmasks.is_cuda and gtmasks.dim() == 3:
            return ops.detail_boundary_targets(gtmasks, self.boundary_threshold)   # one kernel (cvhip_detail_boundary_targets)
        return self.boundary_targets_torch(gtmasks)

    @torch.no_grad()
    def boundary_targets_torch(self, gtmasks):
        g = gtmasks.unsqueeze(1).float()
        thr = self.boundary_threshold
        size = g.shape[2:]
        levels = []
        for s in (1, 2, 4):
            t = F.conv2d(g, self.laplacian_kernel, stride=s, padding=1).clamp(min=0)
            if s > 1:
                t = F.interpolate(t, size, mode="nearest")
            levels.append((t > thr).float())
        pyr = F.conv2d(torch.cat(levels, 1), self.fuse_kernel)
        return (pyr > thr).float()

    def forward(self, boundary_logits, gtmasks):
        tgt = self.boundary_targets(gtmasks)
        if boundary_logits.shape[-1] != tgt.shape[-1]:
            boundary_logits = F.interpolate(boundary_logits, tgt.shape[2:], mode="bilinear", align_corners=True)
        bce = F.binary_cross_entropy_with_logits(boundary_logits, tgt)
        dice = dice_loss_func(torch.sigmoid(boundary_logits), tgt)
        return self.loss_weight * (self.bce_loss_weight * bce + self.dice_loss_weight * dice)


_HEADS = {"FCNHead": FCNHead, "STDCHead": STDCHead}
_LOSSES = {"OhemCrossEntropyLoss2d": OhemCrossEntropyLoss2d, "DetailAggregateLoss": DetailAggregateLoss, "CrossEntropyLoss2d": CrossEntropyLoss2d}


def build_head(cfg):
    cfg = dict(cfg)
    return _HEADS[cfg.pop("name")](**cfg)


def build_loss(cfg):
    cfg = dict(cfg)
    return _LOSSES[cfg.pop("name")](**cfg)


# conf/seg/stdc/cityscapes_stdc1.yml:55-68
STDC1_CFG = dict(
    BACKBONE=dict(subtype="stdc1", out_channels=[32, 64, 256, 512, 1024], layers=[2, 2, 2], out_stages=[2, 3, 4]),
    NECK=dict(),
    HEAD=dict(name="FCNHead", num_classes=19, in_channels=256, channels=256, num_convs=1, is_concat=False),
    AUX_HEAD=[dict(name="STDCHead", num_classes=1, in_channels=256, channels=64, num_convs=1, is_concat=False),
              dict(name="FCNHead", num_classes=19, in_channels=128, channels=64, num_convs=1, is_concat=False),
              dict(name="FCNHead", num_classes=19, in_channels=128, channels=64, num_convs=1, is_concat=False)],
    LOSS=dict(name="OhemCrossEntropyLoss2d"),
    AUX_LOSS=[dict(name="DetailAggregateLoss"), dict(name="OhemCrossEntropyLoss2d"), dict(name="OhemCrossEntropyLoss2d")],
)


class STDCEncoderDecoder(nn.Module):
    """encoder_decoder.py:21-150 for the STDC configuration: backbone -> neck (-> (feats, aux_feats)) -> head; in train mode every
    head's logits are resized to label size (bilinear, align_corners False: encoder_decoder.py:96) and fed to its loss; auxiliary
    losses are prefixed "aux<i>_" (utils/misc.py add_prefix) and `loss` is the sum of all entries."""

    def __init__(self, cfg=None, min_kept=None):
        super().__init__()
        cfg = dict(STDC1_CFG if cfg is None else cfg)
        # Round 6: with the OHEM selection on the device the loss has no sort and no host read left (the detail loss's BCE / dice are
        # plain fixed-shape torch ops): arena.FlatTrainStep captures the whole step as ONE hipGraph. CVHIP_STDC_CAPTURE=0 keeps the
        # two graphs around an eager loss island (and is what runs when the selection falls back to torch.topk: CVHIP_OHEM_SELECT=0)
        self.loss_capturable = ops._OHEM_SELECT and __import__("os").environ.get("CVHIP_STDC_CAPTURE", "1") != "0"
        self.backbone = STDCNet(**cfg["BACKBONE"])
        self.neck = STDCNeck(**cfg.get("NECK", {}))
        self.head = build_head(cfg["HEAD"])
        self.auxiliary_head = nn.ModuleList(build_head(h) for h in cfg.get("AUX_HEAD", []))
        losses = cfg["LOSS"] if isinstance(cfg["LOSS"], (list, tuple)) else [cfg["LOSS"]]
        self.loss = nn.ModuleList(build_loss(l) for l in losses)
        self.auxiliary_loss = nn.ModuleList(build_loss(l) for l in cfg.get("AUX_LOSS", []))
        assert len(self.auxiliary_head) == len(self.auxiliary_loss)
        if min_kept is not None:   # (small test maps have fewer pixels than the default min_kept = 100000)
            for l in list(self.loss) + list(self.auxiliary_loss):
                if isinstance(l, OhemCrossEntropyLoss2d):
                    l.min_kept = int(min_kept)

    @property
    def with_auxiliary_head(self):
        return len(self.auxiliary_head) > 0

    def forward_features(self, imgs):
        feats = self.neck(self.backbone(imgs))
        aux_feats = None
        if isinstance(feats, tuple):
            feats, aux_feats = feats
        if not (self.training and self.with_auxiliary_head):
            return None, [self.head(feats)]
        # the fused map feeds the main head and (when the neck returns no auxiliary maps) every auxiliary head
        if aux_feats is None:
            fs = ops.fanout(feats, 1 + len(self.auxiliary_head))
            return None, [self.head(fs[0])] + [h(f) for h, f in zip(self.auxiliary_head, fs[1:])]
        return None, [self.head(feats)] + [h(f) for h, f in zip(self.auxiliary_head, aux_feats)]

    @staticmethod
    def _loss_forward(pred, targets, loss, into, prefix=""):
        """encoder_decoder.py:89-107: resize to the label size, then every loss of the list (same-named entries add up). OHEM losses
        take the low-resolution logits (resize + cross-entropy in one kernel, round 6) when the geometry allows; the others get the
        label-resolution fp32 logits, resized once"""
        ls = list(loss) if isinstance(loss, (list, tuple, nn.ModuleList)) else [loss]
        full = None
        for l in ls:
            k = prefix + l.loss_name
            if isinstance(l, OhemCrossEntropyLoss2d) and ops.ohem_cross_entropy_resized_ok(pred, targets):
                v = l.forward_lowres(pred, targets)
            else:
                if full is None:
                    full = ops.to_nchw_f32(ops.resize_bilinear(pred, targets.shape[-2:], False))
                v = l(full, targets)
            into[k] = into[k] + v if k in into else v

    def loss_from_features(self, feats, targets):
        losses = {}
        self._loss_forward(feats[0], targets, self.loss, losses)
        for i, (pred, l) in enumerate(zip(feats[1:], self.auxiliary_loss)):
            self._loss_forward(pred, targets, l, losses, "aux%d_" % i)
        losses["loss"] = sum(losses.values())
        return losses

    def forward(self, imgs, targets=None, mode="infer", **kwargs):
        _, feats = self.forward_features(imgs)
        if mode == "train":
            return self.loss_from_features(feats, targets)
        size = targets.shape[-2:] if targets is not None else imgs.shape[-2:]
        return torch.argmax(ops.to_nchw_f32(ops.resize_bilinear(feats[0], size, False)), dim=1)
