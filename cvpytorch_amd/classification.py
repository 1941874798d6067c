"""Classification model on the HIP engine (BASELINE config 1 plumbing: ResNet-50 + fc + CE).

  model : src/models/classification.py:26-67 — `model(imgs, targets, mode)`: 'infer' -> softmax, 'val' -> (losses, preds),
          'train' -> {'loss', 'loss_<label>'...} with the per-class weighted CE terms of :61-65
  loss  : src/losses/seg_loss.py:39-45
  net   : src/models/backbones/seg/resnet.py:96-99,149-153 (classifier=True)
Everything differentiable runs on libcvhip: the backbone, the 2048 -> classes fc (a 1x1 convolution on the pooled map) and the
cross-entropy (cvhip_seg_ce_* on the (N, classes, 1, 1) logits). The per-class REPORTING terms of :61-65 are a handful of torch ops
on the (N, classes) fp32 logits; non-uniform class weights (never used by the reference configs: every dictionary entry is 1.0) keep
torch's weighted cross-entropy.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .deeplab import ResNet


class Classification(nn.Module):
    def __init__(self, dictionary=None, subtype="resnet50", num_classes=None):
        super().__init__()
        if dictionary is None:
            dictionary = [{"class%d" % i: 1.0} for i in range(num_classes)]
        self.dictionary = dictionary
        self.num_classes = len(dictionary)
        self.category = [v for d in dictionary for v in d.keys()]
        self.weight = [d[v] for d in dictionary for v in d.keys()]
        self.backbone = ResNet(subtype, classifier=True, num_classes=self.num_classes)
        self.register_buffer("class_weight", torch.tensor(self.weight).float(), persistent=False)

    def forward(self, imgs, targets=None, mode="infer", **kwargs):
        raw = self.backbone(imgs)
        # the engine's backbone hands over NHWC logits (N, classes, 1, 1); a foreign backbone plain (N, classes)
        outputs = ops.to_nchw_f32(raw).flatten(1) if raw.dim() == 4 else raw
        if mode == "infer":
            return F.softmax(outputs, dim=1)
        targets = targets.long()
        if raw.dim() == 4 and raw.is_cuda and len(set(self.weight)) == 1:
            # uniform class weights: weighted mean == mean (nn.CrossEntropyLoss divides by the sum of the weights of the valid targets)
            losses = {"loss": ops.seg_cross_entropy(raw, targets.view(-1, 1, 1), ignore_index=255)}
        else:
            losses = {"loss": F.cross_entropy(outputs, targets, weight=self.class_weight, ignore_index=255, reduction="mean")}
        if mode == "val":
            return losses, torch.max(outputs, 1)[1]
        # per-class terms (:61-65): mean CE over the samples of class c, times the class weight; only present classes get a key
        outputs = outputs.detach() if raw.dim() == 4 else outputs   # reporting terms only (the reference never back-propagates them)
        ce = F.cross_entropy(outputs, targets.clamp(max=self.num_classes - 1), reduction="none")
        onehot = F.one_hot(targets.clamp(max=self.num_classes - 1), self.num_classes).to(ce.dtype) * (targets < self.num_classes)[:, None]
        cnt = onehot.sum(0)
        per = (onehot * ce[:, None]).sum(0) / cnt.clamp(min=1.0) * self.class_weight
        present = cnt.gt(0).cpu()  # one host sync, as the reference's `if targets[cognize].size(0)` per class
        for idx, label in enumerate(self.category):
            if bool(present[idx]):
                losses["loss_" + label] = per[idx]
        return losses
