"""Classification model on the HIP engine (BASELINE config 1 plumbing: ResNet-50 + fc + CE).

  model : src/models/classification.py:26-67 — `model(imgs, targets, mode)`: 'infer' -> softmax, 'val' -> (losses, preds),
          'train' -> {'loss', 'loss_<label>'...} with the per-class weighted CE terms of :61-65
  loss  : src/losses/seg_loss.py:39-45
  net   : src/models/backbones/seg/resnet.py:96-99,149-153 (classifier=True)
The backbone (all conv/BN/ReLU/pool work) runs on libcvhip; the 2048->classes fc and the cross-entropy on (N, classes)
logits are a few kFLOP and stay on torch's device ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

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
        outputs = self.backbone(imgs)
        if mode == "infer":
            return F.softmax(outputs, dim=1)
        targets = targets.long()
        losses = {"loss": F.cross_entropy(outputs, targets, weight=self.class_weight, ignore_index=255, reduction="mean")}
        if mode == "val":
            return losses, torch.max(outputs, 1)[1]
        # per-class terms (:61-65): mean CE over the samples of class c, times the class weight; only present classes get a key
        ce = F.cross_entropy(outputs, targets.clamp(max=self.num_classes - 1), reduction="none")
        onehot = F.one_hot(targets.clamp(max=self.num_classes - 1), self.num_classes).to(ce.dtype) * (targets < self.num_classes)[:, None]
        cnt = onehot.sum(0)
        per = (onehot * ce[:, None]).sum(0) / cnt.clamp(min=1.0) * self.class_weight
        present = cnt.gt(0).cpu()  # one host sync, as the reference's `if targets[cognize].size(0)` per class
        for idx, label in enumerate(self.category):
            if bool(present[idx]):
                losses["loss_" + label] = per[idx]
        return losses
