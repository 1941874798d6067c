"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — CPU fp32 restatement of the reference's Classification model
(BASELINE config 1: mini-imagenet ResNet50 224x224 bs=8).

  model : src/models/classification.py:26-67 (criterion :41, per-class terms :61-65)
  loss  : src/losses/seg_loss.py:39-45 (CrossEntropyLoss2d = nn.CrossEntropyLoss(weight, ignore_index=255, 'mean'))
  net   : src/models/backbones/seg/resnet.py:96-99,149-153 (classifier=True -> torchvision avgpool + fc)
PARITY UNPINNED for the assembled model: the reference class cannot be constructed here (`.cuda()` in the constructor,
classification.py:41; torchvision absent) — the restatement is anchored on nn.CrossEntropyLoss semantics (same torch build)
and on the ResNet-50 restatement of oracle/torch_ref.py.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .torch_ref import ResNet50


class Classification(nn.Module):
    def __init__(self, dictionary, subtype="resnet50"):
        super().__init__()
        self.dictionary = dictionary
        self.num_classes = len(dictionary)
        self.weight = [w for d in dictionary for w in d.values()]
        self.backbone = ResNet50(subtype, classifier=True, num_classes=self.num_classes)
        self.criterion = nn.CrossEntropyLoss(weight=torch.tensor(self.weight).float(), ignore_index=255, reduction="mean")

    def forward(self, imgs, targets=None, mode="infer"):
        outputs = self.backbone(imgs)
        if mode == "infer":
            return F.softmax(outputs, dim=1)
        losses = {"loss": self.criterion(outputs, targets.long())}
        if mode == "val":
            return losses, torch.max(outputs, 1)[1]
        for idx, d in enumerate(self.dictionary):
            for label, w in d.items():
                sel = targets == idx
                if targets[sel].size(0):
                    losses["loss_" + label] = F.cross_entropy(outputs[sel], targets[sel]) * w
        return losses


def synthetic_cls_batch(batch=8, size=224, num_classes=100, seed=1029):
    """SURVEY §8d config 1."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, 3, size, size, generator=g), torch.randint(0, num_classes, (batch,), generator=g)
