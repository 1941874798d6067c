"""Synthetic batches with the reference's batch format (src/data/datasets/coco.py:131-141 collate:
{'image': (N,3,H,W) float32 normalised, 'target': list[dict(labels (n,), boxes (n,4) cxcywh in [0,1])]}).
SURVEY.md §8(d): randn images; per image U{1..max_boxes} boxes, labels U{0..nc-1}, cx,cy ~ U(.1,.9),
w,h ~ U(.02,.5) clipped to the image; seed 1029 (trainer.py:55)."""
import torch


def synthetic_detection_batch(batch, size=640, num_classes=80, seed=1029, max_boxes=20, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    h, w = (size, size) if isinstance(size, int) else size
    imgs = torch.randn(batch, 3, h, w, generator=g)
    targets = []
    for _ in range(batch):
        n = int(torch.randint(1, max_boxes + 1, (1,), generator=g))
        labels = torch.randint(0, num_classes, (n,), generator=g)
        cxy = torch.rand(n, 2, generator=g) * 0.8 + 0.1
        wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
        wh = torch.min(wh, 2 * torch.min(cxy, 1 - cxy))
        targets.append({"labels": labels.to(device), "boxes": torch.cat([cxy, wh], 1).to(device)})
    return imgs.to(device), targets


def synthetic_segmentation_batch(batch, size=(512, 1024), num_classes=19, seed=1029, ignore_frac=0.05, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    imgs = torch.randn(batch, 3, size[0], size[1], generator=g)
    tgt = torch.randint(0, num_classes, (batch, size[0], size[1]), generator=g)
    ign = torch.rand(batch, size[0], size[1], generator=g) < ignore_frac
    tgt[ign] = 255
    return imgs.to(device), tgt.to(device)
