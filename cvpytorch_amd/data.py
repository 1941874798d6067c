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


class DevicePrefetcher:
    """Input pipeline on device. Mirrors the reference's `DataPrefetcher`
    (src/data/datasets/prefetch_dataLoader.py:20-60: a side stream uploads batch i+1 while batch i trains) with two changes:
      * the loader hands over **uint8 NHWC** images (what cv2 / the CPU augmentations produce before ToTensor): 4x fewer PCIe
        bytes than the fp32 NCHW batch of `trainer.py:157-175` (78.6 MB instead of 315 MB for 64x640x640x3);
      * `ToTensor` + `Normalize(mean, std)` (conf/coco_yolov5_s.yml:36-37) run on the device inside the relayout kernel
        `cvhip_u8_nhwc_to_bf16_norm`, which writes the bf16 NHWC (channels padded to 8) tensor the stem conv consumes.
    `loader` yields (images uint8 [N,H,W,C] host tensor, targets) — targets are moved with `.to(device, non_blocking=True)` when
    they are tensors and passed through otherwise. Staging buffers are pinned once and reused."""

    def __init__(self, loader, device, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)):
        import torch
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        m = torch.tensor(mean, dtype=torch.float32)
        s = torch.tensor(std, dtype=torch.float32)
        self.scale = (1.0 / (255.0 * s)).to(self.device)
        self.shift = (-m / s).to(self.device)
        self._pinned = [None, None]
        self._slot = 0
        self.next_input = self.next_target = None

    def __len__(self):
        return len(self.loader)

    def _upload(self, imgs, target):
        import torch
        from . import lib as L
        if imgs.dtype != torch.uint8 or imgs.dim() != 4:
            raise L.CvhipError("DevicePrefetcher expects uint8 [N,H,W,C] images")
        N, H, W, Cc = imgs.shape
        pin = self._pinned[self._slot]
        if pin is None or pin.shape != imgs.shape:
            pin = torch.empty(imgs.shape, dtype=torch.uint8).pin_memory()
            self._pinned[self._slot] = pin
        self._slot ^= 1
        pin.copy_(imgs)
        with torch.cuda.stream(self.stream):
            dev_u8 = pin.to(self.device, non_blocking=True)
            cp = (Cc + 7) // 8 * 8
            out = torch.empty((N, H, W, cp), dtype=torch.bfloat16, device=self.device)
            L.call("cvhip_u8_nhwc_to_bf16_norm", dev_u8.data_ptr(), N * H * W, Cc, out.data_ptr(), cp, self.scale.data_ptr(),
                   self.shift.data_ptr(), self.stream.cuda_stream)
            tgt = target.to(self.device, non_blocking=True) if torch.is_tensor(target) else target
        return out.permute(0, 3, 1, 2), tgt  # logical (N, 8, H, W) NHWC view; channels >= C are zero

    def _preload(self):
        try:
            imgs, target = next(self._it)
        except StopIteration:
            self.next_input = self.next_target = None
            return
        self.next_input, self.next_target = self._upload(imgs, target)

    def __iter__(self):
        import torch
        self._it = iter(self.loader)
        self._preload()
        while self.next_input is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
            x, t = self.next_input, self.next_target
            x.record_stream(torch.cuda.current_stream(self.device))
            self._preload()
            yield x, t
