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
        self._events = [None, None]   # per staging slot: the H2D copy that last READ the pinned buffer
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
        slot = self._slot
        pin = self._pinned[slot]
        if pin is None or pin.shape != imgs.shape:
            pin = torch.empty(imgs.shape, dtype=torch.uint8).pin_memory()
            self._pinned[slot] = pin
        self._slot ^= 1
        if self._events[slot] is not None:
            self._events[slot].synchronize()   # the async copy issued from this slot two batches ago must have read it
        pin.copy_(imgs)
        with torch.cuda.stream(self.stream):
            dev_u8 = pin.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
            self._events[slot] = ev
            cp = (Cc + 7) // 8 * 8
            from . import ops
            out = torch.empty((N, H, W, cp), dtype=ops.ACT_DTYPE, device=self.device)
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
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(torch.cuda.current_stream(self.device))   # allocated on the side stream, consumed here
            self._preload()
            yield x, t


class GraphFeed:
    """The same on-device input pipeline for a hipGraph-replayed step, whose inputs live at FIXED addresses (the graph's static
    image / target tensors): host uint8 NHWC batch (pinned) --copy stream--> one of two device staging buffers; right before the
    replay the MAIN stream waits for that copy and runs `cvhip_u8_nhwc_to_bf16_norm` from the staging buffer into the static
    image tensor (ToTensor + Normalize + layout in one pass), then the targets' device copy. While step i replays, batch i+1
    is already crossing PCIe (78.6 MB instead of the 315 MB of the reference's fp32 batch: trainer.py:157-175 / prefetch
    dataloader), so the H2D time hides behind the step.

        feed = GraphFeed(step.static_imgs, step.static_targets, mean, std)
        feed.stage(u8_host, tgt_host)            # batch 0
        for i in range(K):
            feed.commit()                        # main stream: staged batch -> static tensors
            feed.stage(u8_host_next, tgt_next)   # copy stream: next batch (overlaps the replay below)
            step(step.static_imgs, step.static_targets)
    """

    def __init__(self, static_imgs, static_targets, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)):
        from . import lib as L
        from .ops import nhwc_ld
        from . import ops
        if static_imgs.dtype != ops.ACT_DTYPE or nhwc_ld(static_imgs) is None or nhwc_ld(static_imgs) % 8:
            raise L.CvhipError("GraphFeed: the static image tensor must be the 16-bit NHWC (channels padded to 8) tensor the stem consumes")
        self.imgs, self.targets = static_imgs, static_targets
        self.device = static_imgs.device
        self.ld = nhwc_ld(static_imgs)
        self.stream = torch.cuda.Stream(device=self.device)
        m = torch.tensor(mean, dtype=torch.float32)
        sd = torch.tensor(std, dtype=torch.float32)
        self.scale = (1.0 / (255.0 * sd)).to(self.device)
        self.shift = (-m / sd).to(self.device)
        N, _, H, W = static_imgs.shape
        self.shape = (N, H, W, len(mean))
        self._u8 = [torch.empty(self.shape, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self._tg = [torch.empty_like(static_targets) for _ in range(2)]
        self._copied = [torch.cuda.Event(), torch.cuda.Event()]    # H2D into slot finished
        self._consumed = [torch.cuda.Event(), torch.cuda.Event()]  # main stream finished reading slot
        self._used = [False, False]
        self._staged = [False, False]   # staged and not yet committed
        self._w = self._r = 0

    def stage(self, u8_host, tgt_host):
        """Start the asynchronous upload of one host batch (pinned uint8 [N,H,W,C] + pinned target tensor)."""
        from . import lib as L
        if u8_host.dtype != torch.uint8 or tuple(u8_host.shape) != self.shape or not u8_host.is_pinned():
            raise L.CvhipError("GraphFeed.stage expects a pinned uint8 %s batch" % (self.shape,))
        s = self._w
        if self._staged[s]:
            raise L.CvhipError("GraphFeed.stage: slot %d was staged and not committed yet (stage and commit must alternate, at most two batches ahead)" % s)
        self._w ^= 1
        if self._used[s]:
            self.stream.wait_event(self._consumed[s])   # the normalise kernel of two batches ago has read this slot
        with torch.cuda.stream(self.stream):
            self._u8[s].copy_(u8_host, non_blocking=True)
            self._tg[s].copy_(tgt_host, non_blocking=True)
            self._copied[s].record(self.stream)
        self._used[s] = True
        self._staged[s] = True
        # the caller may refill these pinned host buffers once this event has completed (`host_done(slot).synchronize()`): the copies
        # above are asynchronous and still read them
        return self._copied[s]

    def host_done(self, slot=None):
        """event that completes when the H2D copies of the most recently staged batch (or of `slot`) have read their host buffers"""
        return self._copied[(self._w ^ 1) if slot is None else slot]

    def _noop(self):
        pass

    def commit(self):
        """Main stream: wait for the oldest staged batch, normalise it into the static image tensor, copy its targets."""
        from . import lib as L
        s = self._r
        if not self._staged[s]:
            raise L.CvhipError("GraphFeed.commit: nothing staged in slot %d" % s)
        self._staged[s] = False
        self._r ^= 1
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._copied[s])
        N, H, W, Cc = self.shape
        L.call("cvhip_u8_nhwc_to_bf16_norm", self._u8[s].data_ptr(), N * H * W, Cc, self.imgs.data_ptr(), self.ld, self.scale.data_ptr(),
               self.shift.data_ptr(), cur.cuda_stream)
        self.targets.copy_(self._tg[s], non_blocking=True)
        self._consumed[s].record(cur)
