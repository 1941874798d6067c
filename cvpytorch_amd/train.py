"""Train-step harness reproducing the reference's `Trainer.run_step` semantics around the HIP engine.

Reference: trainer_det_yolov5.py:145-207 (step), src/optimizers/__init__.py:21-86 (param groups),
src/utils/ema.py:13-39 (EMA), trainer.py:312-313 (DDP wrap -> here: `GradBucketer`, bucketed RCCL
all-reduce launched from autograd hooks on a side stream, overlapped with backward).

bf16 needs no loss scaling, so `GradScaler` (trainer.py:189-201) is a no-op here; the fp16 path of
config 5 keeps a scaler hook (`grad_scale`).
"""
import math
from copy import deepcopy

import torch
import torch.distributed as dist
import torch.nn as nn


def build_param_groups(model, lr=0.01, backbone_lr=None, weight_decay=5e-4, bias_lr_mult=1.0):
    """src/optimizers/__init__.py:36-56: biases and norm weights get no decay; other weights decay."""
    groups = []
    bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))
    for k, v in model.named_modules():
        base_lr = backbone_lr if ("backbone" in k and backbone_lr is not None) else lr
        if hasattr(v, "bias") and isinstance(v.bias, nn.Parameter) and v.bias.requires_grad:
            groups.append({"params": [v.bias], "lr": base_lr * bias_lr_mult, "weight_decay": 0.0})
        if isinstance(v, bn):
            if isinstance(getattr(v, "weight", None), nn.Parameter) and v.weight.requires_grad:
                groups.append({"params": [v.weight], "lr": base_lr, "weight_decay": 0.0})
        elif hasattr(v, "weight") and isinstance(v.weight, nn.Parameter) and v.weight.requires_grad:
            groups.append({"params": [v.weight], "lr": base_lr, "weight_decay": weight_decay})
    return groups


def build_optimizer(model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4, backbone_lr=None):
    """conf/coco_yolov5_s.yml:98-108 -> torch.optim.SGD(momentum .937, nesterov)."""
    groups = build_param_groups(model, lr, backbone_lr, weight_decay)
    # the reference builds one group per parameter (177 for YOLOv5-s); groups with identical hyper-parameters
    # are merged (same arithmetic, 3 multi-tensor launches instead of ~500 per step)
    merged = {}
    for g in groups:
        merged.setdefault((g["lr"], g["weight_decay"]), []).extend(g["params"])
    groups = [{"params": ps, "lr": k[0], "weight_decay": k[1]} for k, ps in merged.items()]
    return torch.optim.SGD(groups, lr=lr, momentum=momentum, nesterov=nesterov)


class ModelEMA:
    """src/utils/ema.py:13-39 — EMA of every floating-point state_dict entry; decay ramps d*(1-exp(-x/2000)).
    The per-tensor python loop of the reference is replaced by two multi-tensor (_foreach) launches."""

    def __init__(self, model, decay=0.9999, updates=0):
        self.ema = deepcopy(model.module if hasattr(model, "module") else model).eval()
        self.updates = updates
        self.decay = lambda x: decay * (1 - math.exp(-x / 2000))
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._pairs = None

    def _collect(self, model):
        msd = (model.module if hasattr(model, "module") else model).state_dict()
        dst, src = [], []
        for k, item in self.ema.state_dict().items():
            if item.dtype.is_floating_point:
                dst.append(item)
                src.append(msd[k].detach())
        self._pairs = (dst, src)

    def update(self, model):
        with torch.no_grad():
            self.updates += 1
            d = self.decay(self.updates)
            if self._pairs is None:
                self._collect(model)
            dst, src = self._pairs
            torch._foreach_mul_(dst, d)
            torch._foreach_add_(dst, src, alpha=1 - d)


class GradBucketer:
    """DistributedDataParallel's gradient averaging, MI355X-style (trainer.py:312-313).

    Parameters are packed into flat fp32 buckets in REVERSE registration order (~ the order backward
    produces gradients). A post-accumulate-grad hook copies each gradient into its bucket slot; when
    a bucket is full its all-reduce (RCCL over xGMI on GPU, gloo in the CPU tests) is launched
    asynchronously on a side stream, overlapping the rest of backward. `finish()` waits, divides by
    world size and scatters the averaged values back into `.grad`.

    Bucket size: xGMI is point-to-point (7 links x ~153 GB/s); YOLOv5-s has 29 MB of fp32 gradients, so
    the default 8 MiB gives ~4 collectives — large enough to be bandwidth- rather than latency-bound,
    small enough to start while most of backward is still running.
    """

    def __init__(self, model, bucket_bytes=8 << 20, process_group=None, comm=None):
        from . import comm as CM
        self.comm = comm if comm is not None else CM.default_comm(process_group)
        self.world = self.comm.world if self.comm is not None else 1
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.buckets = []  # (flat tensor, [(param, offset, numel)])
        self._slot = {}
        cur, cur_n, order = [], 0, list(reversed(self.params))
        cap = max(1, bucket_bytes // 4)
        for p in order:
            if cur and cur_n + p.numel() > cap:
                self._close(cur, cur_n)
                cur, cur_n = [], 0
            cur.append((p, cur_n, p.numel()))
            cur_n += p.numel()
        if cur:
            self._close(cur, cur_n)
        if self.world > 1:
            # DDP's construction-time broadcast of rank 0's parameters and buffers (trainer.py:312-313); tensors that are not
            # contiguous (channels_last weights) travel through a contiguous copy
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):   # (integer buffers too: num_batches_tracked)
                    d = t.data
                    if d.is_contiguous():
                        self.comm.broadcast_(d, 0)
                    else:
                        tmp = d.contiguous()
                        self.comm.broadcast_(tmp, 0)
                        d.copy_(tmp)
            self.comm.wait()
        self._pending = [0] * len(self.buckets)
        self._launched = []
        self._stream = None
        self._hooks = []
        if self.world > 1:
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.reset()

    def _close(self, items, n):
        dev = items[0][0].device
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        bi = len(self.buckets)
        for p, off, ne in items:
            self._slot[p] = (bi, off, ne)
        self.buckets.append((flat, items))

    def reset(self):
        self._pending = [len(items) for _, items in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._got = set()
        self._next = 0

    def _on_grad(self, p):
        bi, off, ne = self._slot[p]
        flat = self.buckets[bi][0]
        flat[off:off + ne].copy_(p.grad.reshape(-1))
        self._got.add(p)
        self._pending[bi] -= 1
        self._launch_in_order()

    def _launch_in_order(self):
        """Buckets go out in INDEX order only (as DDP's reducer does): a full bucket waits until all lower-numbered ones have
        been launched, so the collective sequence is the same on every rank whatever order gradients arrive in."""
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def _launch(self, bi):
        flat = self.buckets[bi][0]
        self._launched[bi] = True
        if flat.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            self.comm.allreduce_(flat, stream=self._stream)
        else:
            self.comm.allreduce_(flat)

    def finish(self):
        """Wait for all buckets, average, write back into .grad. Call after backward().
        EVERY bucket is reduced on every rank, the ones backward did not complete in fixed index order with zero-filled slots for
        the parameters that got no gradient on this rank (data-dependent branches): the collective sequence never depends on
        which gradients a rank happened to produce, and a rank without a local gradient still receives the average."""
        if self.world == 1:
            return
        for bi in range(self._next, len(self.buckets)):
            flat, items = self.buckets[bi]
            for p, off, ne in items:
                if p not in self._got:
                    flat[off:off + ne].zero_()
            self._launch(bi)
        self._next = len(self.buckets)
        self.comm.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)
        inv = 1.0 / self.world
        for flat, items in self.buckets:
            flat.mul_(inv)
            for p, off, ne in items:
                if p.grad is None:
                    p.grad = flat[off:off + ne].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + ne].view_as(p.grad))
        self.reset()


def broadcast_buffers(model, src=0, comm=None):
    """DDP(broadcast_buffers=True): BN running stats follow rank 0 every iteration."""
    from . import comm as CM
    comm = comm if comm is not None else CM.default_comm()
    if comm is None or comm.world == 1:
        return
    bufs = [b for b in model.buffers() if b.dtype.is_floating_point]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    comm.broadcast_(flat, src)
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()


class TrainStep:
    """One reference train step: forward -> loss -> backward (+ bucketed grad all-reduce) -> SGD -> zero_grad -> EMA."""

    def __init__(self, model, optimizer, ema=None, bucketer=None, sync_buffers=False):
        self.model, self.optimizer, self.ema, self.bucketer = model, optimizer, ema, bucketer
        self.sync_buffers = sync_buffers

    def __call__(self, imgs, targets):
        if self.sync_buffers:
            broadcast_buffers(self.model)
        losses = self.model(imgs, targets, "train")
        losses["loss"].backward()
        if self.bucketer is not None:
            self.bucketer.finish()
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        if self.ema is not None:
            self.ema.update(self.model)
        return losses
