"""Flat parameter / gradient / momentum / EMA arenas and the fused optimizer step.

The reference's step tail is python-loop heavy: torch.optim.SGD over 177 one-parameter groups
(src/optimizers/__init__.py:36-68), `ModelEMA.update` looping over every state_dict entry
(src/utils/ema.py:30-39), DDP's bucket copy-in / copy-out (trainer.py:312-313). Here all trainable
parameters live in ONE fp32 arena (each nn.Parameter is re-pointed at a view of it, shape and strides
unchanged), their gradients in a second arena of identical layout (`p.grad` is a permanent view), so:

  * backward writes weight / BN gradients straight into the gradient arena (cvhip_conv2d_wgrad with
    accumulate=1, cvhip_bn_bwd_finalize(accumulate=1)) — no per-parameter .grad tensors, no clones;
  * gradient all-reduce runs IN PLACE on contiguous arena ranges (buckets in reverse layer order,
    launched on a side stream as soon as the layers inside have produced their gradients; the 1/world
    average is folded into the optimizer's grad_scale);
  * SGD(momentum, nesterov, weight decay) + EMA for all parameters is ONE kernel
    (cvhip_sgd_nesterov_ema), the EMA of the BN running statistics one more (cvhip_ema_update);
  * zero_grad is one memset.

Semantics are torch.optim.SGD's and ModelEMA's (checked by tests/test_gpu_arena.py against the stock
optimizer path).
"""
import math
from copy import deepcopy

import torch

from . import comm as CM
from . import lib as L
from . import ops
from .train import build_param_groups

_PREP_PLAN = __import__("os").environ.get("CVHIP_PREP_PLAN", "1") != "0"
_ALIGN = 8  # floats (32 B): keeps every parameter 16-byte aligned for the vectorised packers


def _dense_view(flat, off, like):
    return flat[off:off + like.numel()].as_strided(like.shape, like.stride())


def paired_arena_order(items, follow):
    """Arena order of `items` (tensors, in registration order) such that `follow[id(a)]` — the sibling tensor that must sit
    directly behind `a` (ops.ConvBnActPair views the two as one tensor) — comes right after `a`. Returns (ordered items,
    permutation: ordered[i] = items[perm[i]]). Tensors without a partner keep their relative order; a partner that is not in
    `items` (frozen parameter) or was already placed is left where it is (the pair then simply runs unfused)."""
    present = {id(t) for t in items}
    out, placed = [], set()
    for t in items:
        if id(t) in placed:
            continue
        out.append(t)
        placed.add(id(t))
        nxt = follow.get(id(t))
        if nxt is not None and id(nxt) in present and id(nxt) not in placed:
            out.append(nxt)
            placed.add(id(nxt))
    pos = {id(t): i for i, t in enumerate(items)}
    return out, [pos[id(t)] for t in out]


def build_buckets(seg, cap):
    """Gradient buckets over the arena: `seg[i] = (lo, hi)` is parameter i's float range (arena order), `cap` the bucket size in
    floats. Buckets are contiguous arena ranges filled from the LAST parameter backwards (gradients arrive in roughly reverse
    registration order, as DDP's reducer assumes); a parameter larger than `cap` is a bucket of its own. Pure arithmetic: every rank
    derives the same list from the same model. Returns [(lo, hi, first_param, last_param)], bucket 0 = the last parameters."""
    buckets = []
    hi_i = len(seg) - 1
    i = hi_i
    while i >= 0:
        lo_i = i
        while lo_i - 1 >= 0 and (seg[hi_i][1] - seg[lo_i - 1][0]) <= cap:
            lo_i -= 1
        buckets.append((seg[lo_i][0], seg[hi_i][1], lo_i, hi_i))
        i = hi_i = lo_i - 1
    return buckets


class BucketSchedule:
    """Host-side bookkeeping of the bucketed gradient exchange (the part of DistributedDataParallel's reducer that decides WHEN a
    bucket's collective is issued; trainer.py:312-313 wraps the model in DDP). Pure Python — no device, no transport: `_launch(bi)`
    is the subclass's (FlatTrainState issues the collective on its side stream; tests/test_ddp_gloo.py records / runs it over gloo).
    Invariant the tests pin: buckets are launched in INDEX order 0, 1, 2, ... on every rank — whatever order the gradients complete in
    and whichever parameters receive no gradient at all on a rank — so the collective sequence (and, under hipGraph capture, the
    side-stream fork / collective / join node sequence) is identical on all ranks and cannot deadlock on order."""

    def _init_schedule(self, buckets):
        self.buckets = list(buckets)  # [lo, hi, first_param, last_param]
        self.bucket_of = {}
        for bi, (_, _, lo_i, hi_i) in enumerate(self.buckets):
            for k in range(lo_i, hi_i + 1):
                self.bucket_of[k] = bi
        self._reset_buckets()

    def _reset_buckets(self):
        self._pending = [hi - lo + 1 for (_, _, lo, hi) in self.buckets]
        self._seen = set()
        self._uses = {}
        self._next_bucket = 0

    def note_use(self, i):
        """Forward-side count of the ops that will accumulate into parameter i's gradient slot this step: a layer applied twice
        must not release its bucket after the FIRST backward use (the second one is still accumulating into the same slot)."""
        if self.multi:
            for k in (i if isinstance(i, tuple) else (i,)):
                if k is not None:
                    self._uses[k] = self._uses.get(k, 0) + 1

    def mark_ready(self, i):
        """Called when an op has finished writing its contribution to parameter i's gradient in the arena; the parameter is
        complete once every forward use (note_use) has reported."""
        if not self.multi or self.defer_allreduce or i in self._seen:
            return
        left = self._uses.get(i, 0)
        if left > 1:
            self._uses[i] = left - 1
            return
        self._seen.add(i)
        bi = self.bucket_of[i]
        self._pending[bi] -= 1
        # buckets go out in INDEX order only (as DDP's reducer does): the collective sequence is identical on every rank
        # whatever order gradients complete in
        while self._next_bucket < len(self.buckets) and self._pending[self._next_bucket] == 0:
            self._launch(self._next_bucket)
            self._next_bucket += 1

    def _launch_rest(self):
        """end of backward: buckets still waiting for parameters that received no gradient this step (their slots are zero) go out
        now, in index order"""
        for bi in range(self._next_bucket, len(self.buckets)):
            self._pending[bi] = 0
            self._launch(bi)
        self._next_bucket = len(self.buckets)

    def _launch(self, bi):
        raise NotImplementedError


class FlatTrainState(BucketSchedule):
    def __init__(self, model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4, backbone_lr=None, ema_decay=0.9999,
                 use_ema=True, bucket_bytes=8 << 20, process_group=None, comm=None, force_collectives=False, loss_scaling=None,
                 init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, optimizer="sgd", betas=(0.9, 0.999),
                 eps=1e-8, grad_exchange="allreduce"):
        """optimizer: "sgd" (momentum / nesterov: src/optimizers/__init__.py:60-68) or "adamw" (decoupled weight decay,
        src/optimizers/__init__.py:71-73 — what conf/mini-imagenet.yml:91-99 trains config 1 with): both are ONE fused kernel over the
        flat arenas with the EMA update folded in."""
        self.model = model
        self.momentum, self.nesterov = float(momentum), bool(nesterov)
        # how a gradient bucket is summed over the ranks: "allreduce" (one ring all-reduce per bucket) or "rsag" (reduce-scatter +
        # all-gather of the same range: cvhip_comm_reduce_scatter_f32 / _all_gather_f32)
        self.grad_exchange = str(grad_exchange).lower()
        if self.grad_exchange not in ("allreduce", "rsag"):
            raise L.CvhipError("FlatTrainState: grad_exchange must be 'allreduce' or 'rsag'")
        self.bucket_bytes = int(bucket_bytes)
        self.optimizer = str(optimizer).lower()
        if self.optimizer not in ("sgd", "adamw"):
            raise L.CvhipError("FlatTrainState: optimizer must be 'sgd' or 'adamw'")
        self.betas, self.adam_eps = (float(betas[0]), float(betas[1])), float(eps)
        groups = build_param_groups(model, lr, backbone_lr, weight_decay)
        self._groups = groups   # one group per parameter, in the order the reference's optimizer enumerates them (optimizer_state_dict)
        hyper = {id(g["params"][0]): (g["lr"], g["weight_decay"]) for g in groups}
        # sibling layers that train as one convolution (ops.ConvBnActPair) need their tensors back to back: reorder so that the
        # second layer's weight / BN weight / BN bias / running statistics directly follow the first layer's
        follow = {}
        for m in model.modules():
            if hasattr(m, "hip_sibling_pairs"):
                for a, b in m.hip_sibling_pairs():
                    na, nb = getattr(a, "norm", None), getattr(b, "norm", None)
                    follow[id(a.conv.weight)] = b.conv.weight
                    if na is not None and nb is not None:
                        for name in ("weight", "bias", "running_mean", "running_var"):
                            ta, tb = getattr(na, name, None), getattr(nb, name, None)
                            if ta is not None and tb is not None:
                                follow[id(ta)] = tb

        def paired_order(items):
            return paired_arena_order(items, follow)

        self.params, perm_p = paired_order([p for p in model.parameters() if p.requires_grad])
        dev = self.params[0].device
        if dev.type != "cuda":
            raise L.CvhipError("FlatTrainState needs the model on the GPU (no CPU fallback)")
        offs, total = [], 0
        for p in self.params:
            if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
                raise L.CvhipError("parameter is neither contiguous nor channels_last")
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.offsets, self.total = offs, total
        self.param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.mom = torch.zeros(total, dtype=torch.float32, device=dev)      # SGD momentum buffer / AdamW exp_avg
        self.mom2 = torch.zeros(total, dtype=torch.float32, device=dev) if self.optimizer == "adamw" else None   # AdamW exp_avg_sq
        self.adam_step = torch.zeros(1, dtype=torch.float32, device=dev)     # AdamW step count t, advanced on the device
        self.index = {}
        for i, (p, off) in enumerate(zip(self.params, offs)):
            v = _dense_view(self.param, off, p)
            v.copy_(p.data)
            p.data = v
            p.grad = _dense_view(self.grad, off, p)
            p._hip_grad = p.grad
            p._hip_arena = (self, i)
            self.index[id(p)] = i
        seg = [[off, off + (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN] for p, off in zip(self.params, offs)]
        self.seg_bounds = torch.tensor(seg, dtype=torch.int64, device=dev)
        self.base_lr = [hyper[id(p)][0] for p in self.params]
        self.seg_lr = torch.tensor(self.base_lr, dtype=torch.float32, device=dev)
        self.seg_wd = torch.tensor([hyper[id(p)][1] for p in self.params], dtype=torch.float32, device=dev)
        self.steps = 0
        # wgrad kernels on a side stream (ops._Side): measured no gain on MI355X (YOLOv5-s 3220 -> 3134 img/s, DeepLabv3+ 394 -> 392:
        # every kernel already fills the chip), so it is opt-in: CVHIP_ASYNC_WGRAD=1
        import os
        ops.enable_async_wgrad(os.environ.get("CVHIP_ASYNC_WGRAD", "0") in ("1", "2"), after_dgrad=os.environ.get("CVHIP_ASYNC_WGRAD", "0") == "2")
        # BatchNorm step counters: one multi-tensor add per step (bricks.bn_tick) instead of one tiny kernel per layer
        nbt_mods = [m for m in model.modules()
                    if isinstance(m, torch.nn.BatchNorm2d) and m.track_running_stats and m.num_batches_tracked is not None]
        self._nbt = torch.zeros(len(nbt_mods), dtype=torch.int64, device=dev) if nbt_mods else None
        for i, m in enumerate(nbt_mods):   # the counters become views of ONE int64 buffer (state_dict keys unchanged)
            self._nbt[i] = m.num_batches_tracked
            m._buffers["num_batches_tracked"] = self._nbt[i]
            m._nbt_deferred = True
        # dynamic loss scaling (torch.cuda.amp.GradScaler, trainer.py:189-201) — on by default exactly when the engine stores
        # activations in fp16 (ops.set_precision("fp16")), as the reference pairs autocast(fp16) with GradScaler(enabled=AMP).
        # State lives on the device: {scale, growth_tracker, found_inf, skipped_steps}; ls_dyn = {1/scale or 0, skip}
        self.loss_scaling = (ops.precision() == "fp16") if loss_scaling is None else bool(loss_scaling)
        self.ls_state = torch.tensor([float(init_scale), 0.0, 0.0, 0.0], dtype=torch.float32, device=dev)
        self.ls_dyn = torch.tensor([1.0 / float(init_scale), 0.0], dtype=torch.float32, device=dev)
        self.ls_hyper = (float(growth_factor), float(backoff_factor), int(growth_interval))
        self.prep_plan = None
        self._seeds = {}
        self.lr_scale = 1.0
        self.dyn = torch.tensor([0.0, 1.0], dtype=torch.float32, device=dev)  # {ema_decay, lr_scale}, read by the kernels
        self._dyn_host = torch.zeros((256, 2), dtype=torch.float32).pin_memory()
        # floating-point buffers (BN running statistics) in their own arena
        self.bufs, perm_b = paired_order([b for b in model.buffers() if b.dtype == torch.float32])
        self.buf = torch.zeros(sum(b.numel() for b in self.bufs), dtype=torch.float32, device=dev)
        o = 0
        owners = {}
        for m in model.modules():
            for name, b in m._buffers.items():
                if b is not None and b.dtype == torch.float32:
                    owners[id(b)] = (m, name)
        for b in self.bufs:
            v = self.buf[o:o + b.numel()].view(b.shape)
            v.copy_(b)
            m, name = owners[id(b)]
            m._buffers[name] = v
            o += b.numel()
        # EMA (trainer.py:293: rank 0 only — the caller decides via use_ema)
        self.ema_model = None
        self.ema_param = self.ema_buf = None
        self.ema_decay = ema_decay
        self.ema_updates = 0
        if use_ema:
            self.ema_model = deepcopy(model).eval()
            for p in self.ema_model.parameters():
                p.requires_grad_(False)
                p.grad = None
            self.ema_param = self.param.clone()
            self.ema_buf = self.buf.clone()
            eparams = [p for p, q in zip(self.ema_model.parameters(), model.parameters()) if q.requires_grad]
            eparams = [eparams[i] for i in perm_p]  # same arena order as the live parameters
            for p, off in zip(eparams, offs):
                p.data = _dense_view(self.ema_param, off, p)
            o = 0
            eowners = {}
            for m in self.ema_model.modules():
                for name, b in m._buffers.items():
                    if b is not None and b.dtype == torch.float32:
                        eowners.setdefault(id(b), (m, name))
            ebufs = [b for b in self.ema_model.buffers() if b.dtype == torch.float32]
            for b in [ebufs[i] for i in perm_b]:
                m, name = eowners[id(b)]
                m._buffers[name] = self.ema_buf[o:o + b.numel()].view(b.shape)
                o += b.numel()
        # gradient buckets: contiguous arena ranges, filled in reverse registration order
        # transport: the RCCL communicator behind the C ABI (comm.RcclComm) in production; a torch.distributed (gloo) group
        # only as the test transport for the multi-rank bookkeeping (comm.TorchDistComm)
        self.comm = comm if comm is not None else CM.default_comm(process_group)
        self.world = self.comm.world if self.comm is not None else 1
        # `multi`: the collective machinery is live. force_collectives runs it over a 1-rank communicator too (every collective is
        # then the identity): how the captured-RCCL path is exercised on a single GPU (tests/test_gpu_comm.py)
        self.multi = self.world > 1 or (self.comm is not None and force_collectives)
        buckets = build_buckets(seg, max(1, bucket_bytes // 4))
        self._pending = None
        self._stream = None
        self._uses = {}
        # defer mode (hipGraph replay at world > 1 over a transport that cannot be captured): gradients are NOT reduced while
        # backward runs; finish_allreduce() then reduces the whole gradient arena with ONE in-place collective on the current
        # stream. With the RCCL communicator the bucket collectives are captured INSIDE the graph (side-stream branch) instead.
        self.defer_allreduce = False
        if self.multi:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._hook)
        self._init_schedule(buckets)
        # BatchNorm statistic accumulators (ops._layer_acc): fp64 [2 (forward, backward)][shards][2][K] per BN layer and one more of
        # K1 + K2 channels per sibling pair, in ONE buffer that step_kernels() zero-fills with a single launch
        want = []
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                want.append((m, "_hip_acc", m.num_features))
            if hasattr(m, "hip_sibling_pairs"):
                for a, b in m.hip_sibling_pairs():
                    na, nb = getattr(a, "norm", None), getattr(b, "norm", None)
                    if isinstance(na, torch.nn.BatchNorm2d) and isinstance(nb, torch.nn.BatchNorm2d):
                        want.append((na, "_hip_acc_pair", na.num_features + nb.num_features))
        per = 2 * L.BN_ACC_SHARDS * 2
        self.stat_acc = torch.zeros(sum(k for _, _, k in want) * per, dtype=torch.float64, device=dev)
        o = 0
        self._acc_epoch = [0]   # this state's own "accumulators zeroed" counter (ops._layer_acc)
        for m, attr, k in want:
            m.__dict__[attr] = [self.stat_acc[o:o + per * k].view(2, L.BN_ACC_SHARDS, 2, k), -1, self._acc_epoch, id(m)]
            o += per * k
        if self.world > 1:
            # DistributedDataParallel broadcasts rank 0's parameters and buffers when it wraps the model (trainer.py:312-313):
            # replicas that start from different weights would train apart silently, because only gradients are averaged
            for t in (self.param, self.mom, self.buf) + ((self.mom2,) if self.mom2 is not None else ()):
                if t.numel():
                    self.comm.broadcast_(t, 0)
            # ... and everything of the state_dict that does not live in an arena: frozen parameters, integer buffers
            # (num_batches_tracked) — DDP broadcasts the whole module state
            rest = [p.data for p in model.parameters() if not p.requires_grad] + [b for b in model.buffers() if b.dtype != torch.float32]
            if self._nbt is not None:
                rest = [t for t in rest if t.data_ptr() < self._nbt.data_ptr() or t.data_ptr() >= self._nbt.data_ptr() + 8 * self._nbt.numel()]
                rest.append(self._nbt)
            for t in rest:
                if t.numel() and t.is_cuda:
                    c = t if t.is_contiguous() else t.contiguous()
                    self.comm.broadcast_(c, 0)
                    if c is not t:
                        t.copy_(c)
            self.comm.wait()
            ops._stats_epoch[0] += 1    # running statistics were rewritten through raw pointers: cached eval-mode scale / shift are stale
            ops.bump_weights_epoch()
            if self.ema_param is not None:
                self.ema_param.copy_(self.param)
                self.ema_buf.copy_(self.buf)

    # ---- gradient readiness / all-reduce (bookkeeping: BucketSchedule) -----------------------------------------
    def _hook(self, p):
        self.mark_ready(self.index[id(p)])

    def _launch(self, bi):
        lo, hi = self.buckets[bi][0], self.buckets[bi][1]
        flat = self.grad[lo:hi]
        if self._stream is None:
            self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        if self.grad_exchange == "rsag":   # (a graph branch under capture, like the all-reduce)
            self.comm.reduce_scatter_allgather_(flat, stream=self._stream)
        else:
            self.comm.allreduce_(flat, stream=self._stream)   # cvhip_allreduce_bucket on the side stream (a graph branch under capture)

    def finish_allreduce(self):
        if not self.multi:
            return
        if self.defer_allreduce:
            if self.grad_exchange == "rsag":
                self.comm.reduce_scatter_allgather_(self.grad)
            else:
                self.comm.allreduce_(self.grad)
            self.comm.wait()
            return
        self._launch_rest()   # parameters without a gradient this step: their slots are zero
        self.comm.wait()
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    # ---- optimizer ------------------------------------------------------------------------------------------
    # ---- checkpoint of the optimizer (src/utils/checkpoints.py:39-57 stores / reloads optimizer.state_dict()) ------------------------
    def optimizer_state_dict(self):
        """The state of the fused optimizer in the layout of `torch.optim.SGD(...)` / `torch.optim.AdamW(...)`.state_dict() built over
        the reference's parameter groups (one group per parameter, src/optimizers/__init__.py:36-73): {"state": {i: {"momentum_buffer"}
        | {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [...]}. Tensors are copies in the parameters' logical (OIHW) layout, so
        the dict loads into a stock torch optimizer built the same way, and back (load_optimizer_state_dict). As torch's schedulers
        do, "lr" is the CURRENT learning rate (base x lr_scale) and "initial_lr" the unscaled one; loading prefers "initial_lr".
        NOTE: the fused kernels update EVERY arena segment each step — a parameter that received no gradient in a step
        still sees its (decoupled) weight decay and moment decay, where torch.optim skips parameters whose .grad is None; all
        parameters of the assembled models receive gradients every step, so the trajectories coincide."""
        state, pgs = {}, []
        for gi, g in enumerate(self._groups):
            p = g["params"][0]
            i = self.index[id(p)]
            off = self.offsets[i]
            if self.optimizer == "adamw":
                state[gi] = {"step": self.adam_step.detach().clone().reshape(()),
                             "exp_avg": _dense_view(self.mom, off, p).detach().clone().contiguous(),
                             "exp_avg_sq": _dense_view(self.mom2, off, p).detach().clone().contiguous()}
                pgs.append({"lr": float(self.seg_lr[i]) * self.lr_scale, "initial_lr": float(self.seg_lr[i]), "betas": self.betas, "eps": self.adam_eps, "weight_decay": float(self.seg_wd[i]),
                            "amsgrad": False, "params": [gi]})
            else:
                if self.steps > 0:   # (torch creates the buffer at the first step)
                    state[gi] = {"momentum_buffer": _dense_view(self.mom, off, p).detach().clone().contiguous()}
                pgs.append({"lr": float(self.seg_lr[i]) * self.lr_scale, "initial_lr": float(self.seg_lr[i]), "momentum": self.momentum, "dampening": 0,
                            "weight_decay": float(self.seg_wd[i]), "nesterov": self.nesterov, "params": [gi]})
        return {"state": state, "param_groups": pgs, "cvhip": {"steps": self.steps, "ema_updates": self.ema_updates, "lr_scale": self.lr_scale}}

    def load_optimizer_state_dict(self, sd):
        """Inverse of optimizer_state_dict (also accepts the state_dict of a stock torch.optim.SGD / AdamW over the same one-parameter
        groups): moments, AdamW step count, per-group lr / weight decay. At world > 1 the arenas are re-broadcast from rank 0, as after
        construction."""
        groups = sd["param_groups"]
        if len(groups) != len(self._groups):
            raise L.CvhipError("optimizer state has %d parameter groups, the model has %d" % (len(groups), len(self._groups)))
        ops.zero_fill(self.mom)
        if self.mom2 is not None:
            ops.zero_fill(self.mom2)
        step = None
        lrs, wds = self.seg_lr.tolist(), self.seg_wd.tolist()
        for gi, (g, mine) in enumerate(zip(groups, self._groups)):
            if len(g["params"]) != 1:
                raise L.CvhipError("optimizer state: one parameter per group expected (src/optimizers/__init__.py:36-56)")
            p = mine["params"][0]
            i = self.index[id(p)]
            off = self.offsets[i]
            #  a stock torch / reference checkpoint (utils/checkpoints.py:39-57) stores the SCHEDULER-ADJUSTED lr in "lr" and the
            # unscaled one in "initial_lr"; this engine keeps the unscaled lr per segment and applies lr_scale on the device. Taking a
            # mid-schedule "lr" as the base would apply the schedule factor twice at the next set_lr_scale()
            lrs[i], wds[i] = float(g.get("initial_lr", g["lr"])), float(g.get("weight_decay", wds[i]))
            ent = sd["state"].get(g["params"][0], sd["state"].get(gi))
            if not ent:
                continue
            if self.optimizer == "adamw":
                for key, arena in (("exp_avg", self.mom), ("exp_avg_sq", self.mom2)):
                    t = ent[key]
                    if tuple(t.shape) != tuple(p.shape):
                        raise L.CvhipError("optimizer state %s of group %d has shape %s, the parameter %s" % (key, gi, tuple(t.shape), tuple(p.shape)))
                    _dense_view(arena, off, p).copy_(t.to(arena.device, torch.float32))
                step = float(ent["step"]) if step is None else max(step, float(ent["step"]))
            elif "momentum_buffer" in ent and ent["momentum_buffer"] is not None:
                t = ent["momentum_buffer"]
                if tuple(t.shape) != tuple(p.shape):
                    raise L.CvhipError("momentum buffer of group %d has shape %s, the parameter %s" % (gi, tuple(t.shape), tuple(p.shape)))
                _dense_view(self.mom, off, p).copy_(t.to(self.mom.device, torch.float32))
        self.seg_lr.copy_(torch.tensor(lrs, dtype=torch.float32))
        self.seg_wd.copy_(torch.tensor(wds, dtype=torch.float32))
        self.base_lr = list(lrs)
        if step is not None:
            self.adam_step.fill_(step)
        extra = sd.get("cvhip", {})
        self.steps = int(extra.get("steps", max(self.steps, 1 if sd["state"] else 0)))
        self.ema_updates = int(extra.get("ema_updates", self.ema_updates))
        self.lr_scale = float(extra.get("lr_scale", self.lr_scale))
        if self.world > 1:
            for t in (self.mom,) + ((self.mom2, self.adam_step) if self.mom2 is not None else ()):
                self.comm.broadcast_(t, 0)
            self.comm.wait()

    def set_lr_scale(self, scale):
        """lr = base_lr * scale for every parameter (warm-up / scheduler hook: lr_schedulers/__init__.py); takes
        effect at the next step through device memory, so it also works under hipGraph replay."""
        self.lr_scale = float(scale)

    def zero_grad(self):
        ops.zero_fill(self.grad)  # a kernel, not a memset node (hipGraph-safe)

    def zero_stats(self):
        """one launch zeroes every layer's BatchNorm statistic accumulators for the next step"""
        if self.stat_acc.numel():
            ops.zero_fill(self.stat_acc)
        self._acc_epoch[0] += 1

    def backward(self, loss):
        """scaler.scale(loss).backward() with a PRE-ALLOCATED seed gradient: autograd's implicit ones_like(loss) is an ATen fill
        kernel in every (replayed) step"""
        loss = self.scale_loss(loss)
        key = (tuple(loss.shape), loss.dtype)
        seed = self._seeds.get(key)
        if seed is None:
            if torch.cuda.is_current_stream_capturing():
                ops._OWNED_BACKWARD[0] = True
                try:
                    loss.backward()
                finally:
                    ops._OWNED_BACKWARD[0] = False
                ops.check_parked()
                return None
            seed = self._seeds[key] = torch.ones_like(loss)
        ops._OWNED_BACKWARD[0] = True   # the engine drives this backward pass itself: its ops may work in place on gradients (ops.SppfChain)
        try:
            loss.backward(gradient=seed)
        finally:
            ops._OWNED_BACKWARD[0] = False
        ops.check_parked()   # a gradient parked for a consumer whose backward never ran would be dropped silently

    def scale_loss(self, loss):
        """scaler.scale(loss): the backward pass is seeded with the CURRENT loss scale, read from device memory (so a replayed
        hipGraph follows the scale as it grows / backs off). Identity when loss scaling is off."""
        return loss * self.ls_state[0] if self.loss_scaling else loss

    def loss_scale(self):
        """(scale, skipped steps so far) — host copies (synchronises; diagnostics only)."""
        v = self.ls_state.tolist()
        return v[0], int(v[3])

    def pre_step(self):
        """Host-side bookkeeping of one optimizer step + upload of the dynamic scalars {ema_decay, lr_scale}
        to device memory. Runs eagerly (outside any hipGraph) right before the step's kernels."""
        d = 0.0
        if self.ema_param is not None:
            self.ema_updates += 1
            d = self.ema_decay * (1 - math.exp(-self.ema_updates / 2000))
        slot = self._dyn_host[self.steps % self._dyn_host.shape[0]]  # ring: the async H2D below may still be pending
        slot[0], slot[1] = d, self.lr_scale
        self.dyn.copy_(slot, non_blocking=True)

    def step_kernels(self):
        """The device work of one step (capturable): wait for the bucketed all-reduce, fused SGD+EMA, EMA of the
        BN statistics, zero the gradient arena."""
        ops.join_side()  # side-stream wgrad kernels (ops._Side) must have landed in the gradient arena
        self.finish_allreduce()
        ema_ptr = self.ema_param.data_ptr() if self.ema_param is not None else None
        st = ops._stream()
        if self.loss_scaling:
            # scaler.step(optimizer) + scaler.update(): inf/NaN check of the (all-reduced) gradient arena, scale update and the
            # unscale-or-skip decision, all on the device (three launches, capturable)
            g, b, iv = self.ls_hyper
            L.call("cvhip_loss_scale_check", self.grad.data_ptr(), self.total, self.ls_state.data_ptr(), st)
            L.call("cvhip_loss_scale_update", self.ls_state.data_ptr(), self.ls_dyn.data_ptr(), g, b, iv, st)
        if self.optimizer == "adamw":
            L.call("cvhip_adamw_ema", self.param.data_ptr(), self.grad.data_ptr(), self.mom.data_ptr(), self.mom2.data_ptr(), ema_ptr, self.total,
                   self.seg_bounds.data_ptr(), self.seg_lr.data_ptr(), self.seg_wd.data_ptr(), len(self.params), self.betas[0], self.betas[1],
                   self.adam_eps, self.adam_step.data_ptr(), 0.0, 1.0 / self.world, self.dyn.data_ptr(),
                   self.ls_dyn.data_ptr() if self.loss_scaling else None, st)
        elif self.loss_scaling:
            L.call("cvhip_sgd_nesterov_ema_scaled", self.param.data_ptr(), self.grad.data_ptr(), self.mom.data_ptr(), ema_ptr, self.total,
                   self.seg_bounds.data_ptr(), self.seg_lr.data_ptr(), self.seg_wd.data_ptr(), len(self.params), self.momentum,
                   int(self.nesterov), 0, 0.0, 1.0 / self.world, self.dyn.data_ptr(), self.ls_dyn.data_ptr(), st)
        else:
            L.call("cvhip_sgd_nesterov_ema", self.param.data_ptr(), self.grad.data_ptr(), self.mom.data_ptr(), ema_ptr, self.total,
                   self.seg_bounds.data_ptr(), self.seg_lr.data_ptr(), self.seg_wd.data_ptr(), len(self.params), self.momentum,
                   int(self.nesterov), 0, 0.0, 1.0 / self.world, self.dyn.data_ptr(), st)
        if self.ema_buf is not None and self.buf.numel():
            L.call("cvhip_ema_update", self.ema_buf.data_ptr(), self.buf.data_ptr(), self.buf.numel(), 0.0, self.dyn.data_ptr(), st)
        if self._nbt is not None and self.model.training:
            L.call("cvhip_i64_add", self._nbt.data_ptr(), self._nbt.numel(), 1, st)
        self.zero_grad()
        self.zero_stats()

    def prepare_weights(self):
        """Batched re-pack of every conv layer's bf16 operand images (ops.PrepPlan: one launch instead of ~2 per layer). The plan
        is built once the layers have been through one ordinary step (their image buffers exist at fixed addresses); weights
        whose version counter moved (load_state_dict, manual edits) simply fall back to the per-layer path."""
        if not _PREP_PLAN:
            return
        plan = self.prep_plan
        if plan is None or not plan.valid():
            states = [s for s in ops.conv_states_of(self.model) if s.w_fprop is not None and s.rec is not None]
            if not states:
                return  # first step: nothing prepared yet (layers that never ran keep the per-layer path)
            if torch.cuda.is_current_stream_capturing():
                return  # never build (host->device copy) inside a capture
            plan = self.prep_plan = ops.PrepPlan(states)
        plan.run()

    def post_step(self):
        self.steps += 1
        ops.bump_weights_epoch()  # parameters changed behind torch's version counters
        self._reset_buckets()

    def step(self):
        self.pre_step()
        self.step_kernels()
        self.post_step()


class FlatTrainStep:
    """forward -> loss -> backward (grads land in the arena, buckets all-reduce as they fill) -> fused SGD+EMA.

    `capture(imgs, targets)` records the step into TWO hipGraphs over static buffers:
        G1 = model.forward_features(imgs)                       (backbone / neck / head: ~700 kernels)
        -- eager: loss on the head outputs and its gradient w.r.t. them (small torch ops, runs while G1 executes) --
        G2 = backward from the head outputs + fused optimizer   (~1200 kernels)
    so the ~2000 launches of a YOLOv5-s step cost no host time. libcvhip never allocates, synchronises or issues
    memset/memcpy calls, so both graphs contain kernel nodes only (hipMemset nodes proved unreliable under replay on
    ROCm 7.2, and capturing the torch-op loss backward crashed the runtime — hence the eager island).

    The model must provide `forward_features(imgs) -> (aux, [tensors])` and
    `loss_from_features([tensors], targets) -> dict with 'loss'`.
    """

    def __init__(self, model, state, sync_buffers=False):
        self.model, self.state, self.sync_buffers = model, state, sync_buffers
        self.g1 = self.g2 = None
        self.static_losses = None
        self.eager_tail = False
        self.static_imgs = self.static_targets = None
        self.feats = self.g_feats = None

    @property
    def graph(self):
        return self.g1

    @graph.setter
    def graph(self, v):
        if v is None:
            self.g1 = self.g2 = None
            self.state.defer_allreduce = False
            self.eager_tail = False

    def _eager(self, imgs, targets):
        self.state.prepare_weights()
        losses = self.model(imgs, targets, "train")
        self.state.backward(losses["loss"])
        self.state.step_kernels()
        return losses

    def capture(self, imgs, targets, warmup=2):
        from .bricks import sync_of
        multi = self.state.multi
        # RCCL collectives are captured with the kernels (bucket all-reduces become a parallel branch of the graph, SyncBN
        # exchanges nodes of the main chain); over the test transport they stay outside: backward is replayed, then ONE
        # all-reduce of the arena + the optimizer run eagerly
        capturable = multi and self.state.comm.capturable
        if not capturable and any(sync_of(mm) is not None for mm in self.model.modules()):
            raise L.CvhipError("the step contains active HipSyncBN layers over a transport that cannot be captured: run it eagerly")
        self.state.defer_allreduce = multi and not capturable
        self.eager_tail = multi and not capturable
        if not torch.is_tensor(targets):
            raise L.CvhipError("capture needs the fixed-shape target tensor (e.g. yolov5.targets_to_tensor)")
        m, st = self.model, self.state
        self.static_imgs, self.static_targets = imgs.clone(), targets.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                st.pre_step()
                self._eager(self.static_imgs, self.static_targets)
                st.post_step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        pool = torch.cuda.graph_pool_handle()
        if getattr(m, "loss_capturable", False):
            # the loss is libcvhip kernels too (no torch autograd ops): the whole step is ONE graph
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                st.prepare_weights()
                _, feats = m.forward_features(self.static_imgs)
                losses = m.loss_from_features(feats, self.static_targets)
                st.backward(losses["loss"])
                ops.join_side()
                if not self.eager_tail:
                    st.step_kernels()
            self.static_losses = losses
            self.g1, self.g2 = g, None
            return
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, pool=pool, capture_error_mode="thread_local"):
            st.prepare_weights()
            _, feats = m.forward_features(self.static_imgs)
        self.feats = list(feats)
        self.g_feats = [torch.zeros_like(f) for f in self.feats]
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, pool=pool, capture_error_mode="thread_local"):
            torch.autograd.backward(self.feats, grad_tensors=self.g_feats)
            ops.join_side()
            if not self.eager_tail:
                st.step_kernels()
        self.g1, self.g2 = g1, g2

    def __call__(self, imgs, targets):
        st = self.state
        if self.g1 is not None:
            if imgs is not self.static_imgs:
                self.static_imgs.copy_(imgs, non_blocking=True)
            if targets is not self.static_targets:
                self.static_targets.copy_(targets, non_blocking=True)
            if self.sync_buffers and st.multi and st.buf.numel():
                st.comm.broadcast_(st.buf, 0)  # DDP broadcast_buffers: ONE collective
            st.pre_step()
            self.g1.replay()
            if self.g2 is None:  # single-graph step
                if self.eager_tail:
                    st.step_kernels()
                st.post_step()
                return self.static_losses
            p = [f.detach().requires_grad_(True) for f in self.feats]
            losses = self.model.loss_from_features(p, self.static_targets)
            grads = torch.autograd.grad(st.scale_loss(losses["loss"]), p)
            for d, g in zip(self.g_feats, grads):
                d.copy_(g)
            self.g2.replay()
            if self.eager_tail:
                st.step_kernels()
            st.post_step()
            return losses
        if self.sync_buffers and st.multi and st.buf.numel():
            st.comm.broadcast_(st.buf, 0)  # DDP broadcast_buffers: ONE collective
        st.pre_step()
        losses = self._eager(imgs, targets)
        st.post_step()
        return losses
