"""torch.autograd.Function wrappers over the libcvhip C ABI (include/cvhip.h).

Tensor convention ("NHWC view"): logical shape (N, C, H, W), dtype bf16, device cuda, element
(n, c, h, w) at offset ((n*H + h)*W + w)*ld + c with ld >= C. A plain channels_last tensor has
ld == C; a channel slice of a wider concat buffer has ld == C_total. Every kernel takes the pitch
explicitly, so slices are first-class operands and `torch.cat` never has to materialise.

There is deliberately no CPU / eager fallback in this module: an op that cannot run on the HIP
engine raises (lib.CvhipError).

Reference call sites replaced (file:line in the reference tree):
  conv+BN+act  : src/models/bricks/conv_module.py:201-214
  max-pool     : src/models/modules/yolo_modules.py:176-192
  up x2 + cat  : src/models/modules/yolo_modules.py:147,152
  cat / add    : src/models/modules/yolo_modules.py:102,139,162,190
  head permute : src/models/detects/yolov5_detect.py:43-44
"""
import ctypes as C

import torch

from . import lib as L

ACT_DTYPE = torch.bfloat16  # storage dtype of activations / operand images / activation gradients (set_precision)
_weights_epoch = 0  # bumped by the fused optimizer (it updates parameters behind torch's back)
_stats_epoch = [0]  # bumped by every training-mode BatchNorm forward (running statistics are updated by kernels, not torch ops)


def set_precision(p):
    """Storage precision of the engine: "bf16" (default) or "fp16" (reference: torch.cuda.amp.autocast fp16, trainer.py:179; pair it
    with the dynamic loss scaling of arena.FlatTrainState(loss_scaling=True) as the reference pairs autocast with GradScaler).
    Process-wide; cached operand images are dropped (they are re-packed in the new precision on the next forward)."""
    global ACT_DTYPE
    L.set_precision(p)
    ACT_DTYPE = torch.float16 if p == "fp16" else torch.bfloat16
    bump_weights_epoch()
    _desc_cache.clear()


def precision():
    return L.PRECISION


def bump_weights_epoch():
    global _weights_epoch
    _weights_epoch += 1


# ---- BatchNorm statistic accumulators (include/cvhip.h CVHIP_BN_ACC_SHARDS) -------------------------------------------------------
# Training-mode BN layers fold their batch sums into an fp64 accumulator with atomics (conv epilogue / backward reduction) and the
# consuming pass derives the per-channel constants in its prologue: no partial rows, no finalize launches (~115 five-microsecond
# kernels per YOLOv5-s step). A layer's accumulator pair [2 (forward, backward)][shards][2][K] lives in the flat training state
# (arena.FlatTrainState: ONE zero-fill per step for all layers); without it — or when a layer runs a second time inside one step
# (yolov7 FeatureFusion.conv4) — a scratch accumulator is zero-filled per call. CVHIP_BN_ACC=0 restores the partial-row path.
# Inference (no operand requires a gradient, BatchNorm in eval mode or folded away): the whole ConvModule — conv, bias, BN scale/shift,
# activation, residual — is ONE cvhip_conv2d_fprop_fused launch (conv_module.py:201-214 in eval mode; utils/fuse.py:32-54).
# CVHIP_STEM_IMAGE=0: image stems get their input through the explicit fp32 NCHW -> 16-bit NHWC pass again (A/B switch).
_STEM_IMAGE = __import__("os").environ.get("CVHIP_STEM_IMAGE", "1") != "0"
# CVHIP_EPI_FUSE=0 restores conv + a separate BN/activation pass (A/B switch).
_EPI_FUSE = __import__("os").environ.get("CVHIP_EPI_FUSE", "1") != "0"
_BN_ACC = __import__("os").environ.get("CVHIP_BN_ACC", "1") != "0"
_DW_STATS = __import__("os").environ.get("CVHIP_DW_STATS", "1") != "0"   # depthwise 3x3 forward emits its BatchNorm sums (A/B: 0 = reduction pass)
_BN_ACC_MAX_C = 2048
_acc_epoch = 0


# ---- deterministic mode --------------------------------------------------------------------------------------------------------------
# set_deterministic(True): every floating-point reduction of the train step runs in a FIXED order — dense weight gradients through
# cvhip_conv2d_wgrad_det (per-split slabs + ordered fold instead of fp32 atomics), BatchNorm statistics through the partial-row +
# finalize kernels instead of the fp64 accumulators, no fused 1x1 backward and no stem patch-wgrad (their dW flushes are atomic).
# Two runs from the same state then give bit-identical gradient arenas (tests/test_gpu_deterministic.py). Costs one fold launch per
# layer and the finalize launches back: opt-in, like torch.use_deterministic_algorithms. (Depthwise weight gradients keep their
# atomic epilogue.)
_DETERMINISTIC = __import__("os").environ.get("CVHIP_DETERMINISTIC", "0") == "1"


def set_deterministic(flag=True):
    global _DETERMINISTIC
    _DETERMINISTIC = bool(flag)


def is_deterministic():
    return _DETERMINISTIC


def _wgrad(kname, geom, desc, x, dy, dst, accumulate, st):
    """dense weight gradient into `dst` (fp32 KRSC): the atomic split-K kernel, or its deterministic two-stage form"""
    if _DETERMINISTIC:
        nb = L.load().cvhip_conv2d_wgrad_det_workspace_bytes(C.byref(desc))
        if nb < 0:
            L.check(int(nb), "cvhip_conv2d_wgrad_det_workspace_bytes")
        ws = torch.empty((max(int(nb), 16),), dtype=torch.uint8, device=x.device)
        _timed_call(kname, geom, "cvhip_conv2d_wgrad_det", C.byref(desc), x.data_ptr(), dy.data_ptr(), dst.data_ptr(), int(accumulate),
                    ws.data_ptr(), int(nb), st)
    else:
        _timed_call(kname, geom, "cvhip_conv2d_wgrad", C.byref(desc), x.data_ptr(), dy.data_ptr(), dst.data_ptr(), int(accumulate), st)


def bump_acc_epoch():
    """kept for callers that zero accumulators themselves (process-wide epoch of accumulators WITHOUT an owning state)"""
    global _acc_epoch
    _acc_epoch += 1


def _layer_acc(cfg, K, dev):
    """(forward, backward) accumulators [shards][2][K] fp64, zeroed, for ONE application of the layer behind `cfg`.

    A layer's persistent accumulator entry is [tensor (2, shards, 2, K), epoch of its last use, epoch cell of the OWNING state,
    id of the module it was made for]. It is clean exactly when the owning FlatTrainState has zeroed its accumulator arena since
    the entry's last use (state.zero_stats bumps the state's own cell: another state's step must not make this one look clean —
    teacher / student, GAN, back-to-back tests), and it belongs to THIS module (a deep copy carries a copy of the list whose tensor
    no state ever zeroes). Everything else gets a scratch accumulator that is zero-filled per call."""
    owner = cfg.acc_owner
    if owner is not None:
        ent = owner.__dict__.get(cfg.acc_attr)
        if ent is not None and len(ent) >= 4 and ent[3] == id(owner) and ent[1] != ent[2][0] and ent[0].shape[-1] == K and ent[0].device == dev:
            ent[1] = ent[2][0]
            return ent[0][0], ent[0][1]
    t = zero_fill(torch.empty((2, L.BN_ACC_SHARDS, 2, K), dtype=torch.float64, device=dev))
    return t[0], t[1]


def _stream():
    return torch.cuda.current_stream().cuda_stream


# ---- lazy activations (round 5) ---------------------------------------------------------------------------------------------------
# `ConvModule.forward` is act(norm(conv(x))) (conv_module.py:201-214). In training the BatchNorm statistics need the whole batch, so
# the apply pass z = act(scale*y + shift) cannot ride in the convolution's own epilogue — but it can ride in the CONSUMER's load: a
# layer asked for a lazy result (ConvCfg.lazy_out) only finalizes its statistics (one tiny launch) and hands out its RAW convolution
# output y tagged with a LazyAct; a consumer that reads every input element exactly once through registers — the streaming 1x1
# kernel (conv1x1_stream.hip PRO), the fused 1x1 backward (conv1x1_bwd.hip XPRO: the weight gradient needs the activated input), the
# residual operand of a BN+act pass (ew_kernel RLZ) — applies the transform on load, bit-identically to the stand-alone pass. The
# activated tensor then never exists in HBM (one read + one write of the tensor less per edge). Every other consumer (3x3 implicit
# GEMMs stage their operand by LDS-DMA and re-read each element per tap: an on-load SiLU there is VALU-infeasible)
# calls `materialize`, which runs exactly the pass the producer skipped. CVHIP_LAZY=0 switches the whole mechanism off.
_LAZY = __import__("os").environ.get("CVHIP_LAZY", "1") != "0"
_LAZY_ACTS = (L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY, L.ACT_SILU)


# True while arena.FlatTrainState.backward runs: the engine owns every gradient tensor between its ops, so backward passes may work in
# place on the gradients they receive (SppfChain). Anything else (plain loss.backward(), hooks, retain_grad) gets copies.
_OWNED_BACKWARD = [False]


def set_lazy(flag=True):
    global _LAZY
    _LAZY = bool(flag)


class LazyAct:
    """tag of a tensor that holds the RAW output y of a training-mode Conv-BN-act layer: its logical value is act(scale*y + shift) —
    on the channel range [lo, hi) of the tagged tensor (a concat buffer whose other slices were materialised: lo > 0 or hi < C);
    `scale` / `shift` hold hi - lo values (channel lo first)"""
    __slots__ = ("scale", "shift", "act", "ap", "z", "lo", "hi")

    def __init__(self, scale, shift, act, ap, lo=0, hi=None):
        self.scale, self.shift, self.act, self.ap = scale, shift, int(act), float(ap)
        self.lo = int(lo)
        self.hi = int(hi) if hi is not None else self.lo + int(scale.numel())
        self.z = None   # the materialised tensor, once some consumer needed it

    def full(self, Cc):
        return self.lo == 0 and self.hi == Cc

    def scale_ptr(self):
        """base pointers indexed by the ABSOLUTE channel of the tagged tensor (only [lo, hi) is ever dereferenced)"""
        return self.scale.data_ptr() - 4 * self.lo

    def shift_ptr(self):
        return self.shift.data_ptr() - 4 * self.lo


def act_id_of(conv_module):
    """activation id of a Hip ConvModule (L.ACT_*; -1 when it is not one of the engine's fused activations)"""
    fus = conv_module._fusable(True, True) if hasattr(conv_module, "_fusable") else None
    return fus[1][0] if fus is not None else -1


def lazy_of(t):
    return getattr(t, "_hip_lazy", None) if t is not None else None


# names of the torch.Tensor methods / property getters that do NOT read a tensor's elements: metadata the engine's own ops (and
# autograd) ask a lazy tensor for. Everything else is a stock torch op about to read RAW convolution outputs.
_LAZY_META = frozenset((
    "size", "stride", "dim", "ndimension", "numel", "nelement", "data_ptr", "storage_offset", "is_contiguous", "element_size", "__len__",
    "shape", "device", "dtype", "is_cuda", "requires_grad", "grad_fn", "is_leaf", "layout", "names", "ndim", "_version", "_base", "grad",
    "is_floating_point", "is_complex", "get_device", "is_sparse", "is_quantized", "is_meta", "untyped_storage", "_is_view", "output_nr",
    "register_hook", "retain_grad", "retains_grad", "__hash__", "__repr__", "__format__", "_backward_hooks", "is_pinned", "is_shared"))


class LazyRaw(torch.Tensor):
    """The tensor type of a LAZY activation: the storage holds the RAW output y of a training-mode Conv-BN-act layer
    (conv_module.py:201-214 with the norm / activation deferred into the consumers' loads), the tag `_hip_lazy` says how to read it.
    The engine's ops (autograd Functions, `as_nhwc(..., lazy_ok=True)`) see the raw storage; ANY stock torch function — a view, a
    slice, `.float()`, `detach`, an `nn.Conv2d` that `convert_to_hip` left in place, a forward hook's arithmetic, torch.utils.checkpoint —
    is given the ACTIVATED tensor instead (`ops.materialize`: the pass the producer skipped, computed once and shared), so a consumer
    that does not know about the tag can never read un-normalised, un-activated data."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        if name == "__get__" or name == "__set__":   # property access: func.__self__ is the descriptor
            name = getattr(getattr(func, "__self__", None), "__name__", name)
        if name in _LAZY_META:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        from torch.utils._pytree import tree_map

        def fix(a):
            if isinstance(a, LazyRaw):
                with torch._C.DisableTorchFunctionSubclass():
                    return materialize(a) if lazy_of(a) is not None else a.as_subclass(torch.Tensor)
            return a

        args, kwargs = tree_map(fix, args), tree_map(fix, kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)


def tag_lazy(t, lz):
    """`t` as a LazyRaw carrying the tag (same storage, same autograd node); other engine tags on the tensor object move along"""
    with torch._C.DisableTorchFunctionSubclass():
        z = t.as_subclass(LazyRaw)
    for k in ("_hip_prod",):
        v = getattr(t, k, None)
        if v is not None:
            setattr(z, k, v)
    z._hip_lazy = lz
    return z


class Materialize(torch.autograd.Function):
    """z = act(scale*y + shift): the BN-apply + activation pass a lazy producer skipped, for a consumer that cannot transform on load"""

    @staticmethod
    def forward(ctx, y, lz):
        yy, y_ld = as_nhwc(y, lazy_ok=True)
        N, K, P, Q = yy.shape
        z = empty_nhwc(N, K, P, Q, yy.device)
        _materialize_into(yy, y_ld, z, K, lz)
        return z

    @staticmethod
    def backward(ctx, dz):
        return dz, None


def _materialize_into(x, x_ld, z, z_ld, lz):
    """z = the activated form of the lazy tensor x (N, C, H, W): the pass its producer skipped on [lo, hi), a copy elsewhere"""
    N, Cc, H, W = x.shape
    M = N * H * W
    st = _stream()
    if not lz.full(Cc):
        L.call("cvhip_copy2d", x.data_ptr(), x_ld, z.data_ptr(), z_ld, M, Cc, st)
    kh = lz.hi - lz.lo
    _timed_ew("bn_act_fwd(ew_kernel<0>)", 4.0 * M * kh, "cvhip_bn_act_fwd", x.data_ptr() + 2 * lz.lo, x_ld, z.data_ptr() + 2 * lz.lo, z_ld, M, kh,
              lz.scale.data_ptr(), lz.shift.data_ptr(), lz.act, lz.ap, None, 0, st)


def materialize(x):
    """the activated tensor behind a lazy one (computed once, shared by all consumers that need it); any other tensor unchanged"""
    lz = lazy_of(x)
    if lz is None:
        return x
    if lz.z is None:
        lz.z = Materialize.apply(x, lz)
    return lz.z


def _lazy_in(lz):
    """cvhip_lazy_in of a lazy operand (its channel range [lo, hi), constants indexed by the absolute channel)"""
    return L.LazyIn(lz.scale_ptr(), lz.shift_ptr(), lz.act, lz.ap, lz.lo, lz.hi)


# ---- producer records: BN-backward sums from the kernel that writes the gradient ------------------------------------------------------
# The gradient dz at the output of a Conv-BN-act layer P is, for most layers, written by exactly one kernel: the dgrad (or fused
# 1x1 backward) of the layer that consumed P's output. That kernel can fold P's BatchNorm-backward sums (sum du, sum du*xhat) into
# P's accumulator in its epilogue (include/cvhip.h cvhip_bn_tail) — P's backward then skips its reduction pass over (dz, y).
# Plumbing: P's forward hangs a ProdInfo on its output tensor; the consuming op picks it up in its forward, hands the tail to its
# dgrad in backward and remembers WHICH tensor the sums were taken over; P's backward uses them only if the gradient it receives is
# that very tensor (autograd summed nothing else into it) — otherwise it clears the accumulator and reduces as before.
# MEASURED AND LEFT OFF (CVHIP_BN_TAIL, default 0; profiles/r03_bn_tail_ab.log): every form is a net loss on YOLOv5-s — implicit-GEMM
# dgrad tail +0.2 ms/step, fused 1x1 backward tail +0.28, streaming dgrad tail +0.03 (32 reduction launches deleted, but the epilogues'
# extra y read in fragment layout and a second SiLU' per element cost more than the 25-30 us reductions they replace).
_BN_TAIL_MASK = int(__import__("os").environ.get("CVHIP_BN_TAIL", "0"))   # bit 0: implicit-GEMM dgrad, 1: streaming 1x1 dgrad, 2: fused 1x1 backward
_BN_TAIL = _BN_TAIL_MASK != 0
_TAIL_ACTS = (L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY, L.ACT_SILU)


class ProdInfo:
    __slots__ = ("y", "y_ld", "stats", "act", "ap", "acc", "acc_ld", "c_off", "kh", "shape", "consumers", "fused_dx", "done", "parent", "halves")

    def __init__(self, y, y_ld, stats, act, ap, acc, acc_ld, c_off, kh, shape, parent=None):
        self.y, self.y_ld, self.stats, self.act, self.ap, self.acc, self.acc_ld = y, y_ld, stats, act, ap, acc, acc_ld
        self.c_off, self.kh, self.shape = c_off, kh, tuple(shape)
        self.consumers = 0      # Hip conv ops that took the tensor as their input in forward
        self.fused_dx = None    # the gradient tensor whose rows the sums were taken over (kept alive until P's backward compares)
        self.done = False
        self.parent = parent
        self.halves = None      # sibling pairs: the records of the two output tensors

    def half(self, off, kh):
        return ProdInfo(self.y, self.y_ld, self.stats, self.act, self.ap, self.acc, self.acc_ld, self.c_off + off, kh,
                        (self.shape[0], kh, self.shape[2], self.shape[3]), parent=self)

    def tail(self):
        o4 = 4 * self.c_off
        st = self.stats
        return L.BnTail(self.y.data_ptr() + 2 * self.c_off, self.y_ld, st[2].data_ptr() + o4, st[3].data_ptr() + o4, st[0].data_ptr() + o4,
                        st[1].data_ptr() + o4, self.act, self.ap, self.acc.data_ptr() + 8 * self.c_off, self.acc_ld)


def _take_prod(x):
    """the producer record of an op's input tensor (None unless a training-mode Conv-BN-act layer of this engine made it)"""
    if not _BN_TAIL or not torch.is_grad_enabled():
        return None
    pi = getattr(x, "_hip_prod", None)
    if pi is not None:
        pi.consumers += 1
    return pi


def _tail_for(pi, N, Cc, H, W):
    """cvhip_bn_tail for the dgrad that is about to write dx (N, Cc, H, W dense) of a tensor made by `pi`, or None"""
    if pi is None or pi.done or pi.consumers != 1 or pi.kh != Cc or pi.shape != (N, Cc, H, W):
        return None
    if Cc % 8 or pi.c_off % 8 or pi.y_ld % 8 or (pi.y.data_ptr() + 2 * pi.c_off) % 16:
        return None
    return pi.tail()


def _sums_already_done(pi, dz):
    """True when the BN-backward sums of this layer (half) were folded by the kernel that wrote `dz`; if sums were folded for some
    OTHER tensor (autograd accumulated several gradients) the caller must clear the accumulator and reduce itself -> 'dirty'"""
    if pi is None or not pi.done:
        return False
    fd, pi.fused_dx = pi.fused_dx, None
    if fd is not None and fd.data_ptr() == dz.data_ptr() and tuple(fd.shape) == tuple(dz.shape) and fd.stride() == dz.stride():
        return True
    return "dirty"


class KernelTimer:
    """Optional per-launch HIP-event timing of the conv kernels (used by bench.py's roofline leg).
    Events are recorded on torch's current stream — the stream every libcvhip launch goes to."""

    def __init__(self):
        self.enabled = False
        self.records = []  # (kernel name, algorithmic flops, algorithmic bytes, ev0, ev1)
        self.detail = []   # (entry point, geometry) per record

    def reset(self):
        self.records = []
        self.detail = []

    def summary(self):
        out = {}
        for name, fl, by, e0, e1 in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


TIMER = KernelTimer()


def _igemm_name(nout, m=0, ktot=0, pointwise=False, ld=0, stats=False):
    """label of the kernel / tile configuration launch_igemm picks (conv_igemm.hip: igemm_block_m; conv1x1_stream.hip for
    1x1 stride-1 unpadded passes) for `nout` output channels, m rows"""
    if pointwise and ld % 8 == 0 and L.load().cvhip_conv1x1_stream_blocks(nout, ktot, m, int(stats)) > 0:
        wide = nout > 128 and not stats and 256 * (((ktot + 31) // 32 * 32) * 2 + 16) <= 72 * 1024 and nout <= 256
        return "conv1x1_stream_kernel<%d>" % (32 if nout <= 32 else 64 if nout <= 64 else 256 if wide else 128)
    if nout <= 32:
        return "igemm_kernel<256,32,64,32>"
    if nout <= 64:
        return "igemm_kernel<256,64,64,64>"
    big = ktot >= 512 and ((m + 255) // 256) * ((nout + 127) // 128) >= 384
    return "igemm_kernel<256,128,128,64>" if big else "igemm_kernel<128,128,64,64>"


def _pointwise(R, S, cfg):
    return R == 1 and S == 1 and tuple(cfg.stride) == (1, 1) and tuple(cfg.pad) == (0, 0)


def _wgrad_name(k, r=3, s=3, c=0, m=0, desc=None):
    """label of the kernel / tile configuration launch_wgrad picks (conv_wgrad.hip; conv_stem.hip for image stems)"""
    if desc is not None and L.load().cvhip_conv_stem_blocks(C.byref(desc)) > 0 and desc.y_ld % 8 == 0:
        return "stem_wgrad_kernel"
    tn = 32 if k <= 32 else 64 if k <= 64 else 128
    if r == 1 and s == 1 and float(m) * k * c <= 7.5e9:
        if k >= 256 and c >= 256:   # launcher policy 4 (conv_wgrad.hip): the 128-wide tile as one two-group block per CU — its own kernel
            return "wgrad_kernel<128,64,64,2>"   # instance in rocprof (wgrad_kernel<128, 64, 64, 2, 3, 0>), so its own label here
        tn = 32
    return {32: "wgrad_kernel<32,32,32>", 64: "wgrad_kernel<64,32,64>", 128: "wgrad_kernel<128,64,64>"}[tn]


def _timed_call(kname, geom, fname, *args, passes=1, nbytes=None):
    """geom = (N, C, H, W, K, R, S, P, Q): algorithmic work of one conv pass = 2*M*K*R*S*C flop and
    one read of the gathered operand + one write of the result + the weights (bf16). `passes` > 1 / `nbytes`: fused kernels
    that do several passes' work in one launch state their own algorithmic totals."""
    if not TIMER.enabled:
        L.call(fname, *args)
        return
    N, Cc, H, W, K, R, S, P, Q = geom
    flops = 2.0 * N * P * Q * K * R * S * Cc * passes
    if nbytes is None:
        nbytes = 2.0 * (N * H * W * Cc + N * P * Q * K + K * R * S * Cc)
    if kname.startswith("igemm_kernel") and (fname.startswith("cvhip_conv2d_fprop") or fname.startswith("cvhip_conv2d_dgrad")):
        # launch_igemm hands some multi-tap problems to the patch-resident kernel (conv_patch.hip): label those launches as what runs,
        # so that a row of the bench line's kernel table is ONE device kernel family (as in rocprofv3's per-kernel statistics)
        try:
            dg = 1 if fname.startswith("cvhip_conv2d_dgrad") else 0
            bb = (C.c_int32 * L.BAND_PLAN_INTS)()
            buf = (C.c_int32 * (L.PATCH_CLASS_INTS * 4))()
            if L.load().cvhip_conv2d_band_plan(args[0], dg, bb) > 0:      # launch_igemm's order: band, patch, per-tap
                kname = "conv_band_kernel<%d ch/wave>" % (16 * bb[0])
            elif L.load().cvhip_conv2d_patch_plan(args[0], dg, buf, 4) > 0:
                kname = "conv_patch_kernel<128>"
        except Exception:
            pass
    elif kname.startswith("wgrad_kernel") and fname == "cvhip_conv2d_wgrad":
        # stride-1 3x3 "same" layers run on the tap-resident weight-gradient kernel (conv_wgrad_band.hip) where its plan fits
        try:
            wb = (C.c_int32 * L.WGRAD_BAND_PLAN_INTS)()
            if L.load().cvhip_conv2d_wgrad_band_plan(args[0], wb) > 0:
                kname = "wgrad_band_kernel<%d>" % wb[0]
        except Exception:
            pass
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.call(fname, *args)
    e1.record()
    TIMER.records.append((kname, flops, nbytes, e0, e1))
    TIMER.detail.append((fname, geom))



def _timed_ew(kname, nbytes, fname, *args):
    """Elementwise / reduction passes (BN + activation forward, BN-backward sums, BN-backward apply) under the same HIP-event
    timing as the conv launches: algorithmic bytes = every operand read or written once (bf16), no flops credited."""
    if not TIMER.enabled:
        L.call(fname, *args)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.call(fname, *args)
    e1.record()
    TIMER.records.append((kname, 0.0, float(nbytes), e0, e1))
    TIMER.detail.append((fname, None))


def _ptr(t, off=0):
    return None if t is None else t.data_ptr() + off


def empty_nhwc(N, Cc, H, W, device, ld=None):
    """Fresh NHWC-view tensor of logical shape (N,C,H,W); ld > C gives a padded pitch."""
    if ld is None or ld == Cc:
        return torch.empty((N, Cc, H, W), dtype=ACT_DTYPE, device=device, memory_format=torch.channels_last)
    buf = torch.empty((N, H, W, ld), dtype=ACT_DTYPE, device=device)
    return buf.permute(0, 3, 1, 2)[:, :Cc]


def nhwc_ld(t):
    """Return the pixel pitch of an NHWC-view tensor, or None if `t` is not one."""
    if t.dim() != 4 or t.dtype != ACT_DTYPE:
        return None
    N, Cc, H, W = t.shape
    sN, sC, sH, sW = t.stride()
    if Cc > 1 and sC != 1:
        return None
    ld = sW if W > 1 else (sH if H > 1 else (sN if N > 1 else max(Cc, 1)))
    if W > 1 and sW != ld:
        return None
    if H > 1 and sH != W * ld:
        return None
    if N > 1 and sN != H * W * ld:
        return None
    if ld < Cc:
        return None
    return ld


def as_nhwc(t, lazy_ok=False):
    """NHWC view of `t` (no copy when it already is one; otherwise one channels_last relayout)."""
    if not t.is_cuda:
        raise L.CvhipError("cvpytorch_amd ops need CUDA/HIP tensors (no CPU fallback); got %s" % t.device)
    if not lazy_ok and getattr(t, "_hip_lazy", None) is not None:
        # a lazy tensor holds RAW convolution outputs: only ops that apply its transform on load may read it
        raise L.CvhipError("a lazy activation reached an op that does not transform on load (ops.materialize it first)")
    if t.dtype != ACT_DTYPE:
        t = t.to(ACT_DTYPE)
    ld = nhwc_ld(t)
    if ld is None:
        t = t.contiguous(memory_format=torch.channels_last)
        ld = nhwc_ld(t)
    return t, ld


_desc_cache = {}


def conv_desc(N, Cc, H, W, K, R, S, stride, pad, dil, groups, x_ld, y_ld, k_valid=0, c_valid=0):
    key = (N, Cc, H, W, K, R, S, stride, pad, dil, groups, x_ld, y_ld, k_valid, c_valid)
    d = _desc_cache.get(key)
    if d is None:
        d = L.ConvDesc(N, Cc, H, W, K, R, S, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], groups, x_ld, y_ld,
                       k_valid, c_valid)
        _desc_cache[key] = d
    return d


def conv_out_hw(H, W, R, S, stride, pad, dil):
    P = (H + 2 * pad[0] - dil[0] * (R - 1) - 1) // stride[0] + 1
    Q = (W + 2 * pad[1] - dil[1] * (S - 1) - 1) // stride[1] + 1
    return P, Q


def _round8(x):
    return (x + 7) // 8 * 8


def zero_fill(t):
    """Zero a dense tensor with a libcvhip kernel (no hipMemset: see include/cvhip.h cvhip_zero_fill)."""
    L.call("cvhip_zero_fill", t.data_ptr(), t.numel() * t.element_size(), _stream())
    return t


def _krsc_master(weight):
    """fp32 master weight in KRSC physical order. HipConv2d keeps its parameter that way (OIHW shape,
    channels_last strides) so this is normally a no-op view; other layouts cost one relayout kernel."""
    w = weight.detach()
    if w.dtype != torch.float32:
        w = w.float()
    wk = w.permute(0, 2, 3, 1)
    return wk if wk.is_contiguous() else wk.contiguous()


# ---- weight-gradient kernels on a side stream -------------------------------------------------------------------------------
# wgrad of a layer depends only on (x, dy) of that layer and nothing downstream depends on it until the optimizer, while the
# dgrad -> BN-backward chain is the critical path of backward. Issuing wgrad on a second stream lets it fill the CUs the
# critical path leaves idle (tile-quantisation tails, low-occupancy small layers). Enabled by arena.FlatTrainState (the
# gradients land in the arena, nothing on the main stream consumes them before `join_side()`); operands are kept alive until
# the join so the caching allocator cannot hand their memory to a later main-stream kernel. Under hipGraph capture the
# fork/join become parallel branches of the graph.
class _Side:
    enabled = False
    after_dgrad = False
    stream = None
    keep = []
    dirty = False
    # only layers with at most this many output pixel rows go to the side stream (CVHIP_ASYNC_WGRAD_MAXM: the small late layers leave
    # CUs idle, the large early ones fill the chip by themselves)
    max_rows = int(__import__("os").environ.get("CVHIP_ASYNC_WGRAD_MAXM", str(1 << 62)))


def enable_async_wgrad(flag=True, after_dgrad=False):
    _Side.enabled = bool(flag)
    _Side.after_dgrad = bool(after_dgrad)


def _side_begin():
    if _Side.stream is None:
        _Side.stream = torch.cuda.Stream()
    _Side.stream.wait_stream(torch.cuda.current_stream())
    _Side.dirty = True
    return _Side.stream


def join_side():
    """Make the current stream wait for every side-stream wgrad issued so far and release their operands."""
    if _Side.dirty:
        torch.cuda.current_stream().wait_stream(_Side.stream)
        _Side.dirty = False
    _Side.keep = []



class ConvState:
    """Per-layer cache of the bf16 operand images derived from the fp32 master weight."""

    def __init__(self):
        self.key = None
        self.w_fprop = None     # [K][R][S][C] view of the front of _wf_buf (the row-major image)
        self._wf_buf = None     # the whole fprop image buffer: row-major image (+ the band kernel's fragment-ordered copy)
        self.w_dgrad = None
        self.rec = None

    def prepare(self, weight, pdesc, need_dgrad, vkey):
        """`vkey` identifies the VALUE of the master weight (parameter identity/version; the optimizer epoch is
        added here because the fused optimizer updates parameters behind torch's version counters). Channel
        padding (pdesc.k_valid / c_valid) is applied by the packer kernels: no torch ops are involved.
        The image buffers are allocated once and re-used (fixed addresses: PrepPlan re-packs them in one batched launch)."""
        key = (vkey, _weights_epoch, pdesc.key(), L.PRECISION)
        if self.key == key and self.w_fprop is not None and (self.w_dgrad is not None or not need_dgrad):
            return
        dev = weight.device
        master = _krsc_master(weight)
        lib = L.load()
        shape = (pdesc.K, pdesc.R, pdesc.S, pdesc.C)
        # image sizes from the library: stride-1 3x3 layers carry a second, fragment-ordered copy behind each image (the row-band
        # kernel's operand, include/cvhip.h cvhip_conv2d_prep_weights); w_fprop is the [K][R][S][C] view of the front of its buffer
        nf = int(lib.cvhip_conv2d_weight_image_elems(C.byref(pdesc), 0))
        if nf < 0:
            L.check(nf, "cvhip_conv2d_weight_image_elems")
        if (self.w_fprop is None or tuple(self.w_fprop.shape) != shape or self.w_fprop.device != dev or self.w_fprop.dtype != ACT_DTYPE
                or self._wf_buf is None or self._wf_buf.numel() != nf):
            self._wf_buf = torch.empty((nf,), dtype=ACT_DTYPE, device=dev)
            self.w_fprop = self._wf_buf[:shape[0] * shape[1] * shape[2] * shape[3]].view(shape)
            self.w_dgrad = None
        wd = None
        if need_dgrad:
            n = lib.cvhip_conv2d_weight_image_elems(C.byref(pdesc), 1)
            if n < 0:
                L.check(int(n), "cvhip_conv2d_weight_image_elems")
            n = max(int(n), 8)
            wd = self.w_dgrad if (self.w_dgrad is not None and self.w_dgrad.numel() == n) else torch.empty((n,), dtype=ACT_DTYPE, device=dev)
        L.call("cvhip_conv2d_prep_weights", C.byref(pdesc), master.data_ptr(), self.w_fprop.data_ptr(), _ptr(wd), _stream())
        self._master_ref = master  # keep a relayout copy (if any) alive until the kernels have run
        if wd is not None:
            self.w_dgrad = wd
        self.key = key
        # what PrepPlan needs to redo this preparation: only when the master is the parameter's own memory (no relayout copy)
        self.rec = (master.data_ptr(), pdesc, vkey) if master.data_ptr() == weight.data_ptr() else None


class PrepPlan:
    """ONE launch (cvhip_prep_plan_run) re-packs the bf16 operand images of every ConvState in `states` from the fp32 masters —
    instead of one cast + one pack launch per layer per step. Built after the states have been prepared once the ordinary way;
    `run()` marks them fresh for the current optimizer epoch so the per-layer `prepare` calls become no-ops."""

    def __init__(self, states):
        lib = L.load()
        self.states = [s for s in states if s.rec is not None and s.w_fprop is not None]
        n = len(self.states)
        self.n = n
        if n == 0:
            return
        entries = (L.PrepEntry * n)()
        self.ptrs = []
        for e, s in zip(entries, self.states):
            master, pdesc, _ = s.rec
            C.memmove(C.byref(e.desc), C.byref(pdesc), C.sizeof(L.ConvDesc))
            e.master, e.w_fprop = master, s.w_fprop.data_ptr()
            e.w_dgrad = s.w_dgrad.data_ptr() if s.w_dgrad is not None else None
            self.ptrs.append((master, e.w_fprop, e.w_dgrad))
        item = lib.cvhip_prep_plan_item_bytes()
        host = torch.zeros((n * item,), dtype=torch.uint8)
        blocks = C.c_int32(0)
        L.call("cvhip_prep_plan_build", C.cast(entries, C.c_void_p), n, host.data_ptr(), C.byref(blocks))
        self.blocks = int(blocks.value)
        self.table = host.to(self.states[0].w_fprop.device)

    def valid(self):
        return all(s.rec is not None and s.w_fprop is not None and (s.rec[0], s.w_fprop.data_ptr(), s.w_dgrad.data_ptr() if s.w_dgrad is not None else None) == p
                   for s, p in zip(self.states, self.ptrs))

    def run(self):
        if self.n == 0:
            return
        L.call("cvhip_prep_plan_run", self.table.data_ptr(), self.n, self.blocks, _stream())
        for s in self.states:
            s.key = (s.rec[2], _weights_epoch, s.rec[1].key(), L.PRECISION)


def conv_states_of(model):
    """every ConvState a model's layers own (HipConv2d._hip_state, the fused sibling-pair states)"""
    out = []
    for m in model.modules():
        for name in ("_hip_state", "_hip_pair_state"):
            s = m.__dict__.get(name)
            if isinstance(s, ConvState):
                out.append(s)
    return out


class ConvCfg:
    """Static configuration of one conv(+BN+act) layer (python-side)."""
    __slots__ = ("stride", "pad", "dil", "groups", "act", "act_param", "has_bn", "bn_training", "momentum", "eps",
                 "state", "track", "vkey", "gw", "gb", "gg", "gbeta", "arena", "idx_w", "idx_b", "idx_bn", "sync", "out", "out_split", "dx_link", "res_link", "res_pre",
                 "acc_owner", "acc_attr", "prod", "in_prod", "no_grad", "lazy_out", "lazy_half1", "lazy_half2", "lazy_made", "lazy_made2", "in_lazy", "res_lazy")

    def __init__(self, stride, pad, dil, groups=1, act=L.ACT_NONE, act_param=0.0, has_bn=False, bn_training=True,
                 momentum=0.1, eps=1e-5, state=None, track=True):
        self.stride, self.pad, self.dil, self.groups = tuple(stride), tuple(pad), tuple(dil), groups
        self.act, self.act_param = act, float(act_param)
        self.has_bn, self.bn_training = has_bn, bn_training
        self.momentum, self.eps = float(momentum), float(eps)
        self.state = state if state is not None else ConvState()
        self.track = track
        self.vkey = None  # set by the calling module each forward: (id(param), param._version)
        # flat gradient arena hooks (cvpytorch_amd/arena.py): views to accumulate weight / bias / BN gradients into
        self.gw = self.gb = self.gg = self.gbeta = None
        self.arena, self.idx_w, self.idx_b, self.idx_bn = None, None, None, ()
        self.sync = None  # SyncBN: (process_group or None for the default group, world_size) when statistics are shared across ranks
        # concat elimination: `out` = channel-slice view (of a larger NHWC buffer) that receives the layer's result instead of a
        # fresh tensor; `out_split` = (k1, view): channels [k1, K) go to `view`, channels [0, k1) to a fresh tensor (sibling pairs)
        self.out = None
        self.out_split = None
        # skip-connection gradient fusion (GradLink): `res_link` on the layer whose BN+act pass adds the residual parks that
        # branch's gradient in the link instead of returning it; `dx_link` on the layer that consumes the same tensor folds it
        # into its dgrad epilogue (cvhip_conv2d_dgrad_add)
        self.dx_link = None
        self.res_link = None
        # residual joins BEFORE the activation: z = act(bn(conv(x)) + residual) (ResNet bottleneck tail) instead of after it
        self.res_pre = False
        # module (the BatchNorm layer) that may carry this layer's persistent statistic accumulators, and under which attribute
        self.acc_owner = None
        self.acc_attr = "_hip_acc"
        # producer records (ProdInfo): `prod` = what this layer's forward made for its output, `in_prod` = its input's record
        self.prod = None
        self.in_prod = None
        self.no_grad = False   # conv_bn_act notes whether autograd was recording when the layer was called (inference fast path)
        # lazy activations (LazyAct above): `lazy_out` = the caller wants the RAW output (its consumers transform on load); `lazy_half1`
        # = the same for the first sibling of a pair; `lazy_made` = what forward actually made (None: the activated tensor, as ever);
        # `in_lazy` / `res_lazy` = the input's / residual's tag when THIS layer's kernels take them on load (set by the wrappers)
        self.lazy_out = False
        self.lazy_half1 = False
        self.lazy_half2 = False   # sibling pairs: the SECOND sibling's raw output goes straight into its concat slice (split store) and stays lazy
        self.lazy_made = None
        self.lazy_made2 = None
        self.in_lazy = None
        self.res_lazy = None


# ---- SyncBatchNorm plumbing (trainer.py:126-127 -> torch.nn.SyncBatchNorm semantics) -------------------------------------
def _sync_fwd_totals(partial, rows, K, M, sync):
    """Local partial rows [rows][2][K] -> global (sum, sum of squares) over all ranks as a 1-row partial buffer and the
    global element count. One all-reduce of 2K floats (equal per-rank batches, as DistributedSampler gives)."""
    comm, world = sync
    tot = torch.empty((1 + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=partial.device)
    L.call("cvhip_bn_bwd_finalize", partial.data_ptr(), rows, K, tot[0, 1].data_ptr(), tot[0, 0].data_ptr(), None, None, _stream())
    comm.allreduce_(tot[0])   # RCCL through the C ABI on the current stream (capturable); gloo only as the test transport
    return tot, 1, M * world


def _sync_bwd_sums(dgamma, dbeta, sync):
    """Global (sum dy*xhat, sum dy) scaled by 1/world: cvhip_bn_act_bwd_apply divides by the LOCAL row count, and
    global_sum / M_total == (global_sum / world) / M_local. The parameter gradients stay local (DDP averages them later)."""
    comm, world = sync
    g = torch.stack([dgamma, dbeta])
    comm.allreduce_(g)
    g = g / world
    return g[0], g[1]


def _colreduce_rows(M, Cc):
    return L.load().cvhip_colreduce_rows(M, Cc)


class GradLink:
    """One skip connection x -> (conv_a -> ... -> + x): carries the gradient of the identity branch from the op that performs the
    add (which returns None for that input, so autograd has nothing to accumulate) to conv_a's backward, whose dgrad kernel adds
    it in its epilogue. Created per forward call by the block that owns the skip connection; only when x requires grad.
    A parked gradient (`g` set through the property) is entered in `_PARKED`; the layer that folds it takes it out again by setting
    `g = None`. `check_parked()` at the end of a backward pass finds gradients nobody picked up."""
    __slots__ = ("_g", "ok", "consumed", "__weakref__")

    def __init__(self):
        self._g = None
        self.ok = False   # set by the consuming layer's forward once it is certain to run a dense dgrad on the unpadded input
        self.consumed = False   # set when the consuming layer's backward has started (a gradient parked after that would be lost)

    @property
    def g(self):
        return self._g

    @g.setter
    def g(self, v):
        self._g = v
        if v is None:
            _PARKED.discard(self)
        else:
            _PARKED.add(self)


_PARKED = __import__("weakref").WeakSet()   # links holding a gradient that no dgrad epilogue has folded yet


def check_parked(clear=True):
    """Raise if a gradient parked in a GradLink (skip connections, ops.fanout_linked's side branches) was never folded by its main
    consumer — the main consumer's output did not contribute to the loss (partial losses, a pruned or frozen branch,
    autograd.grad on a subset), so its backward never ran and the parked gradient would be silently DROPPED. Called by
    arena.FlatTrainState.backward after autograd returns; callers that drive loss.backward() themselves can call it too."""
    left = [l for l in list(_PARKED) if l.g is not None]
    if clear:
        for l in left:
            l.g = None
    if left:
        raise L.CvhipError("%d parked gradient(s) were never folded: the main consumer of a linked fan-out / skip connection did not run its "
                           "backward (its output does not reach the loss?). Set CVHIP_FANOUT_LINK=0 / CVHIP_GRAD_LINK=0 for such graphs." % len(left))


def _check_out(out, N, K, P, Q):
    """`out=` destination of a layer: an NHWC bf16 (channel-slice) view of the right logical shape"""
    if tuple(out.shape) != (N, K, P, Q) or out.dtype != ACT_DTYPE:
        raise L.CvhipError("out= has shape %s / %s, the layer produces %s bf16" % (tuple(out.shape), out.dtype, (N, K, P, Q)))
    ld = nhwc_ld(out)
    if ld is None:
        raise L.CvhipError("out= must be an NHWC (channels_last) tensor or a channel slice of one")
    return out, ld


def _mark(arena, idx):
    for i in (idx if isinstance(idx, tuple) else (idx,)):
        arena.mark_ready(i)


# Column sums a PRODUCER of a gradient map already knows (the YOLOv5 loss: cvhip_yolov5_loss_level_bwd_bias): keyed by the map's address,
# taken by the bias gradient of the convolution that receives exactly that map (one backward pass: the producer clears the table first).
_COLSUMS = {}   # (one table: the producer's backward and the consumer's run on autograd's device thread, tests look from the main thread)
_COLSUM_OFFER = __import__("os").environ.get("CVHIP_COLSUM_OFFER", "1") != "0"   # 0: consumers compute their column sums themselves (A/B switch)


def offer_colsum(draw, partial, rows, K):
    """`partial`: fp32 [rows + scratch][2][K] partial rows of the column sums of the NHWC gradient map `draw` (channel 0 at its address)"""
    if _COLSUM_OFFER:
        _COLSUMS[draw.data_ptr()] = (partial, int(rows), int(K), draw)   # (the map stays alive with its sums: its address cannot be re-used meanwhile)


def clear_colsums():
    _COLSUMS.clear()


def _take_colsum(ptr, K):
    hit = _COLSUMS.pop(ptr, None) if _COLSUMS else None
    if hit is None or hit[2] != K:
        return None
    return hit[0], hit[1]


def _conv_grads(ctx, x, weight, dy, dy_ld, need_dx, need_dw, need_db):
    """bias / weight / input gradients of the convolution itself from dy (the gradient at the conv output, NHWC bf16 with pitch
    dy_ld): shared by ConvBnAct.backward and ConvBnActPair.backward."""
    cfg = ctx.cfg
    N, Cc, H, W, K, R, S, P, Q, Kp, x_ld, Cg = ctx.geom
    dev = x.device
    M = N * P * Q
    st = _stream()
    kv = K if Kp != K else 0
    cv = Cg if (not ctx.depthwise and Cg != Cc) else 0
    arena = cfg.arena
    dbias = None
    if ctx.has_bias and need_db and not ctx.train_bn:
        pre = _take_colsum(dy.data_ptr(), K)
        if pre is not None:   # the producer of this gradient map handed its column sums over (detect heads: the fused YOLOv5 loss)
            partial, rows = pre
        else:
            rows = _colreduce_rows(M, K)
            partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
            L.call("cvhip_colsum_partial", dy.data_ptr(), M, K, dy_ld, partial.data_ptr(), st)
        if arena is not None and cfg.gb is not None:
            L.call("cvhip_colsum_finalize", partial.data_ptr(), rows, K, cfg.gb.data_ptr(), 1, st)
            arena.mark_ready(cfg.idx_b)
        else:
            dbias = torch.empty((K,), dtype=torch.float32, device=dev)
            L.call("cvhip_colsum_finalize", partial.data_ptr(), rows, K, dbias.data_ptr(), 0, st)
    elif ctx.has_bias and need_db:
        dbias = zero_fill(torch.empty((K,), dtype=torch.float32, device=dev))  # bias before train-mode BN: zero gradient
    dx = dw = None
    pending_side = None
    direct_w = arena is not None and cfg.gw is not None and tuple(cfg.gw.shape) == tuple(weight.shape)
    if ctx.depthwise:
        desc = conv_desc(N, Cc, H, W, K, R, S, cfg.stride, cfg.pad, cfg.dil, cfg.groups, x_ld, dy_ld)
        if need_dw and direct_w:
            L.call("cvhip_dwconv2d_wgrad", C.byref(desc), x.data_ptr(), dy.data_ptr(), cfg.gw.data_ptr(), 1, st)
            _mark(arena, cfg.idx_w)
        elif need_dw:
            dwm = torch.empty((K, R, S), dtype=torch.float32, device=dev)
            L.call("cvhip_dwconv2d_wgrad", C.byref(desc), x.data_ptr(), dy.data_ptr(), dwm.data_ptr(), 0, st)
            dw = dwm.reshape(K, 1, R, S)
        if need_dx:
            wm = weight.detach()
            wm = (wm if wm.dtype == torch.float32 else wm.float()).reshape(K, R, S)
            wm = wm if wm.is_contiguous() else wm.contiguous()
            dx = empty_nhwc(N, Cc, H, W, dev)
            ddesc = conv_desc(N, Cc, H, W, K, R, S, cfg.stride, cfg.pad, cfg.dil, cfg.groups, Cc, dy_ld)
            L.call("cvhip_dwconv2d_dgrad", C.byref(ddesc), dy.data_ptr(), wm.data_ptr(), dx.data_ptr(), st)
    else:
        if need_dw:
            desc = conv_desc(N, Cc, H, W, Kp, R, S, cfg.stride, cfg.pad, cfg.dil, 1, x_ld, dy_ld, kv, cv)
            padded = (Kp != K) or (Cg != Cc)
            geom = (N, Cc, H, W, K, R, S, P, Q)
            if not padded and direct_w and _Side.enabled and N * P * Q <= _Side.max_rows and not TIMER.enabled and not _DETERMINISTIC and (not arena.multi or arena.defer_allreduce):
                # same, on the side stream (see _Side): runs concurrently with the BN-backward chain of the layers below. With
                # _Side.after_dgrad the fork is taken AFTER this layer's dgrad launch, so wgrad (MFMA / LDS bound) shares the chip
                # with the HBM-bound BN passes that follow instead of with the dgrad kernel (same resources: both slowed down)
                def side_wgrad(desc=desc):
                    side = _side_begin()
                    with torch.cuda.stream(side):
                        L.call("cvhip_conv2d_wgrad", C.byref(desc), x.data_ptr(), dy.data_ptr(), cfg.gw.data_ptr(), 1, side.cuda_stream)
                    _Side.keep.append((x, dy))
                if _Side.after_dgrad and need_dx:
                    pending_side = side_wgrad
                else:
                    side_wgrad()
            elif not padded and direct_w:
                # accumulate straight into the parameter's KRSC slot of the flat gradient arena
                _wgrad(_wgrad_name(Kp, R, S, Cc, N * P * Q, desc), geom, desc, x, dy, cfg.gw, 1, st)
            elif not padded:
                # logical OIHW, KRSC (channels_last) memory: a fresh non-view tensor autograd can adopt as .grad
                dw = torch.empty((K, Cc, R, S), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
                _wgrad(_wgrad_name(Kp, R, S, Cc, N * P * Q, desc), geom, desc, x, dy, dw, 0, st)
            else:
                # padded problem: wgrad into a [Kp][R][S][Cc] scratch, then fold the valid block into the real gradient
                tmp = torch.empty((Kp, R, S, Cc), dtype=torch.float32, device=dev)
                planes = getattr(ctx, "image_planes", 0)
                done = False
                if planes and not _DETERMINISTIC:
                    # image stem whose forward read the fp32 NCHW batch directly: so does its weight gradient (x IS that batch) — unless
                    # the stem weight-gradient kernel refuses the operands (CVHIP_STEM_WGRAD=0, pitch / alignment of dy): then the
                    # conversion pass + the generic kernel below, as in deterministic mode
                    zero_fill(tmp)
                    try:
                        _timed_call("stem_wgrad_kernel", geom, "cvhip_conv2d_wgrad_image", C.byref(desc), x.data_ptr(), planes, dy.data_ptr(),
                                    tmp.data_ptr(), st)
                        done = True
                    except L.CvhipError as e:
                        if "unsupported" not in str(e):
                            raise
                if not done:
                    if planes:
                        x, _ = as_nhwc(images_to_nhwc(x, cpad=Cc))
                    _wgrad(_wgrad_name(Kp, R, S, Cc, N * P * Q, desc), geom, desc, x, dy, tmp, 0, st)
                if direct_w:
                    dst = cfg.gw
                else:
                    dw = zero_fill(torch.empty((K, Cg, R, S), dtype=torch.float32, device=dev, memory_format=torch.channels_last))
                    dst = dw
                L.call("cvhip_f32_unpad_add", tmp.data_ptr(), dst.data_ptr(), K, R * S, Cc, Cg, st)
            if direct_w:
                _mark(arena, cfg.idx_w)
        if need_dx:
            if ctx.w_dgrad is None:
                raise L.CvhipError("dgrad weight image missing (input started requiring grad after forward)")
            dx = empty_nhwc(N, Cc, H, W, dev)
            ddesc = conv_desc(N, Cc, H, W, Kp, R, S, cfg.stride, cfg.pad, cfg.dil, 1, Cc, dy_ld, kv, cv)
            link = cfg.dx_link
            pin = getattr(ctx, "in_prod", None)
            tail = _tail_for(pin, N, Cc, H, W) if ctx.c_orig == Cc else None
            if tail is not None:
                tname = _igemm_name(Cc, N * H * W, -(-R // cfg.stride[0]) * -(-S // cfg.stride[1]) * Kp, _pointwise(R, S, cfg), dy_ld, True)
                if not (_BN_TAIL_MASK & (2 if tname.startswith("conv1x1_stream") else 1)):
                    tail = None
            if tail is not None:
                # dx is the output gradient of the layer that made x: its BN-backward sums come out of this dgrad's epilogue
                g, g_ld = None, 0
                if link is not None and link.g is not None:
                    g, g_ld = as_nhwc(link.g)
                    link.g = None
                    if tuple(g.shape) != (N, Cc, H, W):
                        raise L.CvhipError("GradLink: skip-connection gradient %s does not match the layer input %s" % (tuple(g.shape), (N, Cc, H, W)))
                _timed_call(_igemm_name(Cc, N * H * W, -(-R // cfg.stride[0]) * -(-S // cfg.stride[1]) * Kp, _pointwise(R, S, cfg), dy_ld, True), (N, Cc, H, W, K, R, S, P, Q),
                            "cvhip_conv2d_dgrad_tail", C.byref(ddesc), dy.data_ptr(), ctx.w_dgrad.data_ptr(), _ptr(g), g_ld, dx.data_ptr(), C.byref(tail), st)
                pin.fused_dx, pin.done = dx, True
                if pending_side is not None:
                    pending_side()
                return dx, dw, dbias
            if link is not None and link.g is not None and ctx.c_orig == Cc:
                g, g_ld = as_nhwc(link.g)
                link.g = None
                if tuple(g.shape) != (N, Cc, H, W):
                    raise L.CvhipError("GradLink: skip-connection gradient %s does not match the layer input %s" % (tuple(g.shape), (N, Cc, H, W)))
                _timed_call(_igemm_name(Cc, N * H * W, -(-R // cfg.stride[0]) * -(-S // cfg.stride[1]) * Kp, _pointwise(R, S, cfg), dy_ld), (N, Cc, H, W, K, R, S, P, Q),
                            "cvhip_conv2d_dgrad_add", C.byref(ddesc), dy.data_ptr(), ctx.w_dgrad.data_ptr(), g.data_ptr(), g_ld, dx.data_ptr(), st)
                if pending_side is not None:
                    pending_side()
                return dx, dw, dbias
            _timed_call(_igemm_name(Cc, N * H * W, -(-R // cfg.stride[0]) * -(-S // cfg.stride[1]) * Kp, _pointwise(R, S, cfg), dy_ld), (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv2d_dgrad", C.byref(ddesc), dy.data_ptr(),
                        ctx.w_dgrad.data_ptr(), dx.data_ptr(), st)
            if pending_side is not None:
                pending_side()
    if dw is not None and dw.dtype != weight.dtype:
        dw = dw.to(weight.dtype)
    return dx, dw, dbias


def _materialize_tmp(x, x_ld, lz):
    """backward-side fallback: the activated copy of a lazily consumed input (same shape and pitch), for kernels without an on-load
    transform — costs the pass the forward saved, never more"""
    N, Cc, H, W = x.shape
    z = empty_nhwc(N, Cc, H, W, x.device, ld=x_ld)
    _materialize_into(x, x_ld, z, x_ld, lz)
    return z


_TAIL_MERGE = __import__("os").environ.get("CVHIP_TAIL_MERGE", "1") != "0"   # 0: residual tails as mask pass + reduction pass (A/B switch)
_STEM_BN = __import__("os").environ.get("CVHIP_STEM_BN", "1") != "0"   # 0: apply pass + plain stem weight gradient (A/B switch)


def _stem_wgrad_bn(ctx, cfg, x, y, weight, dz, stats, acc_b, g_out, b_out, accum, act, act_param):
    """dW of an image-stem Conv-BN-act layer straight from dz (BN + activation backward on load). Returns the weight gradient
    (None when it went into the flat gradient arena), or False when the stem kernel does not run this problem."""
    N, Cc, H, W, K, R, S, P, Q, Kp, x_ld, Cg = ctx.geom
    dev = dz.device
    st = _stream()
    cv = Cg if Cg != Cc else 0
    desc = conv_desc(N, Cc, H, W, Kp, R, S, cfg.stride, cfg.pad, cfg.dil, 1, x_ld, Kp, 0, cv)
    if L.load().cvhip_conv_stem_blocks(C.byref(desc)) <= 0:
        return False
    planes = getattr(ctx, "image_planes", 0)
    tmp = zero_fill(torch.empty((Kp, R, S, Cc), dtype=torch.float32, device=dev))
    try:
        _timed_call("stem_wgrad_kernel", (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv2d_wgrad_stem_bn", C.byref(desc),
                    None if planes else x.data_ptr(), x.data_ptr() if planes else None, planes, dz.data_ptr(), y.data_ptr(),
                    stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), acc_b.data_ptr(), K,
                    g_out.data_ptr(), b_out.data_ptr(), int(accum), act, act_param, tmp.data_ptr(), st,
                    nbytes=2.0 * (N * H * W * Cc + 2 * N * P * Q * K))
    except L.CvhipError as e:
        if "unsupported" not in str(e):
            raise
        return False
    arena = cfg.arena
    direct_w = arena is not None and cfg.gw is not None and tuple(cfg.gw.shape) == tuple(weight.shape)
    if direct_w:
        L.call("cvhip_f32_unpad_add", tmp.data_ptr(), cfg.gw.data_ptr(), K, R * S, Cc, Cg, st)
        _mark(arena, cfg.idx_w)
        return None
    dw = zero_fill(torch.empty((K, Cg, R, S), dtype=torch.float32, device=dev, memory_format=torch.channels_last))
    L.call("cvhip_f32_unpad_add", tmp.data_ptr(), dw.data_ptr(), K, R * S, Cc, Cg, st)
    return dw if dw.dtype == weight.dtype else dw.to(weight.dtype)


def _bwd1x1_ok(ctx, cfg, x, need_dx, need_dw, need_db, segs):
    """True when the fused 1x1 backward kernel (conv1x1_bwd.hip: BN/act backward on load + dgrad + wgrad in one pass) takes
    this layer: dense 1x1 stride-1, K in {32, 64, 128}, unpadded channels, both gradients wanted, 16-byte aligned operands."""
    N, Cc, H, W, K, R, S, P, Q, Kp, x_ld, Cg = ctx.geom
    if _DETERMINISTIC:   # the fused kernel flushes dW with fp32 atomics
        return False
    if ctx.depthwise or R != 1 or S != 1 or Kp != K or Cg != Cc or ctx.c_orig != Cc or not (need_dx and need_dw):
        return False
    if (ctx.has_bias and need_db) or ctx.w_dgrad is None or x.data_ptr() % 16 or x_ld % 8:
        return False
    for d, d_ld in segs:
        if d_ld % 8 or d.data_ptr() % 16:
            return False
    desc = conv_desc(N, Cc, H, W, K, R, S, cfg.stride, cfg.pad, cfg.dil, 1, x_ld, Kp)
    return bool(L.load().cvhip_conv1x1_bwd_fused_ok(C.byref(desc)))


def _bwd1x1(ctx, cfg, x, y, weight, segs, k_split, stats, with_mean, ag, ab, act, act_param, acc=None, g_out=None, b_out=None, accumulate=0,
            xin=None, y1=None):
    """dx, dw of a 1x1 Conv-BN-act layer from the gradient(s) at its OUTPUT in one launch (`segs`: one (tensor, pitch), or two
    for sibling pairs; `stats` rows: mean, invstd, scale, shift; ag / ab: sum du*xhat / sum du — or `acc`: the layer's backward
    accumulator, folded by the kernel itself, which then also stores dgamma / dbeta into g_out / b_out)."""
    N, Cc, H, W, K, R, S, P, Q, Kp, x_ld, Cg = ctx.geom
    dev = x.device
    st = _stream()
    arena = cfg.arena
    direct_w = arena is not None and cfg.gw is not None and tuple(cfg.gw.shape) == tuple(weight.shape)
    dw = None
    if direct_w:
        dst = cfg.gw
    else:
        dw = zero_fill(torch.empty((K, Cc, 1, 1), dtype=torch.float32, device=dev, memory_format=torch.channels_last))
        dst = dw
    dx = empty_nhwc(N, Cc, H, W, dev)
    g, g_ld = None, 0
    link = cfg.dx_link
    if link is not None and link.g is not None:
        g, g_ld = as_nhwc(link.g)
        link.g = None
        if tuple(g.shape) != (N, Cc, H, W):
            raise L.CvhipError("GradLink: skip-connection gradient %s does not match the layer input %s" % (tuple(g.shape), (N, Cc, H, W)))
        if g_ld % 8 or g.data_ptr() % 16:
            g = g.contiguous(memory_format=torch.channels_last)
            g_ld = Cc
    desc = conv_desc(N, Cc, H, W, K, R, S, cfg.stride, cfg.pad, cfg.dil, 1, x_ld, Kp)
    (d0, d0_ld) = segs[0]
    (d1, d1_ld) = segs[1] if len(segs) > 1 else (None, 0)
    sc = stats[2].data_ptr() if stats is not None else None
    sh = stats[3].data_ptr() if stats is not None else None
    mu = stats[0].data_ptr() if with_mean else None
    isd = stats[1].data_ptr() if with_mean else None
    M = N * H * W
    if y1 is not None:
        # sibling pair whose second half was stored straight into its concat slice (split store): the raw output has two homes
        if acc is None:
            raise L.CvhipError("split raw output without the accumulator form of the fused 1x1 backward")
        li = _lazy_in(xin) if xin is not None else None
        _timed_call("bwd1x1_kernel", (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv1x1_bwd_fused_split", C.byref(desc), d0.data_ptr(), d0_ld,
                    _ptr(d1), d1_ld, k_split, y.data_ptr(), y1[0].data_ptr(), y1[1], x.data_ptr(), ctx.w_dgrad.data_ptr(), sc, sh, mu, isd,
                    acc.data_ptr(), K, _ptr(g_out), _ptr(b_out), int(accumulate), act, act_param, _ptr(g), g_ld,
                    dx.data_ptr(), Cc, dst.data_ptr(), C.byref(li) if li is not None else None, st, passes=2,
                    nbytes=2.0 * M * (2 * K + 2 * Cc))
    elif xin is not None:
        # x is the RAW output of the layer that made it (lazy activation): the kernel transforms the rows it stages for the weight gradient
        if acc is None:
            raise L.CvhipError("lazy input without the accumulator form of the fused 1x1 backward")
        li = _lazy_in(xin)
        _timed_call("bwd1x1_kernel", (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv1x1_bwd_fused_lazy", C.byref(desc), d0.data_ptr(), d0_ld,
                    _ptr(d1), d1_ld, k_split, y.data_ptr(), x.data_ptr(), ctx.w_dgrad.data_ptr(), sc, sh, mu, isd,
                    acc.data_ptr(), K, _ptr(g_out), _ptr(b_out), int(accumulate), act, act_param, _ptr(g), g_ld,
                    dx.data_ptr(), Cc, dst.data_ptr(), C.byref(li), st, passes=2, nbytes=2.0 * M * (2 * K + 2 * Cc))
    elif acc is not None:
        pin = getattr(ctx, "in_prod", None)
        tail = _tail_for(pin, N, Cc, H, W) if (_BN_TAIL_MASK & 4) else None
        _timed_call("bwd1x1_kernel", (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv1x1_bwd_fused_acc", C.byref(desc), d0.data_ptr(), d0_ld,
                    _ptr(d1), d1_ld, k_split, y.data_ptr(), x.data_ptr(), ctx.w_dgrad.data_ptr(), sc, sh, mu, isd,
                    acc.data_ptr(), K, _ptr(g_out), _ptr(b_out), int(accumulate), act, act_param, _ptr(g), g_ld,
                    dx.data_ptr(), Cc, dst.data_ptr(), C.byref(tail) if tail is not None else None, st, passes=2,
                    nbytes=2.0 * M * (2 * K + 2 * Cc + (Cc if tail is not None else 0)))
        if tail is not None:
            pin.fused_dx, pin.done = dx, True
    else:
        _timed_call("bwd1x1_kernel", (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv1x1_bwd_fused", C.byref(desc), d0.data_ptr(), d0_ld,
                    _ptr(d1), d1_ld, k_split, y.data_ptr(), x.data_ptr(), ctx.w_dgrad.data_ptr(), sc, sh, mu, isd,
                    ag.data_ptr() if with_mean else None, ab.data_ptr() if with_mean else None, act, act_param, _ptr(g), g_ld,
                    dx.data_ptr(), Cc, dst.data_ptr(), st, passes=2, nbytes=2.0 * M * (2 * K + 2 * Cc))
    if direct_w:
        _mark(arena, cfg.idx_w)
    if dw is not None and dw.dtype != weight.dtype:
        dw = dw.to(weight.dtype)
    return dx, dw


def _bn_fwd_acc(y, y_ld, z, z_ld, M, kh, off, K, acc_f, gamma, beta, running_mean, running_var, cfg, stats, residual, res_ld, res_pre, st):
    """BN + activation apply for channels [off, off + kh) of a K-channel layer whose (sum, sum of squares) sit in `acc_f`: the pass
    derives scale / shift itself, stores the layer's statistics for backward and updates the running statistics (block 0)."""
    o4, o8 = 4 * off, 8 * off
    rm = running_mean if cfg.track else None
    rv = running_var if cfg.track else None
    g = gamma.detach() if gamma is not None else None
    bt = beta.detach() if beta is not None else None
    rl = cfg.res_lazy if residual is not None else None
    if rl is not None:
        # the residual is a LAZY activation (raw output of the layer that made it): its transform rides in this pass
        _timed_ew("bn_act_fwd(ew_kernel<0>)", 6.0 * M * kh, "cvhip_bn_act_fwd_acc_lazyres",
                  y.data_ptr() + 2 * off, y_ld, z.data_ptr(), z_ld, M, kh, acc_f.data_ptr() + o8, K, M,
                  (g.data_ptr() + o4) if g is not None else None, (bt.data_ptr() + o4) if bt is not None else None,
                  (rm.data_ptr() + o4) if rm is not None else None, (rv.data_ptr() + o4) if rv is not None else None,
                  cfg.momentum, cfg.eps, stats[0].data_ptr() + o4, stats[1].data_ptr() + o4, stats[2].data_ptr() + o4, stats[3].data_ptr() + o4,
                  cfg.act, cfg.act_param, residual.data_ptr(), res_ld, rl.scale.data_ptr(), rl.shift.data_ptr(), st)
        return
    _timed_ew("bn_act_fwd(ew_kernel<0>)", 2.0 * M * kh * (3 if residual is not None else 2), "cvhip_bn_act_fwd_acc",
              y.data_ptr() + 2 * off, y_ld, z.data_ptr(), z_ld, M, kh, acc_f.data_ptr() + o8, K, M,
              (g.data_ptr() + o4) if g is not None else None, (bt.data_ptr() + o4) if bt is not None else None,
              (rm.data_ptr() + o4) if rm is not None else None, (rv.data_ptr() + o4) if rv is not None else None,
              cfg.momentum, cfg.eps, stats[0].data_ptr() + o4, stats[1].data_ptr() + o4, stats[2].data_ptr() + o4, stats[3].data_ptr() + o4,
              cfg.act, cfg.act_param, _ptr(residual), res_ld, int(bool(res_pre)), st)


def _eval_scale_shift(cfg, gamma, beta, running_mean, running_var, K, dev, st):
    """(scale, shift) of an eval-mode BatchNorm, cached on the layer's ConvState until one of its tensors is written again"""
    # (_weights_epoch: the fused optimizer step and the BN kernels write parameters / running statistics through raw pointers,
    # behind torch's version counters)
    key = (running_mean.data_ptr(), running_mean._version, running_var.data_ptr(), running_var._version,
           None if gamma is None else (gamma.data_ptr(), gamma._version), None if beta is None else (beta.data_ptr(), beta._version),
           float(cfg.eps), _weights_epoch, _stats_epoch[0])
    cache = getattr(cfg.state, "ep_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    ss = torch.empty((2, K), dtype=torch.float32, device=dev)
    L.call("cvhip_bn_eval_scale_shift", K, _ptr(gamma.detach() if gamma is not None else None),
           _ptr(beta.detach() if beta is not None else None), running_mean.data_ptr(), running_var.data_ptr(), cfg.eps,
           ss[0].data_ptr(), ss[1].data_ptr(), st)
    cfg.state.ep_cache = (key, ss)
    return ss


def _dw_folded(cfg, wm, bias, gamma, beta, running_mean, running_var, K, dev, st):
    """(weights [K][R][S], bias [K]) of a depthwise layer with its eval-mode BatchNorm folded in (fp32), cached on the layer's
    ConvState until the weight or a BatchNorm tensor is written again"""
    if not cfg.has_bn:
        return wm, bias
    key = (wm.data_ptr(), wm._version, running_mean.data_ptr(), running_mean._version, running_var.data_ptr(), running_var._version,
           None if gamma is None else (gamma.data_ptr(), gamma._version), None if beta is None else (beta.data_ptr(), beta._version),
           None if bias is None else (bias.data_ptr(), bias._version), float(cfg.eps), _weights_epoch, _stats_epoch[0])
    cache = getattr(cfg.state, "dw_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    ss = _eval_scale_shift(cfg, gamma, beta, running_mean, running_var, K, dev, st)
    wf = (wm * ss[0].view(K, 1, 1)).contiguous()
    bf = ss[1].clone() if bias is None else (bias * ss[0] + ss[1])
    cfg.state.dw_cache = (key, wf, bf)
    return wf, bf


def _conv_fused_inference(x, x_ld, weight, bias, gamma, beta, running_mean, running_var, residual, cfg, geom, kv, cv, st):
    """z = act(bn_eval(conv(x, W) + b)) (+ residual) in ONE launch: the convolution's epilogue applies the bias, the eval-mode
    BatchNorm's scale / shift, the activation and the (post-activation) residual; the result goes straight to `cfg.out` when the
    caller handed a concat slice. No tensor is saved: this path only runs when nothing requires a gradient."""
    N, Cc, H, W, K, R, S, P, Q = geom
    dev = x.device
    if cfg.out is not None:
        z, z_ld = _check_out(cfg.out, N, K, P, Q)
    else:
        z, z_ld = empty_nhwc(N, K, P, Q, dev), K
    desc = conv_desc(N, Cc, H, W, K, R, S, cfg.stride, cfg.pad, cfg.dil, 1, x_ld, z_ld, kv, cv)
    f = L.ConvFuse()
    keep = [bias]
    f.bias = _ptr(bias)
    if cfg.has_bn:
        ss = _eval_scale_shift(cfg, gamma, beta, running_mean, running_var, K, dev, st)
        keep.append(ss)
        f.ep_scale, f.ep_shift = ss[0].data_ptr(), ss[1].data_ptr()
    f.ep_act, f.ep_act_param = int(cfg.act), float(cfg.act_param)
    if residual is not None:
        residual, res_ld = as_nhwc(residual)
        keep.append(residual)
        f.residual, f.residual_ld = residual.data_ptr(), res_ld
        f.residual_pre = int(bool(cfg.res_pre))   # ResNet bottleneck tail: relu(bn3(conv3) + identity)
    kname = "conv_fused_inference"
    _timed_call(kname, geom, "cvhip_conv2d_fprop_fused", C.byref(desc), x.data_ptr(), cfg.state.w_fprop.data_ptr(), z.data_ptr(), C.byref(f), st)
    cfg.prod = None
    return z


class ConvBnAct(torch.autograd.Function):
    """z = act(bn(conv(x, W) + b)) (+ residual)   — any of bn / act / bias / residual optional.

    Channel counts that are not multiples of 8 are handled by padding INSIDE the engine: x may arrive with its
    channels zero-padded to 8 (image stem), K is padded to 8 internally (255-channel detect head); the fp32
    master weight / bias / their gradients keep the layer's real shapes."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, residual, cfg):
        lib = L.load()
        K, Cg, R, S = weight.shape
        depthwise = cfg.groups != 1
        # image stems: an fp32 NCHW image batch (what the reference's dataloader delivers, <= 4 channels) can be read by the stem
        # kernel itself — no layout / precision pass in front of the first convolution (decided below, once the descriptor is known)
        image = None
        if (_STEM_IMAGE and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == Cg <= 4 and not depthwise and x.is_contiguous()
                and not x.requires_grad and x.is_cuda):
            image = x
            N, planes, H, W = x.shape
            Cc, x_ld = 8, 8
        else:
            x, x_ld = as_nhwc(x, lazy_ok=cfg.in_lazy is not None)
            N, Cc, H, W = x.shape
        dev = x.device
        c_orig = Cc
        if not depthwise and Cc == Cg and (Cc % 8 != 0 or x_ld % 8 != 0 or x.data_ptr() % 16 != 0):
            # the MFMA gather reads 16-byte channel vectors: repack odd channel counts / pitches / slice offsets into a
            # zero-padded [N,H,W,round8(C)] buffer (the packers pad the weight to match: cvhip_conv_desc.c_valid)
            Cp = _round8(Cc)
            xp = torch.empty((N, H, W, Cp), dtype=ACT_DTYPE, device=dev)
            if Cp != Cc:
                zero_fill(xp)
            L.call("cvhip_copy2d", x.data_ptr(), x_ld, xp.data_ptr(), Cp, N * H * W, Cc, _stream())
            x, x_ld, Cc = xp.permute(0, 3, 1, 2), Cp, Cp
        if depthwise and not (cfg.groups == Cc == K and Cg == 1):
            raise L.CvhipError("grouped conv other than depthwise is not supported by the HIP engine")
        if not depthwise and Cc != Cg and Cc != _round8(Cg):
            raise L.CvhipError("conv input has %d channels, weight expects %d" % (Cc, Cg))
        P, Q = conv_out_hw(H, W, R, S, cfg.stride, cfg.pad, cfg.dil)
        M = N * P * Q
        Kp = _round8(K)
        kv = K if Kp != K else 0
        cv = Cg if (not depthwise and Cg != Cc) else 0
        need_dx = ctx.needs_input_grad[0]
        if cfg.dx_link is not None:
            cfg.dx_link.ok = bool(need_dx and not depthwise and c_orig == Cc)   # (grad mode is checked where the link is created)
        y = empty_nhwc(N, K, P, Q, dev, ld=Kp)
        st = _stream()
        train_bn = cfg.has_bn and cfg.bn_training
        if train_bn:
            _stats_epoch[0] += 1
        if cfg.has_bn and bias is not None:
            raise L.CvhipError("conv bias followed by BatchNorm is not supported (ConvModule never builds it)")
        stats = None
        partial = None
        rows = 0
        use_acc = False
        acc_f = acc_b = None
        split2 = None
        epilogue_stats = train_bn and not depthwise and Kp == K  # BN sums straight from the MFMA accumulators
        b = bias.detach() if bias is not None else None
        if b is not None and b.dtype != torch.float32:
            b = b.float()
        if depthwise:
            desc = conv_desc(N, Cc, H, W, K, R, S, cfg.stride, cfg.pad, cfg.dil, cfg.groups, x_ld, Kp)
            wm = weight.detach()
            wm = (wm if wm.dtype == torch.float32 else wm.float()).reshape(K, R, S)
            wm = wm if wm.is_contiguous() else wm.contiguous()
            if (_EPI_FUSE and (cfg.no_grad or not any(ctx.needs_input_grad)) and not train_bn and residual is None and cfg.out_split is None
                    and (cfg.has_bn or cfg.act != L.ACT_NONE)):
                # inference: depthwise conv + eval-mode BatchNorm + activation in ONE pass. The BatchNorm's scale / shift are folded
                # into the (tiny, fp32) depthwise weights and bias on the way — utils/fuse.py:32-54 per call, cached per layer
                wf, bf = _dw_folded(cfg, wm, b, gamma, beta, running_mean, running_var, K, dev, st)
                if cfg.out is not None:
                    z, z_ld = _check_out(cfg.out, N, K, P, Q)
                else:
                    z, z_ld = (y, Kp) if Kp == K else (empty_nhwc(N, K, P, Q, dev), K)
                zdesc = conv_desc(N, Cc, H, W, K, R, S, cfg.stride, cfg.pad, cfg.dil, cfg.groups, x_ld, z_ld)
                _timed_ew("dw_fused_inference", 2.0 * (N * H * W * Cc + M * K), "cvhip_dwconv2d_fprop_act", C.byref(zdesc), x.data_ptr(), wf.data_ptr(),
                          _ptr(bf), int(cfg.act), float(cfg.act_param), z.data_ptr(), st)
                cfg.prod = None
                return z
            dw_rows = 0
            if train_bn and _DW_STATS and Kp == K:
                # BatchNorm sums from the strip kernel's fp32 outputs (one partial row per strip): no reduction pass over the stored output
                dw_rows = int(lib.cvhip_dwconv2d_fprop_stats_rows(C.byref(desc), x.data_ptr(), y.data_ptr()))
                if dw_rows < 0:
                    L.check(dw_rows, "cvhip_dwconv2d_fprop_stats_rows")
            if dw_rows > 0:
                rows = dw_rows
                partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
                L.call("cvhip_dwconv2d_fprop_stats", C.byref(desc), x.data_ptr(), wm.data_ptr(), _ptr(b), y.data_ptr(), partial.data_ptr(), st)
            else:
                L.call("cvhip_dwconv2d_fprop", C.byref(desc), x.data_ptr(), wm.data_ptr(), _ptr(b), y.data_ptr(), st)
        else:
            desc = conv_desc(N, Cc, H, W, Kp, R, S, cfg.stride, cfg.pad, cfg.dil, 1, x_ld, Kp, kv, cv)
            # pack descriptor: contiguous pitches (the packed images do not depend on activation pitches)
            pdesc = conv_desc(N, Cc, H, W, Kp, R, S, cfg.stride, cfg.pad, cfg.dil, 1, Cc, Kp, kv, cv)
            cfg.state.prepare(weight, pdesc, need_dx, cfg.vkey)
            infer = (_EPI_FUSE and (cfg.no_grad or not any(ctx.needs_input_grad)) and not train_bn and Kp == K and cfg.out_split is None
                     and (cfg.has_bn or cfg.act != L.ACT_NONE or residual is not None))
            if image is not None and (infer or residual is not None or lib.cvhip_conv_stem_blocks(C.byref(desc)) <= 0):
                # not a problem the image-stem kernel runs (or the one-launch inference form): the explicit conversion pass
                x, x_ld = as_nhwc(images_to_nhwc(image, cpad=8))
                image = None
            if infer:
                return _conv_fused_inference(x, x_ld, weight, b, gamma, beta, running_mean, running_var, residual, cfg,
                                             (N, Cc, H, W, K, R, S, P, Q), kv, cv, st)
            use_acc = _BN_ACC and not _DETERMINISTIC and epilogue_stats and cfg.sync is None and K <= _BN_ACC_MAX_C
            if use_acc:
                acc_f, acc_b = _layer_acc(cfg, K, dev)
            elif epilogue_stats:
                rows = lib.cvhip_conv2d_fprop_stats_rows(C.byref(desc))
                if rows < 0:
                    L.check(rows, "cvhip_conv2d_fprop_stats_rows")
                partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
            kname = "stem_fprop_kernel" if lib.cvhip_conv_stem_blocks(C.byref(desc)) > 0 else _igemm_name(Kp, N * P * Q, R * S * Cc, _pointwise(R, S, cfg), x_ld, epilogue_stats)
            if (cfg.lazy_half2 and cfg.out_split is not None and _LAZY and train_bn and use_acc and residual is None and Kp == K
                    and cfg.act in _LAZY_ACTS and any(ctx.needs_input_grad) and image is None and _pointwise(R, S, cfg)):
                # second sibling lazy: the streaming kernel stores channels [k1, K) of the RAW output straight into the caller's concat
                # slice (split store) and that slice stays raw — no apply pass, no copy; the concat's consumer transforms it on load
                k1s, z2s = cfg.out_split
                z2s, z2s_ld = _check_out(z2s, N, K - k1s, P, Q)
                if (k1s % 8 == 0 and (K - k1s) % 8 == 0 and z2s_ld % 8 == 0 and z2s.data_ptr() % 16 == 0
                        and lib.cvhip_conv1x1_stream_prologue_ok(C.byref(desc), 1) > 0):   # (= the streaming kernel runs this problem)
                    split2 = (k1s, z2s, z2s_ld)
            if cfg.in_lazy is not None or split2 is not None:
                # lazy input: the producing layer's BN scale / shift + activation are applied on load by the streaming 1x1 kernel
                lzi = cfg.in_lazy
                f = L.ConvFuse()
                if lzi is not None:
                    f.pro_scale, f.pro_shift, f.pro_act, f.pro_act_param = lzi.scale_ptr(), lzi.shift_ptr(), lzi.act, lzi.ap
                    if not lzi.full(Cc):
                        f.pro_lo, f.pro_hi = lzi.lo, lzi.hi
                if split2 is not None:
                    f.y2, f.y2_ld, f.y_split = split2[1].data_ptr(), split2[2], split2[0]
                if use_acc:
                    f.bn_acc = acc_f.data_ptr()
                else:
                    f.bias, f.stats_partial = _ptr(b), _ptr(partial)
                _timed_call(kname, (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv2d_fprop_fused", C.byref(desc), x.data_ptr(),
                            cfg.state.w_fprop.data_ptr(), y.data_ptr(), C.byref(f), st)
            elif image is not None:
                f = L.ConvFuse()
                f.x_image, f.x_image_planes = image.data_ptr(), planes
                if use_acc:
                    f.bn_acc = acc_f.data_ptr()
                else:
                    f.bias, f.stats_partial = _ptr(b), _ptr(partial)
                _timed_call(kname, (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv2d_fprop_fused", C.byref(desc), None,
                            cfg.state.w_fprop.data_ptr(), y.data_ptr(), C.byref(f), st)
            elif use_acc:
                _timed_call(kname, (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv2d_fprop_acc", C.byref(desc), x.data_ptr(),
                            cfg.state.w_fprop.data_ptr(), y.data_ptr(), acc_f.data_ptr(), st)
            else:
                _timed_call(kname, (N, Cc, H, W, K, R, S, P, Q), "cvhip_conv2d_fprop", C.byref(desc), x.data_ptr(),
                            cfg.state.w_fprop.data_ptr(), _ptr(b), y.data_ptr(), _ptr(partial), st)
        if train_bn and use_acc:
            stats = torch.empty((4, K), dtype=torch.float32, device=dev)  # mean, invstd, scale, shift: written by the apply pass
        elif train_bn:
            if partial is None:
                rows = _colreduce_rows(M, K)
                partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
                L.call("cvhip_bn_stats_partial", y.data_ptr(), M, K, Kp, partial.data_ptr(), st)
            stats = torch.empty((4, K), dtype=torch.float32, device=dev)  # mean, invstd, scale, shift
            g = gamma.detach() if gamma is not None else None
            bt = beta.detach() if beta is not None else None
            rm = running_mean if cfg.track else None
            rv = running_var if cfg.track else None
            Mstat = M
            if cfg.sync is not None:
                partial, rows, Mstat = _sync_fwd_totals(partial, rows, K, M, cfg.sync)
            L.call("cvhip_bn_finalize", partial.data_ptr(), rows, K, Mstat, _ptr(g), _ptr(bt), _ptr(rm), _ptr(rv),
                   cfg.momentum, cfg.eps, stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), st)
        elif cfg.has_bn:
            stats = torch.empty((4, K), dtype=torch.float32, device=dev)
            L.call("cvhip_bn_eval_scale_shift", K, _ptr(gamma.detach() if gamma is not None else None),
                   _ptr(beta.detach() if beta is not None else None), running_mean.data_ptr(), running_var.data_ptr(),
                   cfg.eps, stats[2].data_ptr(), stats[3].data_ptr(), st)
        res_ld = 0
        if residual is not None:
            residual, res_ld = as_nhwc(residual, lazy_ok=cfg.res_lazy is not None)
        cfg.lazy_made = cfg.lazy_made2 = None
        lazy_able = _LAZY and train_bn and use_acc and residual is None and Kp == K and cfg.act in _LAZY_ACTS and any(ctx.needs_input_grad)
        if cfg.lazy_out and lazy_able and cfg.out_split is None and cfg.out is None:
            # lazy result: statistics finalized (tiny launch), NO apply pass — the consumers read the raw y and transform on load
            o8 = 0
            L.call("cvhip_bn_finalize_acc", acc_f.data_ptr(), K, K, M, _ptr(gamma.detach() if gamma is not None else None),
                   _ptr(beta.detach() if beta is not None else None), _ptr(running_mean if cfg.track else None),
                   _ptr(running_var if cfg.track else None), cfg.momentum, cfg.eps, stats[0].data_ptr(), stats[1].data_ptr(),
                   stats[2].data_ptr(), stats[3].data_ptr(), st)
            z = y
            cfg.lazy_made = (stats, 0, K)
        elif cfg.out_split is not None and cfg.has_bn and residual is None:
            # two destinations: channels [0, k1) -> a fresh tensor, [k1, K) -> the caller's slice of a concat buffer
            k1, z2 = cfg.out_split
            z2, z2_ld = _check_out(z2, N, K - k1, P, Q)
            lazy_h1 = cfg.lazy_half1 and lazy_able and k1 % 8 == 0
            z1 = y[:, :k1] if lazy_h1 else empty_nhwc(N, k1, P, Q, dev)
            for off, kh, zz, zld in ((0, k1, z1, k1), (k1, K - k1, z2, z2_ld)):
                if (lazy_h1 and off == 0) or (split2 is not None and off):
                    # lazy sibling: its statistics only (each half finalizes its own channel range of the shared accumulator)
                    g = gamma.detach() if gamma is not None else None
                    bt = beta.detach() if beta is not None else None
                    o4, o8 = 4 * off, 8 * off
                    L.call("cvhip_bn_finalize_acc", acc_f.data_ptr() + o8, K, kh, M, _ptr(g, o4), _ptr(bt, o4),
                           _ptr(running_mean if cfg.track else None, o4), _ptr(running_var if cfg.track else None, o4), cfg.momentum, cfg.eps,
                           stats[0].data_ptr() + o4, stats[1].data_ptr() + o4, stats[2].data_ptr() + o4, stats[3].data_ptr() + o4, st)
                    if off:
                        cfg.lazy_made2 = (stats, off, kh)
                    else:
                        cfg.lazy_made = (stats, 0, k1)
                    continue
                if use_acc:
                    _bn_fwd_acc(y, Kp, zz, zld, M, kh, off, K, acc_f, gamma, beta, running_mean, running_var, cfg, stats, None, 0, False, st)
                else:
                    _timed_ew("bn_act_fwd(ew_kernel<0>)", 4.0 * M * kh, "cvhip_bn_act_fwd", y.data_ptr() + 2 * off, Kp, zz.data_ptr(), zld, M, kh, stats[2].data_ptr() + 4 * off,
                           stats[3].data_ptr() + 4 * off, cfg.act, cfg.act_param, None, 0, st)
            z = (z1, z2)
        elif cfg.has_bn or cfg.act != L.ACT_NONE or residual is not None:
            if cfg.out is not None:
                z, z_ld = _check_out(cfg.out, N, K, P, Q)
            else:
                z, z_ld = empty_nhwc(N, K, P, Q, dev), K
            res_pre = bool(cfg.res_pre and residual is not None)
            if res_pre and (cfg.act not in (L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY) or Kp != K):
                raise L.CvhipError("res_pre needs none / ReLU / LeakyReLU and an unpadded channel count")
            if cfg.res_lazy is not None and not (use_acc and not res_pre):
                raise L.CvhipError("lazy residual without the accumulator form of the BN+act pass (the wrapper should have materialised it)")
            if use_acc:
                _bn_fwd_acc(y, Kp, z, z_ld, M, K, 0, K, acc_f, gamma, beta, running_mean, running_var, cfg, stats, residual, res_ld, res_pre, st)
            else:
                _timed_ew("bn_act_fwd(ew_kernel<0>)", 2.0 * M * K * (3 if residual is not None else 2),
                          "cvhip_bn_add_act_fwd" if res_pre else "cvhip_bn_act_fwd", y.data_ptr(), Kp, z.data_ptr(), z_ld, M, K,
                       _ptr(stats[2]) if stats is not None else None, _ptr(stats[3]) if stats is not None else None,
                       cfg.act, cfg.act_param, _ptr(residual), res_ld, st)
        else:
            z = y
        ctx.cfg = cfg
        if cfg.arena is not None and ctx.needs_input_grad[1]:
            # bucketed all-reduce: a parameter's bucket may only go once EVERY op that accumulates into its slot has run backward
            cfg.arena.note_use(cfg.idx_w)
            cfg.arena.note_use(cfg.idx_b)
            cfg.arena.note_use(tuple(cfg.idx_bn))
        ctx.geom = (N, Cc, H, W, K, R, S, P, Q, Kp, x_ld, Cg)
        ctx.c_orig = c_orig
        ctx.train_bn = train_bn
        ctx.depthwise = depthwise
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.w_dgrad = cfg.state.w_dgrad if not depthwise else None
        ctx.acc_b = acc_b if use_acc else None   # this application's backward accumulator (sum du, sum du*xhat), zeroed
        ctx.in_prod = cfg.in_prod if (need_dx and not depthwise and c_orig == Cc) else None
        ctx.in_lazy = cfg.in_lazy   # x (saved below) is then the producer's RAW output: backward transforms it on load too
        cfg.prod = ctx.prod = None
        if use_acc and _BN_TAIL and cfg.act in _TAIL_ACTS and not (cfg.res_pre and residual is not None) and any(ctx.needs_input_grad):  # (grad mode is off inside forward)
            cfg.prod = ctx.prod = ProdInfo(y, Kp, stats, cfg.act, cfg.act_param, acc_b, K, 0, K, (N, K, P, Q))
        ctx.res_pre = bool(cfg.res_pre and residual is not None and not isinstance(z, tuple))
        ctx.split2 = split2 is not None
        if split2 is not None:
            cfg.prod = ctx.prod = None
            ctx.save_for_backward(x, y, stats, weight, z[1])   # y's channels [k1, K) were never written: the raw half lives in the concat slice
        elif ctx.res_pre:
            ctx.save_for_backward(x if image is None else image, y, stats, weight, z)   # the activation's derivative is taken from the OUTPUT's sign
        else:
            ctx.save_for_backward(x if image is None else image, y, stats, weight)
        ctx.image_planes = planes if image is not None else 0
        return z

    @staticmethod
    def backward(ctx, dz):
        cfg = ctx.cfg
        if cfg.dx_link is not None:
            cfg.dx_link.consumed = True   # (ParkGrad: gradients arriving from now on go back to autograd)
        N, Cc, H, W, K, R, S, P, Q, Kp, x_ld, Cg = ctx.geom
        M = N * P * Q
        st = _stream()
        dz, dz_ld = as_nhwc(dz)
        act, act_param = cfg.act, cfg.act_param
        tail_sums_done = False
        if ctx.res_pre:
            # z = act(bn(y) + r): first du = dz * act'(z) from the saved output (shared by the BN branch and the residual),
            # then the BN backward below sees a layer WITHOUT activation
            x, y, stats, weight, zout = ctx.saved_tensors
            zout, z_ld = as_nhwc(zout)
            du = empty_nhwc(N, K, P, Q, x.device)
            acc_t = ctx.acc_b if ctx.train_bn else None
            if (_TAIL_MERGE and acc_t is not None and not getattr(ctx, "_acc_b_used", False) and Kp == K and act in (L.ACT_NONE, L.ACT_RELU, L.ACT_LEAKY)
                    and dz_ld % 8 == 0 and z_ld % 8 == 0 and dz.data_ptr() % 16 == 0 and zout.data_ptr() % 16 == 0):
                # round 5: the mask pass and the BN-backward reduction in ONE pass (read dz, z, y; write du; sums into the accumulator)
                _timed_ew("bn_tail_bwd_sums(colreduce_kernel<3>)", 8.0 * M * K, "cvhip_bn_tail_bwd_sums_acc", dz.data_ptr(), dz_ld, zout.data_ptr(), z_ld,
                          y.data_ptr(), Kp, du.data_ptr(), K, M, K, stats[0].data_ptr(), stats[1].data_ptr(), act, act_param, acc_t.data_ptr(), K, st)
                tail_sums_done = True
            else:
                L.call("cvhip_bn_act_bwd_apply", dz.data_ptr(), dz_ld, zout.data_ptr(), z_ld, du.data_ptr(), K, M, K, None, None, None, None,
                       None, None, act, act_param, st)
            dz, dz_ld, act = du, K, L.ACT_NONE
        else:
            x, y, stats, weight = ctx.saved_tensors
        dev = x.device
        need_dx, need_dw, need_db, need_dg, need_dbeta = (ctx.needs_input_grad[i] for i in range(5))
        dgamma = dbeta = dbias = None
        kv = K if Kp != K else 0
        cv = Cg if (not ctx.depthwise and Cg != Cc) else 0
        pointwise = cfg.has_bn or act != L.ACT_NONE
        arena = cfg.arena
        fused = pointwise and _bwd1x1_ok(ctx, cfg, x, need_dx, need_dw, need_db, ((dz, dz_ld),))
        acc_b = ctx.acc_b if ctx.train_bn else None
        lzi = getattr(ctx, "in_lazy", None)
        if lzi is not None and not (fused and acc_b is not None and K <= 128):
            # (not expected: the wrapper admits a lazy input only when this layer's backward is the fused 1x1 kernel) the pass the forward saved
            x = _materialize_tmp(x, x_ld, lzi)
            lzi = None
        direct_bn = arena is not None and cfg.gg is not None and cfg.gbeta is not None
        if acc_b is not None:
            # (sum du, sum du*xhat) into the layer's accumulator — unless the kernel that wrote dz already folded them in
            # (ProdInfo); the consumer below folds the accumulator and stores dgamma / dbeta
            if getattr(ctx, "_acc_b_used", False):
                zero_fill(acc_b)   # backward(retain_graph=True) a second time: the sums of the first pass must not be added twice
            ctx._acc_b_used = True
            have = _sums_already_done(getattr(ctx, "prod", None), dz)
            if have == "dirty":
                zero_fill(acc_b)
                have = False
            if tail_sums_done:
                have = True   # (the residual-tail pass above reduced while it masked)
            if not have:
                _timed_ew("bn_act_bwd_sums(colreduce_kernel<1>)", 4.0 * M * K, "cvhip_bn_act_bwd_sums_acc", dz.data_ptr(), dz_ld, y.data_ptr(), Kp, M, K,
                          stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), act, act_param, acc_b.data_ptr(), K, st)
            if direct_bn:
                g_out, b_out, accum = cfg.gg, cfg.gbeta, 1
            else:
                g_out = dgamma = torch.empty((K,), dtype=torch.float32, device=dev)
                b_out = dbeta = torch.empty((K,), dtype=torch.float32, device=dev)
                accum = 0
        if (acc_b is not None and _STEM_BN and not fused and not need_dx and need_dw and pointwise and not ctx.depthwise and not _DETERMINISTIC
                and Kp == K and dz_ld == K and dz.data_ptr() % 16 == 0 and Cc == 8 and x_ld == 8 and act in _LAZY_ACTS and not (ctx.has_bias and need_db)
                and lzi is None):
            # image stem: no input gradient, so the weight gradient is the only consumer of dy — the BN + activation backward rides in
            # the stem weight-gradient kernel's loads (cvhip_conv2d_wgrad_stem_bn): no apply pass, no dy tensor
            dw = _stem_wgrad_bn(ctx, cfg, x, y, weight, dz, stats, acc_b, g_out, b_out, accum, act, act_param)
            if dw is not False:
                if direct_bn:
                    need_dg = need_dbeta = False
                    for i in cfg.idx_bn:
                        arena.mark_ready(i)
                dres = dz if ctx.has_res else None
                return None, dw, None, (dgamma if need_dg else None), (dbeta if need_dbeta else None), None, None, dres, None
        if fused and acc_b is not None:
            dx, dw = _bwd1x1(ctx, cfg, x, y, weight, ((dz, dz_ld),), K, stats, True, None, None, act, act_param, acc_b, g_out, b_out, accum, xin=lzi)
            if direct_bn:
                need_dg = need_dbeta = False
                for i in cfg.idx_bn:
                    arena.mark_ready(i)
            dres = dz if ctx.has_res else None
            if dres is not None and cfg.res_link is not None and cfg.res_link.ok:
                cfg.res_link.g = dres
                dres = None
            return dx, dw, None, (dgamma if need_dg else None), (dbeta if need_dbeta else None), None, None, dres, None
        if fused:
            # 1x1 layer: BN/activation backward applied on load inside ONE dgrad + wgrad kernel (no dy tensor, no separate passes)
            ag = ab = None
            if ctx.train_bn:
                rows = _colreduce_rows(M, K)
                partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
                _timed_ew("bn_act_bwd_sums(colreduce_kernel<1>)", 4.0 * M * K, "cvhip_bn_act_bwd_partial", dz.data_ptr(), dz_ld, y.data_ptr(), Kp, M, K, stats[2].data_ptr(),
                       stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), act, act_param,
                       partial.data_ptr(), st)
                dgamma = torch.empty((K,), dtype=torch.float32, device=dev)
                dbeta = torch.empty((K,), dtype=torch.float32, device=dev)
                direct_bn = arena is not None and cfg.gg is not None and cfg.gbeta is not None
                L.call("cvhip_bn_bwd_finalize", partial.data_ptr(), rows, K, dgamma.data_ptr(), dbeta.data_ptr(),
                       cfg.gg.data_ptr() if direct_bn else None, cfg.gbeta.data_ptr() if direct_bn else None, st)
                if direct_bn:
                    need_dg = need_dbeta = False
                    for i in cfg.idx_bn:
                        arena.mark_ready(i)
                ag, ab = (dgamma, dbeta) if cfg.sync is None else _sync_bwd_sums(dgamma, dbeta, cfg.sync)
            dx, dw = _bwd1x1(ctx, cfg, x, y, weight, ((dz, dz_ld),), K, stats, ctx.train_bn, ag, ab, act, act_param)
            dres = dz if ctx.has_res else None
            if dres is not None and cfg.res_link is not None and cfg.res_link.ok:
                cfg.res_link.g = dres
                dres = None
            return dx, dw, None, (dgamma if need_dg else None), (dbeta if need_dbeta else None), None, None, dres, None
        if pointwise:
            if Kp != K:  # pad channels must read as zero in dgrad/wgrad
                dy = zero_fill(torch.empty((N, P, Q, Kp), dtype=ACT_DTYPE, device=dev)).permute(0, 3, 1, 2)[:, :K]
            else:
                dy = empty_nhwc(N, K, P, Q, dev)
            if acc_b is not None:
                _timed_ew("bn_act_bwd_apply(ew_kernel<1>)", 6.0 * M * K, "cvhip_bn_act_bwd_apply_acc", dz.data_ptr(), dz_ld, y.data_ptr(), Kp, dy.data_ptr(), Kp, M, K,
                          stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), acc_b.data_ptr(), K,
                          g_out.data_ptr(), b_out.data_ptr(), accum, act, act_param, st)
                if direct_bn:
                    need_dg = need_dbeta = False
                    for i in cfg.idx_bn:
                        arena.mark_ready(i)
            elif ctx.train_bn:
                rows = _colreduce_rows(M, K)
                partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
                _timed_ew("bn_act_bwd_sums(colreduce_kernel<1>)", 4.0 * M * K, "cvhip_bn_act_bwd_partial", dz.data_ptr(), dz_ld, y.data_ptr(), Kp, M, K, stats[2].data_ptr(),
                       stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), act, act_param,
                       partial.data_ptr(), st)
                dgamma = torch.empty((K,), dtype=torch.float32, device=dev)
                dbeta = torch.empty((K,), dtype=torch.float32, device=dev)
                L.call("cvhip_bn_bwd_finalize", partial.data_ptr(), rows, K, dgamma.data_ptr(), dbeta.data_ptr(),
                       cfg.gg.data_ptr() if direct_bn else None, cfg.gbeta.data_ptr() if direct_bn else None, st)
                if direct_bn:
                    need_dg = need_dbeta = False  # already accumulated into the gradient arena
                    for i in cfg.idx_bn:
                        arena.mark_ready(i)
                ag, ab = (dgamma, dbeta) if cfg.sync is None else _sync_bwd_sums(dgamma, dbeta, cfg.sync)
                _timed_ew("bn_act_bwd_apply(ew_kernel<1>)", 6.0 * M * K, "cvhip_bn_act_bwd_apply", dz.data_ptr(), dz_ld, y.data_ptr(), Kp, dy.data_ptr(), Kp, M, K,
                          stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(),
                       ag.data_ptr(), ab.data_ptr(), act, act_param, st)
            else:
                sc = stats[2].data_ptr() if stats is not None else None
                sh = stats[3].data_ptr() if stats is not None else None
                L.call("cvhip_bn_act_bwd_apply", dz.data_ptr(), dz_ld, y.data_ptr(), Kp, dy.data_ptr(), Kp, M, K, sc, sh,
                       None, None, None, None, act, act_param, st)
            dy_ld = Kp
        else:
            dy, dy_ld = dz, dz_ld
            if dy_ld % 8 != 0 or dy.data_ptr() % 16 != 0 or (Kp != K and dy_ld < Kp):
                # repack into a 16-byte-vectorisable pitch / base address (a channel slice of a cat gradient can start at any
                # element) with zeroed pad channels
                buf = zero_fill(torch.empty((N, P, Q, Kp), dtype=ACT_DTYPE, device=dev))
                L.call("cvhip_copy2d", dy.data_ptr(), dy_ld, buf.data_ptr(), Kp, M, K, st)
                dy, dy_ld = buf.permute(0, 3, 1, 2)[:, :K], Kp
        dx, dw, dbias = _conv_grads(ctx, x, weight, dy, dy_ld, need_dx, need_dw, need_db)
        dres = dz if ctx.has_res else None
        if dres is not None and cfg.res_link is not None and cfg.res_link.ok:
            cfg.res_link.g = dres   # handed to the skip connection's first layer (GradLink); nothing for autograd to add
            dres = None
        if dx is not None and ctx.c_orig != Cc:
            dx = dx[:, :ctx.c_orig]  # channel slice of the padded gradient buffer (an NHWC view with ld = round8(C))
        return dx, dw, dbias, (dgamma if need_dg else None), (dbeta if need_dbeta else None), None, None, dres, None


def _lazy_consumer_ok(x, K, R, S, has_bias, cfg, lz):
    """True when THIS layer can read the lazy tensor x as it is: forward on the streaming 1x1 kernel (transform on load), backward
    on the fused 1x1 kernel (which transforms the rows it stages for the weight gradient)"""
    if not _LAZY or _DETERMINISTIC or not torch.is_grad_enabled() or not x.requires_grad:
        return False
    if cfg.groups != 1 or (R, S) != (1, 1) or tuple(cfg.stride) != (1, 1) or tuple(cfg.pad) != (0, 0) or has_bias:
        return False
    if not (cfg.has_bn and cfg.bn_training) or cfg.sync is not None or not _BN_ACC or K > 128 or K % 8:
        return False
    if lz.act not in _LAZY_ACTS or x.dim() != 4 or x.dtype != ACT_DTYPE:
        return False
    ld = nhwc_ld(x)
    N, Cc, H, W = x.shape
    if ld is None or ld % 8 or x.data_ptr() % 16 or not (0 <= lz.lo < lz.hi <= Cc) or lz.lo % 8 or lz.hi % 8 or lz.scale.numel() != lz.hi - lz.lo:
        return False
    return lazy_edge_ok(N, Cc, H, W, K, lz.act, ld)


def lazy_edge_ok(N, Cc, H, W, K, act, ld=None):
    """Would a training-mode 1x1 Conv-BN-act layer with K outputs read an (N, Cc, H, W) LAZY input on load? Kernel availability
    (streaming 1x1 forward with a prologue, fused 1x1 backward: row-count policies included) AND the measured cost of the on-load
    transform (profiles/r05_lazy_*: an on-load SiLU costs the consumer's VALU about what the stand-alone pass costs in HBM time; net
    gain with 64- and 128-channel inputs, net LOSS with 32-channel inputs — 32 -> 32 @160x160: +23 us forward, +19 us backward against
    a 40 us pass). Producers ask before they skip their apply pass, so that no layer stays raw for a consumer that would only
    materialise it."""
    if not _LAZY or _DETERMINISTIC or not _BN_ACC or act not in _LAZY_ACTS:
        return False
    if Cc % 8 or K % 8 or K > 128 or (act == L.ACT_SILU and Cc < 64):
        return False
    desc = conv_desc(N, Cc, H, W, K, 1, 1, (1, 1), (0, 0), (1, 1), 1, Cc if ld is None else ld, K)
    lib = L.load()
    return bool(lib.cvhip_conv1x1_stream_prologue_ok(C.byref(desc), 1)) and bool(lib.cvhip_conv1x1_bwd_fused_ok(C.byref(desc)))


def _admit_lazy(x, residual, K, R, S, has_bias, cfg):
    """Decide how this layer reads a lazy input / residual: on load (cfg.in_lazy / cfg.res_lazy) or through ops.materialize"""
    cfg.in_lazy = cfg.res_lazy = None
    lz = lazy_of(x)
    if lz is not None:
        if _lazy_consumer_ok(x, K, R, S, has_bias, cfg, lz):
            cfg.in_lazy = lz
        else:
            x = materialize(x)
    rl = lazy_of(residual)
    if rl is not None:
        ok = (_LAZY and cfg.has_bn and cfg.bn_training and cfg.sync is None and _BN_ACC and not _DETERMINISTIC and cfg.groups == 1 and not cfg.res_pre
              and K % 8 == 0 and K <= _BN_ACC_MAX_C and rl.act == cfg.act and rl.ap == cfg.act_param and rl.act in (L.ACT_RELU, L.ACT_LEAKY, L.ACT_SILU)
              and torch.is_grad_enabled() and rl.full(K) and rl.scale.numel() == K and nhwc_ld(residual) is not None and nhwc_ld(residual) % 8 == 0
              and residual.data_ptr() % 16 == 0)
        if ok:
            cfg.res_lazy = rl
        else:
            residual = materialize(residual)
    return x, residual


def conv_bn_act(x, weight, bias, gamma, beta, running_mean, running_var, residual, cfg):
    cfg.in_prod = _take_prod(x)
    cfg.no_grad = not torch.is_grad_enabled()   # (inside Function.forward grad mode is always off: note it here)
    x, residual = _admit_lazy(x, residual, weight.shape[0], weight.shape[2], weight.shape[3], bias is not None, cfg)
    z = ConvBnAct.apply(x, weight, bias, gamma, beta, running_mean, running_var, residual, cfg)
    if cfg.prod is not None and torch.is_tensor(z):
        z._hip_prod = cfg.prod
    if cfg.lazy_made is not None and torch.is_tensor(z):
        st4, off, kh = cfg.lazy_made
        z = tag_lazy(z, LazyAct(st4[2][off:off + kh], st4[3][off:off + kh], cfg.act, cfg.act_param))
    return z


class ConvBnActPair(torch.autograd.Function):
    """Two sibling 1x1 Conv-BN-act layers on the SAME input (CSP conv1 / conv2: yolo_modules.py:131-139) as ONE convolution with
    K1 + K2 output channels: x is read once by fprop and once by wgrad, dgrad produces the complete input gradient (no
    gradient-accumulation add), the BN+act pass covers both halves. Needs the two layers' parameters / running statistics /
    gradient slots ADJACENT in the flat arenas (arena.FlatTrainState places `hip_sibling_pairs()` that way); `pair_operands`
    returns None otherwise and the caller runs the two layers separately. Outputs are the two channel slices of one buffer;
    backward takes the two slice gradients from wherever they are (no re-assembly copy)."""

    @staticmethod
    def forward(ctx, x, wf, gf, bf, rmf, rvf, cfg, k1):
        z = ConvBnAct.forward(ctx, x, wf, None, gf, bf, rmf, rvf, None, cfg)
        ctx.k1 = k1
        if isinstance(z, tuple):  # cfg.out_split: the second half went straight into the caller's concat buffer
            return z
        return z[:, :k1], z[:, k1:]

    @staticmethod
    def backward(ctx, d1, d2):
        y1 = None
        if getattr(ctx, "split2", False):
            x, y, stats, weight, yh2 = ctx.saved_tensors
            y1 = as_nhwc(yh2, lazy_ok=True)   # (tensor, pitch): the second sibling's RAW output, in its concat slice
        else:
            x, y, stats, weight = ctx.saved_tensors
        cfg = ctx.cfg
        N, Cc, H, W, K, R, S, P, Q, Kp, x_ld, Cg = ctx.geom
        dev = x.device
        M = N * P * Q
        st = _stream()
        # per half: (address of its channel 0 in the raw output, pitch)
        yat = [(y.data_ptr(), Kp), (y.data_ptr() + 2 * ctx.k1, Kp) if y1 is None else (y1[0].data_ptr(), y1[1])]
        segs = []
        for d, kh in ((d1, ctx.k1), (d2, K - ctx.k1)):
            if d is None:
                d = zero_fill(empty_nhwc(N, kh, P, Q, dev))
            segs.append(as_nhwc(d))
        acc_b = ctx.acc_b if ctx.train_bn else None
        if acc_b is not None:
            # per-half sums into the pair's accumulator, then either ONE fused kernel or per-half apply passes that fold it themselves
            halves = list(zip(segs, (ctx.k1, K - ctx.k1), (0, ctx.k1)))
            pi = getattr(ctx, "prod", None)
            have = [(_sums_already_done(h, d) if pi is not None and pi.halves is not None else False)
                    for h, ((d, _), _, _) in zip(pi.halves if (pi is not None and pi.halves is not None) else (None, None), halves)]
            if "dirty" in have:   # sums were folded over a tensor that is not the gradient we received: start over
                zero_fill(acc_b)
                have = [False, False]
            for ((d, d_ld), kh, off), hv, (ya, ya_ld) in zip(halves, have, yat):
                if hv:
                    continue
                sc, sh, mean, invstd = (stats[i].data_ptr() + 4 * off for i in (2, 3, 0, 1))
                _timed_ew("bn_act_bwd_sums(colreduce_kernel<1>)", 4.0 * M * kh, "cvhip_bn_act_bwd_sums_acc", d.data_ptr(), d_ld, ya, ya_ld, M, kh,
                          sc, sh, mean, invstd, cfg.act, cfg.act_param, acc_b.data_ptr() + 8 * off, K, st)
            lzi = getattr(ctx, "in_lazy", None)
            if _bwd1x1_ok(ctx, cfg, x, ctx.needs_input_grad[0], True, False, segs) and (lzi is None or K <= 128):
                dx, _ = _bwd1x1(ctx, cfg, x, y, weight, segs, ctx.k1, stats, True, None, None, cfg.act, cfg.act_param, acc_b, cfg.gg, cfg.gbeta, 1, xin=lzi,
                                y1=y1)
            else:
                if lzi is not None:
                    x = _materialize_tmp(x, x_ld, lzi)
                dy = empty_nhwc(N, K, P, Q, dev)
                for ((d, d_ld), kh, off), (ya, ya_ld) in zip(halves, yat):
                    sc, sh, mean, invstd = (stats[i].data_ptr() + 4 * off for i in (2, 3, 0, 1))
                    _timed_ew("bn_act_bwd_apply(ew_kernel<1>)", 6.0 * M * kh, "cvhip_bn_act_bwd_apply_acc", d.data_ptr(), d_ld, ya, ya_ld,
                              dy.data_ptr() + 2 * off, Kp, M, kh, sc, sh, mean, invstd, acc_b.data_ptr() + 8 * off, K,
                              cfg.gg.data_ptr() + 4 * off, cfg.gbeta.data_ptr() + 4 * off, 1, cfg.act, cfg.act_param, st)
                dx, _, _ = _conv_grads(ctx, x, weight, dy, Kp, ctx.needs_input_grad[0], True, False)
            for i in cfg.idx_bn:
                cfg.arena.mark_ready(i)
            return dx, None, None, None, None, None, None, None
        if getattr(ctx, "in_lazy", None) is not None:
            x = _materialize_tmp(x, x_ld, ctx.in_lazy)   # (a lazy input is only admitted with the accumulator form: not reached)
        if ctx.train_bn and cfg.sync is None and _bwd1x1_ok(ctx, cfg, x, ctx.needs_input_grad[0], True, False, segs):
            # fused form: per-half BN sums, then ONE kernel for BN/act backward + dgrad + wgrad of both siblings
            dgamma = torch.empty((K,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((K,), dtype=torch.float32, device=dev)
            off = 0
            for (d, d_ld), kh in zip(segs, (ctx.k1, K - ctx.k1)):
                rows = _colreduce_rows(M, kh)
                partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, kh), dtype=torch.float32, device=dev)
                sc, sh, mean, invstd = (stats[i].data_ptr() + 4 * off for i in (2, 3, 0, 1))
                _timed_ew("bn_act_bwd_sums(colreduce_kernel<1>)", 4.0 * M * kh, "cvhip_bn_act_bwd_partial", d.data_ptr(), d_ld, y.data_ptr() + 2 * off, Kp, M, kh, sc, sh, mean, invstd,
                       cfg.act, cfg.act_param, partial.data_ptr(), st)
                L.call("cvhip_bn_bwd_finalize", partial.data_ptr(), rows, kh, dgamma.data_ptr() + 4 * off, dbeta.data_ptr() + 4 * off,
                       cfg.gg.data_ptr() + 4 * off, cfg.gbeta.data_ptr() + 4 * off, st)
                off += kh
            for i in cfg.idx_bn:
                cfg.arena.mark_ready(i)
            dx, _ = _bwd1x1(ctx, cfg, x, y, weight, segs, ctx.k1, stats, True, dgamma, dbeta, cfg.act, cfg.act_param)
            return dx, None, None, None, None, None, None, None
        dy = empty_nhwc(N, K, P, Q, dev)
        off = 0
        for (d, d_ld), kh in zip(segs, (ctx.k1, K - ctx.k1)):
            rows = _colreduce_rows(M, kh)
            partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, kh), dtype=torch.float32, device=dev)
            sc, sh, mean, invstd = (stats[i].data_ptr() + 4 * off for i in (2, 3, 0, 1))
            _timed_ew("bn_act_bwd_sums(colreduce_kernel<1>)", 4.0 * M * kh, "cvhip_bn_act_bwd_partial", d.data_ptr(), d_ld, y.data_ptr() + 2 * off, Kp, M, kh, sc, sh, mean, invstd,
                   cfg.act, cfg.act_param, partial.data_ptr(), st)
            dgamma = torch.empty((kh,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((kh,), dtype=torch.float32, device=dev)
            L.call("cvhip_bn_bwd_finalize", partial.data_ptr(), rows, kh, dgamma.data_ptr(), dbeta.data_ptr(),
                   cfg.gg.data_ptr() + 4 * off, cfg.gbeta.data_ptr() + 4 * off, st)
            _timed_ew("bn_act_bwd_apply(ew_kernel<1>)", 6.0 * M * kh, "cvhip_bn_act_bwd_apply", d.data_ptr(), d_ld, y.data_ptr() + 2 * off, Kp, dy.data_ptr() + 2 * off, Kp, M, kh,
                   sc, sh, mean, invstd, dgamma.data_ptr(), dbeta.data_ptr(), cfg.act, cfg.act_param, st)
            off += kh
        for i in cfg.idx_bn:
            cfg.arena.mark_ready(i)
        dx, _, _ = _conv_grads(ctx, x, weight, dy, Kp, ctx.needs_input_grad[0], True, False)
        return dx, None, None, None, None, None, None, None


def _adjacent(a, b):
    return (a is not None and b is not None and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
            and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size())


def pair_operands(conv1, bn1, conv2, bn2):
    """Fused (weight, gamma, beta, running_mean, running_var, grad views, arena indices) of two sibling 1x1 layers whose tensors
    sit back to back in the flat arenas, or None."""
    w1, w2 = conv1.weight, conv2.weight
    ar = getattr(w1, "_hip_arena", None)
    if ar is None or getattr(w2, "_hip_arena", None) is None or not torch.is_grad_enabled():
        return None
    if not (bn1.training and bn2.training and bn1.track_running_stats and bn2.track_running_stats):
        return None
    if bn1.momentum is None or bn1.momentum != bn2.momentum or bn1.eps != bn2.eps:
        return None
    k1, k2, cin = w1.shape[0], w2.shape[0], w1.shape[1]
    if k1 % 8 or k2 % 8 or cin % 8 or w2.shape[1] != cin or tuple(w1.shape[2:]) != (1, 1) or tuple(w2.shape[2:]) != (1, 1):
        return None
    flat = lambda t: t.detach().reshape(-1) if t.is_contiguous() else t.detach().permute(0, 2, 3, 1).reshape(-1)  # noqa: E731
    pairs = [(flat(w1), flat(w2)), (flat(w1._hip_grad), flat(w2._hip_grad))]
    for name in ("weight", "bias"):
        a, b = getattr(bn1, name), getattr(bn2, name)
        if a is None or b is None or getattr(a, "_hip_arena", None) is None or getattr(b, "_hip_arena", None) is None:
            return None
        pairs += [(a.detach(), b.detach()), (a._hip_grad, b._hip_grad)]
    pairs += [(bn1.running_mean, bn2.running_mean), (bn1.running_var, bn2.running_var)]
    if not all(_adjacent(a, b) for a, b in pairs):
        return None
    cat = lambda a, b: a.as_strided((a.numel() + b.numel(),), (1,))  # noqa: E731  (b follows a in the same storage)
    wf = cat(*pairs[0]).view(k1 + k2, 1, 1, cin).permute(0, 3, 1, 2).requires_grad_(True)   # OIHW shape, KRSC memory
    gw = cat(*pairs[1]).view(k1 + k2, 1, 1, cin).permute(0, 3, 1, 2)
    gf, gg = cat(*pairs[2]).requires_grad_(True), cat(*pairs[3])
    bf, gb = cat(*pairs[4]).requires_grad_(True), cat(*pairs[5])
    rmf, rvf = cat(*pairs[6]), cat(*pairs[7])
    idx_w = (w1._hip_arena[1], w2._hip_arena[1])
    idx_bn = (bn1.weight._hip_arena[1], bn1.bias._hip_arena[1], bn2.weight._hip_arena[1], bn2.bias._hip_arena[1])
    return wf, gf, bf, rmf, rvf, gw, gg, gb, ar[0], idx_w, idx_bn, k1


def conv_bn_act_pair(x, operands, cfg, out2=None):
    wf, gf, bf, rmf, rvf, gw, gg, gb, arena, idx_w, idx_bn, k1 = operands
    cfg.out_split = (k1, out2) if out2 is not None else None
    cfg.arena, cfg.gw, cfg.gg, cfg.gbeta, cfg.idx_w, cfg.idx_bn = arena, gw, gg, gb, idx_w, idx_bn
    cfg.in_prod = _take_prod(x)
    x, _ = _admit_lazy(x, None, wf.shape[0], 1, 1, False, cfg)
    z1, z2 = ConvBnActPair.apply(x, wf, gf, bf, rmf, rvf, cfg, k1)
    if cfg.lazy_made is not None:   # the first sibling's result is lazy: z1 is the channel slice [0, k1) of the pair's raw output
        st4, off, kh = cfg.lazy_made
        z1 = tag_lazy(z1, LazyAct(st4[2][off:off + kh], st4[3][off:off + kh], cfg.act, cfg.act_param))
    if cfg.lazy_made2 is not None:  # the second sibling's slice of the concat buffer holds its RAW output (split store)
        st4, off, kh = cfg.lazy_made2
        z2 = tag_lazy(z2, LazyAct(st4[2][off:off + kh], st4[3][off:off + kh], cfg.act, cfg.act_param))
    if cfg.prod is not None:
        kt = cfg.prod.kh
        cfg.prod.halves = (cfg.prod.half(0, k1), cfg.prod.half(k1, kt - k1))
        z1._hip_prod, z2._hip_prod = cfg.prod.halves
    return z1, z2


class BnAct(torch.autograd.Function):
    """z = act(bn(y)) (+ residual) for an arbitrary NHWC input (statistics by a separate reduction pass)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, residual, has_bn, training, momentum, eps, act, act_param, track, sync=None):
        y, y_ld = as_nhwc(y)
        N, K, P, Q = y.shape
        M = N * P * Q
        dev = y.device
        st = _stream()
        stats = None
        train_bn = has_bn and training
        if train_bn:
            _stats_epoch[0] += 1   # running statistics are about to be rewritten through raw pointers (cached eval scale / shift go stale)
        if train_bn:
            rows = _colreduce_rows(M, K)
            partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
            L.call("cvhip_bn_stats_partial", y.data_ptr(), M, K, y_ld, partial.data_ptr(), st)
            stats = torch.empty((4, K), dtype=torch.float32, device=dev)
            Mstat = M
            if sync is not None:
                partial, rows, Mstat = _sync_fwd_totals(partial, rows, K, M, sync)
            L.call("cvhip_bn_finalize", partial.data_ptr(), rows, K, Mstat, _ptr(gamma.detach().float() if gamma is not None else None),
                   _ptr(beta.detach().float() if beta is not None else None), _ptr(running_mean if track else None),
                   _ptr(running_var if track else None), float(momentum), float(eps), stats[0].data_ptr(), stats[1].data_ptr(),
                   stats[2].data_ptr(), stats[3].data_ptr(), st)
        elif has_bn:
            stats = torch.empty((4, K), dtype=torch.float32, device=dev)
            L.call("cvhip_bn_eval_scale_shift", K, _ptr(gamma.detach().float() if gamma is not None else None),
                   _ptr(beta.detach().float() if beta is not None else None), running_mean.data_ptr(), running_var.data_ptr(),
                   float(eps), stats[2].data_ptr(), stats[3].data_ptr(), st)
        res_ld = 0
        if residual is not None:
            residual, res_ld = as_nhwc(residual)
        z = empty_nhwc(N, K, P, Q, dev)
        L.call("cvhip_bn_act_fwd", y.data_ptr(), y_ld, z.data_ptr(), K, M, K, _ptr(stats[2]) if stats is not None else None,
               _ptr(stats[3]) if stats is not None else None, act, float(act_param), _ptr(residual), res_ld, st)
        ctx.meta = (N, K, P, Q, y_ld, train_bn, act, float(act_param), residual is not None)
        ctx.sync = sync if train_bn else None
        ctx.save_for_backward(y, stats)
        return z

    @staticmethod
    def backward(ctx, dz):
        y, stats = ctx.saved_tensors
        N, K, P, Q, y_ld, train_bn, act, ap, has_res = ctx.meta
        M = N * P * Q
        dev = y.device
        st = _stream()
        dz, dz_ld = as_nhwc(dz)
        dy = empty_nhwc(N, K, P, Q, dev)
        dgamma = dbeta = None
        if train_bn:
            rows = _colreduce_rows(M, K)
            partial = torch.empty((rows + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=dev)
            L.call("cvhip_bn_act_bwd_partial", dz.data_ptr(), dz_ld, y.data_ptr(), y_ld, M, K, stats[2].data_ptr(),
                   stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), act, ap, partial.data_ptr(), st)
            dgamma = torch.empty((K,), dtype=torch.float32, device=dev)
            dbeta = torch.empty((K,), dtype=torch.float32, device=dev)
            L.call("cvhip_bn_bwd_finalize", partial.data_ptr(), rows, K, dgamma.data_ptr(), dbeta.data_ptr(), None, None, st)
            ag, ab = (dgamma, dbeta) if ctx.sync is None else _sync_bwd_sums(dgamma, dbeta, ctx.sync)
            L.call("cvhip_bn_act_bwd_apply", dz.data_ptr(), dz_ld, y.data_ptr(), y_ld, dy.data_ptr(), K, M, K,
                   stats[2].data_ptr(), stats[3].data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), ag.data_ptr(),
                   ab.data_ptr(), act, ap, st)
        else:
            sc = stats[2].data_ptr() if stats is not None else None
            sh = stats[3].data_ptr() if stats is not None else None
            L.call("cvhip_bn_act_bwd_apply", dz.data_ptr(), dz_ld, y.data_ptr(), y_ld, dy.data_ptr(), K, M, K, sc, sh, None, None,
                   None, None, act, ap, st)
        return (dy, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None, None, None,
                dz if has_res else None, None, None, None, None, None, None, None, None)


def bn_act(y, gamma=None, beta=None, running_mean=None, running_var=None, residual=None, has_bn=True, training=True,
           momentum=0.1, eps=1e-5, act=L.ACT_NONE, act_param=0.0, track=True, sync=None):
    return BnAct.apply(y, gamma, beta, running_mean, running_var, residual, has_bn, training, momentum, eps, act, act_param, track, sync)


class MaxPool2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad):
        x, ld = as_nhwc(x)
        N, Cc, H, W = x.shape
        OH, OW = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        y = empty_nhwc(N, Cc, OH, OW, x.device)
        idx = torch.empty((N, OH, OW, Cc), dtype=torch.uint8, device=x.device)
        L.call("cvhip_maxpool2d_fwd", x.data_ptr(), ld, y.data_ptr(), Cc, idx.data_ptr(), N, Cc, H, W, k, stride, pad, _stream())
        ctx.meta = (N, Cc, H, W, k, stride, pad)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        N, Cc, H, W, k, stride, pad = ctx.meta
        dy, ld = as_nhwc(dy)
        dx = empty_nhwc(N, Cc, H, W, dy.device)
        L.call("cvhip_maxpool2d_bwd", dy.data_ptr(), ld, idx.data_ptr(), dx.data_ptr(), Cc, N, Cc, H, W, k, stride, pad, 0, _stream())
        return dx, None, None, None


class SppfChain(torch.autograd.Function):
    """SPPF's pooling chain + concat (yolo_modules.py:186-194: x, m(x), m(m(x)), m(m(m(x))) concatenated) on ONE buffer.

    `x0` is the [:, :c] slice of an (N, 4c, H, W) NHWC buffer that the 1x1 conv in front already wrote (its `out=`); the three stride-1
    pools write their outputs straight into the other three slices — no concat copy. Backward walks the chain on the incoming concat
    gradient IN PLACE: slice j receives the pool gradient of slice j + 1 through cvhip_maxpool2d_bwd's accumulate form, so the three
    gradient-accumulation adds (ops.Fanout) and their temporaries disappear; the first slice is returned as the gradient of x0."""

    @staticmethod
    def forward(ctx, x0, k):
        x0v, ld = as_nhwc(x0)
        N, c, H, W = x0v.shape
        if ld != 4 * c:
            raise L.CvhipError("SppfChain: x0 must be the first channel slice of an (N, 4c, H, W) NHWC buffer")
        pad = k // 2
        st = _stream()
        idx = torch.empty((3, N, H, W, c), dtype=torch.uint8, device=x0v.device)
        esz = x0v.element_size()
        for j in range(3):
            L.call("cvhip_maxpool2d_fwd", x0v.data_ptr() + j * c * esz, ld, x0v.data_ptr() + (j + 1) * c * esz, ld, idx[j].data_ptr(),
                   N, c, H, W, k, 1, pad, st)
        ctx.meta = (N, c, H, W, k, pad)
        ctx.save_for_backward(idx)
        return x0v.as_strided((N, 4 * c, H, W), (H * W * 4 * c, 1, W * 4 * c, 4 * c))   # fresh alias of the whole buffer (as ops.Cat does)

    @staticmethod
    def backward(ctx, d):
        (idx,) = ctx.saved_tensors
        N, c, H, W, k, pad = ctx.meta
        d, ld = as_nhwc(d)
        if ld != 4 * c or not d.is_contiguous(memory_format=torch.channels_last):
            d = d.contiguous(memory_format=torch.channels_last).clone()
            d, ld = as_nhwc(d)
        elif not _OWNED_BACKWARD[0]:
            # autograd's contract: a Function must not modify the gradient it receives (tensor hooks, retain_grad() on the concat or on
            # conv2's input gradient would see slices 1..3 overwritten). Only a backward pass that the flat train state drives itself
            # (arena.FlatTrainState.backward: no user hooks between the engine's ops) walks the chain on the incoming tensor.
            d = d.clone(memory_format=torch.channels_last)
            d, ld = as_nhwc(d)
        st = _stream()
        esz = d.element_size()
        for j in (2, 1, 0):   # slice j += maxpool_bwd(slice j + 1)
            L.call("cvhip_maxpool2d_bwd", d.data_ptr() + (j + 1) * c * esz, ld, idx[j].data_ptr(), d.data_ptr() + j * c * esz, ld,
                   N, c, H, W, k, 1, pad, 1, st)
        return d[:, :c], None


def sppf_chain(x0, k):
    return SppfChain.apply(x0, int(k))


class SppParallel(torch.autograd.Function):
    """SPP's parallel pools + concat (yolo_modules.py:165-194 with a tuple of kernel sizes: x, m5(x), m9(x), m13(x) concatenated; YOLOX's
    backbone, yolox_csp_darknet.py) on ONE buffer, as SppfChain does for the chained form: `x0` is the [:, :c] slice of an
    (N, (1 + len(ks)) * c, H, W) NHWC buffer that the 1x1 conv in front already wrote; every pool reads slice 0 and writes its own slice —
    no concat copies. Backward adds every pool's gradient into slice 0 of the incoming concat gradient (cvhip_maxpool2d_bwd's accumulate
    form): no gradient-accumulation adds, no temporaries."""

    @staticmethod
    def forward(ctx, x0, ks):
        x0v, ld = as_nhwc(x0)
        N, c, H, W = x0v.shape
        nk = len(ks)
        if ld != (1 + nk) * c:
            raise L.CvhipError("SppParallel: x0 must be the first channel slice of an (N, (1 + len(ks)) * c, H, W) NHWC buffer")
        st = _stream()
        idx = torch.empty((nk, N, H, W, c), dtype=torch.uint8, device=x0v.device)
        esz = x0v.element_size()
        for j, k in enumerate(ks):
            L.call("cvhip_maxpool2d_fwd", x0v.data_ptr(), ld, x0v.data_ptr() + (j + 1) * c * esz, ld, idx[j].data_ptr(), N, c, H, W, int(k), 1, int(k) // 2, st)
        ctx.meta = (N, c, H, W, tuple(int(k) for k in ks))
        ctx.save_for_backward(idx)
        return x0v.as_strided((N, (1 + nk) * c, H, W), (H * W * (1 + nk) * c, 1, W * (1 + nk) * c, (1 + nk) * c))

    @staticmethod
    def backward(ctx, d):
        (idx,) = ctx.saved_tensors
        N, c, H, W, ks = ctx.meta
        nk = len(ks)
        d, ld = as_nhwc(d)
        if ld != (1 + nk) * c or not d.is_contiguous(memory_format=torch.channels_last):
            d = d.contiguous(memory_format=torch.channels_last).clone()
            d, ld = as_nhwc(d)
        elif not _OWNED_BACKWARD[0]:
            d = d.clone(memory_format=torch.channels_last)   # (autograd's contract: see SppfChain.backward)
            d, ld = as_nhwc(d)
        st = _stream()
        esz = d.element_size()
        for j, k in enumerate(ks):   # slice 0 += maxpool_bwd_k(slice j + 1)
            L.call("cvhip_maxpool2d_bwd", d.data_ptr() + (j + 1) * c * esz, ld, idx[j].data_ptr(), d.data_ptr(), ld, N, c, H, W, k, 1, k // 2, 1, st)
        return d[:, :c], None


def spp_parallel(x0, ks):
    return SppParallel.apply(x0, tuple(ks))


def max_pool2d(x, k, stride=None, pad=0):
    return MaxPool2d.apply(x, int(k), int(stride if stride is not None else k), int(pad))


class Upsample2xCat(torch.autograd.Function):
    """cat([nearest_up_x2(a), b], dim=1) in one pass; b optional."""

    @staticmethod
    def forward(ctx, a, b):
        a, ld_a = as_nhwc(a)
        N, Ca, Ha, Wa = a.shape
        Cb, ld_b = 0, 0
        if b is not None:
            b, ld_b = as_nhwc(b)
            Cb = b.shape[1]
            if b.shape[0] != N or b.shape[2] != 2 * Ha or b.shape[3] != 2 * Wa:
                raise L.CvhipError("upsample2x_cat: lateral tensor shape mismatch")
        out = empty_nhwc(N, Ca + Cb, 2 * Ha, 2 * Wa, a.device)
        L.call("cvhip_upsample2x_cat_fwd", a.data_ptr(), ld_a, Ca, _ptr(b), ld_b, Cb, out.data_ptr(), Ca + Cb, N, Ha, Wa, _stream())
        ctx.meta = (N, Ca, Cb, Ha, Wa)
        return out

    @staticmethod
    def backward(ctx, dout):
        N, Ca, Cb, Ha, Wa = ctx.meta
        dout, ld = as_nhwc(dout)
        da = empty_nhwc(N, Ca, Ha, Wa, dout.device)
        L.call("cvhip_upsample2x_bwd", dout.data_ptr(), ld, da.data_ptr(), Ca, Ca, N, Ha, Wa, _stream())
        db = dout[:, Ca:] if Cb else None  # channel-slice view: no copy
        return da, db


def upsample2x_cat(a, b=None):
    return Upsample2xCat.apply(a, b)


class Cat(torch.autograd.Function):
    """Channel concat: copies each input into its slice of one NHWC buffer; backward hands out slice views. Inputs that
    already ARE their slice of the destination (producers called with `out=`) are not copied; `into` names the destination
    when only some inputs were produced in place."""

    @staticmethod
    def forward(ctx, into, *xs):
        lzs = [lazy_of(x) for x in xs]
        xs = [as_nhwc(x, lazy_ok=True) for x in xs]
        N, _, H, W = xs[0][0].shape
        Ct = sum(x.shape[1] for x, _ in xs)
        ctx.sizes = [x.shape[1] for x, _ in xs]
        _cat_lazy[0] = None
        # a lazy input (RAW conv output + pending BN/act) may stay lazy when it already IS its slice of the destination, its whole channel
        # range is lazy and it is the only such input (consumers transform ONE channel range on load); every other lazy input is activated
        # on its way into the destination (the copy this op makes anyway). Raw data is never overwritten: its producer's backward reads it.
        where = None
        if into is not None:
            where = _check_out(into, N, Ct, H, W)[0].data_ptr()
        elif all(ld == Ct for _, ld in xs) and all(b[0].data_ptr() == a[0].data_ptr() + 2 * a[0].shape[1] for a, b in zip(xs, xs[1:])):
            where = xs[0][0].data_ptr()
        offs = [sum(ctx.sizes[:i]) for i in range(len(xs))]
        placed = [where is not None and ld == Ct and x.data_ptr() == where + 2 * o for (x, ld), o in zip(xs, offs)]
        lazy_placed = [i for i, lz in enumerate(lzs) if lz is not None and placed[i]]
        if len(lazy_placed) > 1 or any(not lzs[i].full(ctx.sizes[i]) or offs[i] % 8 or ctx.sizes[i] % 8 for i in lazy_placed):
            into, where, placed, lazy_placed = None, None, [False] * len(xs), []   # (rare) a fresh buffer: everything is copied / activated into it
        if lazy_placed:
            i = lazy_placed[0]
            lz = lzs[i]
            _cat_lazy[0] = LazyAct(lz.scale, lz.shift, lz.act, lz.ap, offs[i], offs[i] + ctx.sizes[i])
        if into is None and where is not None and all(placed):
            # every producer wrote straight into consecutive channel slices of one buffer: nothing to copy
            return xs[0][0].as_strided((N, Ct, H, W), (H * W * Ct, 1, W * Ct, Ct))
        if into is not None:
            out, ld_o = _check_out(into, N, Ct, H, W)
            if ld_o != Ct:
                raise L.CvhipError("cat(into=): destination must be a dense NHWC buffer")
            out = out.as_strided((N, Ct, H, W), (H * W * Ct, 1, W * Ct, Ct))  # fresh alias (the result must not BE an argument)
        else:
            out = empty_nhwc(N, Ct, H, W, xs[0][0].device)
        st = _stream()
        off = 0
        M = N * H * W
        for (x, ld), lz in zip(xs, lzs):
            Cc = x.shape[1]
            if not (ld == Ct and x.data_ptr() == out.data_ptr() + 2 * off):
                if lz is None or not lz.full(Cc):
                    L.call("cvhip_copy2d", x.data_ptr(), ld, out.data_ptr() + 2 * off, Ct, M, Cc, st)
                if lz is not None:   # activate the lazy channel range on its way into the destination
                    kh = lz.hi - lz.lo
                    _timed_ew("bn_act_fwd(ew_kernel<0>)", 4.0 * M * kh, "cvhip_bn_act_fwd", x.data_ptr() + 2 * lz.lo, ld, out.data_ptr() + 2 * (off + lz.lo), Ct,
                              M, kh, lz.scale.data_ptr(), lz.shift.data_ptr(), lz.act, lz.ap, None, 0, st)
            off += Cc
        return out

    @staticmethod
    def backward(ctx, dout):
        outs, off = [], 0
        for c in ctx.sizes:
            outs.append(dout[:, off:off + c])
            off += c
        return (None,) + tuple(outs)


class _CatLazy(__import__("threading").local):
    """what Cat.forward found out about lazy slices, handed to `cat` on the SAME thread right after apply returns (a Function cannot
    return a python object): a thread-local list, so that two threads assembling models cannot see each other's tag"""

    def __init__(self):
        self.v = [None]

    def __getitem__(self, i):
        return self.v[i]

    def __setitem__(self, i, x):
        self.v[i] = x


_cat_lazy = _CatLazy()


def cat(xs, into=None):
    out = Cat.apply(into, *xs)
    if _cat_lazy[0] is not None:
        lz, _cat_lazy[0] = _cat_lazy[0], None
        out = tag_lazy(out, lz)
    return out


class Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, la = as_nhwc(a)
        b, lb = as_nhwc(b)
        N, Cc, H, W = a.shape
        out = empty_nhwc(N, Cc, H, W, a.device)
        L.call("cvhip_add2d", a.data_ptr(), la, b.data_ptr(), lb, out.data_ptr(), Cc, N * H * W, Cc, _stream())
        return out

    @staticmethod
    def backward(ctx, d):
        return d, d


def add(a, b):
    return Add.apply(a, b)


class Fanout(torch.autograd.Function):
    """A tensor with several consumers (backbone outputs that feed the next stage AND the neck, SPPF's chained pools, the neck's side
    outputs): autograd would sum the arriving gradients with an ATen add kernel per extra consumer. This op hands every consumer its own
    alias and sums the gradients with the engine's own kernel (cvhip_add2d) — no ATen kernel in the step."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc, la = as_nhwc(gs[0])
        for g in gs[1:]:
            b, lb = as_nhwc(g)
            N, Cc, H, W = acc.shape
            out = empty_nhwc(N, Cc, H, W, acc.device)
            L.call("cvhip_add2d", acc.data_ptr(), la, b.data_ptr(), lb, out.data_ptr(), Cc, N * H * W, Cc, _stream())
            acc, la = out, Cc
        return acc, None


_FANOUT = __import__("os").environ.get("CVHIP_FANOUT", "1") != "0"   # 0: leave the gradient sums to autograd (A/B switch)
_FANOUT_LINK = __import__("os").environ.get("CVHIP_FANOUT_LINK", "1") != "0"   # 0: every fan-out sums its gradients with cvhip_add2d (A/B switch)


class ParkGrad(torch.autograd.Function):
    """The SIDE alias of a two-consumer fan-out whose MAIN consumer is a Hip convolution (round 5): instead of summing the two arriving
    gradients with an add pass (read, read, write), the side branch's gradient is parked in a GradLink and the main consumer's dgrad
    kernel adds it in its epilogue (cvhip_conv2d_dgrad_add: one extra read). Correct for any execution order: if the main consumer's
    backward has already started (link.consumed) — or it cannot fold an addend (link.ok unset) — the gradient is returned to autograd,
    which sums it as usual. In the detectors the order is fixed by data dependence: the side consumers (neck concats, detect
    convolutions) sit downstream of the main consumer's own outputs, so their gradients exist before its backward can run."""

    @staticmethod
    def forward(ctx, x, link):
        ctx.link = link
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        link = ctx.link
        if g is None or not link.ok or link.consumed or link.g is not None:
            return g, None
        link.g = g
        return None, None


def fanout_linked(x):
    """(main alias, side alias or None, link) of a tensor with two consumers, the first of them a Hip conv module that takes
    `dx_link=link`. Protocol: call the main consumer FIRST, then `side = fanout_side(x, side, link)` for the other one — the side
    alias's autograd node is then the younger of the two, so whenever both are ready the engine parks the side gradient before the main
    consumer's backward runs. link None (nothing records gradients, or CVHIP_FANOUT_LINK=0): the aliases of ops.fanout, whose backward
    sums the two gradients with cvhip_add2d."""
    if not (_FANOUT and _FANOUT_LINK and torch.is_grad_enabled() and x.requires_grad) or nhwc_ld(x) is None:
        a, b2 = fanout(x, 2)
        return a, b2, None
    return x, None, GradLink()


def fanout_side(x, side, link):
    """the side alias of fanout_linked, to be called after the main consumer's forward"""
    return side if link is None else ParkGrad.apply(x, link)


def fanout(x, n=2):
    """n aliases of x for n consumers (training only; a plain tuple of x otherwise)"""
    if n < 2 or not _FANOUT or not (torch.is_grad_enabled() and x.requires_grad) or nhwc_ld(x) is None:
        return (x,) * n
    return Fanout.apply(x, n)


class AddAct(torch.autograd.Function):
    """out = act(a + b) — ResNet bottleneck tail. Backward recomputes act' from the saved OUTPUT (valid for the
    activations whose derivative is a function of the output's sign: ReLU / LeakyReLU)."""

    @staticmethod
    def forward(ctx, a, b, act, act_param, link=None):
        ctx.link = link
        if act not in (L.ACT_RELU, L.ACT_LEAKY, L.ACT_NONE):
            raise L.CvhipError("add_act supports none / ReLU / LeakyReLU")
        a, la = as_nhwc(a)
        b, lb = as_nhwc(b)
        N, Cc, H, W = a.shape
        out = empty_nhwc(N, Cc, H, W, a.device)
        L.call("cvhip_add_act_fwd", a.data_ptr(), la, b.data_ptr(), lb, out.data_ptr(), Cc, N * H * W, Cc, act, float(act_param), _stream())
        ctx.meta = (N, Cc, H, W, act, float(act_param))
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dz):
        (out,) = ctx.saved_tensors
        N, Cc, H, W, act, ap = ctx.meta
        if act == L.ACT_NONE:
            du = dz
        else:
            dz, ld = as_nhwc(dz)
            du = empty_nhwc(N, Cc, H, W, dz.device)
            L.call("cvhip_bn_act_bwd_apply", dz.data_ptr(), ld, out.data_ptr(), Cc, du.data_ptr(), Cc, N * H * W, Cc, None, None, None, None,
                   None, None, act, ap, _stream())
        if ctx.link is not None and ctx.link.ok and ctx.needs_input_grad[1]:
            ctx.link.g = du      # the identity branch's gradient travels through the GradLink (no accumulation add)
            return du, None, None, None, None
        return du, du, None, None, None


def add_act(a, b, act=L.ACT_RELU, act_param=0.0, link=None):
    return AddAct.apply(a, b, act, act_param, link)


class ScaleNC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x, ld = as_nhwc(x)
        N, Cc, H, W = x.shape
        scale = scale.float().contiguous()
        y = empty_nhwc(N, Cc, H, W, x.device)
        L.call("cvhip_scale_nc", x.data_ptr(), ld, scale.data_ptr(), y.data_ptr(), Cc, N, Cc, H * W, _stream())
        ctx.save_for_backward(scale)
        ctx.meta = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        N, Cc, H, W = ctx.meta
        dy, ld = as_nhwc(dy)
        dx = empty_nhwc(N, Cc, H, W, dy.device)
        L.call("cvhip_scale_nc", dy.data_ptr(), ld, scale.data_ptr(), dx.data_ptr(), Cc, N, Cc, H * W, _stream())
        return dx, None


class ChannelScale(torch.autograd.Function):
    """y[n,c,h,w] = x[n,c,h,w] * s[n,c]  with gradients to both (SE / attention-refinement gating:
    src/models/necks/seg/stdc_neck.py:53-58,110-114). ds is a per-(n,c) reduction over the pixels."""

    @staticmethod
    def forward(ctx, x, s):
        x, ld = as_nhwc(x)
        N, Cc, H, W = x.shape
        sf = s.reshape(N, Cc).float().contiguous()
        y = empty_nhwc(N, Cc, H, W, x.device)
        L.call("cvhip_scale_nc", x.data_ptr(), ld, sf.data_ptr(), y.data_ptr(), Cc, N, Cc, H * W, _stream())
        ctx.save_for_backward(x, sf)
        ctx.s_shape, ctx.s_dtype = tuple(s.shape), s.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sf = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dy, ld = as_nhwc(dy)
        dx = ds = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(N, Cc, H, W, dy.device)
            L.call("cvhip_scale_nc", dy.data_ptr(), ld, sf.data_ptr(), dx.data_ptr(), Cc, N, Cc, H * W, _stream())
        if ctx.needs_input_grad[1]:
            x2, ldx = as_nhwc(x)
            if Cc % 8 == 0 and ld % 8 == 0 and ldx % 8 == 0 and dy.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0:
                dsf = torch.empty((N, Cc), dtype=torch.float32, device=dy.device)
                L.call("cvhip_channel_scale_bwd_ds", dy.data_ptr(), ld, x2.data_ptr(), ldx, dsf.data_ptr(), N, Cc, H * W, _stream())
                ds = dsf.reshape(ctx.s_shape).to(ctx.s_dtype)
            else:
                ds = (dy.float() * x.float()).sum(dim=(2, 3)).reshape(ctx.s_shape).to(ctx.s_dtype)
        return dx, ds


def channel_scale(x, s):
    return ChannelScale.apply(x, s)


def dropout2d(x, p, training=True):
    """nn.Dropout2d: whole channels of a sample are zeroed with probability p, survivors scaled by 1/(1-p)."""
    if not training or p <= 0.0:
        return x
    N, Cc = x.shape[0], x.shape[1]
    mask = (torch.rand((N, Cc), device=x.device) >= p).float() * (1.0 / (1.0 - p))
    return ScaleNC.apply(x, mask)


class SegCrossEntropy(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index, reduction='mean') on NHWC bf16 logits (N,C,H,W) and int64 targets (N,H,W)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits, ld = as_nhwc(logits)
        N, Cc, H, W = logits.shape
        M = N * H * W
        target = target.long().contiguous()
        lib = L.load()
        partial = torch.empty((2 * lib.cvhip_seg_ce_rows(M),), dtype=torch.float32, device=logits.device)
        out2 = torch.empty((2,), dtype=torch.float32, device=logits.device)
        L.call("cvhip_seg_ce_fwd", logits.data_ptr(), ld, target.data_ptr(), M, Cc, int(ignore_index), partial.data_ptr(), out2.data_ptr(), _stream())
        ctx.meta = (N, Cc, H, W, ld, int(ignore_index))
        ctx.save_for_backward(logits, target, out2)
        return out2[0].clone()

    @staticmethod
    def backward(ctx, g):
        logits, target, out2 = ctx.saved_tensors
        N, Cc, H, W, ld, ign = ctx.meta
        Cp = _round8(Cc)
        gs = g.detach().float().reshape(1).contiguous()
        buf = torch.empty((N, H, W, Cp), dtype=ACT_DTYPE, device=logits.device)
        L.call("cvhip_seg_ce_bwd", logits.data_ptr(), ld, target.data_ptr(), N * H * W, Cc, ign, out2.data_ptr(), gs.data_ptr(),
               buf.data_ptr(), Cp, _stream())
        return buf.permute(0, 3, 1, 2)[:, :Cc], None, None


def seg_cross_entropy(logits, target, ignore_index=255):
    return SegCrossEntropy.apply(logits, target, ignore_index)


class SegCrossEntropyBilinear(torch.autograd.Function):
    """F.interpolate(logits, size=target.shape[-2:], mode='bilinear', align_corners) followed by nn.CrossEntropyLoss(ignore_index,
    'mean') — encoder_decoder.py:93-107 — as ONE pass forward and one backward (cvhip_seg_ce_bilinear_fwd / _bwd): the
    label-resolution logits and their gradient are never materialised."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index, align_corners):
        logits, ld = as_nhwc(logits)
        N, Cc, Hi, Wi = logits.shape
        target = target.long().contiguous()
        Ho, Wo = int(target.shape[-2]), int(target.shape[-1])
        lib = L.load()
        partial = torch.empty((2 * lib.cvhip_seg_ce_rows(N * Ho * Wo),), dtype=torch.float32, device=logits.device)
        out2 = torch.empty((2,), dtype=torch.float32, device=logits.device)
        L.call("cvhip_seg_ce_bilinear_fwd", logits.data_ptr(), ld, target.data_ptr(), N, Cc, Hi, Wi, Ho, Wo, int(align_corners), int(ignore_index),
               partial.data_ptr(), out2.data_ptr(), _stream())
        ctx.meta = (N, Cc, Hi, Wi, Ho, Wo, ld, int(ignore_index), int(align_corners))
        ctx.save_for_backward(logits, target, out2)
        return out2[0].clone()

    @staticmethod
    def backward(ctx, g):
        logits, target, out2 = ctx.saved_tensors
        N, Cc, Hi, Wi, Ho, Wo, ld, ign, ac = ctx.meta
        Cp = _round8(Cc)
        gs = g.detach().float().reshape(1).contiguous()
        buf = torch.empty((N, Hi, Wi, Cp), dtype=ACT_DTYPE, device=logits.device)
        L.call("cvhip_seg_ce_bilinear_bwd", logits.data_ptr(), ld, target.data_ptr(), N, Cc, Hi, Wi, Ho, Wo, ac, ign, out2.data_ptr(), gs.data_ptr(),
               buf.data_ptr(), Cp, _stream())
        return buf.permute(0, 3, 1, 2)[:, :Cc], None, None, None


class OhemCrossEntropyBilinear(torch.autograd.Function):
    """OhemCrossEntropyLoss2d (src/losses/seg/cross_entropy_loss.py:51-69) of logits that encoder_decoder.py:96 first resizes to the
    label size — on the LOW-resolution logits: the per-pixel losses come from cvhip_seg_ce_bilinear_fwd_px (the label-resolution
    logits never exist), the selection of the hard pixels is the fixed-shape formulation of segmentors.OhemCrossEntropyLoss2d on that
    fp32 vector (one topk, masked sums, where: no host sync), and the backward is cvhip_seg_ce_bilinear_bwd_px with the selection as
    per-pixel weights. forward(logits, target, thresh_nlog (device scalar -log(thresh)), min_kept, ignore_index, loss_weight)."""

    @staticmethod
    def forward(ctx, logits, target, thr, min_kept, ignore_index, loss_weight):
        logits, ld = as_nhwc(logits)
        N, Cc, Hi, Wi = logits.shape
        target = target.long().contiguous()
        Ho, Wo = int(target.shape[-2]), int(target.shape[-1])
        M = N * Ho * Wo
        if M <= min_kept:
            raise IndexError("OhemCrossEntropyLoss2d: %d pixels, min_kept %d (the reference indexes loss[min_kept])" % (M, min_kept))
        per = torch.empty((M,), dtype=torch.float32, device=logits.device)
        L.call("cvhip_seg_ce_bilinear_fwd_px", logits.data_ptr(), ld, target.data_ptr(), N, Cc, Hi, Wi, Ho, Wo, 0, int(ignore_index),
               per.data_ptr(), _stream())
        lw = float(loss_weight)
        loss = per * lw if lw != 1.0 else per
        v = torch.topk(loss, min_kept + 1, sorted=True).values[min_kept]   # = sorted(loss, descending)[min_kept]
        thr = thr.to(loss.dtype)
        above = (loss > thr).to(loss.dtype)
        n_above = above.sum().clamp_min(1.0)
        mean_above = (loss * above).sum() / n_above
        gt = (loss > v).to(loss.dtype)
        n_gt = gt.sum()
        mean_top = ((loss * gt).sum() + (min_kept - n_gt) * v) / float(min_kept)
        ties = (loss == v).to(loss.dtype)
        hard = v > thr
        # d out / d loss[m]: `above` / n_above on the threshold branch; 1 / min_kept for the pixels above v and the tied pixels' common
        # share (min_kept - n_gt) / n_ties / min_kept on the other (segmentors.OhemCrossEntropyLoss2d, the tie term) — as weights in
        # [0, 1] times one scalar
        w = torch.where(hard, above, gt + ties * ((min_kept - n_gt) / ties.sum().clamp_min(1.0)).clamp(0.0, 1.0))
        scal = torch.where(hard, 1.0 / n_above, torch.full_like(n_above, 1.0 / float(min_kept))) * lw
        ctx.meta = (N, Cc, Hi, Wi, Ho, Wo, ld, int(ignore_index))
        ctx.save_for_backward(logits, target, w, scal)
        return torch.where(hard, mean_above, mean_top)

    @staticmethod
    def backward(ctx, g):
        logits, target, w, scal = ctx.saved_tensors
        N, Cc, Hi, Wi, Ho, Wo, ld, ign = ctx.meta
        Cp = _round8(Cc)
        gs = (g.detach().float().reshape(1) * scal.reshape(1)).contiguous()
        buf = torch.empty((N, Hi, Wi, Cp), dtype=ACT_DTYPE, device=logits.device)
        L.call("cvhip_seg_ce_bilinear_bwd_px", logits.data_ptr(), ld, target.data_ptr(), N, Cc, Hi, Wi, Ho, Wo, 0, ign, w.data_ptr(), gs.data_ptr(),
               buf.data_ptr(), Cp, _stream())
        return buf.permute(0, 3, 1, 2)[:, :Cc], None, None, None, None, None


class OhemCrossEntropyBilinearFused(torch.autograd.Function):
    """OhemCrossEntropyBilinear with the selection on the device too (cvhip_ohem_select: three-pass radix select of the cut value + one
    reduction instead of torch.topk and ~15 element-wise / reduction launches; the backward kernel derives every pixel's weight from its
    forward loss and the selection record). forward(logits, target, thresh_nlog (python float), min_kept, ignore_index, loss_weight)."""

    @staticmethod
    def forward(ctx, logits, target, thr, min_kept, ignore_index, loss_weight):
        logits, ld = as_nhwc(logits)
        N, Cc, Hi, Wi = logits.shape
        target = target.long().contiguous()
        Ho, Wo = int(target.shape[-2]), int(target.shape[-1])
        M = N * Ho * Wo
        if M <= min_kept:
            raise IndexError("OhemCrossEntropyLoss2d: %d pixels, min_kept %d (the reference indexes loss[min_kept])" % (M, min_kept))
        dev = logits.device
        st = _stream()
        per = torch.empty((M,), dtype=torch.float32, device=dev)
        L.call("cvhip_seg_ce_bilinear_fwd_px", logits.data_ptr(), ld, target.data_ptr(), N, Cc, Hi, Wi, Ho, Wo, 0, int(ignore_index),
               per.data_ptr(), st)
        ws = torch.empty((int(L.load().cvhip_ohem_select_workspace_bytes()),), dtype=torch.uint8, device=dev)
        sel = torch.empty((8,), dtype=torch.float32, device=dev)
        L.call("cvhip_ohem_select", per.data_ptr(), M, int(min_kept), float(thr), float(loss_weight), ws.data_ptr(), sel.data_ptr(), st)
        ctx.meta = (N, Cc, Hi, Wi, Ho, Wo, ld, int(ignore_index), float(loss_weight))
        ctx.save_for_backward(logits, target, per, sel)
        return sel[0].clone()

    @staticmethod
    def backward(ctx, g):
        logits, target, per, sel = ctx.saved_tensors
        N, Cc, Hi, Wi, Ho, Wo, ld, ign, lw = ctx.meta
        Cp = _round8(Cc)
        gs = g.detach().float().reshape(1).contiguous()
        buf = torch.empty((N, Hi, Wi, Cp), dtype=ACT_DTYPE, device=logits.device)
        L.call("cvhip_seg_ce_bilinear_bwd_ohem", logits.data_ptr(), ld, target.data_ptr(), N, Cc, Hi, Wi, Ho, Wo, 0, ign, per.data_ptr(), lw,
               sel.data_ptr(), gs.data_ptr(), buf.data_ptr(), Cp, _stream())
        return buf.permute(0, 3, 1, 2)[:, :Cc], None, None, None, None, None


_OHEM_SELECT = __import__("os").environ.get("CVHIP_OHEM_SELECT", "1") != "0"   # 0: the selection as torch ops (OhemCrossEntropyBilinear)


def ohem_cross_entropy_resized_ok(logits, target):
    """the fused OHEM path runs this geometry (half-pixel bilinear up-sampling to the label size, <= 32 classes, tile fits the LDS)"""
    N, Cc, Hi, Wi = logits.shape
    return bool(_SEG_CE_FUSED and logits.is_cuda and L.load().cvhip_seg_ce_bilinear_ok(Cc, Hi, Wi, int(target.shape[-2]), int(target.shape[-1]), 0))


def detail_boundary_targets(labels, threshold=0.1):
    """detail_loss.py:37-79 on an int64 label map (N, H, W) -> fp32 (N, 1, H, W) in {0, 1} (cvhip_detail_boundary_targets)"""
    labels = labels.long().contiguous()
    N, H, W = labels.shape
    out = torch.empty((N, 1, H, W), dtype=torch.float32, device=labels.device)
    L.call("cvhip_detail_boundary_targets", labels.data_ptr(), N, H, W, float(threshold), out.data_ptr(), _stream())
    return out


_SEG_CE_FUSED = __import__("os").environ.get("CVHIP_SEG_CE_FUSED", "1") != "0"


def seg_cross_entropy_resized(logits, target, ignore_index=255, align_corners=False):
    """CE of `logits` bilinearly resized to the label size; the fused kernels when the geometry allows, else the two ops"""
    N, Cc, Hi, Wi = logits.shape
    Ho, Wo = int(target.shape[-2]), int(target.shape[-1])
    if _SEG_CE_FUSED and L.load().cvhip_seg_ce_bilinear_ok(Cc, Hi, Wi, Ho, Wo, int(bool(align_corners))):
        return SegCrossEntropyBilinear.apply(logits, target, ignore_index, align_corners)
    return seg_cross_entropy(resize_bilinear(logits, (Ho, Wo), align_corners), target, ignore_index)


class ResizeBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Ho, Wo, align_corners, out=None):
        x, ld = as_nhwc(x)
        N, Cc, Hi, Wi = x.shape
        if out is None:
            yld = _round8(Cc)  # padded pitch: odd channel counts (19-class logits) keep the 16-byte vector path
            y = empty_nhwc(N, Cc, Ho, Wo, x.device, ld=yld)
        else:  # `out=`: a channel slice of the buffer the consumer reads (concat elimination, as for the conv layers)
            y, yld = _check_out(out, N, Cc, Ho, Wo)
            y = y.as_strided(y.shape, y.stride())   # fresh alias (the result must not BE an argument)
        L.call("cvhip_resize_bilinear_fwd", x.data_ptr(), ld, y.data_ptr(), yld, N, Cc, Hi, Wi, Ho, Wo, int(align_corners), _stream())
        ctx.meta = (N, Cc, Hi, Wi, Ho, Wo, int(align_corners))
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, Hi, Wi, Ho, Wo, ac = ctx.meta
        dy, ld = as_nhwc(dy)
        xld = _round8(Cc)
        dx = empty_nhwc(N, Cc, Hi, Wi, dy.device, ld=xld)
        nb = int(L.fn("cvhip_resize_bilinear_bwd_workspace_bytes")(N, Cc, Hi, Wi, Ho, Wo))   # > 0: the separable form pays (ratio >= 4)
        if nb > 0:
            ws = torch.empty((nb,), dtype=torch.uint8, device=dy.device)
            L.call("cvhip_resize_bilinear_bwd_ws", dy.data_ptr(), ld, dx.data_ptr(), xld, N, Cc, Hi, Wi, Ho, Wo, ac, ws.data_ptr(), nb, _stream())
        else:
            L.call("cvhip_resize_bilinear_bwd", dy.data_ptr(), ld, dx.data_ptr(), xld, N, Cc, Hi, Wi, Ho, Wo, ac, _stream())
        return dx, None, None, None, None


def resize_bilinear(x, size, align_corners=False, out=None):
    return ResizeBilinear.apply(x, int(size[0]), int(size[1]), bool(align_corners), out)


class ResizeNearest(torch.autograd.Function):
    """F.interpolate(x, size, mode="nearest") for arbitrary sizes (cvhip_resize_nearest_fwd / _bwd: exact copy forward,
    deterministic gather-sum backward)."""

    @staticmethod
    def forward(ctx, x, Ho, Wo):
        x, ld = as_nhwc(x)
        N, Cc, Hi, Wi = x.shape
        yld = _round8(Cc)
        y = empty_nhwc(N, Cc, Ho, Wo, x.device, ld=yld)
        L.call("cvhip_resize_nearest_fwd", x.data_ptr(), ld, y.data_ptr(), yld, N, Cc, Hi, Wi, Ho, Wo, _stream())
        ctx.meta = (N, Cc, Hi, Wi, Ho, Wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, Hi, Wi, Ho, Wo = ctx.meta
        dy, ld = as_nhwc(dy)
        xld = _round8(Cc)
        dx = empty_nhwc(N, Cc, Hi, Wi, dy.device, ld=xld)
        L.call("cvhip_resize_nearest_bwd", dy.data_ptr(), ld, dx.data_ptr(), xld, N, Cc, Hi, Wi, Ho, Wo, _stream())
        return dx, None, None


def resize_nearest(x, size):
    return ResizeNearest.apply(x, int(size[0]), int(size[1]))


class GlobalAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x, ld = as_nhwc(x)
        N, Cc, H, W = x.shape
        y = empty_nhwc(N, Cc, 1, 1, x.device)
        L.call("cvhip_global_avgpool_fwd", x.data_ptr(), ld, y.data_ptr(), N, Cc, H * W, _stream())
        ctx.meta = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W = ctx.meta
        dy = dy.to(ACT_DTYPE).reshape(N, Cc).contiguous()
        dx = empty_nhwc(N, Cc, H, W, dy.device)
        L.call("cvhip_global_avgpool_bwd", dy.data_ptr(), dx.data_ptr(), Cc, N, Cc, H * W, _stream())
        return dx


def global_avg_pool(x):
    return GlobalAvgPool.apply(x)


def images_to_nhwc(x, cpad=8, focus=False):
    """fp32 NCHW image batch -> bf16 NHWC view with channels zero-padded to `cpad` (no autograd)."""
    if not x.is_cuda:
        raise L.CvhipError("cvpytorch_amd ops need CUDA/HIP tensors (no CPU fallback)")
    x = x.detach().float().contiguous()
    N, Cc, H, W = x.shape
    if focus:
        y = torch.empty((N, cpad, H // 2, W // 2), dtype=ACT_DTYPE, device=x.device, memory_format=torch.channels_last)
        L.call("cvhip_focus_nchw_f32_to_nhwc_bf16", x.data_ptr(), y.data_ptr(), N, Cc, H, W, cpad, _stream())
    else:
        y = torch.empty((N, cpad, H, W), dtype=ACT_DTYPE, device=x.device, memory_format=torch.channels_last)
        L.call("cvhip_nchw_f32_to_nhwc_bf16", x.data_ptr(), y.data_ptr(), N, Cc, H, W, cpad, _stream())
    return y


class NhwcToNchwF32(torch.autograd.Function):
    """bf16 NHWC activations -> fp32 NCHW contiguous (what torch-side losses consume) and its adjoint."""

    @staticmethod
    def forward(ctx, x):
        x, ld = as_nhwc(x)
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, H, W), dtype=torch.float32, device=x.device)
        L.call("cvhip_nhwc_bf16_to_nchw_f32", x.data_ptr(), ld, y.data_ptr(), N, Cc, H, W, _stream())
        ctx.meta = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, H, W = ctx.meta
        dy = dy.float().contiguous()
        Cp = _round8(Cc)
        buf = torch.empty((N, H, W, Cp), dtype=ACT_DTYPE, device=dy.device)
        if Cp != Cc:
            zero_fill(buf)
        L.call("cvhip_nchw_f32_to_nhwc_bf16_ld", dy.data_ptr(), buf.data_ptr(), Cp, N, Cc, H, W, _stream())
        return buf.permute(0, 3, 1, 2)[:, :Cc]


def to_nchw_f32(x):
    return NhwcToNchwF32.apply(x)


class HeadPermute(torch.autograd.Function):
    """(N, A*NO, H, W) bf16 NHWC -> (N, A, H, W, NO) fp32 contiguous."""

    @staticmethod
    def forward(ctx, x, A, NO):
        x, ld = as_nhwc(x)
        N, Cc, H, W = x.shape
        if Cc != A * NO:
            raise L.CvhipError("head_permute: channels != A*NO")
        y = torch.empty((N, A, H, W, NO), dtype=torch.float32, device=x.device)
        L.call("cvhip_head_permute_fwd", x.data_ptr(), ld, y.data_ptr(), N, A, NO, H, W, _stream())
        ctx.meta = (N, A, NO, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, A, NO, H, W = ctx.meta
        dy = dy.float().contiguous()
        Cp = _round8(A * NO)
        buf = torch.empty((N, H, W, Cp), dtype=ACT_DTYPE, device=dy.device)
        L.call("cvhip_head_permute_bwd", dy.data_ptr(), buf.data_ptr(), Cp, N, A, NO, H, W, _stream())
        return buf.permute(0, 3, 1, 2)[:, :A * NO], None, None


def head_permute(x, A, NO):
    return HeadPermute.apply(x, A, NO)


class YoloV5LossFused(torch.autograd.Function):
    """YOLOv5 loss (build_targets + CIoU + class/objectness BCE) straight on the bf16 NHWC head maps — libcvhip
    cvhip_yolov5_loss_* (src/losses/yolov5_loss.py:173-278). forward(targets, cfg, *raw_maps) -> (total (1,), stats (3,)).
    cfg: object with num_classes, num_anchors, anchors (L, A, 2) list, anchor_t, hyp_box/obj/cls, balance."""

    @staticmethod
    def forward(ctx, targets, cfg, *raws):
        dev = raws[0].device
        st = _stream()
        tg = targets.detach()
        if tg.dtype != torch.float32 or not tg.is_contiguous():
            tg = tg.float().contiguous()
        T = tg.shape[0]
        A, NO = cfg.num_anchors, cfg.num_classes + 5
        nl = len(raws)
        sums = torch.empty((nl, 4), dtype=torch.float32, device=dev)
        descs, wss, maps, ncells = [], [], [], []
        raws = [as_nhwc(r) for r in raws]
        assign = None
        ota = getattr(cfg, "ota", None)
        if ota is not None:
            # YOLOv7 OTA: the positives are decided on the device first (cvhip_ota_assign: find_3_positive candidates pooled per image,
            # dynamic-k matching, conflict resolution); the per-level kernels below then take them from `assign`
            N = raws[0][0].shape[0]
            od = L.OtaDesc()
            od.L, od.N, od.A, od.NO, od.T, od.G = nl, N, A, NO, T, int(ota["G"])
            od.anchor_t, od.img_size = float(cfg.anchor_t), float(ota["img_size"])
            for i, (r, ld) in enumerate(raws):
                od.H[i], od.W[i], od.ld[i], od.stride[i] = r.shape[2], r.shape[3], ld, float(ota["stride"][i])
                for j, v in enumerate([float(x) for pair in cfg.anchors[i] for x in pair]):
                    od.anchors[i][j] = v
            nb = L.load().cvhip_ota_workspace_bytes(C.byref(od))
            if nb < 0:
                L.check(int(nb), "cvhip_ota_workspace_bytes")
            ota_ws = torch.empty((int(nb),), dtype=torch.uint8, device=dev)
            assign = torch.empty((nl, 5 * A * T), dtype=torch.int32, device=dev)
            ptrs = (C.c_void_p * nl)(*[r.data_ptr() for r, _ in raws])
            L.call("cvhip_ota_assign", C.byref(od), ptrs, tg.data_ptr(), ota_ws.data_ptr(), assign.data_ptr(), st)
            ctx.ota_keep = (ota_ws, od)
            cfg.last_assign = assign   # diagnostics / tests
        for i, (r, ld) in enumerate(raws):
            N, Cc, H, W = r.shape
            if Cc != A * NO:
                raise L.CvhipError("yolov5 loss: head map has %d channels, expected %d" % (Cc, A * NO))
            d = L.YoloLossDesc(N, A, NO, H, W, ld, T, float(cfg.anchor_t))
            for j, v in enumerate([float(x) for pair in cfg.anchors[i] for x in pair]):
                d.anchors[j] = v
            nbytes = L.load().cvhip_yolov5_loss_workspace_bytes(C.byref(d))
            if nbytes < 0:
                L.check(int(nbytes), "cvhip_yolov5_loss_workspace_bytes")
            ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev)
            if assign is not None:
                L.call("cvhip_yolov5_loss_level_fwd_assigned", C.byref(d), r.data_ptr(), tg.data_ptr(), assign[i].data_ptr(), ws.data_ptr(),
                       sums[i].data_ptr(), st)
            else:
                L.call("cvhip_yolov5_loss_level_fwd", C.byref(d), r.data_ptr(), tg.data_ptr(), ws.data_ptr(), sums[i].data_ptr(), st)
            descs.append(d)
            wss.append(ws)
            maps.append(r)
            ncells.append(float(N * A * H * W))
        key = (tuple(ncells), tuple(cfg.balance[:nl]), str(dev))
        consts = cfg._const_cache.get(key)
        if consts is None:  # built once per shape/device (a host->device copy here would break hipGraph capture)
            consts = (torch.tensor(ncells, dtype=torch.float32, device=dev), torch.tensor(list(cfg.balance[:nl]), dtype=torch.float32, device=dev))
            cfg._const_cache[key] = consts
        total = torch.empty((1,), dtype=torch.float32, device=dev)
        stats = torch.empty((3,), dtype=torch.float32, device=dev)
        bs = float(maps[0].shape[0])
        L.call("cvhip_yolov5_loss_finalize", sums.data_ptr(), nl, consts[0].data_ptr(), consts[1].data_ptr(), float(cfg.hyp_box),
               float(cfg.hyp_obj), float(cfg.hyp_cls), cfg.num_classes, bs, total.data_ptr(), stats.data_ptr(), st)
        ctx.cfg, ctx.descs, ctx.wss, ctx.ncells, ctx.bs = cfg, descs, wss, ncells, bs
        ctx.save_for_backward(tg, sums, *maps)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)   # no zeros tensor (an ATen fill kernel) for the unused gradient of `stats`
        return total, stats

    @staticmethod
    def backward(ctx, g_total, g_stats):
        tg, sums, *maps = ctx.saved_tensors
        cfg = ctx.cfg
        st = _stream()
        g = g_total.detach()
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.float().contiguous()
        outs = []
        nc = cfg.num_classes
        clear_colsums()
        for i, r in enumerate(maps):
            d = ctx.descs[i]
            draw = torch.empty((d.N, d.H, d.W, d.ld), dtype=ACT_DTYPE, device=r.device)
            K = d.A * d.NO
            bp = torch.empty((L.YOLO_BIAS_ROWS + L.REDUCE_SCRATCH_ROWS, 2, K), dtype=torch.float32, device=r.device)
            L.call("cvhip_yolov5_loss_level_bwd_bias", C.byref(d), r.data_ptr(), tg.data_ptr(), ctx.wss[i].data_ptr(), sums[i].data_ptr(),
                   g.data_ptr(), float(cfg.hyp_box) * ctx.bs, (float(cfg.hyp_cls) * ctx.bs / nc) if nc > 1 else 0.0,
                   float(cfg.hyp_obj) * float(cfg.balance[i]) * ctx.bs / ctx.ncells[i], draw.data_ptr(), bp.data_ptr(), st)
            # the detect convolution that receives this map takes its bias gradient from these rows (ops._conv_grads)
            offer_colsum(draw, bp, L.YOLO_BIAS_ROWS, K)
            outs.append(draw.permute(0, 3, 1, 2)[:, :K])
        return (None, None, *outs)


def yolov5_loss_fused(raws, targets, cfg):
    return YoloV5LossFused.apply(targets, cfg, *raws)


class SimotaLossFused(torch.autograd.Function):
    """YOLOX loss (SimOTA assignment + 5*IoU^2 + obj + cls BCE) straight on the bf16 NHWC head maps — libcvhip
    cvhip_simota_loss_* (src/losses/det/yolox_loss.py:73-435). forward(targets (B,G,5) pixels, cfg, *raw_maps) -> out5 =
    [loss, conf_loss, cls_loss, 5*iou_loss, num_fg/num_gts]; only out5[0] is differentiable."""

    @staticmethod
    def forward(ctx, targets, cfg, *raws):
        dev = raws[0].device
        st = _stream()
        tg = targets.detach()
        if tg.dtype != torch.float32 or not tg.is_contiguous():
            tg = tg.float().contiguous()
        B, G = int(tg.shape[0]), int(tg.shape[1])
        nc = cfg.num_classes
        maps = []
        d = L.SimotaDesc()
        d.L, d.B, d.G, d.nc = len(raws), B, G, nc
        A = 0
        for i, r in enumerate(raws):
            r, ld = as_nhwc(r)
            N, Cc, H, W = r.shape
            if Cc != nc + 5 or N != B:
                raise L.CvhipError("simota loss: head map %d has shape %s, expected (%d, %d, H, W)" % (i, tuple(r.shape), B, nc + 5))
            d.H[i], d.W[i], d.ld[i], d.stride[i] = H, W, ld, float(cfg.strides[i])
            A += H * W
            maps.append(r)
        d.A = A
        nbytes = L.load().cvhip_simota_workspace_bytes(C.byref(d))
        if nbytes < 0:
            L.check(int(nbytes), "cvhip_simota_workspace_bytes")
        ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * len(maps))(*[m.data_ptr() for m in maps])
        out5 = torch.empty((5,), dtype=torch.float32, device=dev)
        L.call("cvhip_simota_loss_fwd", C.byref(d), ptrs, tg.data_ptr(), ws.data_ptr(), out5.data_ptr(), st)
        ctx.desc, ctx.ws = d, ws
        ctx.save_for_backward(tg, *maps)
        return out5

    @staticmethod
    def backward(ctx, g5):
        tg, *maps = ctx.saved_tensors
        d = ctx.desc
        g = g5.detach().float()[0:1].contiguous()  # d total / d out5[0]; the other entries are reporting-only
        draws = [torch.empty((d.B, d.H[i], d.W[i], d.ld[i]), dtype=ACT_DTYPE, device=maps[i].device) for i in range(d.L)]
        ptrs = (C.c_void_p * d.L)(*[m.data_ptr() for m in maps])
        dptrs = (C.c_void_p * d.L)(*[t.data_ptr() for t in draws])
        L.call("cvhip_simota_loss_bwd", C.byref(d), ptrs, tg.data_ptr(), ctx.ws.data_ptr(), g.data_ptr(), dptrs, _stream())
        return (None, None, *[t.permute(0, 3, 1, 2)[:, :d.nc + 5] for t in draws])


def simota_loss_fused(raws, targets, cfg):
    return SimotaLossFused.apply(targets, cfg, *raws)


def simota_read_assignment(fn_ctx_desc, ws):
    """(matched gt (B, A) int32 with -1 = background, matched IoU (B, A)) of a finished cvhip_simota_loss_fwd — tests."""
    d = fn_ctx_desc
    m = torch.empty((d.B, d.A), dtype=torch.int32, device=ws.device)
    u = torch.empty((d.B, d.A), dtype=torch.float32, device=ws.device)
    L.call("cvhip_simota_read_assignment", C.byref(d), ws.data_ptr(), m.data_ptr(), u.data_ptr(), _stream())
    return m, u


# ---- non-differentiable post-processing ---------------------------------------------------------------

def yolov5_decode(levels, strides, anchors_px, A, NO):
    """levels: list of (N, A*NO, H, W) head outputs; returns (N, sum(A*H*W), NO) fp32 — detects/yolov5_detect.py:48-57."""
    N = levels[0].shape[0]
    tot = sum(A * l.shape[2] * l.shape[3] for l in levels)
    out = torch.empty((N, tot, NO), dtype=torch.float32, device=levels[0].device)
    off = 0
    for l, s, anc in zip(levels, strides, anchors_px):
        l, ld = as_nhwc(l.detach())
        H, W = l.shape[2], l.shape[3]
        anc = anc.to(device=l.device, dtype=torch.float32).contiguous()
        L.call("cvhip_yolov5_decode", l.data_ptr(), ld, out.data_ptr(), N, A, NO, H, W, float(s), anc.data_ptr(), tot * NO, off, _stream())
        off += A * H * W
    return out


def nms(boxes, scores, iou_thr):
    """torchvision.ops.nms contract: returns int64 indices of kept boxes, in decreasing score order."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if not boxes.is_cuda:
        raise L.CvhipError("cvpytorch_amd.nms needs CUDA/HIP tensors (no CPU fallback)")
    order = torch.sort(scores.float(), descending=True, stable=True)[1]
    b = boxes.float()[order].contiguous()
    n = b.shape[0]
    lib = L.load()
    ws = torch.empty((int(lib.cvhip_nms_workspace_bytes(n)),), dtype=torch.uint8, device=b.device)
    keep = torch.empty((n,), dtype=torch.int32, device=b.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=b.device)
    L.call("cvhip_nms_sorted", b.data_ptr(), n, float(iou_thr), ws.data_ptr(), keep.data_ptr(), cnt.data_ptr(), _stream())
    k = int(cnt.item())
    return order[keep[:k].long()]


def box_iou(a, b):
    a = a.float().contiguous()
    b = b.float().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L.call("cvhip_box_iou", a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(), _stream())
    return out
