"""TEST INFRASTRUCTURE — a CPU emulator of the engine's STORAGE precision on top of the oracle's modules.

Why: end-to-end gradients of a randomly initialised 60-conv network are noise-limited when activations and activation
gradients are stored in 16 bits (the fp32 oracle and the engine agree only to a median per-parameter cosine of ~0.86, and so
does the oracle with itself under CPU bf16 autocast). VERDICT r1 asked to PROVE that the residual is storage rounding and not
kernel error. An fp32-storage build of the kernels is not possible (every tile shape, DMA width and transpose read is built on
2-byte elements), so the proof goes the other way round: this module re-states, in plain fp32 torch on the CPU, exactly WHERE the
engine rounds to 16 bits —

    y   = r16(conv(x16, r16(W)))                  BN statistics from the fp32 accumulators, before the rounding
    z   = r16(act(scale * y + shift) (+ residual))
    du  = r16-valued dz * act'(scale * y + shift);   dgamma / dbeta in fp32
    dy  = r16(scale * (du - dbeta/M - xhat * dgamma/M)),   xhat from the ROUNDED y with the unrounded statistics
    dx  = r16(dgrad(dy, r16(W)) (+ skip-connection gradient))      dW = wgrad(x16, dy) in fp32
    sibling 1x1 pairs (CSP conv1 / conv2) as ONE convolution (one dgrad sum, one rounding)

— and nothing else differs from the oracle (same module tree, same loss). Rounding is discontinuous, so two correct
implementations of these rounding points do NOT agree bit for bit: fp32 summation order alone flips ~1 % of the bf16 roundings by
a whole ulp (emulate_storage(acc64=True) is the second realisation that measures this ceiling, ~0.955 median per-parameter cosine
for YOLOv5-s in bf16). tests/test_gpu_storage_emulator.py therefore asserts what is decidable: the engine deviates from the fp32
oracle by as much as this model does, and agrees with this model as well as this model agrees with its second realisation.
Only tests import this file."""
import types

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.grad import conv2d_input, conv2d_weight

from oracle import torch_ref as R


def r16(t, dt):
    return t.to(dt).float()


def _act(u, kind, slope):
    if kind == "silu":
        return u * torch.sigmoid(u)
    if kind == "relu":
        return torch.relu(u)
    if kind == "leaky":
        return F.leaky_relu(u, slope)
    return u


def _dact(u, kind, slope):
    if kind == "silu":
        s = torch.sigmoid(u)
        return s * (1 + u * (1 - s))
    if kind == "relu":
        return (u > 0).float()
    if kind == "leaky":
        return torch.where(u > 0, torch.ones_like(u), torch.full_like(u, slope))
    return torch.ones_like(u)


class GradLink:
    def __init__(self):
        self.g = None


class QConvBnAct(torch.autograd.Function):
    """One engine layer with its rounding points. `meta` = dict(stride, padding, dilation, groups, eps, act, slope, dt, has_bn,
    link_in: GradLink whose gradient is added inside this layer's dgrad, link_out: GradLink that receives the residual's gradient)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, res, meta):
        dt = meta["dt"]
        xq, wq = r16(x, dt), r16(w, dt)
        if meta.get("acc64"):     # same rounding points, different fp32 summation (exactly-rounded sums): see emulate_storage(acc64=)
            acc = F.conv2d(xq.double(), wq.double(), None, meta["stride"], meta["padding"], meta["dilation"], meta["groups"]).float()
        else:
            acc = F.conv2d(xq, wq, None, meta["stride"], meta["padding"], meta["dilation"], meta["groups"])
        K = w.shape[0]
        if meta["has_bn"]:
            mean = acc.double().mean((0, 2, 3))
            var = (acc.double() ** 2).mean((0, 2, 3)) - mean ** 2
            invstd = (1.0 / torch.sqrt(var.clamp(min=0) + meta["eps"])).float()
            mean = mean.float()
            yq = r16(acc, dt)
            scale = gamma * invstd
            shift = beta - mean * scale
            u = yq * scale.view(1, K, 1, 1) + shift.view(1, K, 1, 1)
        else:
            yq = r16(acc + (bias.view(1, K, 1, 1) if bias is not None else 0.0), dt)
            mean = invstd = scale = None
            u = yq
        z = _act(u, meta["act"], meta["slope"])
        if res is not None:
            z = z + res
        ctx.meta = meta
        ctx.has_res = res is not None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(xq, wq, yq, u, mean, invstd, scale)
        return r16(z, dt)

    @staticmethod
    def backward(ctx, dz):
        meta = ctx.meta
        dt = meta["dt"]
        xq, wq, yq, u, mean, invstd, scale = ctx.saved_tensors
        K = wq.shape[0]
        dzq = r16(dz, dt)
        du = dzq * _dact(u, meta["act"], meta["slope"])
        dgamma = dbeta = dbias = None
        if meta["has_bn"]:
            M = du.numel() // K
            xh = (yq - mean.view(1, K, 1, 1)) * invstd.view(1, K, 1, 1)
            dbeta = du.double().sum((0, 2, 3)).float()
            dgamma = (du.double() * xh.double()).sum((0, 2, 3)).float()
            dy = scale.view(1, K, 1, 1) * (du - (dbeta / M).view(1, K, 1, 1) - xh * (dgamma / M).view(1, K, 1, 1))
        else:
            dy = du
            if ctx.has_bias:
                dbias = dy.sum((0, 2, 3))
        dyq = r16(dy, dt)
        if meta.get("acc64"):
            dx = conv2d_input(xq.shape, wq.double(), dyq.double(), meta["stride"], meta["padding"], meta["dilation"], meta["groups"]).float()
        else:
            dx = conv2d_input(xq.shape, wq, dyq, meta["stride"], meta["padding"], meta["dilation"], meta["groups"])
        link = meta.get("link_in")
        if link is not None and link.g is not None:   # the skip connection's gradient joins in fp32, ONE rounding (cvhip_conv2d_dgrad_add)
            dx = dx + link.g
            link.g = None
        dxq = r16(dx, dt)
        dw = conv2d_weight(xq, wq.shape, dyq, meta["stride"], meta["padding"], meta["dilation"], meta["groups"])
        dres = None
        if ctx.has_res:
            lo = meta.get("link_out")
            if lo is not None:
                lo.g = dzq
            else:
                dres = dzq
        return dxq, dw, dbias, dgamma, dbeta, dres, None


def _kind(m):
    a = getattr(m, "act", None) if getattr(m, "with_act", False) else None
    if a is None:
        return "none", 0.0
    if isinstance(a, (nn.SiLU, R.Swish)):
        return "silu", 0.0
    if isinstance(a, nn.ReLU):
        return "relu", 0.0
    if isinstance(a, nn.LeakyReLU):
        return "leaky", float(a.negative_slope)
    raise NotImplementedError(type(a))


def _meta(conv, bn, kind, slope, dt, **kw):
    return dict(stride=conv.stride, padding=conv.padding, dilation=conv.dilation, groups=conv.groups, eps=(bn.eps if bn is not None else 0.0),
                act=kind, slope=slope, dt=dt, has_bn=bn is not None, **kw)


def _convmodule_forward(self, x, res=None, link_in=None, link_out=None):
    kind, slope = _kind(self)
    bn = self.bn if self.with_norm else None
    meta = _meta(self.conv, bn, kind, slope, self._emu_dt, acc64=self._emu_acc64, link_in=link_in, link_out=link_out)
    return QConvBnAct.apply(x, self.conv.weight, self.conv.bias, bn.weight if bn is not None else None, bn.bias if bn is not None else None, res, meta)


def _bottleneck_forward(self, x):
    if self.shortcut and isinstance(self.conv2, R.ConvModule):
        link = GradLink() if x.requires_grad else None
        h = self.conv1(x, link_in=link)
        return self.conv2(h, res=x, link_out=link)
    out = self.conv2(self.conv1(x))
    return out + x if self.shortcut else out


def _csp_forward(self, x):
    # conv1 / conv2 on the same input train as ONE convolution with K1 + K2 outputs (ops.ConvBnActPair): one dgrad, one rounding
    c1, c2 = self.conv1, self.conv2
    kind, slope = _kind(c1)
    w = torch.cat((c1.conv.weight, c2.conv.weight), 0)
    g = torch.cat((c1.bn.weight, c2.bn.weight), 0)
    b = torch.cat((c1.bn.bias, c2.bn.bias), 0)
    meta = _meta(c1.conv, c1.bn, kind, slope, self._emu_dt, acc64=self._emu_acc64)
    z = QConvBnAct.apply(x, w, None, g, b, None, meta)
    k1 = c1.conv.weight.shape[0]
    x_1 = self.m(z[:, :k1])
    return self.conv3(torch.cat((x_1, z[:, k1:]), dim=1))


def _detect_forward(self, x):
    x = list(x)
    for i in range(self.num_layers):
        conv = self.m[i]
        x[i] = QConvBnAct.apply(x[i], conv.weight, conv.bias, None, None, None, _meta(conv, None, "none", 0.0, self._emu_dt, acc64=self._emu_acc64))
        bs, _, ny, nx = x[i].shape
        x[i] = x[i].view(bs, self.num_anchors, self.num_outputs, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
    return None, x


def _v7conv_forward(self, x):
    # oracle/yolov7_ref.py Conv: conv -> bn -> SiLU | Identity
    kind = "silu" if isinstance(self.act, nn.SiLU) else "none"
    meta = _meta(self.conv, self.bn, kind, 0.0, self._emu_dt, acc64=self._emu_acc64)
    return QConvBnAct.apply(x, self.conv.weight, None, self.bn.weight, self.bn.bias, None, meta)


def emulate_storage(model, dt=torch.bfloat16, fuse_pairs=True, acc64=False):
    """Patch an ORACLE model (oracle/torch_ref.py classes) in place so that its training forward/backward rounds where the engine
    rounds. Returns the model. acc64=True keeps every rounding point but evaluates the convolution sums in fp64 (then fp32): a second
    REALISATION of the same storage format whose pre-rounding values differ from the default one by fp32 summation order only —
    the experiment that measures how far two exact implementations of the same rounding points can agree at all."""
    from oracle import yolov7_ref as R7
    for m in model.modules():
        m._emu_acc64 = acc64
        if isinstance(m, R7.Conv):
            m._emu_dt = dt
            m.forward = types.MethodType(_v7conv_forward, m)
        elif isinstance(m, R.ConvModule):
            assert m.order == ("conv", "norm", "act")
            m._emu_dt = dt
            m.forward = types.MethodType(_convmodule_forward, m)
        elif isinstance(m, R.DarknetBottleneck):
            m.forward = types.MethodType(_bottleneck_forward, m)
        elif isinstance(m, R.CSPLayer) and fuse_pairs:
            m._emu_dt = dt
            m.forward = types.MethodType(_csp_forward, m)
        elif isinstance(m, R.YOLOv5Detect):
            m._emu_dt = dt
            m.forward = types.MethodType(_detect_forward, m)
    return model


class _Round16(torch.autograd.Function):
    """Straight-through 16-bit storage: the value is rounded going forward, the gradient is rounded going backward."""

    @staticmethod
    def forward(ctx, x, dt):
        ctx.dt = dt
        y = x.to(dt).float()
        return y.clone() if y.data_ptr() == x.data_ptr() else y

    @staticmethod
    def backward(ctx, g):
        return g.to(ctx.dt).float(), None


def emulate_storage_generic(model, dt=torch.bfloat16):
    """A module-type-agnostic approximation for models whose blocks are not ConvModules (ResNet bottlenecks, heads built from plain
    nn.Conv2d / nn.BatchNorm2d): every convolution, normalisation, activation, pooling and linear module rounds its output (and the
    gradient arriving at it) to 16 bits. This rounds in a few more places than the engine does (the engine keeps BN + activation in
    one pass), so it is a slightly PESSIMISTIC estimate of what 16-bit storage alone does to the gradients: good for a floor, not
    for the block-by-block equality that emulate_storage supports. Weights are rounded through the same op. Returns the model."""
    kinds = (nn.Conv2d, nn.BatchNorm2d, nn.ReLU, nn.SiLU, nn.LeakyReLU, nn.Sigmoid, nn.Hardswish, nn.MaxPool2d, nn.AvgPool2d,
             nn.AdaptiveAvgPool2d, nn.Linear, nn.Upsample, nn.UpsamplingNearest2d, R.Swish)
    def round_out(mod, inp, out, dt=dt):
        return _Round16.apply(out, dt)

    def round_w(mod, inp, dt=dt):      # the engine multiplies 16-bit weight images; the fp32 master keeps receiving the gradient
        mod._w_saved = mod.weight.data.clone()
        mod.weight.data.copy_(mod.weight.data.to(dt).float())
        return None

    def restore_w(mod, inp, out):
        mod.weight.data.copy_(mod._w_saved)
        return None

    for m in model.modules():
        if isinstance(m, kinds):
            if getattr(m, "inplace", False):
                m.inplace = False
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                m.register_forward_pre_hook(round_w)
                m.register_forward_hook(restore_w)
            m.register_forward_hook(round_out)
    return model
