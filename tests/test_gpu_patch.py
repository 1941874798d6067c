"""GPU parity tests of the patch-resident convolution kernel (csrc/conv_patch.hip) and of the fused convolution entry point
(cvhip_conv2d_fprop_fused: BatchNorm-apply + activation PROLOGUE, bias / scale+shift / activation EPILOGUE) through the C ABI, against
fp32 CPU arithmetic on the same 16-bit-rounded operands — the semantics of `act(norm(conv(x)))`, conv_module.py:201-214, and of the
folded conv+act of utils/fuse.py:32-54.

Tolerances: outputs stored in 16 bits: max |err| <= 2^-7 max|ref|, relative L2 <= 4e-3 (one rounding of an fp32-accumulated value);
the prologue's stored activation z_out must be BIT-identical to the stand-alone BN+activation pass (cvhip_bn_act_fwd) it replaces;
fp32 BatchNorm sums: relative L2 <= 1e-3.

CVHIP_PATCH=2 (read per launch) makes the launcher pick the patch kernel for every geometry it can tile, so the small / ragged /
image-spanning cases below run on it instead of falling to the per-tap kernel by the "too much padding" policy."""
import ctypes as C
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import test_gpu_kernels as K
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

dev, rel_l2, max_rel, to_nhwc_dev = K.dev, K.rel_l2, K.max_rel, K.to_nhwc_dev


def rnd(x):
    return x.to(K.BF).float()


@pytest.fixture(autouse=True)
def force_patch(monkeypatch):
    monkeypatch.setenv("CVHIP_PATCH", "2")
    monkeypatch.setenv("CVHIP_BAND", "0")   # (the row-band kernel would take the stride-1 3x3 cases first: tests/test_gpu_band.py)
    yield


PATCH_CASES = [
    # N, C, H, W, K, R, S, stride, pad, dil
    (2, 64, 20, 24, 128, 3, 3, 1, 1, 1),
    (2, 128, 17, 19, 128, 3, 3, 1, 1, 1),    # ragged, two channel chunks
    (3, 64, 20, 20, 64, 3, 3, 1, 1, 1),      # tiles span images
    (1, 32, 23, 37, 32, 3, 3, 1, 1, 1),      # 32-channel chunks, BN 32
    (2, 96, 12, 12, 48, 3, 3, 1, 1, 1),      # Cin % 64 != 0 -> 32-deep chunks x 3, K padded tile
    (2, 64, 16, 16, 48, 3, 3, 1, 2, 2),      # dilation 2
    (2, 64, 12, 12, 32, 5, 5, 1, 2, 1),
    (1, 64, 10, 14, 72, 3, 1, 1, 0, 1),      # no padding, 3x1, K = 72: one ragged 128-wide channel tile
    (2, 64, 9, 300, 64, 3, 3, 1, 1, 1),      # wider than one tile
    (2, 256, 10, 10, 255, 3, 3, 1, 1, 1),    # four chunks, K 255 -> padded 256 = two n-tiles
    (62, 128, 40, 40, 128, 3, 3, 1, 1, 1),   # the dominant YOLOv5-s shape (ragged batch)
    (9, 64, 80, 80, 64, 3, 3, 1, 1, 1),
    (5, 32, 160, 160, 32, 3, 3, 1, 1, 1),
]
DGRAD_S2 = [
    (2, 32, 32, 32, 64, 3, 3, 2, 1, 1),
    (3, 64, 21, 23, 64, 3, 3, 2, 1, 1),      # odd sizes: parity classes of unequal extent
    (26, 128, 80, 80, 256, 3, 3, 2, 1, 1),
]


def _takes(case, dgrad=False):
    N, Cc, H, W, Kk, R, S, s, p, d = case
    Kp = (Kk + 7) // 8 * 8
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp)
    buf = (C.c_int32 * (4 * L.PATCH_CLASS_INTS))()
    return L.load().cvhip_conv2d_patch_plan(C.byref(desc), 1 if dgrad else 0, buf, 4)


@pytest.mark.parametrize("case", PATCH_CASES)
def test_patch_cases_take_the_patch_kernel(case):
    assert _takes(case) == 1
    Kp = (case[4] + 7) // 8 * 8
    assert _takes(case, True) == (1 if Kp % 32 == 0 else 0)   # dgrad gathers dy: its channel count decides


@pytest.mark.parametrize("case", PATCH_CASES)
def test_patch_fprop(case):
    K.test_conv_fprop(case)


@pytest.mark.parametrize("case", [c for c in PATCH_CASES if c[4] % 8 == 0])
def test_patch_fprop_bn_stats(case):
    K.test_conv_fprop_bn_stats(case)


@pytest.mark.parametrize("case", PATCH_CASES + DGRAD_S2)
def test_patch_dgrad(case):
    K.test_conv_dgrad(case)


@pytest.mark.parametrize("case", [PATCH_CASES[1], PATCH_CASES[3], DGRAD_S2[1]])
def test_patch_dgrad_add(case):
    """dgrad with the skip-connection gradient added in the epilogue (cvhip_conv2d_dgrad_add)"""
    N, Cc, H, W, Kk, R, S, s, p, d = case
    x, w = K._mk(case, 2)
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(xr, w, None, stride=s, padding=p, dilation=d)
    dy = rnd(torch.randn(y.shape, generator=torch.Generator().manual_seed(3)))
    add = rnd(torch.randn(x.shape, generator=torch.Generator().manual_seed(4)))
    (gx,) = torch.autograd.grad(y, xr, dy)
    gx = gx + add
    st, Kp = K._prep(case, w, True)
    dyd = to_nhwc_dev(dy)
    addd = to_nhwc_dev(add)
    dx = ops.empty_nhwc(N, Cc, H, W, dev())
    dx.fill_(float("nan"))
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp)
    L.call("cvhip_conv2d_dgrad_add", C.byref(desc), dyd.data_ptr(), st.w_dgrad.data_ptr(), addd.data_ptr(), Cc, dx.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    got = dx.float().cpu()
    assert torch.isfinite(got).all()
    assert max_rel(got, gx) < 2 ** -7 and rel_l2(got, gx) < 4e-3


def test_patch_channel_slice_operands_and_acc_stats():
    """x read from / y written into channel slices of wider buffers; BatchNorm sums into the fp64 accumulator form"""
    torch.manual_seed(0)
    N, Cc, H, W, Kk = 3, 64, 14, 18, 64
    xbig = rnd(torch.randn(N, 160, H, W))
    w = rnd(torch.randn(Kk, Cc, 3, 3) / 24)
    ref = F.conv2d(xbig[:, 64:128], w, None, padding=1)
    xb = to_nhwc_dev(xbig)
    ybig = torch.zeros((N, 192, H, W), dtype=K.BF, device=dev()).contiguous(memory_format=torch.channels_last)
    st = ops.ConvState()
    pdesc = ops.conv_desc(N, Cc, H, W, Kk, 3, 3, (1, 1), (1, 1), (1, 1), 1, Cc, Kk)
    st.prepare(w.to(dev()).contiguous(memory_format=torch.channels_last), pdesc, False, ("slice", 0))
    desc = ops.conv_desc(N, Cc, H, W, Kk, 3, 3, (1, 1), (1, 1), (1, 1), 1, 160, 192)
    acc = torch.zeros((L.BN_ACC_SHARDS, 2, Kk), dtype=torch.float64, device=dev())
    esz = 2
    L.call("cvhip_conv2d_fprop_acc", C.byref(desc), xb.data_ptr() + 64 * esz, st.w_fprop.data_ptr(), ybig.data_ptr() + 32 * esz, acc.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    got = ybig[:, 32:96].float().cpu()
    assert rel_l2(got, ref) < 4e-3
    assert float(ybig[:, :32].abs().max()) == 0.0 and float(ybig[:, 96:].abs().max()) == 0.0
    s = acc.sum(0).cpu()
    assert rel_l2(s[0], ref.double().sum((0, 2, 3))) < 1e-3 or float((s[0] - ref.double().sum((0, 2, 3))).abs().max()) < 1e-2
    assert rel_l2(s[1], (ref.double() ** 2).sum((0, 2, 3))) < 1e-3


def _act(u, act, ap=0.1):
    if act == L.ACT_SILU:
        return u * torch.sigmoid(u)
    if act == L.ACT_RELU:
        return torch.relu(u)
    if act == L.ACT_LEAKY:
        return torch.where(u > 0, u, u * ap)
    if act == L.ACT_SIGMOID:
        return torch.sigmoid(u)
    if act == L.ACT_HSWISH:
        return u * torch.clamp(u + 3, 0, 6) / 6
    return u


def _fused(desc, xd, wimg, y, **kw):
    f = L.ConvFuse()
    keep = []
    for k, v in kw.items():
        if torch.is_tensor(v):
            keep.append(v)
            v = v.data_ptr()
        setattr(f, k, v)
    return L.fn("cvhip_conv2d_fprop_fused")(C.byref(desc), xd.data_ptr(), wimg.data_ptr(), y.data_ptr(), C.byref(f), ops._stream())


@pytest.mark.parametrize("act", [L.ACT_SILU, L.ACT_RELU, L.ACT_NONE])
@pytest.mark.parametrize("case", [PATCH_CASES[0], PATCH_CASES[1], PATCH_CASES[2], PATCH_CASES[3], PATCH_CASES[4], PATCH_CASES[5], PATCH_CASES[10]])
def test_patch_prologue_bn_act(case, act):
    """conv(act(scale*y + shift)) with the raw producer output y as the operand; padding is zero AFTER the activation; z_out is
    bit-identical to the stand-alone BN+activation pass"""
    N, Cc, H, W, Kk, R, S, s, p, d = case
    g = torch.Generator().manual_seed(7)
    yraw = rnd(torch.randn(N, Cc, H, W, generator=g) * 1.5)
    w = rnd(torch.randn(Kk, Cc, R, S, generator=g) / math.sqrt(Cc * R * S))
    sc = torch.rand(Cc, generator=g) + 0.5
    sh = torch.randn(Cc, generator=g) * 0.5 + 0.3    # a non-zero shift: act(shift) != 0, so a halo activated on load would show
    z = rnd(_act(yraw * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), act))
    ref = F.conv2d(z, w, None, stride=s, padding=p, dilation=d)
    st, Kp = K._prep(case, w, False)
    yd = to_nhwc_dev(yraw)
    P, Q = ref.shape[2:]
    out = ops.empty_nhwc(N, Kk, P, Q, dev(), ld=Kp)
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp, Kk if Kp != Kk else 0, 0)
    same = (P, Q) == (H, W)
    zo = torch.full((N, H, W, Cc), float("nan"), dtype=K.BF, device=dev()) if same else None
    assert L.load().cvhip_conv2d_fprop_prologue_ok(C.byref(desc), int(same)) == 1
    scd, shd = sc.to(dev()), sh.to(dev())
    kw = dict(pro_scale=scd, pro_shift=shd, pro_act=act, pro_act_param=0.1)
    if same:
        kw.update(z_out=zo, z_ld=Cc)
    L.check(_fused(desc, yd, st.w_fprop, out, **kw), "cvhip_conv2d_fprop_fused")
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert max_rel(got, ref) < 2 ** -7, max_rel(got, ref)
    assert rel_l2(got, ref) < 4e-3
    if same:
        zref = torch.empty((N, H, W, Cc), dtype=K.BF, device=dev())
        L.call("cvhip_bn_act_fwd", yd.data_ptr(), Cc, zref.data_ptr(), Cc, N * H * W, Cc, scd.data_ptr(), shd.data_ptr(), act, 0.1, None, 0, ops._stream())
        torch.cuda.synchronize()
        assert torch.equal(zo.view(torch.int16), zref.view(torch.int16)), "z_out differs from the stand-alone BN+activation pass"


EPI_CASES = [
    PATCH_CASES[0],                          # patch kernel
    PATCH_CASES[4],                          # patch kernel, padded K
    (2, 32, 32, 32, 64, 3, 3, 2, 1, 1),      # stride 2: per-tap implicit GEMM (256x64 tile)
    (26, 128, 80, 80, 256, 3, 3, 2, 1, 1),   # stride 2, 256x128 tile, staged epilogue
    (2, 64, 16, 16, 64, 1, 1, 2, 0, 1),      # 1x1 stride 2: per-tap kernel
    (2, 256, 10, 10, 255, 1, 1, 1, 0, 1),    # small 1x1: general kernel, direct (unpacked) stores
    (7, 64, 60, 60, 32, 1, 1, 1, 0, 1),      # streaming 1x1 kernel
    (7, 128, 60, 60, 255, 1, 1, 1, 0, 1),    # streaming 1x1, 256-wide tile, padded K
    (8, 8, 250, 250, 32, 6, 6, 2, 2, 1),     # stem kernel
]


@pytest.mark.parametrize("act", [L.ACT_SILU, L.ACT_RELU, L.ACT_LEAKY, L.ACT_HSWISH])
@pytest.mark.parametrize("case", EPI_CASES)
def test_fused_epilogue_every_kernel(case, act):
    """y = act((conv(x) + bias) * scale + shift) in the convolution's own pass, on every dense kernel of the library"""
    N, Cc, H, W, Kk, R, S, s, p, d = case
    x, w = K._mk(case, 11)
    g = torch.Generator().manual_seed(12)
    bias = torch.randn(Kk, generator=g) * 0.2
    sc = torch.rand(Kk, generator=g) + 0.5
    sh = torch.randn(Kk, generator=g) * 0.3
    conv = F.conv2d(x, w, bias, stride=s, padding=p, dilation=d)
    ref = _act(conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1), act)
    st, Kp = K._prep(case, w, False)
    xd = to_nhwc_dev(x)
    P, Q = ref.shape[2:]
    out = ops.empty_nhwc(N, Kk, P, Q, dev(), ld=Kp)
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp, Kk if Kp != Kk else 0, 0)
    pad = lambda v: torch.cat([v, torch.zeros(Kp - Kk)]).to(dev())
    L.check(_fused(desc, xd, st.w_fprop, out, bias=bias.to(dev()), ep_scale=pad(sc), ep_shift=pad(sh), ep_act=act, ep_act_param=0.1),
            "cvhip_conv2d_fprop_fused")
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert max_rel(got, ref) < 2 ** -7, max_rel(got, ref)
    assert rel_l2(got, ref) < 4e-3
    # activation only (a bias-free ConvModule without norm)
    out2 = ops.empty_nhwc(N, Kk, P, Q, dev(), ld=Kp)
    L.check(_fused(desc, xd, st.w_fprop, out2, ep_act=act, ep_act_param=0.1), "cvhip_conv2d_fprop_fused")
    torch.cuda.synchronize()
    ref2 = _act(F.conv2d(x, w, None, stride=s, padding=p, dilation=d), act)
    assert rel_l2(out2.float().cpu(), ref2) < 4e-3


def test_fused_refusals():
    lib = L.load()
    case = PATCH_CASES[0]
    N, Cc, H, W, Kk, R, S, s, p, d = case
    x, w = K._mk(case)
    st, Kp = K._prep(case, w, False)
    xd = to_nhwc_dev(x)
    out = ops.empty_nhwc(N, Kk, H, W, dev())
    desc = ops.conv_desc(N, Cc, H, W, Kk, R, S, (s, s), (p, p), (d, d), 1, Cc, Kk)
    v = torch.ones(Kk, device=dev())
    part = torch.zeros((4096, 2, Kk), device=dev())
    assert _fused(desc, xd, st.w_fprop, out, ep_scale=v) == L.ERR_INVALID                               # scale without shift
    assert _fused(desc, xd, st.w_fprop, out, stats_partial=part, ep_act=L.ACT_RELU) == L.ERR_INVALID   # sums + epilogue
    assert _fused(desc, xd, st.w_fprop, out, z_out=out, z_ld=Kk) == L.ERR_INVALID                      # z_out without a prologue
    # a prologue on a geometry only the per-tap kernels run is refused, not ignored
    d1 = ops.conv_desc(N, Cc, H, W, Kk, 3, 3, (2, 2), (1, 1), (1, 1), 1, Cc, Kk)   # stride-2 input
    out = ops.empty_nhwc(N, Kk, H // 2, W // 2, dev())
    vc = torch.ones(Cc, device=dev())
    assert lib.cvhip_conv2d_fprop_prologue_ok(C.byref(d1), 0) == 0
    assert _fused(d1, xd, st.w_fprop, out, pro_scale=vc, pro_shift=vc, pro_act=L.ACT_RELU) == L.ERR_UNSUPPORTED
