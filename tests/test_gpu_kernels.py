"""GPU parity tests (run with `-m gpu` on an MI355X): every HIP kernel, called through the C ABI
(cvpytorch_amd.lib / ops), against fp32 CPU arithmetic on the SAME bf16-rounded operands.

Tolerances (stated per SURVEY.md §8(a)):
  * conv / BN / activation outputs stored as bf16: one bf16 rounding of an fp32-accumulated value,
    i.e. max |err| <= 2^-8 * max|ref| and relative L2 <= 4e-3; fp32 outputs (wgrad, BN stats):
    relative L2 <= 1e-3 (summation order only).
  * max-pool values + argmax routing, nearest upsample, concat, Focus, head permute, NMS keep lists:
    bit-exact.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

BF = torch.bfloat16


def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


def max_rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def bf(x):
    return x.to(BF).float()


def to_nhwc_dev(x):
    return x.to(dev()).to(BF).contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------------------------------------------
# hardware layout probes
# ------------------------------------------------------------------------------------------------------
def test_probe_mfma_layout():
    torch.manual_seed(0)
    a = torch.randn(16, 32).to(BF)
    b = torch.randn(32, 16).to(BF)  # asymmetric on purpose (transpose-detecting)
    d = torch.zeros(16, 16, device=dev())
    ad, bd = a.to(dev()), b.to(dev())  # keep the device copies alive across the launch
    L.call("cvhip_probe_mfma_16x16x32", ad.data_ptr(), bd.data_ptr(), d.data_ptr(), None)
    torch.cuda.synchronize()
    ref = a.float() @ b.float()
    assert rel_l2(d, ref) < 1e-5


def test_probe_ds_read_tr16():
    src = torch.arange(256, dtype=torch.float32).to(BF)
    out = torch.zeros(256, dtype=BF, device=dev())
    srcd = src.to(dev())
    L.call("cvhip_probe_ds_read_tr16", srcd.data_ptr(), out.data_ptr(), None)
    torch.cuda.synchronize()
    got = out.float().cpu().view(64, 4)
    exp = torch.empty(64, 4)
    for l in range(64):
        for j in range(4):
            exp[l, j] = (l & 15) + j * 16 + (l >> 4) * 64
    assert torch.equal(got, exp), got[:20]


# ------------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, C, H, W, K, R, S, stride, pad, dil
    (2, 32, 20, 20, 32, 1, 1, 1, 0, 1),
    (2, 64, 20, 24, 128, 3, 3, 1, 1, 1),
    (2, 128, 17, 19, 128, 3, 3, 1, 1, 1),
    (2, 32, 32, 32, 64, 3, 3, 2, 1, 1),
    (3, 64, 21, 23, 64, 3, 3, 2, 1, 1),
    (2, 8, 64, 64, 32, 6, 6, 2, 2, 1),
    (2, 256, 10, 10, 255, 1, 1, 1, 0, 1),
    (1, 512, 8, 8, 256, 1, 1, 1, 0, 1),
    (2, 64, 16, 16, 48, 3, 3, 1, 2, 2),
    (2, 64, 16, 16, 64, 1, 1, 2, 0, 1),
    (1, 16, 40, 40, 16, 3, 3, 1, 1, 1),
    (2, 128, 12, 12, 19, 1, 1, 1, 0, 1),
    (1, 40, 9, 9, 24, 3, 3, 1, 1, 1),
    (64, 32, 8, 8, 32, 3, 3, 1, 1, 1),
    # large enough for the 256x128 block / 128x64 wave-tile configuration (>= 384 tiles), ragged last tile
    (62, 128, 40, 40, 128, 3, 3, 1, 1, 1),
    (25, 64, 64, 64, 256, 1, 1, 1, 0, 1),
    (26, 128, 80, 80, 256, 3, 3, 2, 1, 1),
    # 1x1 / stride 1 with >= 192 row tiles: the persistent streaming kernel (conv1x1_stream.hip) — ragged last tile, reduced
    # channels not a multiple of 32, two 128-channel pipeline chunks, 32/64/128/256-wide output tiles (fprop and dgrad roles)
    (7, 64, 60, 60, 32, 1, 1, 1, 0, 1),
    (7, 40, 60, 60, 128, 1, 1, 1, 0, 1),
    (7, 128, 60, 60, 255, 1, 1, 1, 0, 1),
    (7, 24, 60, 60, 64, 1, 1, 1, 0, 1),
    (4, 256, 80, 80, 128, 1, 1, 1, 0, 1),
    (8, 64, 80, 80, 512, 1, 1, 1, 0, 1),   # 512 outputs without BN sums: four 128-wide column tiles (grid.y) of the streaming kernel
    # 8-channel image stems with >= 512 4x64 output tiles: the direct patch kernel (conv_stem.hip) — k6 s2 (YOLOv5), k3 s1
    # (YOLOv7), k7 s2 with 24 output channels; ragged tiles in both directions
    (8, 8, 250, 250, 32, 6, 6, 2, 2, 1),
    (5, 8, 150, 150, 32, 3, 3, 1, 1, 1),
    (8, 8, 250, 250, 24, 7, 7, 2, 3, 1),
]
STREAM_CASES = CONV_CASES[-9:-3]
STEM_CASES = CONV_CASES[-3:]


def test_stem_cases_take_the_stem_kernel():
    lib = L.load()
    for N, Cc, H, W, K, R, S, s, p, d in STEM_CASES:
        desc = ops.conv_desc(N, Cc, H, W, K, R, S, (s, s), (p, p), (d, d), 1, Cc, K)
        assert lib.cvhip_conv_stem_blocks(C.byref(desc)) > 0
        assert lib.cvhip_conv2d_fprop_stats_rows(C.byref(desc)) == lib.cvhip_conv_stem_blocks(C.byref(desc))
    small = ops.conv_desc(2, 8, 64, 64, 32, 6, 6, (2, 2), (2, 2), (1, 1), 1, 8, 32)
    assert lib.cvhip_conv_stem_blocks(C.byref(small)) == 0


def test_stream1x1_cases_take_the_streaming_kernel():
    lib = L.load()
    for N, Cc, H, W, K, R, S, s, p, d in STREAM_CASES:
        Kp = (K + 7) // 8 * 8
        assert lib.cvhip_conv1x1_stream_blocks(Kp, Cc, N * H * W, 0) > 0, (K, Cc)        # fprop + bias
        assert lib.cvhip_conv1x1_stream_blocks(Cc, Kp, N * H * W, 0) > 0, (K, Cc)        # dgrad
        if K % 8 == 0 and K <= 128:   # with BN sums only single-column-tile problems stream
            desc = ops.conv_desc(N, Cc, H, W, K, R, S, (s, s), (p, p), (d, d), 1, Cc, K)
            assert lib.cvhip_conv2d_fprop_stats_rows(C.byref(desc)) == lib.cvhip_conv1x1_stream_blocks(K, Cc, N * H * W, 1) > 0
    assert lib.cvhip_conv1x1_stream_blocks(512, 64, 8 * 80 * 80, 1) == 0   # wide outputs + BN sums: general kernel
    assert lib.cvhip_conv1x1_stream_blocks(128, 512, 1 << 20, 1) == 0      # weight tile does not fit the LDS budget
    assert lib.cvhip_conv1x1_stream_blocks(64, 64, 4096, 1) == 0            # too few row tiles to cover the chip


def _mk(case, seed=0):
    N, Cc, H, W, K, R, S, s, p, d = case
    g = torch.Generator().manual_seed(seed)
    x = bf(torch.randn(N, Cc, H, W, generator=g))
    w = bf(torch.randn(K, Cc, R, S, generator=g) / math.sqrt(Cc * R * S))
    return x, w


def _prep(case, w, need_dgrad=True):
    N, Cc, H, W, K, R, S, s, p, d = case
    Kp = (K + 7) // 8 * 8
    st = ops.ConvState()
    pdesc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp, K if Kp != K else 0, 0)
    st.prepare(w.to(dev()).contiguous(memory_format=torch.channels_last), pdesc, need_dgrad, ("test", id(w)))
    return st, Kp


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fprop(case):
    N, Cc, H, W, K, R, S, s, p, d = case
    x, w = _mk(case)
    bias = torch.randn(K)
    ref = F.conv2d(x, w, bias, stride=s, padding=p, dilation=d)
    st, Kp = _prep(case, w, False)
    xd = to_nhwc_dev(x)
    P, Q = ref.shape[2:]
    y = ops.empty_nhwc(N, K, P, Q, dev(), ld=Kp)
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp, K if Kp != K else 0, 0)
    b = bias.to(dev())  # K real entries: the kernel guards the padded channels (cvhip_conv_desc.k_valid)
    L.call("cvhip_conv2d_fprop", C.byref(desc), xd.data_ptr(), st.w_fprop.data_ptr(), b.data_ptr(), y.data_ptr(), None, ops._stream())
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert max_rel(got, ref) < 2 ** -7, max_rel(got, ref)
    assert rel_l2(got, ref) < 4e-3


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fprop_bn_stats(case):
    N, Cc, H, W, K, R, S, s, p, d = case
    if K % 8:
        pytest.skip("epilogue statistics need K % 8 == 0 (host falls back to the reduction pass)")
    x, w = _mk(case, 1)
    ref = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    st, Kp = _prep(case, w, False)
    xd = to_nhwc_dev(x)
    P, Q = ref.shape[2:]
    y = ops.empty_nhwc(N, K, P, Q, dev())
    desc = ops.conv_desc(N, Cc, H, W, K, R, S, (s, s), (p, p), (d, d), 1, Cc, K)
    rows = L.load().cvhip_conv2d_fprop_stats_rows(C.byref(desc))
    part = torch.full((rows, 2, K), float("nan"), device=dev())
    L.call("cvhip_conv2d_fprop", C.byref(desc), xd.data_ptr(), st.w_fprop.data_ptr(), None, y.data_ptr(), part.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    s1 = part[:, 0].double().sum(0).cpu()
    s2 = part[:, 1].double().sum(0).cpu()
    r = ref.double()
    assert rel_l2(s1, r.sum((0, 2, 3))) < 1e-3 or float((s1 - r.sum((0, 2, 3))).abs().max()) < 1e-2
    assert rel_l2(s2, (r * r).sum((0, 2, 3))) < 1e-3
    assert rel_l2(y.float().cpu(), ref) < 4e-3


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(case):
    N, Cc, H, W, K, R, S, s, p, d = case
    x, w = _mk(case, 2)
    xr = x.clone().requires_grad_(True)
    y = F.conv2d(xr, w, None, stride=s, padding=p, dilation=d)
    dy = bf(torch.randn(y.shape, generator=torch.Generator().manual_seed(3)))
    (gx,) = torch.autograd.grad(y, xr, dy)
    st, Kp = _prep(case, w, True)
    P, Q = y.shape[2:]
    dyd = torch.zeros((N, P, Q, Kp), dtype=BF, device=dev())
    dyd[..., :K] = dy.permute(0, 2, 3, 1).to(dev()).to(BF)
    dx = ops.empty_nhwc(N, Cc, H, W, dev())
    dx.fill_(float("nan"))
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp)
    L.call("cvhip_conv2d_dgrad", C.byref(desc), dyd.data_ptr(), st.w_dgrad.data_ptr(), dx.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    got = dx.float().cpu()
    assert torch.isfinite(got).all()
    assert max_rel(got, gx) < 2 ** -7, max_rel(got, gx)
    assert rel_l2(got, gx) < 4e-3


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(case):
    N, Cc, H, W, K, R, S, s, p, d = case
    x, w = _mk(case, 4)
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(x, wr, None, stride=s, padding=p, dilation=d)
    dy = bf(torch.randn(y.shape, generator=torch.Generator().manual_seed(5)))
    (gw,) = torch.autograd.grad(y, wr, dy)
    Kp = (K + 7) // 8 * 8
    P, Q = y.shape[2:]
    xd = to_nhwc_dev(x)
    dyd = torch.zeros((N, P, Q, Kp), dtype=BF, device=dev())
    dyd[..., :K] = dy.permute(0, 2, 3, 1).to(dev()).to(BF)
    dw = torch.full((Kp, R, S, Cc), float("nan"), device=dev())
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp)
    L.call("cvhip_conv2d_wgrad", C.byref(desc), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), 0, ops._stream())
    torch.cuda.synchronize()
    got = dw[:K].permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, gw) < 1e-3, rel_l2(got, gw)
    if Kp != K:
        assert float(dw[K:].abs().max()) == 0.0


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad_deterministic(case):
    """cvhip_conv2d_wgrad_det: per-split slabs + ordered fold — same values as the atomic kernel to rounding, BIT-identical
    from call to call, and accumulate=1 adds onto the destination"""
    N, Cc, H, W, K, R, S, s, p, d = case
    x, w = _mk(case, 4)
    wr = w.clone().requires_grad_(True)
    y = F.conv2d(x, wr, None, stride=s, padding=p, dilation=d)
    dy = bf(torch.randn(y.shape, generator=torch.Generator().manual_seed(5)))
    (gw,) = torch.autograd.grad(y, wr, dy)
    Kp = (K + 7) // 8 * 8
    P, Q = y.shape[2:]
    xd = to_nhwc_dev(x)
    dyd = torch.zeros((N, P, Q, Kp), dtype=BF, device=dev())
    dyd[..., :K] = dy.permute(0, 2, 3, 1).to(dev()).to(BF)
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp)
    nb = int(L.load().cvhip_conv2d_wgrad_det_workspace_bytes(C.byref(desc)))
    assert nb >= 0
    ws = torch.empty((max(nb, 16),), dtype=torch.uint8, device=dev())
    outs = []
    for rep in range(3):
        dw = torch.full((Kp, R, S, Cc), float("nan"), device=dev())
        ws.random_(0, 255)                                   # stale workspace contents must not matter
        L.call("cvhip_conv2d_wgrad_det", C.byref(desc), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), 0, ws.data_ptr(), nb, ops._stream())
        torch.cuda.synchronize()
        outs.append(dw.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    got = outs[0][:K].permute(0, 3, 1, 2).cpu()
    assert torch.isfinite(got).all()
    assert rel_l2(got, gw) < 1e-3, rel_l2(got, gw)
    base = torch.full((Kp, R, S, Cc), 0.5, device=dev())
    L.call("cvhip_conv2d_wgrad_det", C.byref(desc), xd.data_ptr(), dyd.data_ptr(), base.data_ptr(), 1, ws.data_ptr(), nb, ops._stream())
    torch.cuda.synchronize()
    assert torch.allclose(base, outs[0] + 0.5, rtol=2e-6, atol=1e-6 * float(outs[0].abs().max()) + 1e-5)      # (the fold adds the old value in its own fixed order)
    # a workspace that is too small is refused, not overrun
    if nb > 16:
        rc = L.load().cvhip_conv2d_wgrad_det(C.byref(desc), xd.data_ptr(), dyd.data_ptr(), base.data_ptr(), 0, ws.data_ptr(), nb - 4, ops._stream())
        assert rc != 0


def test_conv_channel_slice_operands():
    """x read from / y written into channel slices of wider buffers (pitch != channels)."""
    torch.manual_seed(0)
    N, Cc, H, W, K = 2, 32, 12, 12, 64
    xbig = bf(torch.randn(N, 96, H, W))
    w = bf(torch.randn(K, Cc, 3, 3) / 17)
    ref = F.conv2d(xbig[:, 32:64], w, None, padding=1)
    xb = to_nhwc_dev(xbig)
    ybig = torch.zeros((N, 160, H, W), dtype=BF, device=dev()).contiguous(memory_format=torch.channels_last)
    st = ops.ConvState()
    st.prepare(w.to(dev()).contiguous(memory_format=torch.channels_last), ops.conv_desc(N, Cc, H, W, K, 3, 3, (1, 1), (1, 1), (1, 1), 1, Cc, K), False, ("slice",))
    desc = ops.conv_desc(N, Cc, H, W, K, 3, 3, (1, 1), (1, 1), (1, 1), 1, 96, 160)
    L.call("cvhip_conv2d_fprop", C.byref(desc), xb.data_ptr() + 2 * 32, st.w_fprop.data_ptr(), None, ybig.data_ptr() + 2 * 64, None, ops._stream())
    torch.cuda.synchronize()
    out = ybig.float().cpu()
    assert rel_l2(out[:, 64:128], ref) < 4e-3
    assert float(out[:, :64].abs().max()) == 0.0 and float(out[:, 128:].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------------
# depthwise
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Cc,H,W,s,p,d", [(32, 12, 14, 1, 1, 1), (64, 10, 10, 1, 3, 3), (48, 11, 9, 2, 1, 1), (20, 8, 8, 1, 1, 1),
                                         # 3x3/s1/d1 register-window fast path: several row segments, ragged tails, pad 0 / 2, > 256 channels
                                         (304, 21, 200, 1, 1, 1), (16, 9, 130, 1, 0, 1), (24, 9, 70, 1, 2, 1), (2304, 5, 7, 1, 1, 1),
                                         # LDS-ring kernel (dw3x3_lds_kernel): 32-vector chunks, two chunks with a ragged second one (65 vectors),
                                         # pad 0 and pad 2, strips that do not divide the width, row blocks that do not divide the height
                                         (256, 40, 64, 1, 1, 1), (520, 17, 40, 1, 1, 1), (304, 33, 50, 1, 0, 1), (128, 19, 120, 1, 2, 1),
                                         # 3x3 / stride 2 / pad 1 input gradient by tap parity (dw3x3_s2_dgrad_kernel, round 6): even and odd
                                         # sizes (the last odd row / column has no (i + 1) / 2 output), > 256 channels, one-pixel outputs
                                         (64, 16, 20, 2, 1, 1), (128, 9, 8, 2, 1, 1), (2304, 6, 5, 2, 1, 1), (32, 2, 2, 2, 1, 1), (8, 1, 7, 2, 1, 1)])
def test_depthwise(Cc, H, W, s, p, d):
    torch.manual_seed(0)
    N = 2
    x = bf(torch.randn(N, Cc, H, W)).requires_grad_(True)
    w = torch.randn(Cc, 1, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, None, stride=s, padding=p, dilation=d, groups=Cc)
    dy = bf(torch.randn_like(y))
    gx, gw = torch.autograd.grad(y, (x, w), dy)
    P, Q = y.shape[2:]
    desc = ops.conv_desc(N, Cc, H, W, Cc, 3, 3, (s, s), (p, p), (d, d), Cc, Cc, Cc)
    xd, dyd = to_nhwc_dev(x.detach()), to_nhwc_dev(dy)
    wd = w.detach().reshape(Cc, 3, 3).contiguous().to(dev())
    yd = ops.empty_nhwc(N, Cc, P, Q, dev())
    L.call("cvhip_dwconv2d_fprop", C.byref(desc), xd.data_ptr(), wd.data_ptr(), None, yd.data_ptr(), ops._stream())
    dxd = ops.empty_nhwc(N, Cc, H, W, dev())
    L.call("cvhip_dwconv2d_dgrad", C.byref(desc), dyd.data_ptr(), wd.data_ptr(), dxd.data_ptr(), ops._stream())
    dwd = torch.full((Cc, 3, 3), float("nan"), device=dev())
    L.call("cvhip_dwconv2d_wgrad", C.byref(desc), xd.data_ptr(), dyd.data_ptr(), dwd.data_ptr(), 0, ops._stream())
    torch.cuda.synchronize()
    assert rel_l2(yd.float().cpu(), y.detach()) < 4e-3
    assert rel_l2(dxd.float().cpu(), gx) < 4e-3
    assert rel_l2(dwd.cpu().reshape(Cc, 1, 3, 3), gw) < 1e-3


# ------------------------------------------------------------------------------------------------------
# BN + activation
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("act,name", [(L.ACT_SILU, "silu"), (L.ACT_RELU, "relu"), (L.ACT_LEAKY, "leaky"), (L.ACT_NONE, "none")])
@pytest.mark.parametrize("Cc,M_hw", [(32, (9, 11)), (256, (5, 5)), (24, (7, 3)), (20, (4, 4))])
def test_bn_act_fwd_bwd(act, name, Cc, M_hw):
    torch.manual_seed(0)
    N = 3
    H, W = M_hw
    y = bf(torch.randn(N, Cc, H, W) * 2 + 0.5)
    res = bf(torch.randn(N, Cc, H, W))
    gamma, beta = torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.3
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    fn = {L.ACT_SILU: F.silu, L.ACT_RELU: F.relu, L.ACT_LEAKY: lambda t: F.leaky_relu(t, 0.1), L.ACT_NONE: lambda t: t}[act]
    yr = y.clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    zr = fn(F.batch_norm(yr, rm_ref, rv_ref, g_, b_, True, 0.03, 1e-3)) + res
    dz = bf(torch.randn_like(zr))
    gy, gg, gb = torch.autograd.grad(zr, (yr, g_, b_), dz)
    yd = to_nhwc_dev(y).requires_grad_(True)
    gd, bd = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    rmd, rvd = rm.to(dev()), rv.to(dev())
    z = ops.bn_act(yd, gd, bd, rmd, rvd, to_nhwc_dev(res), True, True, 0.03, 1e-3, act, 0.1, True)
    z.backward(to_nhwc_dev(dz))
    torch.cuda.synchronize()
    assert rel_l2(z.detach().float().cpu(), zr.detach()) < 4e-3
    assert rel_l2(yd.grad.float().cpu(), gy) < 6e-3
    assert rel_l2(gd.grad.cpu(), gg) < 2e-3 and rel_l2(bd.grad.cpu(), gb) < 2e-3
    assert rel_l2(rmd.cpu(), rm_ref) < 1e-4 and rel_l2(rvd.cpu(), rv_ref) < 1e-4


# ------------------------------------------------------------------------------------------------------
# exact glue ops
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("k,s,p,Cc,H,W", [(5, 1, 2, 32, 13, 11), (9, 1, 4, 16, 10, 10), (13, 1, 6, 8, 12, 12), (2, 2, 0, 24, 12, 10), (3, 2, 1, 20, 11, 13)])
def test_maxpool_exact(k, s, p, Cc, H, W):
    torch.manual_seed(0)
    # coarse values => many ties: exercises the first-max tie rule
    x = (torch.randint(-3, 4, (2, Cc, H, W)).float() * 0.5)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, k, s, p)
    dy = bf(torch.randn_like(yr))
    (gx,) = torch.autograd.grad(yr, xr, dy)
    xd = to_nhwc_dev(x).requires_grad_(True)
    y = ops.max_pool2d(xd, k, s, p)
    y.backward(to_nhwc_dev(dy))
    torch.cuda.synchronize()
    assert torch.equal(y.detach().float().cpu(), yr.detach())
    # gradient routing is exact; sums of bf16 cotangents are rounded once to bf16
    assert rel_l2(xd.grad.float().cpu(), gx) < 4e-3
    assert torch.equal(xd.grad.float().cpu() != 0, bf(gx) != 0) or rel_l2(xd.grad.float().cpu(), gx) < 1e-3


@pytest.mark.parametrize("k,N,Cc,H,W", [(5, 16, 512, 20, 20), (5, 8, 1024, 13, 17), (3, 32, 256, 32, 32), (5, 130, 8, 20, 20),
                                         # round 6: k = 9 / 13 (SPP / SPPCSPC), and maps that fit the LDS with 2 or 1 channel vectors per block only
                                         (9, 16, 256, 20, 20), (13, 16, 256, 20, 20), (13, 4, 512, 40, 40), (9, 4, 512, 40, 40), (5, 4, 512, 40, 40),
                                         (13, 4, 512, 48, 64), (9, 8, 128, 11, 7)])
def test_maxpool_lds_path_values_argmax_and_gradient(k, N, Cc, H, W):
    """stride-1 "same" pools of small maps with >= 128 (image, 64-channel) blocks run through the LDS kernels (pool_resize.hip,
    round 4): values AND arg-max bytes (tap index kh * k + kw) equal to ATen's max_pool2d_with_indices — ties (coarse values), -inf
    borders and NaNs included — and the backward routes every cotangent to the same input pixel."""
    import ctypes as C
    from cvpytorch_amd import lib as L
    torch.manual_seed(k * 100 + N)
    x = (torch.randint(-3, 4, (N, Cc, H, W)).float() * 0.5)
    x[0, 0, 3, 4] = float("nan")
    x[1, 1, 0, 0] = float("nan")
    x[1, 1, 0, 1] = float("nan")
    x[2, 2] = float("-inf")
    p = k // 2
    yr, ir = F.max_pool2d(x, k, 1, p, return_indices=True)
    xd = to_nhwc_dev(x)
    x4, ld = ops.as_nhwc(xd)
    y = ops.empty_nhwc(N, Cc, H, W, xd.device)
    idx = torch.empty((N, H, W, Cc), dtype=torch.uint8, device=xd.device)
    L.call("cvhip_maxpool2d_fwd", x4.data_ptr(), ld, y.data_ptr(), Cc, idx.data_ptr(), N, Cc, H, W, k, 1, p, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    yc = y.float().cpu()
    assert torch.equal(torch.nan_to_num(yc, nan=123.0), torch.nan_to_num(yr, nan=123.0))
    # ATen's index is the flat input position ih * W + iw; ours the tap (kh, kw) of the window of output (oh, ow)
    tap = idx.permute(0, 3, 1, 2).cpu().long()
    oh = torch.arange(H).view(1, 1, H, 1)
    ow = torch.arange(W).view(1, 1, 1, W)
    flat = (oh + tap // k - p) * W + (ow + tap % k - p)
    assert torch.equal(flat, ir)
    dy = bf(torch.randn(N, Cc, H, W))
    xr = x.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(F.max_pool2d(xr, k, 1, p), xr, dy)
    dyd = to_nhwc_dev(dy)
    d4, ldd = ops.as_nhwc(dyd)
    dx = ops.empty_nhwc(N, Cc, H, W, xd.device)
    L.call("cvhip_maxpool2d_bwd", d4.data_ptr(), ldd, idx.data_ptr(), dx.data_ptr(), Cc, N, Cc, H, W, k, 1, p, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rel_l2(dx.float().cpu(), gx) < 4e-3
    assert torch.equal(dx.float().cpu() != 0, bf(gx) != 0) or rel_l2(dx.float().cpu(), gx) < 1e-3


def test_upsample_cat_exact():
    torch.manual_seed(0)
    a, b = bf(torch.randn(2, 32, 5, 7)), bf(torch.randn(2, 24, 10, 14))
    ad, bd = to_nhwc_dev(a).requires_grad_(True), to_nhwc_dev(b).requires_grad_(True)
    out = ops.upsample2x_cat(ad, bd)
    ref = torch.cat([F.interpolate(a, scale_factor=2, mode="nearest"), b], 1)
    assert torch.equal(out.detach().float().cpu(), ref)
    dout = bf(torch.randn_like(ref))
    out.backward(to_nhwc_dev(dout))
    torch.cuda.synchronize()
    ga = dout[:, :32].view(2, 32, 5, 2, 7, 2).sum((3, 5))
    assert rel_l2(ad.grad.float().cpu(), ga) < 4e-3
    assert torch.equal(bd.grad.float().cpu(), dout[:, 32:])


def test_cat_add_exact():
    torch.manual_seed(0)
    xs = [bf(torch.randn(2, c, 6, 5)) for c in (16, 8, 40)]
    xd = [to_nhwc_dev(x).requires_grad_(True) for x in xs]
    out = ops.cat(xd)
    assert torch.equal(out.detach().float().cpu(), torch.cat(xs, 1))
    g = bf(torch.randn(2, 64, 6, 5))
    out.backward(to_nhwc_dev(g))
    for t, (o, c) in zip(xd, [(0, 16), (16, 8), (24, 40)]):
        assert torch.equal(t.grad.float().cpu(), g[:, o:o + c])
    s = ops.add(xd[0].detach(), xd[0].detach())
    assert torch.equal(s.float().cpu(), bf(xs[0] * 2))


def test_images_and_focus_layout():
    torch.manual_seed(0)
    x = torch.randn(2, 3, 16, 20)
    y = ops.images_to_nhwc(x.to(dev()), cpad=8)
    assert torch.equal(y[:, :3].float().cpu(), bf(x)) and float(y[:, 3:].abs().max()) == 0
    f = ops.images_to_nhwc(x.to(dev()), cpad=16, focus=True)
    ref = torch.cat((x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]), 1)
    assert torch.equal(f[:, :12].float().cpu(), bf(ref)) and float(f[:, 12:].abs().max()) == 0


def test_head_permute_exact():
    torch.manual_seed(0)
    N, A, NO, H, W = 2, 3, 85, 6, 5
    x = bf(torch.randn(N, A * NO, H, W))
    buf = torch.zeros((N, H, W, 256), dtype=BF, device=dev())
    buf[..., :255] = x.permute(0, 2, 3, 1).to(dev()).to(BF)
    xd = buf.permute(0, 3, 1, 2)[:, :255].requires_grad_(True)
    out = ops.head_permute(xd, A, NO)
    ref = x.view(N, A, NO, H, W).permute(0, 1, 3, 4, 2).contiguous()
    assert torch.equal(out.detach().cpu(), ref)
    g = torch.randn_like(ref)
    (gx,) = torch.autograd.grad(out, xd, g.to(dev()))
    exp = bf(g.permute(0, 1, 4, 2, 3).reshape(N, A * NO, H, W))
    assert torch.equal(gx.float().cpu(), exp)


@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("Hi,Wi,Ho,Wo", [(4, 8, 16, 32), (1, 1, 4, 8), (5, 7, 20, 21), (16, 16, 8, 8),
                                         (4, 6, 32, 48), (3, 5, 13, 27), (2, 3, 19, 12)])   # x8, ragged x4.3 / x5.4 (separable backward), x9.5 / x4
def test_bilinear(align, Hi, Wi, Ho, Wo):
    torch.manual_seed(0)
    x = bf(torch.randn(2, 24, Hi, Wi))
    xr = x.clone().requires_grad_(True)
    yr = F.interpolate(xr, size=(Ho, Wo), mode="bilinear", align_corners=align)
    dy = bf(torch.randn_like(yr))
    (gx,) = torch.autograd.grad(yr, xr, dy)
    xd = to_nhwc_dev(x).requires_grad_(True)
    y = ops.resize_bilinear(xd, (Ho, Wo), align)
    y.backward(to_nhwc_dev(dy))
    torch.cuda.synchronize()
    assert rel_l2(y.detach().float().cpu(), yr.detach()) < 4e-3
    assert rel_l2(xd.grad.float().cpu(), gx) < 4e-3
    # the separable two-pass backward (ratios >= 4 on both axes) and the gather form are the same sum: equal to fp32 rounding
    nb = int(L.load().cvhip_resize_bilinear_bwd_workspace_bytes(2, 24, Hi, Wi, Ho, Wo))
    assert (nb > 0) == (Ho >= 4 * Hi and Wo >= 4 * Wi)
    if nb > 0:
        dyd = to_nhwc_dev(dy)
        a = ops.empty_nhwc(2, 24, Hi, Wi, dev())
        b = ops.empty_nhwc(2, 24, Hi, Wi, dev())
        ws = torch.empty((nb,), dtype=torch.uint8, device=dev())
        L.call("cvhip_resize_bilinear_bwd", dyd.data_ptr(), 24, a.data_ptr(), 24, 2, 24, Hi, Wi, Ho, Wo, int(align), ops._stream())
        L.call("cvhip_resize_bilinear_bwd_ws", dyd.data_ptr(), 24, b.data_ptr(), 24, 2, 24, Hi, Wi, Ho, Wo, int(align), ws.data_ptr(), nb, ops._stream())
        torch.cuda.synchronize()
        assert rel_l2(b.float().cpu(), a.float().cpu()) < 2e-3
        assert rel_l2(b.float().cpu(), gx) < 4e-3


@pytest.mark.parametrize("shape", [(3, 40, 7, 9), (2, 128, 32, 64), (2, 1024, 16, 32), (3, 72, 17, 31), (2, 40, 16, 16)])
def test_global_avg_pool(shape):
    """(the larger maps run the 1024-thread vector kernel of round 6: >= 256 pixels, C % 8 == 0; ragged row counts and channel tails)"""
    torch.manual_seed(0)
    N, Cc, H, W = shape
    x = bf(torch.randn(N, Cc, H, W) + 0.3)
    xd = to_nhwc_dev(x).requires_grad_(True)
    y = ops.global_avg_pool(xd)
    assert rel_l2(y.detach().float().cpu(), x.mean((2, 3), keepdim=True)) < 4e-3
    g = bf(torch.randn(N, Cc, 1, 1))
    y.backward(to_nhwc_dev(g))
    assert rel_l2(xd.grad.float().cpu(), (g / (H * W)).expand(N, Cc, H, W)) < 4e-3


# ------------------------------------------------------------------------------------------------------
# post-processing
# ------------------------------------------------------------------------------------------------------
def _boxes(n, seed, grid=False):
    g = torch.Generator().manual_seed(seed)
    xy = torch.rand(n, 2, generator=g) * 200
    wh = torch.rand(n, 2, generator=g) * 60 + 1
    if grid:  # coarse coordinates => exact IoU ties at the threshold
        xy, wh = (xy / 8).round() * 8, (wh / 8).round() * 8 + 8
    b = torch.cat([xy, xy + wh], 1)
    s = torch.rand(n, generator=g)
    if grid:
        s = (s * 20).round() / 20  # duplicate scores: exercises the stable sort
    return b, s


@pytest.mark.parametrize("n,seed,grid", [(1, 0, False), (63, 1, False), (64, 2, False), (65, 3, True), (1000, 4, False), (1000, 5, True), (5000, 6, False)])
def test_nms_bit_exact(n, seed, grid):
    from oracle import torch_ref as R
    b, s = _boxes(n, seed, grid)
    for thr in (0.45, 0.6, 0.5):
        keep = ops.nms(b.to(dev()), s.to(dev()), thr).cpu()
        ref = R.nms(b, s, thr)
        assert keep.dtype == torch.int64
        assert torch.equal(keep, ref), (n, thr, keep[:10], ref[:10])


def test_nms_degenerate_and_empty():
    from oracle import torch_ref as R
    assert ops.nms(torch.zeros((0, 4), device=dev()), torch.zeros((0,), device=dev()), 0.5).numel() == 0
    b = torch.tensor([[0., 0., 0., 0.], [0., 0., 0., 0.], [1., 1., 2., 2.], [1., 1., 2., 2.], [5., 5., 5., 9.]])
    s = torch.tensor([0.5, 0.5, 0.9, 0.9, 0.1])
    assert torch.equal(ops.nms(b.to(dev()), s.to(dev()), 0.5).cpu(), R.nms(b, s, 0.5))


def test_box_iou_matches_reference_vectors(golden_dir):
    z = np.load(golden_dir + "/box_iou.npz")
    a, b = torch.from_numpy(z["a"]), torch.from_numpy(z["b"])
    got = ops.box_iou(a.to(dev()), b.to(dev())).cpu()
    assert rel_l2(got, torch.from_numpy(z["iou"])) < 1e-6
    k = np.load(golden_dir + "/bbox_overlaps_kat.npz")  # the reference's own known-answer vector
    got = ops.box_iou(torch.from_numpy(k["b1"]).to(dev()), torch.from_numpy(k["b2"]).to(dev())).cpu()
    assert torch.allclose(got, torch.tensor([[0.5, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 0.0]]), atol=1e-6)


def test_yolov5_decode():
    from oracle import torch_ref as R
    torch.manual_seed(0)
    det = R.YOLOv5Detect(80, in_channels=(32, 64, 128))
    det.eval()
    feats = [bf(torch.randn(2, 255, s, s)) for s in (8, 4, 2)]
    # oracle decode of the same (already convolved) tensors
    zs = []
    for i, f in enumerate(feats):
        bs, _, ny, nx = f.shape
        t = f.view(bs, 3, 85, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
        yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
        grid = torch.stack((xv, yv), 2).expand((1, 3, ny, nx, 2)).float()
        ag = (det.anchors[i] * det.stride[i]).view(1, 3, 1, 1, 2)
        y = t.sigmoid()
        y[..., 0:2] = (y[..., 0:2] * 2 - 0.5 + grid) * det.stride[i]
        y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * ag
        zs.append(y.view(bs, -1, 85))
    ref = torch.cat(zs, 1)
    lv = []
    for f in feats:
        buf = torch.zeros((2, f.shape[2], f.shape[3], 256), dtype=BF, device=dev())
        buf[..., :255] = f.permute(0, 2, 3, 1).to(dev()).to(BF)
        lv.append(buf.permute(0, 3, 1, 2)[:, :255])
    got = ops.yolov5_decode(lv, det.stride, [det.anchors[i] * det.stride[i] for i in range(3)], 3, 85).cpu()
    assert got.shape == ref.shape
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5


def test_padded_channels_in_engine():
    """3-channel image stem (C padded to 8) and 21-channel output (K padded to 24): master weight, bias and
    their gradients keep the REAL shapes; padding happens inside libcvhip (cvhip_conv_desc.k_valid / c_valid)."""
    torch.manual_seed(0)
    from cvpytorch_amd import bricks
    m = bricks.HipConv2d(3, 21, 3, stride=2, padding=1).to(dev())
    x = torch.randn(2, 3, 16, 20)
    ref = F.conv2d(bf(x), bf(m.weight.detach().cpu()), m.bias.detach().cpu(), stride=2, padding=1)
    y = m(x.to(dev()))
    assert tuple(y.shape) == (2, 21, 8, 10)
    assert rel_l2(y.float().cpu(), ref) < 4e-3
    g = bf(torch.randn_like(ref))
    y.backward(g.to(dev()).to(BF))
    wr = bf(m.weight.detach().cpu()).requires_grad_(True)
    br = m.bias.detach().cpu().clone().requires_grad_(True)
    F.conv2d(bf(x), wr, br, stride=2, padding=1).backward(g)
    assert tuple(m.weight.grad.shape) == (21, 3, 3, 3)
    assert rel_l2(m.weight.grad.cpu(), wr.grad) < 2e-3
    assert rel_l2(m.bias.grad.cpu(), br.grad) < 2e-3


def test_zero_fill_and_unpad_add():
    t = torch.full((1000,), 7.0, device=dev())
    ops.zero_fill(t)
    assert float(t.abs().max()) == 0.0
    src = torch.randn(8, 9, 16, device=dev())
    dst = torch.ones(5, 9, 3, device=dev())
    L.call("cvhip_f32_unpad_add", src.data_ptr(), dst.data_ptr(), 5, 9, 16, 3, ops._stream())
    assert torch.allclose(dst, 1 + src[:5, :, :3])
