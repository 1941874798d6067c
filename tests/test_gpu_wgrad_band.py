"""GPU parity tests of the tap-resident 3x3 weight-gradient kernel (csrc/conv_wgrad_band.hip: nine-tap accumulator tiles, four pixel
replicas folded through the LDS, 256-pixel ranges staged once by LDS-DMA and addressed through a virtual tall image) through the C
ABI (cvhip_conv2d_wgrad), against fp32 CPU arithmetic on the same 16-bit-rounded operands — `aten::convolution_backward(weight)` of
the stride-1 3x3 "same" layers (trainer.py:189; modules/yolo_modules.py:95-104).

Tolerance as for the general weight-gradient kernel (test_gpu_kernels.py): fp32 outputs, relative L2 <= 1e-3 (summation order only).
The geometry list is chosen for the kernel's addressing: ranges that start mid-row, ranges that run across image boundaries (maps of
fewer than 256 pixels), odd widths (fragments wrap rows), dilation 2 / 3, splits with a partial last range, operands that are channel
slices of wider buffers, 32- and 64-wide output tiles, accumulate into a non-zero gradient."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import test_gpu_kernels as K
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

dev, rel_l2 = K.dev, K.rel_l2

WGB_CASES = [
    # N, C, H, W, K, dil
    (3, 128, 40, 40, 128, 1),    # the dominant YOLOv5-s shape (KT 64, eight tiles)
    (2, 64, 80, 80, 64, 1),
    (5, 256, 20, 20, 256, 1),    # 400-pixel maps: ranges cross image boundaries
    (64, 32, 8, 8, 32, 1),       # 64-pixel maps: a range spans four images (KT 32)
    (2, 128, 17, 19, 128, 1),    # odd sizes: 8-pixel groups wrap rows, last range partial
    (1, 32, 23, 37, 96, 1),      # KT 32 with three output-channel tiles
    (2, 64, 16, 16, 64, 2),      # dilation 2
    (3, 64, 13, 21, 96, 3),      # dilation 3, odd sizes, KT 32 x 3
    (2, 96, 12, 12, 64, 1),      # three input-channel tiles
    (1, 64, 24, 100, 64, 1),     # wide rows
    (7, 512, 16, 32, 512, 2),    # 128 tiles: every block owns the whole pixel range
    (40, 128, 40, 40, 128, 1),   # 250 ranges over 31 splits: uneven splits
]


def _took(desc):
    buf = (C.c_int32 * 8)()
    return L.load().cvhip_conv2d_wgrad_band_plan(C.byref(desc), buf), list(buf)


@pytest.mark.parametrize("case", WGB_CASES)
@pytest.mark.parametrize("form", ["plain", "slices_accumulate"])
def test_wgrad_band(case, form, monkeypatch):
    monkeypatch.setenv("CVHIP_WGRAD_BAND", "2")
    N, Cc, H, W, Kk, d = case
    g = torch.Generator().manual_seed(11)
    x = K.bf(torch.randn(N, Cc, H, W, generator=g))
    dy = K.bf(torch.randn(N, Kk, H, W, generator=g))
    w = torch.zeros(Kk, Cc, 3, 3, requires_grad=True)
    y = F.conv2d(x, w, None, stride=1, padding=d, dilation=d)
    (gw,) = torch.autograd.grad(y, w, dy)
    if form == "plain":
        xd = K.to_nhwc_dev(x)
        dyd = K.to_nhwc_dev(dy)
        x_ld, y_ld = Cc, Kk
        xp, dyp = xd.data_ptr(), dyd.data_ptr()
        base = None
        dw = torch.full((Kk, 3, 3, Cc), float("nan"), device=dev())
    else:
        # operands are channel slices [8, 8 + C) / [16, 16 + K) of wider NHWC buffers (concat slices: DESIGN.md §2); dw += ...
        xb = torch.randn(N, H, W, Cc + 24, device=dev()).to(K.BF)
        xb[..., 8:8 + Cc] = x.permute(0, 2, 3, 1).to(dev()).to(K.BF)
        dyb = torch.randn(N, H, W, Kk + 16, device=dev()).to(K.BF)
        dyb[..., 16:16 + Kk] = dy.permute(0, 2, 3, 1).to(dev()).to(K.BF)
        xd, dyd = xb, dyb
        x_ld, y_ld = Cc + 24, Kk + 16
        xp, dyp = xb.data_ptr() + 16, dyb.data_ptr() + 32
        base = torch.randn(Kk, 3, 3, Cc, generator=g)
        dw = base.to(dev()).clone()
    desc = ops.conv_desc(N, Cc, H, W, Kk, 3, 3, (1, 1), (d, d), (d, d), 1, x_ld, y_ld)
    took, plan = _took(desc)
    assert took == 1, ("the tap-resident kernel must take this geometry", case, plan)
    L.call("cvhip_conv2d_wgrad", C.byref(desc), xp, dyp, dw.data_ptr(), 0 if form == "plain" else 1, ops._stream())
    torch.cuda.synchronize()
    got = dw.permute(0, 3, 1, 2).cpu()
    ref = gw if base is None else gw + base.permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert rel_l2(got, ref) < 1e-3, (rel_l2(got, ref), plan)
    # every (k, tap, c) entry individually (a wrong tap offset or a transposed tile keeps the norm plausible on symmetric data)
    err = (got.double() - ref.double()).abs().amax(dim=(0, 1))
    scale = ref.double().abs().amax(dim=(0, 1)).clamp_min(1e-6)
    assert float((err / scale).max()) < 2e-2, (err / scale)


def test_wgrad_band_matches_general_kernel(monkeypatch):
    """the same launch with the kernel switched off (CVHIP_WGRAD_BAND=0 -> wgrad_kernel): equal to summation order"""
    N, Cc, H, W, Kk, d = 6, 128, 40, 40, 128, 1
    g = torch.Generator().manual_seed(3)
    xd = K.to_nhwc_dev(K.bf(torch.randn(N, Cc, H, W, generator=g)))
    dyd = K.to_nhwc_dev(K.bf(torch.randn(N, Kk, H, W, generator=g)))
    desc = ops.conv_desc(N, Cc, H, W, Kk, 3, 3, (1, 1), (d, d), (d, d), 1, Cc, Kk)
    outs = []
    for mode in ("0", "2"):
        monkeypatch.setenv("CVHIP_WGRAD_BAND", mode)
        dw = torch.full((Kk, 3, 3, Cc), float("nan"), device=dev())
        L.call("cvhip_conv2d_wgrad", C.byref(desc), xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), 0, ops._stream())
        torch.cuda.synchronize()
        outs.append(dw.cpu())
    assert rel_l2(outs[1], outs[0]) < 1e-5


def test_wgrad_band_refuses_what_it_cannot_run(monkeypatch):
    monkeypatch.setenv("CVHIP_WGRAD_BAND", "2")
    for (N, Cc, H, W, Kk, R, s, p, d) in [(2, 64, 20, 20, 64, 3, 2, 1, 1),    # stride 2
                                         (2, 64, 20, 20, 64, 1, 1, 0, 1),    # 1x1
                                         (2, 64, 20, 20, 64, 3, 1, 0, 1),    # not "same"
                                         (2, 48, 20, 20, 64, 3, 1, 1, 1),    # C % 32
                                         (2, 64, 20, 20, 40, 3, 1, 1, 1),    # K % 32
                                         (1, 64, 8, 8, 64, 3, 1, 1, 1)]:     # fewer pixels than one range
        desc = ops.conv_desc(N, Cc, H, W, Kk, R, R, (s, s), (p, p), (d, d), 1, Cc, Kk)
        assert _took(desc)[0] == 0
    monkeypatch.setenv("CVHIP_WGRAD_BAND", "0")
    desc = ops.conv_desc(3, 128, 40, 40, 128, 3, 3, (1, 1), (1, 1), (1, 1), 1, 128, 128)
    assert _took(desc)[0] == 0
