"""GPU parity tests of cvhip_conv2d_dgrad_add on the per-tap implicit GEMM (csrc/conv_igemm.hip) — the gradient arriving over a skip
connection folded into the input-gradient kernel's epilogue (yolo_modules.py:102, torchvision Bottleneck; replaces autograd's
accumulation add reached from trainer.py:189).

Round 6: the addend tile goes through the LDS output tile (coalesced 16-byte row reads) instead of being read in the MFMA fragment
layout. Same arithmetic (fp32 accumulator + addend, ONE rounding to 16 bits), so the two forms must be BIT-identical; both are held to
fp32 CPU arithmetic on the same 16-bit-rounded operands: max |err| <= 2^-7 max|ref|, relative L2 <= 4e-3.

CVHIP_PATCH=0 / CVHIP_BAND=0 (read per launch) keep the multi-tap cases on the per-tap kernel."""
import ctypes as C

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


CASES = [
    # N, C, H, W, K, R, S, stride, pad, dil
    (2, 32, 32, 32, 64, 3, 3, 2, 1, 1),       # 256 x 32 tiles, 2-deep ring
    (3, 64, 21, 23, 64, 3, 3, 2, 1, 1),       # odd sizes: parity classes of unequal extent (no interleaved tile order), 256 x 64 tiles
    (5, 128, 40, 40, 256, 3, 3, 2, 1, 1),     # 128-wide tiles, ragged last tile
    (26, 128, 80, 80, 256, 3, 3, 2, 1, 1),    # 256 x 128 tiles (YOLOv5-s stage convolution)
    (3, 256, 20, 20, 512, 3, 3, 2, 1, 1),     # two channel tiles
    (2, 256, 32, 64, 512, 1, 1, 2, 0, 1),     # 1x1 stride 2: three of the four classes have no tap (addend only)
    (2, 1024, 16, 32, 256, 1, 1, 1, 0, 1),    # deep 1x1 (ResNet conv1 of a bottleneck)
    (2, 512, 20, 20, 512, 1, 1, 1, 0, 1),
    (1, 136, 9, 11, 64, 3, 3, 1, 1, 1),       # C % 8 == 0 but not a multiple of the tile: ragged channel tile
    (2, 64, 12, 12, 72, 5, 5, 1, 2, 1),
]


@pytest.fixture(autouse=True)
def per_tap(monkeypatch):
    monkeypatch.setenv("CVHIP_PATCH", "0")
    monkeypatch.setenv("CVHIP_BAND", "0")
    yield


def _run(case, monkeypatch, res_lds, add_ld_extra=0):
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
    if add_ld_extra:   # the addend as a channel slice of a wider buffer (pitch != C)
        wide = torch.zeros((N, H, W, Cc + add_ld_extra), dtype=K.BF, device=dev())
        wide[..., add_ld_extra // 2:add_ld_extra // 2 + Cc] = add.permute(0, 2, 3, 1).to(K.BF).to(dev())
        addd = wide[..., add_ld_extra // 2:add_ld_extra // 2 + Cc]
        add_ld = Cc + add_ld_extra
    else:
        addd = to_nhwc_dev(add)
        add_ld = Cc
    dx = ops.empty_nhwc(N, Cc, H, W, dev())
    dx.fill_(float("nan"))
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp)
    monkeypatch.setenv("CVHIP_IGEMM_RES_LDS", "1" if res_lds else "0")
    L.call("cvhip_conv2d_dgrad_add", C.byref(desc), dyd.data_ptr(), st.w_dgrad.data_ptr(), addd.data_ptr(), add_ld, dx.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    return dx, gx


@pytest.mark.parametrize("case", CASES)
def test_dgrad_add_lds_addend(case, monkeypatch):
    got_lds, ref = _run(case, monkeypatch, True)
    got_dir, _ = _run(case, monkeypatch, False)
    a = got_lds.float().cpu()
    assert torch.isfinite(a).all()
    assert max_rel(a, ref) < 2 ** -7 and rel_l2(a, ref) < 4e-3
    assert torch.equal(got_lds.view(torch.int16), got_dir.view(torch.int16)), "LDS-staged addend differs from the direct read"


@pytest.mark.parametrize("case,extra", [(CASES[2], 8), (CASES[2], 4), (CASES[6], 16)])
def test_dgrad_add_addend_slice(case, extra, monkeypatch):
    """addend read from a channel slice of a wider buffer: pitch % 8 == 0 with a 16-byte-aligned base takes the LDS form, anything else
    the direct read — same values either way"""
    got, ref = _run(case, monkeypatch, True, add_ld_extra=extra)
    a = got.float().cpu()
    assert torch.isfinite(a).all()
    assert max_rel(a, ref) < 2 ** -7 and rel_l2(a, ref) < 4e-3


# ---- single-tap plans whose channel count is not a multiple of the 32-deep reduction step (round 6: FAST staging with a partial last
# step; before, the 560 -> 512 1x1 of DeepLabv3+'s decoder — deeplabv3plus_head.py:63-68 — ran the general per-row decode path) -----------
ONE_TAP_RAGGED = [
    (2, 560, 16, 24, 512, 1, 1, 1, 0, 1),     # the decoder's 1x1 (17 full steps + 16 channels)
    (2, 304, 16, 16, 256, 1, 1, 1, 0, 1),     # torchvision-style decoder concat (256 + 48)
    (2, 72, 15, 17, 136, 1, 1, 2, 0, 1),      # stride 2 keeps small 1x1 problems off the streaming kernel
    (1, 24, 9, 9, 40, 1, 1, 2, 0, 1),         # fewer channels than one step
    (3, 40, 12, 12, 64, 1, 1, 2, 0, 1),
]


@pytest.mark.parametrize("case", ONE_TAP_RAGGED)
def test_one_tap_ragged_channels_fprop(case):
    K.test_conv_fprop(case)


@pytest.mark.parametrize("case", ONE_TAP_RAGGED)
def test_one_tap_ragged_channels_dgrad(case):
    K.test_conv_dgrad(case)


@pytest.mark.parametrize("case", [c for c in ONE_TAP_RAGGED if c[4] % 8 == 0])
def test_one_tap_ragged_channels_bn_stats(case):
    K.test_conv_fprop_bn_stats(case)
