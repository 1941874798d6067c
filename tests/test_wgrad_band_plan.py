"""CPU-only checks of the tap-resident 3x3 weight-gradient kernel's HOST-SIDE plan and ADDRESSING (csrc/conv_wgrad_band.hip,
cvhip_conv2d_wgrad_band_plan). The plan is pure host arithmetic; the kernel's addressing — 256-pixel ranges in (image, row, column)
order, the virtual tall image (G = dilation zero rows between images), the patch loader's pixel -> (image, row, column) map, the
fragment gathers' output pixel -> patch pixel map and the tap offsets, the K-element order of a step — is restated here in numpy from
the formulas in the kernel's header and run on small problems against torch's convolution_backward(weight) (trainer.py:189 ->
aten::convolution_backward). The swizzles of the two LDS images are checked for bank-conflict freedom over every tap shift.
The device side of the same plans is tests/test_gpu_wgrad_band.py."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

KEYS = ("KF", "tiles", "splits", "ranges_per_split", "blocks", "lds", "PW", "nxp")
RANGE = 256


def plan(shape, dil=1, mode="2", stride=1, pad=None, R=3):
    N, Cc, H, W, K = shape
    pad = dil if pad is None else pad
    old = os.environ.get("CVHIP_WGRAD_BAND")
    os.environ["CVHIP_WGRAD_BAND"] = mode
    try:
        d = ops.conv_desc(N, Cc, H, W, K, R, R, (stride, stride), (pad, pad), (dil, dil), 1, Cc, K)
        buf = (C.c_int32 * 8)()
        r = L.load().cvhip_conv2d_wgrad_band_plan(C.byref(d), buf)
    finally:
        if old is None:
            os.environ.pop("CVHIP_WGRAD_BAND", None)
        else:
            os.environ["CVHIP_WGRAD_BAND"] = old
    assert r in (0, 1)
    return dict(zip(KEYS, buf)) if r else None


SHAPES = [
    (64, 128, 40, 40, 128), (64, 64, 80, 80, 64), (64, 32, 160, 160, 32), (64, 256, 20, 20, 256),   # YOLOv5-s, batch 64
    (64, 128, 80, 80, 128), (16, 256, 160, 160, 256),                                               # YOLOX-s head, YOLOv7-l
    (16, 256, 32, 64, 256), (16, 512, 16, 32, 512),                                                 # DeepLabv3+ R50, batch 16
    (3, 128, 17, 19, 128), (1, 32, 23, 37, 96), (2, 96, 12, 12, 64), (64, 32, 8, 8, 32),
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dil", [1, 2])
def test_plan_invariants(shape, dil):
    N, Cc, H, W, K = shape
    pl = plan(shape, dil)
    M = N * H * W
    if pl is None:
        # the only legitimate reason for these channel counts: the two [dY | patch] buffers (or the patch pieces) do not fit the CU
        pw = (W + 2 * dil + 7) // 8 * 8
        rows = (RANGE - 1 + W - 1) // W + 1
        ph = rows + (rows + H - 1) // H * dil + 2 * dil
        nxp = (ph * pw + 15) // 16
        assert nxp > 64 or 2 * (nxp * 1024 + RANGE * 32 * 2) > 159 * 1024, (shape, dil)   # not even the 32-channel tile's buffers fit
        return
    kt = 16 * pl["KF"]
    # 64-channel output tiles where K allows AND two [dY | patch] buffers fit; the 32-channel tile otherwise (wide rows: round 6)
    pw0 = (W + 2 * dil + 7) // 8 * 8
    rows0 = (RANGE - 1 + W - 1) // W + 1
    nxp0 = ((rows0 + (rows0 + H - 1) // H * dil + 2 * dil) * pw0 + 15) // 16
    fits64 = 2 * (nxp0 * 1024 + RANGE * 64 * 2) <= 159 * 1024
    assert pl["KF"] == (4 if (K % 64 == 0 and fits64) else 2)
    assert pl["tiles"] == (K // kt) * (Cc // 32)
    assert pl["blocks"] == pl["tiles"] * pl["splits"]
    total_ranges = -(-M // RANGE)
    assert pl["splits"] * pl["ranges_per_split"] >= total_ranges                 # the splits cover every pixel ...
    assert (pl["splits"] - 1) * pl["ranges_per_split"] < total_ranges            # ... and none of them is empty
    assert pl["splits"] <= max(1, 256 // pl["tiles"])                            # one resident block per CU (a single round) where the tiles allow
    assert pl["PW"] % 8 == 0 and pl["PW"] >= W + 2 * dil
    # the patch buffer holds the rows a range can reach: the output rows 256 consecutive pixels touch, dil zero rows per image boundary, dil above and below
    rows = (RANGE - 1 + W - 1) // W + 1
    ph = rows + (rows + H - 1) // H * dil + 2 * dil
    assert pl["nxp"] * 16 >= ph * pl["PW"] and pl["nxp"] <= 64
    need = 2 * (pl["nxp"] * 1024 + RANGE * kt * 2)
    fold = 4 * pl["KF"] * 9 * 64 * 16                                             # four parked accumulator tiles (two replicas x two halves)
    assert pl["lds"] >= need and pl["lds"] >= fold and pl["lds"] <= 159 * 1024


def test_default_policy_and_refusals():
    # default policy: the benchmark shapes whose buffers fit; small problems stay on the general kernel
    for shape in [(64, 128, 40, 40, 128), (64, 64, 80, 80, 64), (64, 256, 20, 20, 256), (64, 32, 160, 160, 32)]:
        assert plan(shape, 1, mode="1") is not None
    assert plan((2, 64, 20, 20, 64), 1, mode="1") is None and plan((2, 64, 20, 20, 64), 1, mode="2") is not None
    assert plan((64, 128, 40, 40, 128), 1, mode="0") is None
    assert plan((2, 64, 20, 20, 64), 1, stride=2) is None                       # stride 2
    assert plan((2, 64, 20, 20, 64), 1, R=1, pad=0) is None                      # 1x1
    assert plan((2, 64, 20, 20, 64), 1, pad=0) is None                           # not "same"
    assert plan((2, 48, 20, 20, 64), 1) is None and plan((2, 64, 20, 20, 40), 1) is None   # channel counts
    assert plan((1, 64, 8, 8, 64), 1) is None                                    # fewer pixels than one range
    # wide rows: the 32-channel-tile fallback is taken by default on the smaller problems only (profiles/r06_wgrad_band_wide.log)
    assert plan((16, 128, 64, 128, 128), 1, mode="1")["KF"] == 2 and plan((16, 64, 160, 160, 64), 1, mode="1")["KF"] == 2
    assert plan((16, 128, 160, 160, 128), 1, mode="1") is None and plan((16, 128, 160, 160, 128), 1, mode="2")["KF"] == 2


# ---- the kernel's addressing, restated -------------------------------------------------------------------------------------------------
def _interp_wgrad(x, dy, dil, pl):
    """x [N][H][W][C], dy [N][H][W][K] (float64 numpy) -> dW [K][3][3][C] the way the kernel walks the problem: splits of
    `ranges_per_split` ranges; per range the patch of the virtual tall image; per output pixel one patch pixel + nine tap offsets."""
    N, H, W, Cc = x.shape
    K = dy.shape[3]
    G, VP, PW = dil, H + dil, pl["PW"]
    M = N * H * W
    dyf = dy.reshape(M, K)
    dw = np.zeros((K, 3, 3, Cc))
    for split in range(pl["splits"]):
        m_begin = split * pl["ranges_per_split"] * RANGE
        m_end = min(M, m_begin + pl["ranges_per_split"] * RANGE)
        for q0 in range(m_begin, m_end, RANGE):
            qlast = min(q0 + RANGE, m_end) - 1
            gr0, grl = q0 // W, qlast // W
            vbase = gr0 + (gr0 // H) * G
            phr = (grl + (grl // H) * G) - vbase + 1 + 2 * dil
            assert phr * PW <= pl["nxp"] * 16, "patch rows of a range exceed the buffer"
            # patch loader: patch pixel (pr, pc) <- virtual row v0 + pr, input column pc - dil; zero page outside
            v0 = vbase - dil
            patch = np.zeros((phr, PW, Cc))
            for pr in range(phr):
                v = v0 + pr
                if v < 0:
                    continue
                n, ih = divmod(v, VP)
                if ih >= H or n >= N:
                    continue                      # a gap row of the virtual tall image (or past the batch)
                patch[pr, dil:dil + W] = x[n, ih]
            # fragment gathers: output pixel m -> patch pixel (tap (0, 0)), taps at (i * dil rows, j * dil columns)
            for m in range(q0, qlast + 1):
                gr, ow = divmod(m, W)
                vrel = gr + (gr // H) * G - vbase
                for i in range(3):
                    for j in range(3):
                        dw[:, i, j, :] += np.outer(dyf[m], patch[vrel + i * dil, ow + j * dil])
    return dw


@pytest.mark.parametrize("case", [
    (2, 32, 11, 13, 32, 1),     # odd sizes: ranges start mid-row, the second image begins inside a range
    (9, 32, 6, 5, 64, 1),       # 30-pixel maps: a range spans nine images (eight shared zero rows)
    (3, 32, 10, 12, 32, 2),     # dilation 2: two zero rows between images, taps two pixels apart
    (1, 64, 20, 24, 32, 3),     # dilation 3
])
def test_addressing_reproduces_conv_wgrad(case):
    N, Cc, H, W, K, dil = case
    pl = plan((N, Cc, H, W, K), dil)
    assert pl is not None
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cc, H, W, generator=g, dtype=torch.float64)
    dy = torch.randn(N, K, H, W, generator=g, dtype=torch.float64)
    w = torch.zeros(K, Cc, 3, 3, dtype=torch.float64, requires_grad=True)
    (gw,) = torch.autograd.grad(F.conv2d(x, w, None, 1, dil, dil), w, dy)
    got = _interp_wgrad(x.permute(0, 2, 3, 1).numpy(), dy.permute(0, 2, 3, 1).numpy(), dil, pl)
    np.testing.assert_allclose(got, gw.permute(0, 2, 3, 1).numpy(), rtol=1e-9, atol=1e-9)


def test_step_element_order_is_a_permutation_and_conflict_free():
    """A step's 32 pixels are read as pl = 16h + 4g + q4 (h: instruction, g: 16-lane group, q4: row of the 4 x 4 transpose block): a
    permutation of 0..31 shared by both operands; lanes 0..31 of one ds_read_b64_tr_b16 (g = 0, 1) touch 8 consecutive pixels whose
    32-byte pieces must fall into 8 different 32-byte bank groups (256 bytes of banks) — dY rows of 128 / 64 bytes with their segment
    swizzle, patch pixels of 64 bytes with theirs, at every tap shift and patch alignment."""
    pls = sorted(16 * h + 4 * g + q for h in range(2) for g in range(4) for q in range(4))
    assert pls == list(range(32))
    for h in range(2):
        for half in range(2):                        # lanes 0..31 / 32..63 are serviced separately
            pix = [16 * h + 4 * g + q for g in (2 * half, 2 * half + 1) for q in range(4)]
            assert pix == list(range(pix[0], pix[0] + 8))
            for kf in range(4):                      # dY, KT = 64: 128-byte rows, segment kf ^ ((pl >> 1) & 3)
                groups = {(pl * 128 + ((kf ^ ((pl >> 1) & 3)) << 5)) // 32 % 8 for pl in pix}
                assert len(groups) == 8
            for kf in range(2):                      # dY, KT = 32: 64-byte rows, segment kf ^ ((pl >> 2) & 1)
                groups = {(pl * 64 + ((kf ^ ((pl >> 2) & 1)) << 5)) // 32 % 8 for pl in pix}
                assert len(groups) == 8
    for base in range(0, 64):                        # patch: any alignment of the 8 consecutive pixels (tap shifts, row offsets % 8 == 0)
        for chh in range(2):
            groups = {(pp * 64 + ((chh ^ ((pp >> 2) & 1)) << 5)) // 32 % 8 for pp in range(base, base + 8)}
            assert len(groups) == 8


def test_loader_swizzle_matches_reader():
    """LDS-DMA writes lane-linearly: lane l of a patch instruction fills physical 16-byte slot l & 3 of pixel l >> 2 and must FETCH the
    logical slot (l & 3) ^ (((pp >> 2) & 1) << 1); the reader looks for logical half `ch` at physical half ch ^ ((pp >> 2) & 1). Same
    for the dY image (KT 64: 8 rows x 8 slots per instruction; KT 32: 16 rows x 4 slots)."""
    for piece in range(4):
        for lane in range(64):
            pp, phys = piece * 16 + (lane >> 2), lane & 3
            logical = phys ^ (((lane >> 4) & 1) << 1)                 # what the kernel computes from the lane alone
            assert logical == phys ^ (((pp >> 2) & 1) << 1)
            assert (logical >> 1) ^ ((pp >> 2) & 1) == phys >> 1      # reader: physical half of logical half
            pl, slot = piece * 8 + (lane >> 3), lane & 7              # dY, KT 64
            kf = (slot >> 1) ^ ((lane >> 4) & 3)
            assert (lane >> 4) & 3 == (pl >> 1) & 3 and kf ^ ((pl >> 1) & 3) == slot >> 1
            pl, slot = piece * 16 + (lane >> 2), lane & 3             # dY, KT 32
            kf = (slot >> 1) ^ ((lane >> 4) & 1)
            assert (lane >> 4) & 1 == (pl >> 2) & 1 and kf ^ ((pl >> 2) & 1) == slot >> 1
