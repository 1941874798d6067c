"""CPU-only check of the row-band convolution kernel's HOST-SIDE plan (csrc/conv_band.hip, cvhip_conv2d_band_plan): band heights, wave
layout, LDS budget and the default policy are pure host arithmetic — every invariant the device code relies on is checked here without a
GPU, for each of the kernel's forms (narrow / wide waves, with and without the LDS read-ahead). The device side of the same plans is
tests/test_gpu_band.py. Replaces aten::convolution's algorithm choice for the stride-1 3x3 layers (conv_module.py:209)."""
import ctypes as C
import os

import pytest

from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

KEYS = ("NF", "WN", "MFW", "PPS", "PF", "TH", "bands", "n_tiles", "total", "lds", "PH", "PW", "NW")
ENV = ("CVHIP_BAND", "CVHIP_BAND_NW")


def plan(shape, dgrad=False, pad=1, dil=1, **env):
    N, Cc, H, W, K = shape
    for k in ENV:
        os.environ.pop(k, None)
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        d = ops.conv_desc(N, Cc, H, W, K, 3, 3, (1, 1), (pad, pad), (dil, dil), 1, Cc, K)
        buf = (C.c_int32 * L.BAND_PLAN_INTS)()
        r = L.load().cvhip_conv2d_band_plan(C.byref(d), 1 if dgrad else 0, buf)
    finally:
        for k in ENV:
            os.environ.pop(k, None)
    assert r in (0, 1)
    return dict(zip(KEYS, buf)) if r else None


SHAPES = [
    (64, 128, 40, 40, 128), (64, 64, 80, 80, 64), (64, 32, 160, 160, 32), (64, 256, 20, 20, 256),   # YOLOv5-s, batch 64
    (16, 64, 128, 256, 64), (16, 128, 64, 128, 128), (16, 256, 32, 64, 256), (16, 512, 32, 64, 512),  # DeepLabv3+ R50, batch 16
    (3, 128, 17, 19, 128), (1, 32, 23, 37, 32), (2, 64, 9, 300, 64), (2, 96, 12, 12, 64), (2, 64, 14, 18, 128),
]
FORMS = [dict(CVHIP_BAND_NW=8), dict(CVHIP_BAND_NW=4)]   # (the wide-wave / read-ahead forms of round 5 are no longer instantiated)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("geom", [(1, 1), (2, 2), (0, 1)])
def test_band_plan_invariants(shape, form, geom):
    pad, dil = geom
    N, Cc, H, W, K = shape
    OH, OW = H + 2 * pad - 2 * dil, W + 2 * pad - 2 * dil
    fOH, fOW, fK, fC = OH, OW, K, Cc
    for dgrad in (False, True):
        # dgrad is the same GEMM with the roles swapped: it writes the H x W input gradient (Cc channels) from the K-channel output gradient
        OH, OW, K, Cc = (H, W, shape[1], shape[4]) if dgrad else (fOH, fOW, fK, fC)
        pl = plan(shape, dgrad, pad, dil, CVHIP_BAND=2, **form)
        if pl is None:
            # the forced form does not fit: wide waves need a 64- or 128-channel tile; rows wider than a block's fragment slots
            # the kernel's channel shapes: output 32, 64 or a multiple of 128, input a multiple of 32
            # ... and even a one-row band must fit the LDS (two patch buffers when there is more than one 32-channel chunk)
            pw1 = (OW + 2 * dil + 7) // 8 * 8
            pieces1 = -(-(1 + 2 * dil) * pw1 * 4 // 512)
            assert not (K in (32, 64) or K % 128 == 0) or Cc % 32 or OW > 16 * 13 * (8 // max(1, min(K, 128) // 32)) or \
                pieces1 > 12 or (2 if Cc > 32 else 1) * pieces1 * 8192 + 8192 > 156 * 1024, (shape, dgrad)
            continue
        assert pl["NF"] == 2   # narrow waves (the only instantiated form)
        BN = min(K, 128)
        assert pl["WN"] * 16 * pl["NF"] == BN and pl["n_tiles"] * BN == K
        assert pl["NW"] in (4, 8) and (pl["NW"] == 8 or form["CVHIP_BAND_NW"] == 4) and pl["WN"] <= pl["NW"]   # (NW = 4 falls back to 8)
        WM = pl["NW"] // pl["WN"]
        assert pl["MFW"] in (7, 10, 13) and (pl["NF"] == 2 or pl["MFW"] == 7)
        # every output pixel of a band has a fragment slot; the bands cover the image; one block per (image, band, channel tile)
        assert pl["MFW"] * WM * 16 >= pl["TH"] * OW
        assert pl["bands"] == -(-OH // pl["TH"]) and pl["total"] == N * pl["bands"] * pl["n_tiles"]
        # even bands: no band is more than one row higher than another would have to be
        assert pl["TH"] == -(-OH // pl["bands"])
        # the patch: TH output rows + the taps' halo, row pitch a multiple of 8 pixels that holds a row and its halo
        assert pl["PH"] == pl["TH"] + 2 * dil and pl["PW"] % 8 == 0 and pl["PW"] >= OW + 2 * dil
        # the DMA pieces (PPS per wave and K step, six steps, 16 pixels each, NW waves) cover it, in one or two buffers + the dummy slots;
        # two co-resident 4-wave blocks share the CU's 160 KB
        pieces = -(-pl["PH"] * pl["PW"] * 4 // (64 * pl["NW"]))
        assert pieces <= pl["PPS"] * 6 and (pl["PPS"] == 1) == (pieces <= 6)
        bufs = 2 if Cc > 32 else 1
        assert pl["lds"] >= bufs * -(-pl["PH"] * pl["PW"] // 16) * 1024 + 1024 and pl["lds"] <= (79 if pl["NW"] == 4 else 156) * 1024
        assert pl["PF"] == 0   # (no LDS read-ahead form in the library)


def test_band_default_policy():
    """the default policy (conv_band.hip band_plan_nw; profiles/r05_band_image_bench.log) takes the YOLOv5-s shapes the kernel measured
    faster on — whole rounds of block slots, nearly full waves, two co-resident 4-wave blocks per CU where that plan qualifies — and leaves
    DeepLabv3+'s (batch 16: idle slots) to the patch-resident / per-tap kernels"""
    want = {(64, 128, 40, 40, 128): (4, 5, 512), (64, 64, 80, 80, 64): (4, 5, 1024), (64, 32, 160, 160, 32): (4, 2, 5120),
            (64, 256, 20, 20, 256): (4, 5, 512)}
    for shape, (nw, th, total) in want.items():
        for dgrad in (False, True):
            pl = plan(shape, dgrad)
            assert pl is not None and pl["NF"] == 2 and pl["PF"] == 0 and (pl["NW"], pl["TH"], pl["total"]) == (nw, th, total), (shape, pl)
            assert pl["total"] % (512 if nw == 4 else 256) == 0
    for shape in ((16, 128, 64, 128, 128), (16, 64, 128, 256, 64)):
        assert plan(shape) is None, shape        # shallow reductions with idle fragment slots: the patch-resident / per-tap kernels stay faster
    # round 6: DEEP reductions (>= 8 chunks of 32 channels) are taken with emptier rounds / waves too (profiles/r06_band_deep_policy.log)
    for shape, dil in (((16, 256, 32, 64, 256), 1), ((16, 512, 16, 32, 512), 1), ((16, 2560, 16, 32, 512), 1), ((16, 512, 32, 64, 512), 2)):
        for dgrad in (False, True):
            pl = plan(shape, dgrad, dil, dil)
            assert pl is not None and pl["NF"] == 2, (shape, dgrad)
    assert plan((64, 128, 40, 40, 128), CVHIP_BAND_NW=8)["NW"] == 8
    assert plan((64, 128, 40, 40, 128), CVHIP_BAND=0) is None
    # stride 2 (round 6, second session): the FORWARD plan of a 3x3 / padding 1 / dilation 1 layer is taken — rows of two column planes,
    # two patch rows per output row; its input gradient (stride-parity classes), other paddings, 1x1 and grouped convolutions never are
    d = ops.conv_desc(64, 256, 40, 40, 512, 3, 3, (2, 2), (1, 1), (1, 1), 1, 256, 512)
    buf = (C.c_int32 * L.BAND_PLAN_INTS)()
    assert L.load().cvhip_conv2d_band_plan(C.byref(d), 0, buf) == 1
    th, ph, pw = buf[5], buf[10], buf[11]
    assert ph == (th - 1) * 2 + 3 and pw == 24 + 24      # odd columns 2k - 1 (k = 0 .. 20, padded to 24) + even columns (20 -> 24)
    assert L.load().cvhip_conv2d_band_plan(C.byref(d), 1, None) == 0
    d = ops.conv_desc(64, 256, 40, 40, 512, 3, 3, (2, 2), (0, 0), (1, 1), 1, 256, 512)
    assert L.load().cvhip_conv2d_band_plan(C.byref(d), 0, None) == 0
    # wide maps stay on the per-tap kernel by default (the stride-2 band is LDS-DMA-bound there: profiles/r06_band_s2_bench.log)
    d = ops.conv_desc(64, 64, 160, 160, 128, 3, 3, (2, 2), (1, 1), (1, 1), 1, 64, 128)
    assert L.load().cvhip_conv2d_band_plan(C.byref(d), 0, None) == 0
    d = ops.conv_desc(64, 128, 40, 40, 128, 1, 1, (1, 1), (0, 0), (1, 1), 1, 128, 128)
    assert L.load().cvhip_conv2d_band_plan(C.byref(d), 0, None) == 0


def test_weight_image_sizes():
    """cvhip_conv2d_weight_image_elems (pure host arithmetic): the fragment-ordered copy exists behind an image exactly when the layer is a
    stride-1 3x3 convolution without channel padding whose GEMM has 32 | 64 | 128k output rows and a reduction width that is a multiple
    of 32 — fprop: rows K, reduction C; dgrad: rows C, reduction K (csrc/conv_plan.h band_image_fprop / band_image_dgrad)"""
    lib = L.load()

    def sizes(Cc, K, R=3, stride=1, k_valid=0, c_valid=0, dil=1):
        d = ops.conv_desc(2, Cc, 24, 24, K, R, R, (stride, stride), (R // 2 * dil, R // 2 * dil), (dil, dil), 1, Cc, K, k_valid, c_valid)
        n = K * R * R * Cc
        f, g = lib.cvhip_conv2d_weight_image_elems(C.byref(d), 0), lib.cvhip_conv2d_weight_image_elems(C.byref(d), 1)
        assert g - (lib.cvhip_conv2d_dgrad_weight_elems(C.byref(d)) - 0) in (0, n)
        return f // n, g // n if stride == 1 else None

    assert sizes(128, 128) == (2, 2) and sizes(64, 64) == (2, 2) and sizes(32, 32) == (2, 2) and sizes(256, 256, dil=2) == (2, 2)
    assert sizes(64, 128) == (2, 2) and sizes(128, 64) == (2, 2)
    assert sizes(96, 64) == (2, 1)          # dgrad would write 96 rows: not a tile shape of the kernel
    assert sizes(64, 96) == (1, 2)
    assert sizes(8, 32) == (1, 1)           # image stem: 8 input channels
    assert sizes(128, 128, R=1) == (1, 1)   # 1x1
    assert sizes(128, 128, stride=2)[0] == 2        # stride-2 3x3 / padding 1: the forward image has the fragment-ordered copy (conv_band.hip s2)
    assert sizes(96, 64, stride=2)[0] == 2 and sizes(64, 96, stride=2)[0] == 1
    assert sizes(128, 256, k_valid=255) == (1, 1)   # padded channels (the detect layers' 255 outputs)
    d = ops.conv_desc(2, 64, 24, 24, 64, 3, 3, (1, 1), (1, 1), (1, 1), 1, 64, 64)
    assert lib.cvhip_conv2d_weight_image_elems(C.byref(d), 2) == L.ERR_INVALID


def _interpret_band_fprop(x, w, pad, dil, pl):
    """numpy interpreter of the band kernel's ADDRESSING (csrc/conv_band.hip header): the DMA loader's lane -> (patch pixel, slot) map with the
    source-side swizzle, the fragment reads' byte addresses (row pitch PW, tap shifts, bit 5 ^= bit 8), the band image's fragment order
    (csrc/conv_plan.h) and the MFMA operand layout (lane = row + 16 * k-group) — executed for one 32-channel chunk at a time exactly as
    the plan lays the work out. x: [N][H][W][C] float, w: [K][3][3][C] float. Returns y [N][OH][OW][K]."""
    import numpy as np
    N, H, W, Cc = x.shape
    K = w.shape[0]
    OH, OW = H + 2 * pad - 2 * dil, W + 2 * pad - 2 * dil
    NW, WN, MFW, TH, PH, PW = pl["NW"], pl["WN"], pl["MFW"], pl["TH"], pl["PH"], pl["PW"]
    WM, NF, BN, NC = NW // WN, pl["NF"], min(K, 128), Cc // 32
    lo = -pad
    # band image of the row-major weights Wt[n][tap*C + c]
    wt = w.reshape(K, 9 * Cc)
    nvec = K * 9 * Cc // 8
    img = np.zeros((nvec, 8), dtype=np.float64)
    for v in range(nvec):
        lane, fr = v & 63, v >> 6
        f, st = fr % (K // 16), fr // (K // 16)
        tap, cc = st // NC, st % NC
        n, c0 = f * 16 + (lane & 15), cc * 32 + (lane >> 4) * 8
        img[v] = wt[n, tap * Cc + c0: tap * Cc + c0 + 8]
    y = np.zeros((N, OH, OW, K))
    npieces = -(-PH * PW * 4 // (64 * NW))
    buf_kb = -(-PH * PW // 16)
    for n_img in range(N):
        for band in range(pl["bands"]):
            oh0 = band * TH
            rows = min(TH, OH - oh0)
            npx = rows * OW
            for ntile in range(pl["n_tiles"]):
                n0 = ntile * BN
                acc = {}
                for c in range(NC):
                    lds = np.full((buf_kb * 1024 // 16, 8), np.nan)          # 16-byte slots of one patch buffer
                    for j in range(npieces):                                   # ---- the loader
                        for wave in range(NW):
                            if ((j * NW + wave) << 4) >= PH * PW:
                                continue                                        # a KB wholly past the patch: the dummy slot
                            for lane in range(64):
                                pp = ((j * NW + wave) << 4) + (lane >> 2)
                                lsl = (lane & 3) ^ (((lane >> 4) & 1) << 1)
                                pr, pc = divmod(pp, PW)
                                ih, iw = oh0 + lo + pr, lo + pc
                                ok = pp < PH * PW and 0 <= ih < H and 0 <= iw < W
                                val = x[n_img, ih, iw, c * 32 + lsl * 8: c * 32 + lsl * 8 + 8] if ok else np.zeros(8)
                                lds[(j * NW + wave) * 64 + lane] = val
                    for wave in range(NW):                                     # ---- the waves
                        wn, wm = wave % WN, wave // WN
                        for b in range(MFW):
                            for ti in range(3):
                                for tj in range(3):
                                    tap = ti * 3 + tj
                                    bfrag = np.zeros((64, 8))                  # pixel fragment: lane = pixel + 16 * k-group
                                    for lane in range(64):
                                        q = min(((wm * MFW + b) << 4) + (lane & 15), max(npx - 1, 0))
                                        r, cc = divmod(q, OW)
                                        ab = ((r * PW + cc) << 6) + ((lane >> 4) << 4)
                                        u = ab + (tj * dil) * 64
                                        aj = u ^ ((u >> 3) & 32)
                                        addr = aj + (ti * dil) * PW * 64
                                        assert addr % 16 == 0 and addr // 16 < lds.shape[0]
                                        bfrag[lane] = lds[addr // 16]
                                    for a in range(NF):
                                        fi = (n0 >> 4) + wn * NF + a
                                        afrag = img[((tap * NC + c) * (K // 16) + fi) * 64: ((tap * NC + c) * (K // 16) + fi) * 64 + 64]
                                        d = acc.setdefault((wave, a, b), np.zeros((16, 16)))
                                        for g in range(4):                     # D[n][pix] += sum_g A[n + 16g] . B[pix + 16g]
                                            d += afrag[16 * g: 16 * g + 16] @ bfrag[16 * g: 16 * g + 16].T
                for (wave, a, b), d in acc.items():                            # ---- the epilogue
                    wn, wm = wave % WN, wave // WN
                    for pix in range(16):
                        q = ((wm * MFW + b) << 4) + pix
                        if q < npx:
                            r, cc = divmod(q, OW)
                            y[n_img, oh0 + r, cc, n0 + wn * 16 * NF + a * 16: n0 + wn * 16 * NF + a * 16 + 16] = d[:, pix]
    return y


@pytest.mark.parametrize("case", [(1, 32, 7, 9, 32, 1, 1, {}), (1, 64, 6, 11, 64, 1, 1, {"CVHIP_BAND_NW": 8}), (1, 32, 9, 6, 64, 2, 2, {}),
                                  (1, 32, 5, 20, 64, 0, 1, {"CVHIP_BAND_NW": 8}), (1, 64, 8, 12, 256, 1, 1, {})])
def test_band_addressing_interpreter(case):
    """the addressing the kernel's header and conv_plan.h state (loader swizzle, fragment addresses, band image order, MFMA operand
    layout), run in numpy on the plan the library returns, reproduces torch's convolution"""
    import numpy as np
    import torch
    import torch.nn.functional as F
    N, Cc, H, W, K, pad, dil, env = case
    pl = plan((N, Cc, H, W, K), False, pad, dil, CVHIP_BAND=2, **env)
    assert pl is not None, case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(N, Cc, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(K, Cc, 3, 3, generator=g, dtype=torch.float64)
    ref = F.conv2d(x, w, None, 1, pad, dil).permute(0, 2, 3, 1).numpy()
    got = _interpret_band_fprop(x.permute(0, 2, 3, 1).numpy(), w.permute(0, 2, 3, 1).contiguous().numpy(), pad, dil, pl)
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-9), float(np.abs(got - ref).max())
