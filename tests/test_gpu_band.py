"""GPU parity tests of the row-band 3x3 convolution kernel (csrc/conv_band.hip: band-resident input patch, weight fragments loaded
straight into registers, one barrier per 32-channel chunk) through the C ABI, against fp32 CPU arithmetic on the same 16-bit-rounded
operands — `aten::convolution` / `convolution_backward(input)` of the stride-1 3x3 layers (conv_module.py:209, trainer.py:189).

Tolerances as for the other convolution kernels (test_gpu_kernels.py): outputs stored in 16 bits: max |err| <= 2^-7 max|ref|, relative
L2 <= 4e-3; fp32 BatchNorm sums: relative L2 <= 1e-3.

CVHIP_BAND=2 (read per launch) makes the launcher pick the band kernel for every geometry it can run."""
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


# the kernel's forms (conv_band.hip header): narrow waves (32 channels x <= 13 pixel fragments) as one 8-wave block per CU or as two
# co-resident 4-wave blocks. (The wide-wave and LDS-read-ahead forms of round 5 — measured, never ahead — left the library in round 6.)
FORMS = {"narrow": ("2", "0", "8"), "narrow_nw4": ("2", "0", "4")}   # (weight fragments per wave, read-ahead, waves per block)


@pytest.fixture(autouse=True, params=sorted(FORMS))
def force_band(request, monkeypatch):
    nf, pf, nw = FORMS[request.param]
    monkeypatch.setenv("CVHIP_BAND_NW", nw)
    monkeypatch.setenv("CVHIP_BAND", "2")
    yield request.param


def _skip_unless_form_runs(case, form, dgrad):
    """a form that does not fit the geometry falls through to the other convolution kernels (tested elsewhere): the case would pass
    without running the kernel under test"""
    N, Cc, H, W, Kk, R, S, s, p, d = case
    Kp = Kk
    desc = ops.conv_desc(N, Cc, H, W, Kp, R, S, (s, s), (p, p), (d, d), 1, Cc, Kp)
    buf = (C.c_int32 * L.BAND_PLAN_INTS)()
    took = L.load().cvhip_conv2d_band_plan(C.byref(desc), 1 if dgrad else 0, buf)
    nf, pf, nw = FORMS[form]
    if took != 1 or buf[0] != int(nf) or (pf == "1") != (buf[4] > 0) or buf[12] != int(nw):
        pytest.skip("form %s does not run %s (plan %s)" % (form, case, list(buf)))


BAND_CASES = [
    # N, C, H, W, K, R, S, stride, pad, dil
    (3, 128, 40, 40, 128, 3, 3, 1, 1, 1),    # the dominant YOLOv5-s shape: 10-row bands, 4 x 2 waves x 13 fragments
    (2, 64, 80, 80, 64, 3, 3, 1, 1, 1),      # 2 x 4 waves
    (2, 32, 160, 160, 32, 3, 3, 1, 1, 1),    # 1 x 8 waves, one chunk (single patch buffer)
    (3, 256, 20, 20, 256, 3, 3, 1, 1, 1),    # two 128-wide channel tiles, 8 chunks
    (2, 128, 17, 19, 128, 3, 3, 1, 1, 1),    # ragged: last band shorter, fragments wrap image rows at odd widths
    (1, 32, 23, 37, 32, 3, 3, 1, 1, 1),
    (2, 64, 16, 16, 64, 3, 3, 1, 2, 2),      # dilation 2
    (2, 64, 14, 18, 128, 3, 3, 1, 0, 1),     # no padding: output smaller than the input
    (2, 64, 9, 300, 64, 3, 3, 1, 1, 1),      # wide rows: one-row bands
    (2, 96, 12, 12, 64, 3, 3, 1, 1, 1),      # three chunks
    (1, 128, 64, 128, 128, 3, 3, 1, 1, 1),   # DeepLabv3+ decoder shape
    (2, 512, 16, 32, 512, 3, 3, 1, 2, 2),    # ASPP-like dilated, four channel tiles, 16 chunks
    # tiny maps (round 6: the small-problem policy sends them here): whole images shorter than one fragment row, 1 - 2 pixel rows
    (4, 128, 4, 4, 128, 3, 3, 1, 1, 1),
    (4, 256, 8, 8, 256, 3, 3, 1, 1, 1),
    (3, 512, 4, 4, 512, 3, 3, 1, 1, 1),
    (2, 64, 2, 2, 64, 3, 3, 1, 1, 1),
    (5, 32, 1, 1, 32, 3, 3, 1, 1, 1),
    (4, 128, 16, 16, 128, 3, 3, 1, 1, 1),
    (2, 64, 3, 5, 32, 3, 3, 1, 1, 1),
]


# stride 2 (round 6, second session): 3x3 / padding 1 forward plans — two patch rows per output row, rows stored as two column planes
BAND_S2_CASES = [
    (3, 64, 160, 160, 128, 3, 3, 2, 1, 1),   # YOLOv5-s stage convolution: one-row bands of 80 pixels
    (3, 128, 80, 80, 256, 3, 3, 2, 1, 1),    # two channel tiles
    (2, 256, 40, 40, 512, 3, 3, 2, 1, 1),    # four channel tiles, 8 chunks
    (2, 32, 64, 64, 64, 3, 3, 2, 1, 1),      # one chunk: single patch buffer
    (2, 64, 21, 23, 64, 3, 3, 2, 1, 1),      # odd sizes: the last even / odd column and the last row fall outside the image
    (3, 128, 17, 38, 128, 3, 3, 2, 1, 1),
    (1, 96, 9, 9, 32, 3, 3, 2, 1, 1),        # three chunks, tiny map
    (4, 128, 2, 2, 128, 3, 3, 2, 1, 1),      # 1 x 1 outputs
]


@pytest.mark.parametrize("case", BAND_S2_CASES)
def test_band_s2_takes_the_kernel(case):
    N, Cc, H, W, Kk, R, S, s, p, d = case
    desc = ops.conv_desc(N, Cc, H, W, Kk, R, S, (s, s), (p, p), (d, d), 1, Cc, Kk)
    assert L.load().cvhip_conv2d_band_plan(C.byref(desc), 0, None) == 1
    assert L.load().cvhip_conv2d_band_plan(C.byref(desc), 1, None) == 0   # the input gradient stays on the stride-parity classes


@pytest.mark.parametrize("case", BAND_CASES + BAND_S2_CASES)
def test_band_fprop(case, force_band):
    """plain forward, no bias (ConvModule convolutions in front of a norm layer carry none, conv_module.py:112-119)"""
    _skip_unless_form_runs(case, force_band, False)
    N, Cc, H, W, Kk, R, S, s, p, d = case
    x, w = K._mk(case)
    ref = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    st, Kp = K._prep(case, w, False)
    xd = to_nhwc_dev(x)
    P, Q = ref.shape[2:]
    y = ops.empty_nhwc(N, Kk, P, Q, dev())
    y.fill_(float("nan"))
    desc = ops.conv_desc(N, Cc, H, W, Kk, R, S, (s, s), (p, p), (d, d), 1, Cc, Kk)
    L.call("cvhip_conv2d_fprop", C.byref(desc), xd.data_ptr(), st.w_fprop.data_ptr(), None, y.data_ptr(), None, ops._stream())
    torch.cuda.synchronize()
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    assert max_rel(got, ref) < 2 ** -7, max_rel(got, ref)
    assert rel_l2(got, ref) < 4e-3


def _act(u, act):
    if act == L.ACT_RELU:
        return torch.relu(u)
    if act == L.ACT_SILU:
        return u * torch.sigmoid(u)
    if act == L.ACT_LEAKY:
        return torch.where(u > 0, u, 0.1 * u)
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


@pytest.mark.parametrize("act", [L.ACT_SILU, L.ACT_RELU, L.ACT_LEAKY, L.ACT_HSWISH, L.ACT_SIGMOID, L.ACT_NONE])
@pytest.mark.parametrize("case", [BAND_CASES[0], BAND_CASES[2], BAND_CASES[4], BAND_CASES[6], BAND_CASES[8], BAND_CASES[10], BAND_S2_CASES[1], BAND_S2_CASES[4]])
def test_band_fused_epilogue(case, act, force_band):
    """inference form (round 6): y = act((conv(x) + bias) * scale + shift) (+ residual before or after the activation) in the band
    kernel's own store pass — the eval-mode ConvModule (conv_module.py:201-214), the Darknet shortcut x + act(bn(conv)) and the ResNet
    tail relu(bn(conv) + identity)"""
    _skip_unless_form_runs(case, force_band, False)
    N, Cc, H, W, Kk, R, S, s, p, d = case
    x, w = K._mk(case, 21)
    g = torch.Generator().manual_seed(22)
    bias = torch.randn(Kk, generator=g) * 0.2
    sc = torch.rand(Kk, generator=g) + 0.5
    sh = torch.randn(Kk, generator=g) * 0.3
    conv = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    P, Q = conv.shape[2:]
    res = rnd(torch.randn(N, Kk, P, Q, generator=g))
    st, Kp = K._prep(case, w, False)
    xd, resd = to_nhwc_dev(x), to_nhwc_dev(res)
    desc = ops.conv_desc(N, Cc, H, W, Kk, R, S, (s, s), (p, p), (d, d), 1, Cc, Kk)
    scd, shd, bd = sc.to(dev()), sh.to(dev()), bias.to(dev())
    v = lambda t: t.view(1, -1, 1, 1)

    def run(**kw):
        out = ops.empty_nhwc(N, Kk, P, Q, dev())
        out.fill_(float("nan"))
        L.check(_fused(desc, xd, st.w_fprop, out, **kw), "cvhip_conv2d_fprop_fused")
        torch.cuda.synchronize()
        return out.float().cpu()

    def close(got, ref):
        assert torch.isfinite(got).all()
        assert max_rel(got, ref) < 2 ** -7, max_rel(got, ref)
        assert rel_l2(got, ref) < 4e-3

    # folded BatchNorm + activation
    close(run(ep_scale=scd, ep_shift=shd, ep_act=act, ep_act_param=0.1), _act(conv * v(sc) + v(sh), act))
    # + bias, + residual after the activation (Darknet shortcut)
    close(run(bias=bd, ep_scale=scd, ep_shift=shd, ep_act=act, ep_act_param=0.1, residual=resd, residual_ld=Kk),
          _act((conv + v(bias)) * v(sc) + v(sh), act) + res)
    # residual before the activation (ResNet tail)
    close(run(ep_scale=scd, ep_shift=shd, ep_act=act, ep_act_param=0.1, residual=resd, residual_ld=Kk, residual_pre=1),
          _act(conv * v(sc) + v(sh) + res, act))
    # activation only / bias only
    close(run(ep_act=act, ep_act_param=0.1), _act(conv, act))
    close(run(bias=bd), conv + v(bias))


@pytest.mark.parametrize("case", BAND_CASES + BAND_S2_CASES)
def test_band_fprop_bn_acc(case, force_band):
    """training form: raw output + BatchNorm sums folded into the layer's fp64 accumulator (cvhip_conv2d_fprop_acc)"""
    _skip_unless_form_runs(case, force_band, False)
    N, Cc, H, W, Kk, R, S, s, p, d = case
    x, w = K._mk(case, 1)
    ref = F.conv2d(x, w, None, stride=s, padding=p, dilation=d)
    st, Kp = K._prep(case, w, False)
    xd = to_nhwc_dev(x)
    P, Q = ref.shape[2:]
    y = ops.empty_nhwc(N, Kk, P, Q, dev())
    desc = ops.conv_desc(N, Cc, H, W, Kk, R, S, (s, s), (p, p), (d, d), 1, Cc, Kk)
    acc = torch.zeros(L.BN_ACC_SHARDS, 2, Kk, dtype=torch.float64, device=dev())
    L.call("cvhip_conv2d_fprop_acc", C.byref(desc), xd.data_ptr(), st.w_fprop.data_ptr(), y.data_ptr(), acc.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    s1, s2 = acc.sum(0).cpu()
    r = ref.double()
    assert rel_l2(s1, r.sum((0, 2, 3))) < 1e-3 or float((s1 - r.sum((0, 2, 3))).abs().max()) < 1e-2
    assert rel_l2(s2, (r * r).sum((0, 2, 3))) < 1e-3
    assert rel_l2(y.float().cpu(), ref) < 4e-3


@pytest.mark.parametrize("case", BAND_CASES)
def test_band_dgrad(case, force_band):
    _skip_unless_form_runs(case, force_band, True)
    K.test_conv_dgrad(case)


@pytest.mark.parametrize("case", [BAND_CASES[0], BAND_CASES[4], BAND_CASES[6]] + BAND_CASES[12:])
def test_band_dgrad_add(case, force_band):
    """dgrad with the skip-connection gradient added in the epilogue (cvhip_conv2d_dgrad_add)"""
    _skip_unless_form_runs(case, force_band, True)
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


def _band_image_expected(rowmajor, nrows, cin):
    """conv_plan.h "band image": 16-byte vector v = ((tap * (cin/32) + c/32) * (nrows/16) + n/16) * 64 + ((c%32)/8) * 16 + n%16 holds
    Wt[n][tap*cin + c .. c+7] of the row-major image Wt[nrows][9*cin]"""
    w = rowmajor.reshape(nrows, 9, cin // 32, 4, 8)            # n, tap, chunk, g, e
    w = w.reshape(nrows // 16, 16, 9, cin // 32, 4, 8)         # f, r, tap, chunk, g, e
    return w.permute(2, 3, 0, 4, 1, 5).contiguous().reshape(-1)  # tap, chunk, f, g, r, e


@pytest.mark.parametrize("shape", [(128, 128), (64, 64), (32, 32), (64, 128), (256, 96), (128, 32)])
@pytest.mark.parametrize("batched", [False, True])
def test_band_image_layout(shape, batched):
    """the fragment-ordered weight copies behind the fprop / dgrad images (cvhip_conv2d_prep_weights and the batched cvhip_prep_plan_run)
    hold exactly the row-major images' values, permuted as csrc/conv_plan.h states"""
    Kk, Cc = shape
    w = torch.randn(Kk, Cc, 3, 3, generator=torch.Generator().manual_seed(Kk + Cc)).to(dev()).contiguous(memory_format=torch.channels_last)
    desc = ops.conv_desc(2, Cc, 12, 12, Kk, 3, 3, (1, 1), (1, 1), (1, 1), 1, Cc, Kk)
    n = Kk * 9 * Cc
    lib = L.load()
    has_f = Cc % 32 == 0 and (Kk in (32, 64) or Kk % 128 == 0)
    has_d = Kk % 32 == 0 and (Cc in (32, 64) or Cc % 128 == 0)
    assert lib.cvhip_conv2d_weight_image_elems(C.byref(desc), 0) == n * (2 if has_f else 1)
    assert lib.cvhip_conv2d_weight_image_elems(C.byref(desc), 1) == n * (2 if has_d else 1)
    st = ops.ConvState()
    st.prepare(w, desc, True, ("img", Kk, Cc))
    if batched:
        plan = ops.PrepPlan([st])
        assert plan.n == 1
        st._wf_buf.fill_(float("nan"))
        st.w_dgrad.fill_(float("nan"))
        plan.run()
    torch.cuda.synchronize()
    wf, wd = st._wf_buf.float().cpu(), st.w_dgrad.float().cpu()
    master = w.permute(0, 2, 3, 1).contiguous().float().cpu()          # K, R, S, C
    ref_f = master.to(K.BF).float().reshape(-1)
    assert torch.equal(wf[:n], ref_f)
    ref_d = master.permute(3, 1, 2, 0).contiguous().to(K.BF).float().reshape(-1)   # C, R, S, K (stride 1: taps in kernel order)
    assert torch.equal(wd[:n], ref_d)
    if has_f:
        assert torch.equal(wf[n:], _band_image_expected(ref_f, Kk, Cc))
    if has_d:
        assert torch.equal(wd[n:2 * n], _band_image_expected(ref_d, Cc, Kk))
