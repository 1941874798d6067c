"""Image stems reading the dataloader's own tensor (round 4): cvhip_conv_fuse.x_image / cvhip_conv2d_wgrad_image let the image-stem
kernel read the fp32 NCHW batch plane by plane, so the fp32 NCHW -> 16-bit NHWC pass in front of the first convolution disappears.
The result must be what the two-step form gives: the same 16-bit rounding of the same values feeds the same MFMAs (outputs and
BatchNorm sums bit-identical); the weight gradient differs only by the order of its fp32 atomics."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from cvpytorch_amd import bricks, lib as L, ops


def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("N,H,W,K,R,s,p", [(8, 256, 256, 32, 6, 2, 2), (4, 257, 390, 32, 3, 2, 1), (16, 128, 256, 16, 3, 1, 1), (2, 512, 512, 32, 7, 2, 3)])
def test_stem_fprop_and_wgrad_from_the_fp32_image_equal_the_two_step_form(N, H, W, K, R, s, p):
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    torch.manual_seed(N * 7 + R)
    img = torch.randn(N, 3, H, W, device=dev())
    desc = ops.conv_desc(N, 8, H, W, K, R, R, (s, s), (p, p), (1, 1), 1, 8, K, 0, 3)
    if lib.cvhip_conv_stem_blocks(C.byref(desc)) <= 0:
        pytest.skip("not an image-stem problem")
    P, Q = ops.conv_out_hw(H, W, R, R, (s, s), (p, p), (1, 1))
    w = (torch.randn(K, 3, R, R, device=dev()) * 0.1).contiguous(memory_format=torch.channels_last)
    state = ops.ConvState()
    state.prepare(w, ops.conv_desc(N, 8, H, W, K, R, R, (s, s), (p, p), (1, 1), 1, 8, K, 0, 3), False, ("t", 0))
    x8 = ops.images_to_nhwc(img, cpad=8)
    rows = lib.cvhip_conv2d_fprop_stats_rows(C.byref(desc))
    ya = torch.empty(N, P, Q, K, dtype=ops.ACT_DTYPE, device=dev())
    yb = torch.empty_like(ya)
    pa = torch.zeros(rows + L.REDUCE_SCRATCH_ROWS, 2, K, device=dev())
    pb = torch.zeros_like(pa)
    L.call("cvhip_conv2d_fprop", C.byref(desc), x8.data_ptr(), state.w_fprop.data_ptr(), None, ya.data_ptr(), pa.data_ptr(), st)
    f = L.ConvFuse()
    f.x_image, f.x_image_planes, f.stats_partial = img.data_ptr(), 3, pb.data_ptr()
    L.call("cvhip_conv2d_fprop_fused", C.byref(desc), None, state.w_fprop.data_ptr(), yb.data_ptr(), C.byref(f), st)
    torch.cuda.synchronize()
    assert torch.equal(ya.view(torch.int16), yb.view(torch.int16))
    assert torch.equal(pa[:rows], pb[:rows])
    dy = torch.randn(N, P, Q, K, device=dev()).to(ops.ACT_DTYPE)
    da = torch.zeros(K, R, R, 8, device=dev())
    db = torch.zeros_like(da)
    L.call("cvhip_conv2d_wgrad", C.byref(desc), x8.data_ptr(), dy.data_ptr(), da.data_ptr(), 1, st)
    L.call("cvhip_conv2d_wgrad_image", C.byref(desc), img.data_ptr(), 3, dy.data_ptr(), db.data_ptr(), st)
    torch.cuda.synchronize()
    assert float((da - db).abs().max()) <= 1e-4 * float(da.abs().max()) + 1e-6
    assert float(db[..., 3:].abs().max()) == 0.0


def test_stem_image_refusals():
    lib = L.load()
    img = torch.randn(2, 3, 64, 64, device=dev())
    y = torch.empty(2, 32, 32, 32, dtype=ops.ACT_DTYPE, device=dev())
    w = torch.zeros(32 * 36 * 8, dtype=ops.ACT_DTYPE, device=dev())
    small = ops.conv_desc(2, 8, 64, 64, 32, 6, 6, (2, 2), (2, 2), (1, 1), 1, 8, 32, 0, 3)   # too few tiles for the stem kernel
    assert lib.cvhip_conv_stem_blocks(C.byref(small)) == 0
    f = L.ConvFuse()
    f.x_image, f.x_image_planes = img.data_ptr(), 3
    assert lib.cvhip_conv2d_fprop_fused(C.byref(small), None, w.data_ptr(), y.data_ptr(), C.byref(f), None) == L.ERR_UNSUPPORTED
    f.x_image_planes = 5
    assert lib.cvhip_conv2d_fprop_fused(C.byref(small), None, w.data_ptr(), y.data_ptr(), C.byref(f), None) == L.ERR_INVALID
    dw = torch.zeros(32, 6, 6, 8, device=dev())
    assert lib.cvhip_conv2d_wgrad_image(C.byref(small), img.data_ptr(), 3, y.data_ptr(), dw.data_ptr(), None) == L.ERR_UNSUPPORTED


def test_stem_module_train_step_with_and_without_the_conversion_pass():
    """HipConvModule(3 -> 32, 6x6 s2) + BN + SiLU, forward + backward in training mode: the image path (default) against
    CVHIP_STEM_IMAGE=0's explicit conversion pass."""
    torch.manual_seed(0)
    m = bricks.HipConvModule(3, 32, 6, stride=2, padding=2, norm_cfg=dict(type="HipBN"), act_cfg=dict(type="HipSiLU")).to(dev()).train()
    img = torch.randn(8, 3, 256, 256, device=dev())
    outs = []
    old = ops._STEM_IMAGE
    try:
        for flag in (True, False):
            ops._STEM_IMAGE = flag
            m.zero_grad(set_to_none=True)
            ops.TIMER.enabled = True
            ops.TIMER.reset()
            z = m(img)
            z.float().square().mean().backward()
            torch.cuda.synchronize()
            names = [r[0] for r in ops.TIMER.records]
            ops.TIMER.enabled = False
            ops.TIMER.reset()
            outs.append((z.detach().float().clone(), m.conv.weight.grad.detach().clone(), names))
    finally:
        ops._STEM_IMAGE = old
        ops.TIMER.enabled = False
    (za, ga, _), (zb, gb, _) = outs
    assert torch.equal(za, zb)
    assert float((ga - gb).abs().max()) <= 1e-4 * float(gb.abs().max()) + 1e-7


def test_stem_wgrad_falls_back_when_the_stem_kernel_refuses(monkeypatch):
    """cvhip_conv2d_wgrad_image may answer CVHIP_ERR_UNSUPPORTED (CVHIP_STEM_WGRAD=0, operand pitch / alignment): the step
    must then take the conversion pass + the generic weight-gradient kernel instead of raising."""
    import torch
    from cvpytorch_amd import bricks, lib as L, ops
    d = torch.device("cuda:0")
    torch.manual_seed(4)
    m = bricks.HipConvModule(3, 32, 6, 2, 2, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")).to(d).train()
    x = torch.rand(4, 3, 128, 128, device=d)
    g = (torch.randn(4, 32, 64, 64, device=d) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)

    def run(refuse):
        for p in m.parameters():
            p.grad = None
        real = L.call

        def call(name, *a):
            if refuse and name == "cvhip_conv2d_wgrad_image":
                raise L.CvhipError("cvhip_conv2d_wgrad_image failed: unsupported shape/feature")
            return real(name, *a)

        monkeypatch.setattr(L, "call", call)
        m(x).backward(g)
        torch.cuda.synchronize()
        monkeypatch.setattr(L, "call", real)
        return m.conv.weight.grad.float().clone()

    a, b = run(False), run(True)
    rel = float((a - b).norm() / a.norm())
    assert rel <= 2e-3, rel     # the image path rounds the fp32 pixels to 16 bits exactly like the conversion pass: same operands


@pytest.mark.parametrize("k,s,pad,act", [(6, 2, 2, "SiLU"), (3, 2, 1, "ReLU"), (3, 1, 1, "SiLU")])
def test_stem_wgrad_with_bn_backward_on_load_equals_apply_pass(k, s, pad, act):
    """Round 5: the image stem's weight gradient taken straight from dz (cvhip_conv2d_wgrad_stem_bn: BN + activation backward applied on
    load, no apply pass, no dy tensor) == apply pass + plain stem weight gradient. Same fp32 sums; dy is formed as sc*du + b1*y + c1
    instead of sc*(du - k1 - xhat*k2), so single values may round to the neighbouring 16-bit number: dW rel-L2 <= 3e-3 (the fused 1x1
    backward's own bound), dgamma / dbeta identical (both read the same accumulator)."""
    import torch
    from cvpytorch_amd import bricks, lib as L, ops
    d = torch.device("cuda:0")
    torch.manual_seed(6)
    m = bricks.HipConvModule(3, 32, k, s, pad, norm_cfg=dict(type="BN"), act_cfg=dict(type=act)).to(d).train()
    with torch.no_grad():
        m.bn.weight.uniform_(0.5, 1.5)
        m.bn.bias.normal_(0, 0.2)
    N, H = (8, 256) if s == 2 else (8, 128)
    x = torch.rand(N, 3, H, H, device=d)
    P = (H + 2 * pad - k) // s + 1
    g = (torch.randn(N, 32, P, P, device=d) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    calls = []
    real = L.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)

    out = {}
    try:
        L.call = spy
        for flag in (False, True):
            ops._STEM_BN = flag
            calls.clear()
            for p_ in m.parameters():
                p_.grad = None
            m(x).backward(g)
            torch.cuda.synchronize()
            out[flag] = (m.conv.weight.grad.float().clone(), m.bn.weight.grad.float().clone(), m.bn.bias.grad.float().clone(), list(calls))
    finally:
        L.call = real
        ops._STEM_BN = True
    assert "cvhip_conv2d_wgrad_stem_bn" in out[True][3] and "cvhip_bn_act_bwd_apply_acc" not in out[True][3]
    assert "cvhip_conv2d_wgrad_stem_bn" not in out[False][3] and "cvhip_bn_act_bwd_apply_acc" in out[False][3]
    rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
    assert rel(out[True][0], out[False][0]) <= 3e-3, rel(out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1]) and torch.equal(out[True][2], out[False][2])
