"""GPU parity tests of the depthwise 3x3 forward that emits its training-mode BatchNorm sums (cvhip_dwconv2d_fprop_stats, csrc/dwconv.hip
LDS strip kernel): one pass instead of the forward + a reduction pass over the stored output — the statistics half of
aten::native_batch_norm behind the depthwise convolution of DepthwiseSeparableConvModule (depthwise_separable_conv_module.py:10-99,
conv_module.py:209-211).

The output must be BIT-identical to cvhip_dwconv2d_fprop's; the partial rows, summed, must match the sums of the fp32 reference
convolution (relative L2 <= 1e-3: the kernel sums its fp32 accumulators before their rounding to 16 bits, as the dense convolutions'
epilogues do). The module-level test holds the whole Conv-BN-act layer (forward, input / weight / BN parameter gradients, running
statistics) to the reduction-pass form (CVHIP_DW_STATS=0 semantics) within the 16-bit storage tolerance."""
import ctypes as C
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import test_gpu_kernels as K
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

dev, rel_l2, to_nhwc_dev = K.dev, K.rel_l2, K.to_nhwc_dev

CASES = [
    # N, C, H, W, pad
    (2, 560, 40, 96, 1),     # DeepLabv3+ decoder width (35 + 35 channel vectors: two chunks at a 1120-byte pixel pitch)
    (2, 512, 33, 70, 1),     # ragged strips / row blocks
    (3, 304, 24, 64, 1),
    (2, 64, 48, 200, 1),
    (1, 256, 16, 128, 1),
    (2, 128, 20, 60, 0),     # no padding: output smaller than the input
]


@pytest.mark.parametrize("case", CASES)
def test_dw_fprop_stats(case):
    N, Cc, H, W, p = case
    torch.manual_seed(1)
    x = K.bf(torch.randn(N, Cc, H, W))
    w = torch.randn(Cc, 1, 3, 3)
    ref = F.conv2d(x, w, None, padding=p, groups=Cc)
    P, Q = ref.shape[2:]
    desc = ops.conv_desc(N, Cc, H, W, Cc, 3, 3, (1, 1), (p, p), (1, 1), Cc, Cc, Cc)
    xd = to_nhwc_dev(x)
    wd = w.reshape(Cc, 3, 3).contiguous().to(dev())
    y0 = ops.empty_nhwc(N, Cc, P, Q, dev())
    y1 = ops.empty_nhwc(N, Cc, P, Q, dev())
    y1.fill_(float("nan"))
    lib = L.load()
    rows = int(lib.cvhip_dwconv2d_fprop_stats_rows(C.byref(desc), xd.data_ptr(), y1.data_ptr()))
    if rows == 0:
        pytest.skip("the strip kernel does not run this geometry")
    assert rows > 0
    part = torch.full((rows + L.REDUCE_SCRATCH_ROWS, 2, Cc), float("nan"), device=dev())
    L.call("cvhip_dwconv2d_fprop", C.byref(desc), xd.data_ptr(), wd.data_ptr(), None, y0.data_ptr(), ops._stream())
    L.call("cvhip_dwconv2d_fprop_stats", C.byref(desc), xd.data_ptr(), wd.data_ptr(), None, y1.data_ptr(), part.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    pr = part[:rows].double().cpu()
    assert torch.isfinite(pr).all()          # every row written by every channel chunk
    s1, s2 = pr[:, 0].sum(0), pr[:, 1].sum(0)
    r = ref.double()
    assert rel_l2(s1, r.sum((0, 2, 3))) < 1e-3 or float((s1 - r.sum((0, 2, 3))).abs().max()) < 1e-2
    assert rel_l2(s2, (r * r).sum((0, 2, 3))) < 1e-3


def test_dw_stats_rows_zero_where_the_strip_kernel_does_not_run():
    lib = L.load()
    for (Cc, H, W, s, p, d) in ((64, 6, 6, 1, 1, 1), (64, 40, 96, 2, 1, 1), (64, 40, 96, 1, 2, 2)):
        desc = ops.conv_desc(2, Cc, H, W, Cc, 3, 3, (s, s), (p, p), (d, d), Cc, Cc, Cc)
        x = torch.zeros((2, H, W, Cc), dtype=K.BF, device=dev())
        P, Q = ops.conv_out_hw(H, W, 3, 3, (s, s), (p, p), (d, d))
        y = torch.zeros((2, P, Q, Cc), dtype=K.BF, device=dev())
        assert int(lib.cvhip_dwconv2d_fprop_stats_rows(C.byref(desc), x.data_ptr(), y.data_ptr())) == 0


def _run_layer(monkeypatch, on):
    """one depthwise Conv-BN-ReLU layer, forward + backward, with the fused statistics on / off"""
    monkeypatch.setattr(ops, "_DW_STATS", on)
    from cvpytorch_amd import bricks
    torch.manual_seed(5)
    m = bricks.ConvModule(64, 64, 3, padding=1, groups=64, norm_cfg=dict(type="BN", requires_grad=True), act_cfg=dict(type="ReLU")).to(dev()).train()
    x = (torch.randn(2, 64, 40, 96, device=dev())).to(K.BF).float().requires_grad_(True)
    y = m(x)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(6)).to(dev())
    (y.float() * g).sum().backward()
    torch.cuda.synchronize()
    bn = m.bn
    return (y.detach().float().cpu(), x.grad.float().cpu(), m.conv.weight.grad.float().cpu(), bn.weight.grad.float().cpu(), bn.bias.grad.float().cpu(),
            bn.running_mean.float().cpu(), bn.running_var.float().cpu())


def test_dw_layer_with_fused_statistics_equals_reduction_pass(monkeypatch):
    a = _run_layer(monkeypatch, True)
    b = _run_layer(monkeypatch, False)
    names = ("y", "dx", "dw", "dgamma", "dbeta", "running_mean", "running_var")
    for n, u, v in zip(names, a, b):
        assert torch.isfinite(u).all(), n
        # the statistics differ by the rounding of the stored output only (sums of fp32 accumulators vs of 16-bit values)
        assert rel_l2(u, v) < 6e-3, (n, rel_l2(u, v))


# ---- channel scaling y = x * s[n][c] (STDC attention-refinement / feature-fusion gates, stdc_neck.py:53-58,110-114): the gate's gradient
# ds = sum_p dy * x through cvhip_channel_scale_bwd_ds (round 6: one pass instead of torch's convert + mul + strided reduce) -------------
@pytest.mark.parametrize("N,Cc,H,W", [(3, 128, 16, 32), (2, 256, 32, 64), (2, 1024, 8, 8), (1, 64, 7, 5), (2, 24, 9, 9)])
def test_channel_scale_gradients(N, Cc, H, W):
    torch.manual_seed(3)
    x = K.bf(torch.randn(N, Cc, H, W))
    s = K.bf(torch.rand(N, Cc, 1, 1))
    g = K.bf(torch.randn(N, Cc, H, W))
    xr, sr = x.clone().requires_grad_(True), s.clone().requires_grad_(True)
    (xr * sr).backward(g)
    xd = x.to(dev()).to(K.BF).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    sd = s.to(dev()).to(K.BF).requires_grad_(True)
    y = ops.channel_scale(xd, sd)
    y.backward(g.to(dev()).to(K.BF).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()
    assert rel_l2(y.detach().float().cpu(), (x * s)) < 4e-3
    assert rel_l2(xd.grad.float().cpu(), xr.grad) < 4e-3
    assert rel_l2(sd.grad.float().cpu().reshape(N, Cc), sr.grad.reshape(N, Cc)) < 4e-3   # fp32 sums, one rounding to the gate's 16 bits
