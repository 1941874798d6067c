"""GPU tests of the two §8(f) rows built this round: (3) the on-device input pipeline (uint8 NHWC -> normalised bf16 NHWC
kernel + DevicePrefetcher) against the reference's CPU ToTensor + Normalize, and (4) the deploy path (conv+BN folding,
RepConv re-parameterisation) against the un-fused eval forward."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvpytorch_amd import bricks, data, deploy, yolo_blocks, yolov5, yolov7
from test_gpu_modules import dev, rel_l2

MEAN, STD = (0.406, 0.456, 0.485), (0.225, 0.224, 0.229)


def _ref_normalize(u8_nhwc):
    x = u8_nhwc.permute(0, 3, 1, 2).float() / 255.0                      # ToTensor
    m = torch.tensor(MEAN).view(1, 3, 1, 1)
    s = torch.tensor(STD).view(1, 3, 1, 1)
    return (x - m) / s                                                   # Normalize (det_transforms.py:102-109)


def test_prefetcher_equals_cpu_totensor_normalize_and_feeds_the_stem():
    g = torch.Generator().manual_seed(0)
    batches = [(torch.randint(0, 256, (3, 32, 40, 3), generator=g, dtype=torch.uint8), torch.arange(4.0) + i) for i in range(3)]
    pf = data.DevicePrefetcher(batches, dev(), MEAN, STD)
    torch.manual_seed(0)
    stem = bricks.HipConvModule(3, 16, 6, 2, 2, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")).to(dev()).eval()
    n = 0
    for (x, t), (u8, tt) in zip(pf, batches):
        n += 1
        assert tuple(x.shape) == (3, 8, 32, 40) and x.dtype == torch.bfloat16 and x.is_cuda
        ref = _ref_normalize(u8)
        got = x[:, :3].float().cpu()
        assert float((got - ref.to(torch.bfloat16).float()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
        assert float(x[:, 3:].float().abs().max()) == 0.0
        assert torch.equal(t.cpu(), tt)
        with torch.no_grad():
            a = stem(x)                          # bf16 NHWC, channels padded to 8: consumed as is
            b = stem(ref.to(dev()))              # fp32 NCHW path (cvhip_nchw_f32_to_nhwc_bf16)
        assert rel_l2(a.float(), b.float()) < 1e-2
    assert n == 3


def test_fuse_model_eval_forward_matches_unfused():
    torch.manual_seed(0)
    m = torch.nn.Sequential(bricks.HipConvModule(16, 32, 3, padding=1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")),
                            yolo_blocks.CSPLayer(32, 32, n=1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")),
                            yolo_blocks.SPPF(32, 32, kernel_sizes=5, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU"))).to(dev())
    g = torch.Generator().manual_seed(1)
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.running_mean.copy_(torch.randn(mm.num_features, generator=g).to(dev()) * 0.2)
            mm.running_var.copy_((torch.rand(mm.num_features, generator=g) + 0.5).to(dev()))
            mm.weight.data.copy_((torch.rand(mm.num_features, generator=g) + 0.5).to(dev()))
            mm.bias.data.copy_(torch.randn(mm.num_features, generator=g).to(dev()) * 0.2)
    m.eval()
    x = torch.randn(2, 16, 12, 10).to(torch.bfloat16).to(dev()).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = m(x).float()
        n_bn = sum(isinstance(mm, torch.nn.BatchNorm2d) for mm in m.modules())
        deploy.fuse_model(m)
        assert n_bn > 0 and sum(isinstance(mm, torch.nn.BatchNorm2d) for mm in m.modules()) == 0
        got = m(x).float()
    assert rel_l2(got, ref) < 2e-2, rel_l2(got, ref)


def test_reparam_repconv_eval_forward_matches():
    torch.manual_seed(0)
    head = yolov7.YOLOv7Head(width_mul=0.125).to(dev())
    g = torch.Generator().manual_seed(2)
    for mm in head.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.running_mean.copy_(torch.randn(mm.num_features, generator=g).to(dev()) * 0.2)
            mm.running_var.copy_((torch.rand(mm.num_features, generator=g) + 0.5).to(dev()))
    head.eval()
    xs = [torch.randn(2, c, s, s).to(torch.bfloat16).to(dev()).contiguous(memory_format=torch.channels_last) for c, s in ((16, 16), (32, 8), (64, 4))]
    with torch.no_grad():
        ref = [o.float() for o in head(xs)]
        deploy.reparam_repconv(head)
        assert all(hasattr(mm, "rbr_reparam") for mm in head.modules() if isinstance(mm, yolov7.RepConv))
        got = [o.float() for o in head(xs)]
    for a, b in zip(got, ref):
        assert rel_l2(a, b) < 2e-2, rel_l2(a, b)


def _randomise_bn(model, seed=1):
    g = torch.Generator().manual_seed(seed)
    for mm in model.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.running_mean.copy_(torch.randn(mm.num_features, generator=g).to(dev()) * 0.2)
            mm.running_var.copy_((torch.rand(mm.num_features, generator=g) + 0.5).to(dev()))
            mm.weight.data.copy_((torch.rand(mm.num_features, generator=g) + 0.5).to(dev()))
            mm.bias.data.copy_(torch.randn(mm.num_features, generator=g).to(dev()) * 0.2)


def test_fuse_model_on_index_based_forwards_deeplab_and_stdc():
    """models whose forward indexes (conv, bn) children directly (deeplab.ResNet stem / Bottleneck downsample and
    sibling convN/bnN, stdc._DwPwSkip) must still run after fuse_model and give the un-fused eval result; every BN is folded."""
    from cvpytorch_amd import deeplab, stdc
    torch.manual_seed(0)
    for build, shape in ((lambda: deeplab.EncoderDecoder(19, output_stride=32), (2, 3, 64, 96)),
                         (lambda: stdc.STDCNet("stdc1"), (2, 3, 64, 64))):
        m = build().to(dev())
        _randomise_bn(m)
        m.eval()
        x = torch.randn(*shape).to(dev())
        with torch.no_grad():
            ref = m.backbone(x) if hasattr(m, "backbone") else m(x)
            if hasattr(m, "backbone"):
                ref = m.head(ref)
        deploy.fuse_model(m)
        assert not any(isinstance(mm, torch.nn.BatchNorm2d) for mm in m.modules()), "a BatchNorm survived the folding"
        with torch.no_grad():
            out = m.head(m.backbone(x)) if hasattr(m, "backbone") else m(x)
        ref = ref if isinstance(ref, (list, tuple)) else [ref]
        out = out if isinstance(out, (list, tuple)) else [out]
        assert len(ref) == len(out)
        for a, b in zip(out, ref):
            assert rel_l2(a.float(), b.float()) < 3e-2, rel_l2(a.float(), b.float())


@pytest.mark.parametrize("Cc,Hi,Wi,Ho,Wo", [(32, 8, 16, 20, 33), (19, 7, 5, 7, 5), (64, 16, 16, 8, 8), (40, 1, 1, 9, 11), (24, 9, 12, 27, 24)])
def test_nearest_resize_matches_torch_exactly(Cc, Hi, Wi, Ho, Wo):
    """cvhip_resize_nearest_fwd / _bwd vs F.interpolate(mode='nearest'): forward is a copy -> bit-exact; backward sums gradient
    runs in fp32 and rounds once."""
    from cvpytorch_amd import ops
    torch.manual_seed(Ho)
    x = torch.randn(3, Cc, Hi, Wi).to(torch.bfloat16)
    xd = x.to(dev()).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = ops.resize_nearest(xd, (Ho, Wo))
    xr = x.float().requires_grad_(True)
    yr = torch.nn.functional.interpolate(xr, (Ho, Wo), mode="nearest")
    assert torch.equal(y.detach().float().cpu(), yr.detach())
    g = torch.randn(3, Cc, Ho, Wo).to(torch.bfloat16)
    y.backward(g.to(dev()).contiguous(memory_format=torch.channels_last))
    yr.backward(g.float())
    torch.cuda.synchronize()
    assert rel_l2(xd.grad.float().cpu(), xr.grad) < 4e-3


def test_eval_forward_is_one_launch_per_convmodule_and_matches_the_oracle_and_the_two_pass_form():
    """Round 4: in inference (nothing requires a gradient) a ConvModule — conv, eval-mode BatchNorm scale / shift, activation,
    Darknet shortcut — is ONE cvhip_conv2d_fprop_fused launch (conv_module.py:201-214 in eval mode); after deploy.fuse_model the
    BatchNorm is folded into the weights (utils/fuse.py:32-54) and only bias + activation remain in the epilogue. Checked on the
    YOLOv5-n backbone + neck + head: (a) no BN / activation element-wise launch is left, (b) outputs equal the fp32 oracle's eval
    forward within the storage tolerance, (c) and the two-pass form (CVHIP_EPI_FUSE off) to bf16 rounding."""
    import importlib
    from cvpytorch_amd import ops, yolov5
    from oracle import torch_ref as R
    torch.manual_seed(0)
    ref = R.YOLOv5(20, "n")
    _randomise_bn(ref)
    ref.eval()
    hip = yolov5.YOLOv5(20, "n").to(dev())
    hip.load_state_dict(ref.state_dict(), strict=False)
    hip.eval()
    x = torch.randn(2, 3, 128, 160)
    with torch.no_grad():
        want = ref.forward_features(x)[0] if hasattr(ref, "forward_features") else ref.detect(ref.neck(ref.backbone(x)))[0]
        ops.TIMER.enabled = True
        ops.TIMER.reset()
        got = hip.forward_features(x.to(dev()))[0]
        torch.cuda.synchronize()
        names = [r[0] for r in ops.TIMER.records]
        ops.TIMER.enabled = False
        ops.TIMER.reset()
        assert not [n for n in names if "ew_kernel" in n or "bn_act" in n], names
        assert sum(n == "conv_fused_inference" for n in names) >= 50
        old = ops._EPI_FUSE
        ops._EPI_FUSE = False
        try:
            two_pass = hip.forward_features(x.to(dev()))[0]
        finally:
            ops._EPI_FUSE = old
        deploy.fuse_model(hip)
        folded = hip.forward_features(x.to(dev()))[0]
    torch.cuda.synchronize()
    w = want.float()
    scale = float(w.abs().max())
    for name, t in (("fused eval BN", got), ("two-pass", two_pass), ("folded", folded)):
        err = float((t.float().cpu() - w).abs().max())
        assert err <= 3e-2 * scale, (name, err, scale)
    assert float((got.float() - two_pass.float()).abs().max()) <= 2e-2 * scale


def test_deeplab_eval_forward_has_no_elementwise_bn_act_pass_and_matches_the_two_pass_form():
    """DeepLabv3+ in inference: the ResNet bottleneck tails relu(bn3(conv3) + identity) (residual BEFORE the activation), the
    dilated / depthwise-separable ASPP and decoder modules (cvhip_dwconv2d_fprop_act) and every other ConvModule run as ONE launch
    each — before and after deploy.fuse_model — and give the two-pass result to 16-bit rounding."""
    from cvpytorch_amd import deeplab, ops
    torch.manual_seed(0)
    m = deeplab.EncoderDecoder(19, output_stride=32).to(dev())
    _randomise_bn(m)
    m.eval()
    x = torch.randn(2, 3, 64, 96).to(dev())

    def run():
        return m.forward_features(x)[1][0].float()

    with torch.no_grad():
        old = ops._EPI_FUSE
        ops._EPI_FUSE = False
        try:
            ref = run()
        finally:
            ops._EPI_FUSE = old
        ops.TIMER.enabled = True
        ops.TIMER.reset()
        got = run()
        torch.cuda.synchronize()
        names = [r[0] for r in ops.TIMER.records]
        ops.TIMER.enabled = False
        ops.TIMER.reset()
        assert not [n for n in names if "ew_kernel" in n or "bn_act" in n], sorted(set(names))
        assert sum(n == "dw_fused_inference" for n in names) >= 4 and sum(n == "conv_fused_inference" for n in names) >= 50
        deploy.fuse_model(m)
        folded = run()
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 3e-2 * scale
    # folding rounds W*scale to 16 bits once per layer (the un-folded forms keep fp32 scale/shift in the epilogue): over ResNet-50's
    # ~60 stacked layers that is a few % at the worst element, ~1 % in the L2 sense
    assert float((folded - ref).abs().max()) <= 8e-2 * scale
    assert float((folded - ref).norm() / ref.norm()) <= 2e-2


@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("Cc,H,W,k,s,p,d", [(32, 12, 14, 3, 1, 1, 1), (48, 11, 9, 3, 2, 1, 1), (64, 16, 16, 3, 1, 6, 6), (24, 9, 9, 5, 1, 2, 1), (128, 40, 48, 3, 1, 1, 1)])
def test_depthwise_fprop_with_fused_activation(Cc, H, W, k, s, p, d, act):
    """cvhip_dwconv2d_fprop_act == act(depthwise conv + bias) on every depthwise forward kernel (register window, LDS ring, taps in
    registers, generic)"""
    import ctypes as C
    import torch.nn.functional as F
    from cvpytorch_amd import lib as L, ops
    g = torch.Generator().manual_seed(Cc + k)
    x = torch.randn(2, Cc, H, W, generator=g).to(torch.bfloat16).float()
    w = torch.randn(Cc, 1, k, k, generator=g) / k
    b = torch.randn(Cc, generator=g) * 0.3
    y = F.conv2d(x, w, b, stride=s, padding=p, dilation=d, groups=Cc)
    ref = {1: torch.relu(y), 2: y * torch.sigmoid(y), 3: torch.where(y > 0, y, 0.1 * y)}[act]
    xd = x.to(dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    P, Q = ref.shape[2:]
    out = ops.empty_nhwc(2, Cc, P, Q, dev())
    desc = ops.conv_desc(2, Cc, H, W, Cc, k, k, (s, s), (p, p), (d, d), Cc, Cc, Cc)
    wd, bd = w.reshape(Cc, k, k).contiguous().to(dev()), b.to(dev())
    L.call("cvhip_dwconv2d_fprop_act", C.byref(desc), xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), act, 0.1, out.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert float((got - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max()) + 1e-3
