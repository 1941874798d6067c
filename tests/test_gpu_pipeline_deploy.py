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
