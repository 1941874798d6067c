"""Deploy-path parameter arithmetic (cvpytorch_amd/deploy.py) checked on CPU against plain torch modules:
conv+BN folding (src/utils/fuse.py:32-54) and RepConv re-parameterisation (yolov7_modules.py:215-300). Only the folded
PARAMETERS are compared here (as fp32 convs evaluated by torch); the HIP forward of the folded modules is a GPU test."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from cvpytorch_amd import deploy, yolov7
from oracle import yolov7_ref as R7


def _rand_bn(bn, g):
    with torch.no_grad():
        bn.weight.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(bn.num_features, generator=g) * 0.3)
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.2)
        bn.running_var.copy_(torch.rand(bn.num_features, generator=g) + 0.3)


def test_fuse_conv_and_bn_matches_eval_conv_bn():
    g = torch.Generator().manual_seed(0)
    for (cin, cout, k, s, p, d, grp, bias) in [(8, 16, 3, 1, 1, 1, 1, False), (16, 16, 3, 2, 1, 1, 16, False), (8, 24, 1, 1, 0, 1, 1, True), (8, 8, 3, 1, 2, 2, 1, False)]:
        conv = nn.Conv2d(cin, cout, k, s, p, d, grp, bias)
        bn = nn.BatchNorm2d(cout)
        _rand_bn(bn, g)
        bn.eval()
        x = torch.randn(2, cin, 9, 10, generator=g)
        ref = bn(conv(x))
        f = deploy.fuse_conv_and_bn(conv, bn)
        got = F.conv2d(x, f.weight, f.bias, s, p, d, grp)
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)
        assert f.weight.is_contiguous(memory_format=torch.channels_last) or cin // grp == 1 or k == 1


def test_repconv_equivalent_kernel_bias():
    g = torch.Generator().manual_seed(1)
    for (c1, c2) in [(16, 16), (16, 24)]:
        ref = R7.RepConv(c1, c2)
        for m in ref.modules():
            if isinstance(m, nn.BatchNorm2d):
                _rand_bn(m, g)
        ref.eval()
        hip = yolov7.RepConv(c1, c2)
        hip.load_state_dict(ref.state_dict())
        hip.eval()
        k, b = deploy.repconv_equivalent_kernel_bias(hip)
        x = torch.randn(2, c1, 7, 9, generator=g)
        got = F.silu(F.conv2d(x, k, b, 1, 1))
        assert torch.allclose(got, ref(x), rtol=1e-4, atol=1e-5)
