"""Pins the ResNet-50 restatements (oracle/torch_ref.py ResNet50 — the checker of every DeepLabv3+ / Classification parity test — and
cvpytorch_amd/deeplab.py ResNet) to the hand-derived known-answer file tests/golden/resnet50_kat.json (tools/gen_resnet_kat.py):
torchvision's published ResNet-50 — parameter counts 25,557,032 / 23,508,032, block layout [3, 4, 6, 3] x expansion 4, the
state_dict key list and tensor shapes, the v1.5 stride placement (on conv2 and on the projection shortcut). torchvision itself
(reference src/models/backbones/seg/resnet.py:11,52-54) is not installed, so the published definition is the admissible pin.
CPU only: structure, not arithmetic (the arithmetic of each layer type is pinned by the ConvModule / BN fixtures)."""
import json
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "resnet50_kat.json")))


def _models():
    from oracle import torch_ref as R
    from cvpytorch_amd import deeplab
    return {"oracle": lambda **kw: R.ResNet50(**kw), "engine": lambda **kw: deeplab.ResNet(**kw)}


def _canon(key):
    """the seg wrapper names torchvision's conv1 / bn1 `stem.0` / `stem.1` (seg/resnet.py:64-80 builds the stem itself)"""
    return key.replace("stem.0.", "conv1.").replace("stem.1.", "bn1.")


@pytest.mark.parametrize("which", ["oracle", "engine"])
def test_plain_resnet50_is_torchvisions_resnet50(which):
    m = _models()[which](subtype="resnet50", classifier=True, num_classes=1000)
    sd = {_canon(k): v for k, v in m.state_dict().items()}
    assert sorted(sd) == sorted(KAT["state_dict_keys"])
    for k, shp in KAT["shapes"].items():
        assert list(sd[k].shape) == shp, k
    n_all = sum(p.numel() for p in m.parameters())
    n_fc = sum(p.numel() for n, p in m.named_parameters() if n.startswith("fc."))
    assert n_all == KAT["total_with_fc_1000"] == 25557032          # the published figure
    assert n_all - n_fc == KAT["total_without_fc"] == 23508032
    for li in range(1, 5):
        layer = getattr(m, "layer%d" % li)
        assert len(layer) == KAT["blocks"][li - 1]
        assert sum(p.numel() for p in layer.parameters()) == KAT["layer_params"]["layer%d" % li]
    mods = dict(m.named_modules())
    for name, s in KAT["conv_strides"].items():
        conv = mods[name]
        assert tuple(conv.stride) == (s, s), name                  # v1.5: stride on the 3x3 and on the projection
        assert conv.bias is None, name
    assert all(type(b).expansion == KAT["expansion"] for li in range(1, 5) for b in getattr(m, "layer%d" % li))


@pytest.mark.parametrize("which", ["oracle", "engine"])
def test_v1c_backbone_differs_only_in_the_stem(which):
    """the benchmark's backbone (resnet50v1c, out_stages [1, 4]): layers 1-4 are the pinned ones, the deep stem is the reference's own
    (seg/resnet.py:67-80: 3 -> 32 -> 32 -> 64, 3x3)"""
    m = _models()[which](subtype="resnet50v1c", out_stages=(1, 4))
    sd = m.state_dict()
    body = sorted(k for k in sd if k.startswith("layer"))
    assert body == sorted(k for k in KAT["state_dict_keys"] if k.startswith("layer"))
    for k in body:
        assert list(sd[k].shape) == KAT["shapes"][k], k
    stem = sum(p.numel() for n, p in m.named_parameters() if n.startswith("stem."))
    assert stem == 3 * 32 * 9 + 64 + 32 * 32 * 9 + 64 + 32 * 64 * 9 + 128
    assert sum(p.numel() for p in m.parameters()) == stem + sum(KAT["layer_params"].values())


def test_oracle_and_engine_agree_on_dilation_rewrites():
    """output_stride 16 / 8 (torchvision's replace_stride_with_dilation): same strides / dilations / paddings in both restatements"""
    from oracle import torch_ref as R
    from cvpytorch_amd import deeplab
    for os_ in (32, 16, 8):
        a = R.ResNet50(subtype="resnet50v1c", output_stride=os_)
        b = deeplab.ResNet(subtype="resnet50v1c", output_stride=os_)
        ca = {n: (tuple(m.stride), tuple(m.dilation), tuple(m.padding)) for n, m in a.named_modules() if isinstance(m, torch.nn.Conv2d)}
        cb = {n: (tuple(m.stride), tuple(m.dilation), tuple(m.padding)) for n, m in b.named_modules() if isinstance(m, torch.nn.Conv2d)}
        assert ca == cb, os_
