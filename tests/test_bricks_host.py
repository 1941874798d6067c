"""Host-side (CPU) checks of the drop-in boundary: registry API mirrors the reference's
(src/models/bricks/*.py, src/utils/registry.py), Hip modules keep the reference's parameter names, and
the product path fails loudly off-GPU instead of falling back."""
import pytest
import torch
import torch.nn as nn

from cvpytorch_amd import bricks, lib, ops, yolov5
from cvpytorch_amd.train import ModelEMA, build_param_groups
from oracle import torch_ref as R


def test_registries_and_builders():
    assert "HipConv2d" in bricks.CONV_LAYERS and "Conv2d" in bricks.CONV_LAYERS
    assert "HipBN" in bricks.NORM_LAYERS and "HipSiLU" in bricks.ACTIVATION_LAYERS and "HipConvModule" in bricks.PLUGIN_LAYERS
    c = bricks.build_conv_layer(dict(type="HipConv2d"), 8, 16, 3, stride=2, padding=1, bias=False)
    assert isinstance(c, bricks.HipConv2d) and isinstance(c, nn.Conv2d) and c.weight.shape == (16, 8, 3, 3) and c.bias is None
    assert type(bricks.build_conv_layer(None, 8, 16, 1)) is nn.Conv2d  # cfg None => 'Conv2d' (bricks/conv.py:29-30)
    name, bn = bricks.build_norm_layer(dict(type="HipBN", momentum=0.03, eps=0.001), 16)
    assert name == "bn" and isinstance(bn, nn.BatchNorm2d) and bn.eps == 0.001 and bn.momentum == 0.03
    name, bn = bricks.build_norm_layer(dict(type="HipBN", requires_grad=False), 16, postfix=2)
    assert name == "bn2" and not bn.weight.requires_grad and bn.eps == 1e-5
    assert isinstance(bricks.build_activation_layer(dict(type="HipSiLU", inplace=True)), nn.SiLU)
    with pytest.raises(KeyError):
        bricks.build_conv_layer(dict(type="NoSuchConv"), 1, 1, 1)
    with pytest.raises(KeyError):
        bricks.build_norm_layer(dict(momentum=0.1), 4)
    with pytest.raises(TypeError):
        bricks.build_conv_layer("HipConv2d", 1, 1, 1)
    with pytest.raises(KeyError):
        bricks.CONV_LAYERS.register_module("HipConv2d", module=bricks.HipConv2d)  # duplicate without force
    bricks.CONV_LAYERS.register_module("HipConv2d", force=True, module=bricks.HipConv2d)

    @bricks.ACTIVATION_LAYERS.register_module(name="_TestAct", force=True)
    class _TestAct(nn.Module):
        pass
    assert bricks.ACTIVATION_LAYERS.get("_TestAct") is _TestAct
    nm, layer = bricks.build_plugin_layer(dict(type="HipConvModule", in_channels=8, out_channels=8, kernel_size=1), postfix="_x")
    assert nm == "conv_block_x" and isinstance(layer, bricks.HipConvModule)


def test_convmodule_attribute_surface():
    m = bricks.HipConvModule(8, 16, 3, stride=2, padding=1, norm_cfg=dict(type="BN", momentum=0.03, eps=0.001), act_cfg=dict(type="SiLU", inplace=True))
    assert (m.in_channels, m.out_channels, m.kernel_size, m.stride, m.padding, m.dilation, m.groups) == (8, 16, (3, 3), (2, 2), 1, (1, 1), 1)
    assert m.with_norm and m.with_act and not m.with_bias and m.norm_name == "bn" and m.norm is m.bn
    assert sorted(m.state_dict()) == sorted(R.ConvModule(8, 16, 3, 2, 1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")).state_dict())
    assert isinstance(m.conv, bricks.HipConv2d) and isinstance(m.bn, bricks.HipBN) and isinstance(m.act, bricks.HipSiLU)
    assert float(m.bn.weight.min()) == 1.0 and float(m.bn.bias.abs().max()) == 0.0
    nb = bricks.HipConvModule(8, 16, 1, norm_cfg=None, act_cfg=None)
    assert nb.with_bias and nb.conv.bias is not None  # bias 'auto' rule (conv_module.py:108-110)


def test_yolov5_state_dict_matches_reference_layout():
    ref, hip = R.YOLOv5(80, "s"), yolov5.YOLOv5(80, "s")
    rk = list(ref.state_dict().keys())
    hk = [k for k in hip.state_dict().keys() if not k.startswith("loss.")]
    assert rk == hk
    assert sum(p.numel() for p in hip.parameters()) == 7235389  # SURVEY.md §2.3 / BASELINE.md §2
    hip.load_state_dict(ref.state_dict(), strict=False)
    for k, v in ref.state_dict().items():
        assert torch.equal(hip.state_dict()[k], v)


def test_convert_to_hip_swaps_and_shares_parameters():
    ref = R.UpsamplingModule(32, 16, 1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU"))
    hip = bricks.convert_to_hip(ref)
    kinds = {type(m) for m in hip.modules()}
    assert bricks.HipConv2d in kinds and bricks.HipBN in kinds and bricks.HipSiLU in kinds and bricks.HipUpsampleNearest2x in kinds
    assert nn.Conv2d not in kinds and nn.BatchNorm2d not in kinds
    assert hip.conv.conv.weight is ref.conv.conv.weight


def test_no_cpu_fallback():
    m = bricks.HipConvModule(8, 8, 1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU"))
    with pytest.raises(lib.CvhipError):
        m(torch.zeros(1, 8, 4, 4))
    with pytest.raises(lib.CvhipError):
        ops.cat([torch.zeros(1, 8, 2, 2, dtype=torch.bfloat16)])
    with pytest.raises(lib.CvhipError):
        ops.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)


def test_param_groups_follow_reference_rules():
    """src/optimizers/__init__.py:36-56: bias + norm weights undecayed, conv weights decayed."""
    m = yolov5.YOLOv5(80, "s")
    groups = build_param_groups(m, lr=0.01, weight_decay=5e-4)
    ids = set()
    for g in groups:
        for p in g["params"]:
            assert id(p) not in ids
            ids.add(id(p))
    assert ids == {id(p) for p in m.parameters()}
    decayed = sum(p.numel() for g in groups if g["weight_decay"] > 0 for p in g["params"])
    conv_w = sum(mm.weight.numel() for mm in m.modules() if isinstance(mm, nn.Conv2d))
    assert decayed == conv_w


def test_nhwc_view_detection():
    x = torch.zeros(2, 16, 4, 5, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    assert ops.nhwc_ld(x) == 16
    assert ops.nhwc_ld(x[:, 8:]) == 16 and ops.nhwc_ld(x[:, :8]) == 16
    assert ops.nhwc_ld(torch.zeros(2, 16, 4, 5, dtype=torch.bfloat16)) is None
    assert ops.nhwc_ld(x.float()) is None
    assert ops.nhwc_ld(torch.zeros(3, 8, 1, 1, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)) == 8


def test_ema_matches_reference_formula():
    """src/utils/ema.py:30-39 on CPU tensors (the EMA is device-agnostic torch code)."""
    import math
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    ema = ModelEMA(m)
    before = {k: v.clone() for k, v in ema.ema.state_dict().items()}
    with torch.no_grad():
        for p in m.parameters():
            p.add_(1.0)
    ema.update(m)
    d = 0.9999 * (1 - math.exp(-1 / 2000))
    for k, v in ema.ema.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.allclose(v, before[k] * d + (1 - d) * m.state_dict()[k])


def test_deeplab_state_dict_matches_oracle_layout():
    from cvpytorch_amd import deeplab
    ref, hip = R.EncoderDecoder(19), deeplab.EncoderDecoder(19)
    assert list(ref.state_dict().keys()) == list(hip.state_dict().keys())
    assert sum(p.numel() for p in hip.parameters()) == 41225187  # 41.23 M
    hip.load_state_dict(ref.state_dict())
    r8 = R.EncoderDecoder(19, output_stride=8)
    assert [tuple(v.shape) for v in r8.state_dict().values()] == [tuple(v.shape) for v in deeplab.EncoderDecoder(19, output_stride=8).state_dict().values()]


def test_convert_to_hip_skips_what_the_engine_refuses(caplog):
    """VERDICT r04 task 9: convert_to_hip leaves layers the engine cannot run (non-zero padding modes, string padding, grouped but not
    depthwise convolutions) in place — as convert_sync_batchnorm leaves foreign modules (trainer.py:127) — with one logged line each,
    and converts their neighbours; parameters stay shared, state_dict keys unchanged."""
    import logging
    import torch.nn as nn
    from cvpytorch_amd import bricks
    m = nn.Sequential(nn.Conv2d(8, 8, 3, padding=1), nn.Conv2d(8, 8, 3, padding=1, padding_mode="reflect"), nn.Conv2d(8, 8, 3, padding="same"),
                      nn.Conv2d(8, 16, 3, padding=1, groups=2), nn.Conv2d(16, 16, 3, padding=1, groups=16), nn.BatchNorm2d(16), nn.SiLU())
    keys = list(m.state_dict().keys())
    w1 = m[1].weight
    with caplog.at_level(logging.WARNING, logger="cvpytorch_amd"):
        out = bricks.convert_to_hip(m)
    assert isinstance(out[0], bricks.HipConv2d) and isinstance(out[4], bricks.HipConv2d)           # dense and depthwise: converted
    for i in (1, 2, 3):
        assert type(out[i]) is nn.Conv2d                                                            # refused geometries: untouched
    assert out[1].weight is w1
    assert isinstance(out[5], bricks.HipBN) and isinstance(out[6], bricks.HipSiLU)
    assert list(out.state_dict().keys()) == keys
    msgs = [r.getMessage() for r in caplog.records if "convert_to_hip" in r.getMessage()]
    assert len(msgs) == 3 and any("reflect" in s for s in msgs) and any("same" in s for s in msgs) and any("groups=2" in s for s in msgs)
    assert bricks.hip_conv_unsupported(nn.Conv2d(8, 8, 1)) is None
