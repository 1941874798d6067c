"""Pin the oracle (oracle/torch_ref.py) to the reference: replay the golden vectors that
tools/gen_golden.py captured from the reference's own modules (/root/reference, build container only).
CPU-only; same torch build => results are required to match to fp32 round-off (rtol 1e-5) and index
tensors exactly."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import torch_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        parts = k.split("/", 1)
        if len(parts) == 1:
            out[k] = z[k]
        else:
            out.setdefault(parts[0], {})[parts[1]] = z[k]
    return out


def T(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= atol + rtol * scale, "max err %g (scale %g)" % (err, scale)


def lst(d):
    return [T(d[str(i)]) for i in range(len(d))]


def load_state(mod, state):
    sd = {k: T(v) for k, v in state.items()}
    missing, unexpected = mod.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


CONV_ACT = {"k1": dict(type="SiLU"), "k3": dict(type="SiLU"), "k3s2": dict(type="Swish"), "k3s2odd": dict(type="SiLU"),
            "k6s2": dict(type="SiLU"), "k3d2": dict(type="ReLU"), "k1bias": None, "dw3d3": dict(type="ReLU"), "k1s2": None}
CONV_NORM = {"k3d2": dict(type="BN"), "k1bias": None, "dw3d3": dict(type="BN"), "k1s2": dict(type="BN")}
BN_YOLO = dict(type="BN", momentum=0.03, eps=0.001)


def run(mod, inputs, cots):
    inputs = [x.clone().requires_grad_(True) for x in inputs]
    out = mod(*inputs)
    outs = [o for o in (list(out) if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(o)]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    named = [(n, p) for n, p in mod.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, inputs + [p for _, p in named], allow_unused=True)
    gpar = {n: (g if g is not None else torch.zeros_like(p)) for (n, p), g in zip(named, grads[len(inputs):])}
    return outs, grads[:len(inputs)], gpar


@pytest.mark.parametrize("name", sorted(CONV_ACT))
def test_convmodule(name):
    g = load("convmodule_" + name)
    cin, cout, k, s, p, d, grp = [int(v) for v in g["meta"]]
    m = R.ConvModule(cin, cout, k, stride=s, padding=p, dilation=d, groups=grp, norm_cfg=CONV_NORM.get(name, BN_YOLO), act_cfg=CONV_ACT[name])
    load_state(m, g["state"])
    m.train()
    outs, gx, gpar = run(m, [T(g["x"])], [T(g["cot"])])
    close(outs[0], g["out"])
    close(gx[0], g["gx"], rtol=1e-4)
    for n, v in g["gparam"].items():
        close(gpar[n], v, rtol=1e-4)
    for n, v in g.get("state_after", {}).items():
        close(m.state_dict()[n], v)


BLOCKS = {
    "bottleneck": lambda: R.DarknetBottleneck(16, 16, 1.0, True, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "csp": lambda: R.CSPLayer(32, 32, n=2, shortcut=True, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "sppf": lambda: R.SPPF(32, 32, kernel_sizes=5, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "spp": lambda: R.SPPF(32, 32, kernel_sizes=(5, 9, 13), norm_cfg=BN_YOLO, act_cfg=dict(type="Swish")),
    "focus": lambda: R.Focus(3, 16, 3, norm_cfg=BN_YOLO, act_cfg=dict(type="Swish")),
    "up": lambda: R.UpsamplingModule(32, 16, 1, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "down": lambda: R.DownsamplingModule(16, 32, 1, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
}


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_block(name):
    g = load("block_" + name)
    m = BLOCKS[name]()
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    load_state(m, g["state"])
    m.train()
    outs, gx, gpar = run(m, lst(g["x"]), lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e)
    for a, e in zip(gx, lst(g["gx"])):
        close(a, e, rtol=1e-4)
    for n, v in g["gparam"].items():
        close(gpar[n], v, rtol=2e-4)


def test_backbone_v5n_full():
    g = load("backbone_v5n_full")
    m = R.YOLOv5CSPDarknet("cspdark_n")
    load_state(m, g["state"])
    m.train()
    outs, _, gpar = run(m, [T(g["x"])], lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e, rtol=1e-4)
    close(gpar["stem.conv.weight"], g["g_stem"], rtol=1e-3)
    for n, v in g["gparam_norms"].items():
        assert abs(float(gpar[n].norm()) - float(v)) <= 1e-3 * max(1.0, float(v)), n


def test_backbone_v5s_structure():
    """Same parameter names / count as the reference's YOLOv5CSPDarknet('cspdark_s') and same output shapes."""
    g = load("backbone_v5s")
    m = R.YOLOv5CSPDarknet("cspdark_s")
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["state_keys"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"][0])
    m.train()
    feats = m(T(g["x"]))
    for f, e in zip(feats, lst(g["feats"])):
        assert tuple(f.shape) == tuple(e.shape)


def test_detect():
    g = load("detect_v5")
    m = R.YOLOv5Detect(80, in_channels=(256, 512, 1024), width_mul=0.125)
    load_state(m, g["state"])
    m.train()
    _, tr = m(lst(g["x"]))
    for o, e in zip(tr, lst(g["train_out"])):
        close(o, e)
    m.eval()
    z, _ = m(lst(g["x"]))
    close(z, g["z"])


def test_bbox_iou_family():
    g = load("bbox_iou")
    b1, b2 = T(g["b1"]), T(g["b2"])
    close(R.bbox_iou(b1, b2, x1y1x2y2=False), g["iou"])
    close(R.bbox_iou(b1, b2, x1y1x2y2=False, GIoU=True), g["giou"])
    close(R.bbox_iou(b1, b2, x1y1x2y2=False, DIoU=True), g["diou"])
    close(R.bbox_iou(b1, b2, x1y1x2y2=False, CIoU=True), g["ciou"])


def test_bbox_overlaps_known_answer():
    """The one known-answer vector the reference holds for this path: src/losses/det/iou_losses.py:35-52."""
    g = load("bbox_overlaps_kat")
    expect = torch.tensor([[0.5, 0.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 0.0]])
    close(T(g["iou"]), expect, atol=1e-4)
    close(R.box_iou(T(g["b1"]), T(g["b2"])), expect, atol=1e-4)


def test_box_iou_xywh2xyxy():
    g = load("box_iou")
    close(R.box_iou(T(g["a"]), T(g["b"])), g["iou"])
    close(R.xywh2xyxy(T(g["a"])), g["xyxy"])


@pytest.mark.parametrize("trial", [0, 1, 2])
def test_yolov5_loss(trial):
    g = load("yolov5_loss_%d" % trial)
    p = [q.requires_grad_(True) for q in lst(g["p"])]
    loss = R.YOLOv5Loss(80)
    total, stats = loss(p, T(g["targets"]))
    close(total, g["total"])
    close(stats, g["stats"])
    grads = torch.autograd.grad(total, p)
    for a, e in zip(grads, lst(g["grads"])):
        close(a, e, atol=1e-7)
    tcls, tbox, indices, anch = loss.build_targets([q.detach() for q in p], T(g["targets"]))
    for i in range(3):
        assert torch.equal(indices[i][0], T(g["b"][str(i)]))
        assert torch.equal(indices[i][1], T(g["a"][str(i)]))
        assert torch.equal(indices[i][2], T(g["gj"][str(i)]))
        assert torch.equal(indices[i][3], T(g["gi"][str(i)]))
        assert torch.equal(tcls[i], T(g["tcls"][str(i)]))
        close(tbox[i], g["tbox"][str(i)])


def test_golden_files_present():
    assert len(glob.glob(os.path.join(GOLD, "*.npz"))) >= 20


def test_deeplabv3plus_head_and_ce():
    """Reference Deeplabv3PlusHead (heads/seg/deeplabv3plus_head.py) + CrossEntropyLoss2d after the bilinear resize
    (segmentors/encoder_decoder.py:93-107)."""
    import torch.nn.functional as F
    g = load("deeplabv3plus_head")
    m = R.Deeplabv3PlusHead(19, in_channels=64, channels=32, dilations=(1, 2, 3, 4), low_in_channels=16, low_channels=8, dropout_ratio=0)
    load_state(m, g["state"])
    m.train()
    xs = [x.requires_grad_(True) for x in lst(g["x"])]
    logits = m(xs)
    close(logits, g["logits"], rtol=1e-4)
    tgt = T(g["target"])
    up = F.interpolate(logits, size=tgt.shape[-2:], mode="bilinear", align_corners=False)
    loss = torch.nn.CrossEntropyLoss(ignore_index=255)(up, tgt.long())
    close(loss, g["loss"], rtol=1e-5)
    named = [(n, q) for n, q in m.named_parameters()]
    grads = torch.autograd.grad(loss, xs + [q for _, q in named])
    for a, e in zip(grads[:2], lst(g["gx"])):
        close(a, e, rtol=1e-3, atol=1e-8)
    for (n, _), a in zip(named, grads[2:]):
        close(a, g["gparam"][n], rtol=2e-3, atol=1e-8)


# ------------------------------------------------------------------------------------------------------
# YOLOX — fixtures from tools/gen_golden_more.py
# ------------------------------------------------------------------------------------------------------
from oracle import yolox_ref as RX  # noqa: E402


class _ListArg(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, *xs):
        return self.m(list(xs))


def test_yolox_backbone_n_full():
    g = load("yolox_backbone_n")
    m = RX.YOLOXCSPDarknet("cspdark_n")
    load_state(m, g["state"])
    m.train()
    outs, _, gpar = run(m, [T(g["x"])], lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e, rtol=1e-4)
    close(gpar["stem.conv.conv.weight"], g["g_stem"], rtol=1e-3)
    for n, v in g["gparam_norms"].items():
        assert abs(float(gpar[n].norm()) - float(v)) <= 1e-3 * max(1.0, float(v)), n


def test_yolox_head():
    g = load("yolox_head_n")
    m = RX.YOLOXHead("yolox_n", num_classes=80, norm_cfg=BN_YOLO)
    load_state(m, g["state"])
    m.train()
    outs, gx, gpar = run(_ListArg(m), lst(g["x"]), lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e, rtol=1e-4)
    for a, e in zip(gx, lst(g["gx"])):
        close(a, e, rtol=1e-4)
    for n, v in g["gparam"].items():
        close(gpar["m." + n], v, rtol=2e-4)


@pytest.mark.parametrize("trial", [0, 1, 2])
def test_yolox_loss(trial):
    g = load("yolox_loss_%d" % trial)
    p = [q.requires_grad_(True) for q in lst(g["p"])]
    out, assigns = RX.YOLOXLoss(80)(p, T(g["targets"]), return_assign=True)
    for k in ("loss", "conf_loss", "cls_loss", "iou_loss", "num_fg"):
        close(out[k], g[k])
    grads = torch.autograd.grad(out["loss"], p)
    for a, e in zip(grads, lst(g["grads"])):
        close(a, e, atol=1e-7)
    rec = [a for a in assigns if a is not None]
    assert len(rec) == len(g.get("fg", {}))
    for i, (fg, mgt, miou) in enumerate(rec):
        assert torch.equal(fg, T(g["fg"][str(i)]))
        assert torch.equal(mgt, T(g["matched_gt"][str(i)]))
        close(miou, g["matched_iou"][str(i)])


def test_yolox_neck_shapes():
    """YOLOXNeck cannot be built in the reference; the restatement is pinned by the shape contract of
    yolox_neck.py:80-105 — three maps of out_channels at strides 8/16/32."""
    m = RX.YOLOXNeck("yolox_n")
    outs = m([torch.randn(1, 64, 8, 8), torch.randn(1, 128, 4, 4), torch.randn(1, 256, 2, 2)])
    assert [tuple(o.shape) for o in outs] == [(1, 64, 8, 8), (1, 64, 4, 4), (1, 64, 2, 2)]


# ------------------------------------------------------------------------------------------------------
# YOLOv7 blocks
# ------------------------------------------------------------------------------------------------------
from oracle import yolov7_ref as R7  # noqa: E402

V7_BLOCKS = {
    "v7_eelan": lambda: R7.EELAN(16, 8, 32),
    "v7_downa": lambda: R7.DownA(16, 8),
    "v7_downb": lambda: R7.DownB(16, 16),
    "v7_sppcspc": lambda: R7.SPPCSPC(32, 16),
    "v7_upsampling": lambda: R7.UpSampling(16, 24, 8),
    "v7_featurefusion": lambda: R7.FeatureFusion(16, 8),
    "v7_repconv_id": lambda: R7.RepConv(16, 16),
    "v7_repconv": lambda: R7.RepConv(16, 24),
    "v7_neck": lambda: _ListArg(R7.YOLOv7Neck(width_mul=0.0625)),
    "v7_head": lambda: _ListArg(R7.YOLOv7Head(width_mul=0.0625)),
}


def _load_maybe_wrapped(m, state):
    target = m.m if isinstance(m, _ListArg) else m
    load_state(target, state)


@pytest.mark.parametrize("name", sorted(V7_BLOCKS))
def test_v7_block(name):
    g = load(name)
    m = V7_BLOCKS[name]()
    R7._bn_fix(m)
    _load_maybe_wrapped(m, g["state"])
    m.train()
    outs, gx, gpar = run(m, lst(g["x"]), lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e, rtol=1e-4)
    for a, e in zip(gx, lst(g["gx"])):
        close(a, e, rtol=2e-4)
    pre = "m." if isinstance(m, _ListArg) else ""
    for n, v in g["gparam"].items():
        close(gpar[pre + n], v, rtol=5e-4)


def test_v7_detect():
    g = load("v7_detect")
    m = R7.YOLOv7Detect(80, width_mul=0.0625)
    load_state(m, g["state"])
    m.train()
    _, tr = m(lst(g["x"]))
    for o, e in zip(tr, lst(g["train_out"])):
        close(o, e)
    m.eval()
    z, _ = m(lst(g["x"]))
    close(z, g["z"])


# ------------------------------------------------------------------------------------------------------
# STDC
# ------------------------------------------------------------------------------------------------------
from oracle import stdc_ref as RS  # noqa: E402


class _FlatNeck(torch.nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, *xs):
        f, aux = self.m(list(xs))
        return [f] + list(aux[1:])


STDC_BLOCKS = {
    "stdc_cat_s2": lambda: RS.CatBottleneck(16, 32, 4, 2),
    "stdc_cat_s1": lambda: RS.CatBottleneck(32, 32, 4, 1),
    "stdc_add_s2": lambda: RS.AddBottleneck(16, 32, 4, 2),
    "stdc_arm": lambda: RS.AttentionRefinementModule(32, 16),
    "stdc_ffm": lambda: RS.FeatureFusionModule(48, 32),
    "stdc_neck": lambda: _FlatNeck(RS.STDCNeck(in_channels=[32, 64, 128], out_channels=32, aux_out_channels=16)),
}


@pytest.mark.parametrize("name", sorted(STDC_BLOCKS))
def test_stdc_block(name):
    g = load(name)
    m = STDC_BLOCKS[name]()
    load_state(m.m if isinstance(m, _FlatNeck) else m, g["state"])
    m.train()
    outs, gx, gpar = run(m, lst(g["x"]), lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e, rtol=1e-4)
    for a, e in zip(gx, lst(g["gx"])):
        close(a, e, rtol=2e-4)
    pre = "m." if isinstance(m, _FlatNeck) else ""
    for n, v in g["gparam"].items():
        close(gpar[pre + n], v, rtol=5e-4)


def test_stdc_net_small_and_structure():
    g = load("stdc_net_small")
    m = RS.STDCNet("stdc1", out_channels=[8, 16, 64, 128, 256], layers=[2, 2, 2], block_num=4)
    load_state(m, g["state"])
    m.train()
    outs, _, gpar = run(m, [T(g["x"])], lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e, rtol=1e-4)
    close(gpar["stem.conv.weight"], g["g_stem"], rtol=1e-3)
    for n, v in g["gparam_norms"].items():
        assert abs(float(gpar[n].norm()) - float(v)) <= 1e-3 * max(1.0, float(v)), n
    s = load("stdc1_structure")
    full = RS.STDCNet("stdc1")
    assert sorted(full.state_dict().keys()) == [str(k) for k in s["state_keys"]]
    assert sum(p.numel() for p in full.parameters()) == int(s["n_params"][0])
    full.train()
    feats = full(torch.randn(1, 3, 64, 128))
    assert [list(f.shape) for f in feats] == [list(map(int, r)) for r in s["shapes"]]


@pytest.mark.parametrize("trial", [0, 1])
def test_v7_ota_loss(trial):
    """YOLOv7Loss with OTA assignment (src/losses/yolov7_loss.py:129-420): totals, gradients and the per-level matched lists
    (image, anchor, gj, gi, class, box) must equal the reference's."""
    g = load("v7_ota_loss_%d" % trial)
    p = [q.requires_grad_(True) for q in lst(g["p"])]
    size = int(g["size"][0])
    imgs = torch.zeros(p[0].shape[0], 3, size, size)
    (total, stats), (bs, as_, gjs, gis, tg) = R7.YOLOv7OTALoss(80)(p, T(g["targets"]), imgs, return_assign=True)
    close(total, g["total"])
    close(stats, g["stats"])
    grads = torch.autograd.grad(total, p)
    for a, e in zip(grads, lst(g["grads"])):
        close(a, e, atol=1e-7)
    for i in range(3):
        assert torch.equal(bs[i], T(g["b"][str(i)]))
        assert torch.equal(as_[i], T(g["a"][str(i)]))
        assert torch.equal(gjs[i], T(g["gj"][str(i)]))
        assert torch.equal(gis[i], T(g["gi"][str(i)]))
        assert torch.equal(tg[i][:, 1], T(g["tcls"][str(i)]))
        close(tg[i][:, 2:6], g["tbox"][str(i)])


@pytest.mark.parametrize("name,kw", [("cspdarknet_n", dict(subtype="cspdark_n")), ("cspdarknet_n_dw", dict(subtype="cspdark_n", depthwise=True))])
def test_generic_cspdarknet(name, kw):
    """oracle restatement of src/models/backbones/det/csp_darknet.py:25-103 == the reference's own class (outputs, stem gradient,
    per-parameter gradient norms), plain and depthwise."""
    g = load(name)
    m = RX.CSPDarknet(**kw)
    load_state(m, g["state"])
    assert list(m.out_channels) == [int(v) for v in g["out_channels"]]
    m.train()
    outs, _, gpar = run(m, [T(g["x"])], lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        close(o, e, rtol=1e-4)
    close(gpar["stem.conv.conv.weight"], g["g_stem"], rtol=1e-3)
    for n, v in g["gparam_norms"].items():
        assert abs(float(gpar[n].norm()) - float(v)) <= 1e-3 * max(1.0, float(v)), n
    assert RX.CSPDarknet("cspdark_t").out_channels == [96, 192, 384]   # the 0.375-width 't' variant only this class has
