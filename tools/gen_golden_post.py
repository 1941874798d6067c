"""Reference-run fixtures for the post-processing loops and the seg ResNet / EncoderDecoder wrappers (VERDICT r04 task 4).

Build container only (needs /root/reference; never runs on the GPU box). The reference's OWN functions are imported and executed:

  src.models.yolov5.non_max_suppression        (models/yolov5.py:62-153)
  src.models.yolox.yolox_post_process          (models/yolox.py:18-68)
  src.models.modules.nms.multiclass_nms / batched_nms   (modules/nms.py:5-132)
  src.models.backbones.seg.resnet.ResNet       (backbones/seg/resnet.py:27-154: deep stem, out_stages, the dilation rewrite)
  src.models.segmentors.encoder_decoder.EncoderDecoder  (segmentors/encoder_decoder.py:21-150: backbone -> head -> resize -> CE)

Their only missing symbols are third-party: `torchvision.ops.nms` / `batched_nms` and `torchvision.models.resnet.resnet50`
(torchvision is neither vendored in the reference nor installed here). Those — and only those — are supplied by the
restatements that are already pinned by hand-derived known-answer vectors: oracle.torch_ref.nms (tests/golden/nms_kat.json),
torchvision's batched_nms strategy (offset boxes by class * (max coordinate + 1), then nms) and the torchvision ResNet-50 topology
(tests/golden/resnet50_kat.json). Everything else that runs is the reference's code.

    python tools/gen_golden_post.py        # writes tests/golden/post_*.npz, seg_resnet_wrapper.npz, seg_encoder_decoder.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gen_golden as GG  # noqa: E402  (stub finder, save helpers)

from oracle import torch_ref as R  # noqa: E402

OUT = GG.OUT


def tv_batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms (third-party; torchvision/ops/boxes.py at the pinned 0.7): class-offset strategy."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + 1)
    return R.nms(boxes + offsets[:, None], scores, iou_threshold)


class _TvResNet50(torch.nn.Module):
    """what `torchvision.models.resnet.resnet50()` returns, as far as the reference touches it (backbones/seg/resnet.py:52-100):
    conv1 / bn1 / relu / maxpool / layer1..4 / avgpool / fc — built from the KAT-pinned restatement of the torchvision topology"""

    def __init__(self):
        super().__init__()
        o = R.ResNet50("resnet50", out_stages=(1, 2, 3, 4), output_stride=32, classifier=True, num_classes=1000)
        self.conv1, self.bn1, self.relu = o.stem[0], o.stem[1], o.stem[2]
        self.maxpool = o.maxpool
        self.layer1, self.layer2, self.layer3, self.layer4 = o.layer1, o.layer2, o.layer3, o.layer4
        self.avgpool, self.fc = o.avgpool, o.fc


def install_third_party():
    GG.install()
    import torchvision  # the stub package
    import torchvision.ops
    torchvision.ops.nms = R.nms
    torchvision.ops.batched_nms = tv_batched_nms
    tvr = GG._StubModule("torchvision.models.resnet")   # (every other name the package imports from it stays an uncalled stub)
    tvr.__path__ = []

    def _no(*a, **k):
        raise RuntimeError("only resnet50 is restated")

    tvr.resnet50 = lambda pretrained=False, **k: _TvResNet50()
    tvr.resnet18 = tvr.resnet34 = tvr.resnet101 = tvr.resnet152 = _no
    import torchvision.models
    torchvision.models.resnet = tvr
    sys.modules["torchvision.models.resnet"] = tvr
    import torch.hub
    torch.hub.load_state_dict_from_url = lambda *a, **k: {"state_dict": {}}   # no network: the "pretrained" file holds nothing to copy


def synthetic_pred(B, n, nc, seed, hot=0.06, clusters=40):
    """decoded YOLOv5-style rows (cx, cy, w, h, obj, cls...): clustered boxes so NMS has work, quantised scores so ties occur"""
    g = torch.Generator().manual_seed(seed)
    ctr = torch.rand(B, clusters, 2, generator=g) * 600 + 20
    which = torch.randint(0, clusters, (B, n), generator=g)
    cxy = torch.gather(ctr, 1, which[..., None].expand(B, n, 2)) + torch.randn(B, n, 2, generator=g) * 6
    wh = torch.rand(B, n, 2, generator=g) * 80 + 10
    obj = torch.rand(B, n, 1, generator=g) * 0.2
    hotm = torch.rand(B, n, 1, generator=g) < hot
    obj = torch.where(hotm, 0.3 + 0.7 * torch.rand(B, n, 1, generator=g), obj)
    cls = torch.rand(B, n, nc, generator=g) ** 3
    obj[:, ::7] = (obj[:, ::7] * 16).round() / 16
    cls[:, ::5] = (cls[:, ::5] * 8).round() / 8
    return torch.cat([cxy, wh, obj, cls], -1)


def pack_list(lst, width):
    """list of (k_i, width) tensors / None -> (sum k, width) array + counts (-1 = None)"""
    counts = np.array([(-1 if t is None else int(t.shape[0])) for t in lst], dtype=np.int64)
    rows = [GG.npy(t).reshape(-1, width) for t in lst if t is not None and t.shape[0]]
    flat = np.concatenate(rows, 0) if rows else np.zeros((0, width), np.float32)
    return flat.astype(np.float32), counts


def gen_nms_v5():
    from src.models import yolov5 as RY
    cases = [
        # name, B, n, nc, seed, hot, conf, iou, classes, agnostic, multi_label, max_det
        ("best", 4, 900, 12, 3, 0.08, 0.25, 0.45, None, False, False, 300),
        ("multi", 4, 900, 12, 4, 0.08, 0.25, 0.45, None, False, True, 300),
        ("agnostic", 3, 900, 6, 5, 0.10, 0.25, 0.45, None, True, False, 300),
        ("classes", 3, 900, 8, 6, 0.10, 0.25, 0.45, [1, 3, 6], False, False, 300),
        # class ids outside [0, nc) match nothing in the reference (yolov5.py:118-119 compares x[:, 5:6] == classes): ADVICE r05
        ("classes_oob", 3, 900, 8, 6, 0.10, 0.25, 0.45, [8, 3, -1, 6], False, False, 300),
        ("classes_oob_multi", 2, 700, 8, 11, 0.10, 0.25, 0.45, [8, 0, -2], False, True, 300),
        ("maxdet", 2, 1500, 4, 7, 0.60, 0.25, 0.70, None, False, False, 20),
        ("val_thresholds", 2, 500, 5, 8, 0.30, 0.001, 0.6, None, False, True, 300),
        ("single_class", 2, 600, 1, 9, 0.20, 0.25, 0.45, None, False, True, 300),
    ]
    only = os.environ.get("GEN_ONLY")   # regenerate a subset (new cases) without touching the committed fixtures
    for (name, B, n, nc, seed, hot, conf, iou, classes, agn, ml, max_det) in cases:
        if only and name not in only.split(","):
            continue
        pred = synthetic_pred(B, n, nc, seed, hot=hot, clusters=60 if name != "maxdet" else 500)
        pred[1, :, 4] = 0.0   # an image without any candidate
        out = RY.non_max_suppression(pred.clone(), conf, iou, classes=classes, agnostic=agn, multi_label=ml, max_det=max_det)
        flat, counts = pack_list(out, 6)
        GG.save("post_nms_v5_" + name, pred=GG.npy(pred), out=flat, counts=counts,
                cfg=np.array([conf, iou, float(agn), float(ml), float(max_det)], np.float64),
                classes=np.array(classes if classes is not None else [], np.int64))


def gen_yolox_post():
    from src.models import yolox as RX
    strides = (8, 16, 32)
    for name, B, nc, seed, conf, thr in [("a", 3, 6, 5, 0.3, 0.5), ("b", 2, 80, 6, 0.25, 0.65), ("c", 2, 1, 7, 0.4, 0.45)]:
        g = torch.Generator().manual_seed(seed)
        hw = [(16, 16), (8, 8), (4, 4)]
        feats = [torch.randn(B, 5 + nc, h, w, generator=g) for h, w in hw]
        for f in feats:
            f[:, 4] += 0.5
            f[:, 2:4] *= 0.5
        for f in feats:
            f[1, 4] = -20.0   # image 1: nothing passes -> None
        out = RX.yolox_post_process([f.clone() for f in feats], strides, nc, conf, thr)
        flat, counts = pack_list(out, 7)
        GG.save("post_yolox_" + name, f0=GG.npy(feats[0]), f1=GG.npy(feats[1]), f2=GG.npy(feats[2]), out=flat, counts=counts,
                cfg=np.array([nc, conf, thr], np.float64), strides=np.array(strides, np.int64))


def gen_mc_nms():
    from src.models.modules import nms as RN
    g = torch.Generator().manual_seed(31)
    n, ncls = 400, 5
    ctr = torch.rand(30, 2, generator=g) * 300
    c = ctr[torch.randint(0, 30, (n,), generator=g)] + torch.randn(n, 2, generator=g) * 4
    wh = torch.rand(n, 2, generator=g) * 40 + 8
    boxes = torch.cat([c - wh / 2, c + wh / 2], 1)
    scores = torch.rand(n, generator=g)
    scores[::9] = (scores[::9] * 8).round() / 8
    idxs = torch.randint(0, ncls, (n,), generator=g)
    # batched_nms: both branches (n < split_thr; per-class split) and class_agnostic
    for name, cfg in [("plain", dict(type="nms", iou_threshold=0.5)), ("split", dict(type="nms", iou_threshold=0.5, split_thr=100)),
                      ("agnostic", dict(type="nms", iou_threshold=0.4, class_agnostic=True))]:
        dets, keep = RN.batched_nms(boxes.clone(), scores.clone(), idxs.clone(), dict(cfg))
        GG.save("post_batched_nms_" + name, boxes=GG.npy(boxes), scores=GG.npy(scores), idxs=GG.npy(idxs), dets=GG.npy(dets), keep=GG.npy(keep),
                iou=np.array([cfg["iou_threshold"], cfg.get("split_thr", 10000), float(cfg.get("class_agnostic", False))], np.float64))
    # multiclass_nms: shared boxes, per-class boxes, score factors, max_num, nothing above the threshold
    ms = torch.rand(n, ncls + 1, generator=g) ** 2
    ms[::6] = (ms[::6] * 8).round() / 8
    mb4 = boxes
    mbc = (boxes[:, None, :] + torch.randn(n, ncls, 4, generator=g) * 1.5).reshape(n, ncls * 4)
    sf = torch.rand(n, generator=g) * 0.5 + 0.5
    for name, mb, thr, cfg, max_num, factors in [
            ("shared", mb4, 0.6, dict(type="nms", iou_threshold=0.45), 50, None),
            ("perclass", mbc, 0.5, dict(type="nms", iou_threshold=0.5), -1, None),
            ("factors", mb4, 0.5, dict(type="nms", iou_threshold=0.5), 100, sf),
            ("agnostic", mb4, 0.5, dict(type="nms", iou_threshold=0.5, class_agnostic=True), 100, None),
            ("empty", mb4, 2.0, dict(type="nms", iou_threshold=0.5), 100, None)]:
        dets, labels = RN.multiclass_nms(mb.clone(), ms.clone(), thr, dict(cfg), max_num=max_num, score_factors=None if factors is None else factors.clone())
        GG.save("post_multiclass_nms_" + name, multi_bboxes=GG.npy(mb), multi_scores=GG.npy(ms), dets=GG.npy(dets).reshape(-1, 5), labels=GG.npy(labels),
                cfg=np.array([thr, cfg["iou_threshold"], float(cfg.get("class_agnostic", False)), float(max_num)], np.float64),
                score_factors=GG.npy(factors) if factors is not None else np.zeros((0,), np.float32))


def _checksums(module):
    """(sum, sum of squares) of every running statistic AFTER the training forward: pins the BatchNorm momentum updates"""
    keys = [k for k in module.state_dict() if k.endswith("running_mean") or k.endswith("running_var")]
    sd = module.state_dict()
    return np.array([[float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in keys], np.float64), keys


def gen_seg_wrappers():
    from src.models.backbones.seg.resnet import ResNet
    from seeded_state import seed_state
    for name, kw in [("os8_cfg", dict(subtype="resnet50v1c", out_stages=[1, 4], output_stride=8, pretrained=True)),
                     ("os16_3stages", dict(subtype="resnet50v1c", out_stages=[2, 3, 4], output_stride=16, pretrained=True)),
                     ("plain_stem", dict(subtype="resnet50", out_stages=[1, 4], output_stride=32, pretrained=False))]:
        m = ResNet(**kw)
        m.train()   # (the reference's train() returns None)
        sig = seed_state(m, 17)          # the state is a function of (seed, key names): the fixture need not carry 100 MB of weights
        g = torch.Generator().manual_seed(29)
        x = torch.randn(2, 3, 64, 96, generator=g).requires_grad_(True)
        # eval-mode outputs first (BatchNorm on the seeded running statistics): the well-conditioned view of the same structure, which a
        # 16-bit engine can be held to at the deepest stages (train-mode BN over 2 x 2 x 3 values amplifies storage rounding chaotically)
        m.eval()
        with torch.no_grad():
            ev = m(x.detach())
        ev = ev if isinstance(ev, (list, tuple)) else [ev]
        m.train()
        outs = m(x)
        outs = outs if isinstance(outs, (list, tuple)) else [outs]
        loss = sum((o.float() ** 2).mean() for o in outs)
        loss.backward()
        arrs = {"x": GG.npy(x), "dx": GG.npy(x.grad), "loss": np.array(float(loss.detach())), "state_seed": np.array(17)}
        for i, o in enumerate(outs):
            arrs["out%d" % i] = GG.npy(o)
        for i, o in enumerate(ev):
            arrs["eval_out%d" % i] = GG.npy(o)
        arrs["state_sig"] = np.array(sig)
        cs, keys = _checksums(m)
        arrs["running_checksums"], arrs["running_keys"] = cs, np.array(keys)
        named = dict(m.named_parameters())
        for k in ("stem.0.weight", "stem.6.weight", "stem.7.weight", "layer1.0.conv1.weight", "layer1.0.downsample.0.weight", "layer2.3.bn1.weight",
                  "layer3.5.bn2.bias", "layer4.2.bn3.weight"):
            if k in named and named[k].grad is not None:
                arrs["grad." + k] = GG.npy(named[k].grad)
        arrs["out_channels"] = np.array(m.out_channels, np.int64)
        # the (non-)dilation rewrite as built: (stride, dilation, padding, downsample stride) of every conv2 in layer3 / layer4
        geo = []
        for ln in ("layer3", "layer4"):
            for b in getattr(m, ln):
                geo.append([b.conv2.stride[0], b.conv2.dilation[0], b.conv2.padding[0], b.downsample[0].stride[0] if b.downsample is not None else 0])
        arrs["conv2_geometry"] = np.array(geo, np.int64)
        GG.save("seg_resnet_wrapper_" + name, **arrs)


def gen_encoder_decoder():
    from src.models.segmentors.encoder_decoder import EncoderDecoder
    from src.utils.config import CommonConfiguration
    from seeded_state import seed_state
    cfg = {"BACKBONE": {"name": "ResNet", "subtype": "resnet50v1c", "out_stages": [1, 4], "output_stride": 8, "pretrained": True},
           "NECK": None,
           "HEAD": {"name": "Deeplabv3PlusHead", "num_classes": 19, "in_channels": 2048, "channels": 512, "dilations": [1, 12, 24, 36],
                    "low_in_channels": 256, "low_channels": 48},
           "AUX_HEAD": None, "LOSS": {"name": "CrossEntropyLoss2d"}, "AUX_LOSS": None}
    try:
        model_cfg = CommonConfiguration.from_dict(cfg) if hasattr(CommonConfiguration, "from_dict") else CommonConfiguration(cfg)
    except Exception:
        model_cfg = types.SimpleNamespace(**cfg)
    dictionary = [{"c%d" % i: 1.0} for i in range(19)]
    m = EncoderDecoder(dictionary, model_cfg)
    m.train()
    sig = seed_state(m, 23)
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout2d, torch.nn.Dropout)):
            mod.p = 0.0   # (dropout draws from the global generator: off, so that the fixture is a function of its inputs)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 64, 128, generator=g).requires_grad_(True)
    tgt = torch.randint(0, 19, (2, 64, 128), generator=g)
    tgt[:, :4] = 255
    m.eval()
    with torch.no_grad():
        eval_logits = m.head(m.backbone(x.detach()))      # (2, 19, 16, 32): eval-mode BatchNorm on the seeded running statistics
    m.train()
    losses = m(x, tgt, mode="train")
    losses["loss"].backward()
    arrs = {"x": GG.npy(x), "target": GG.npy(tgt), "eval_logits": GG.npy(eval_logits), "dx": GG.npy(x.grad), "loss": np.array(float(losses["loss"].detach())), "state_seed": np.array(23)}
    arrs["loss_keys"] = np.array(sorted(losses.keys()))
    arrs["loss_values"] = np.array([float(losses[k].detach()) for k in sorted(losses.keys())], np.float64)
    cs, keys = _checksums(m)
    arrs["running_checksums"], arrs["running_keys"] = cs, np.array(keys)
    arrs["state_sig"] = np.array(sig)
    named = dict(m.named_parameters())
    small = [k for k in named if named[k].grad is not None and named[k].numel() <= 40000]
    pick = [k for k in small if k.startswith("backbone.stem")][:2] + [k for k in small if k.startswith("head.")][:6]
    for k in pick:
        arrs["grad." + k] = GG.npy(named[k].grad)
    m.eval()
    with torch.no_grad():
        arrs["val_argmax"] = GG.npy(m(x.detach(), tgt, mode="val")).astype(np.int16)
    GG.save("seg_encoder_decoder", **arrs)


def main():
    install_third_party()
    torch.set_num_threads(4)
    if "--seg-only" not in sys.argv:
        gen_nms_v5()
        gen_yolox_post()
        gen_mc_nms()
    gen_seg_wrappers()
    gen_encoder_decoder()


if __name__ == "__main__":
    main()
