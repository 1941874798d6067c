"""Pin the oracle's POST-PROCESSING loops and seg ResNet / EncoderDecoder wrappers to the reference (VERDICT r04 task 4, SURVEY §8
rows a9 / a16): replay of the fixtures that tools/gen_golden_post.py captured from the reference's OWN functions
(src/models/yolov5.py:62-153, src/models/yolox.py:18-68, src/models/modules/nms.py:5-132, src/models/backbones/seg/resnet.py:27-154,
src/models/segmentors/encoder_decoder.py:21-150) with only their third-party symbols (torchvision nms / batched_nms / resnet50)
supplied by the KAT-pinned restatements. CPU only. Index / box / score rows: bit-exact; floating-point network outputs: rtol 1e-5."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import nms_ref as RN
from oracle import torch_ref as R
from oracle import yolox_ref as RX
from seeded_state import seed_state

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def T(a):
    return torch.from_numpy(np.asarray(a))


def split_rows(flat, counts):
    out, o = [], 0
    for c in counts.tolist():
        if c < 0:
            out.append(None)
        else:
            out.append(T(flat[o:o + c]))
            o += c
    return out


def close(a, b, rtol=1e-5, atol=1e-6):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    scale = max(1.0, float(b.abs().max())) if b.numel() else 1.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    assert err <= atol + rtol * scale, "max err %g (scale %g)" % (err, scale)


NMS_V5 = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(GOLD, "post_nms_v5_*.npz")))


def test_fixture_inventory():
    assert len(NMS_V5) == 9
    assert len(glob.glob(os.path.join(GOLD, "post_yolox_*.npz"))) == 3
    assert len(glob.glob(os.path.join(GOLD, "post_batched_nms_*.npz"))) == 3
    assert len(glob.glob(os.path.join(GOLD, "post_multiclass_nms_*.npz"))) == 5
    assert len(glob.glob(os.path.join(GOLD, "seg_resnet_wrapper_*.npz"))) == 3


@pytest.mark.parametrize("name", NMS_V5)
def test_non_max_suppression_equals_reference(name):
    z = load(name)
    conf, iou, agn, ml, max_det = z["cfg"].tolist()
    classes = z["classes"].tolist() or None
    ref = split_rows(z["out"], z["counts"])
    got = R.non_max_suppression(T(z["pred"]).clone(), conf, iou, classes, bool(agn), bool(ml), int(max_det))
    assert len(got) == len(ref)
    assert sum(int(r.shape[0]) for r in ref) > 0 and int(ref[1].shape[0]) == 0     # the case has work and its empty image
    for a, r in zip(got, ref):
        assert torch.equal(a, r)                                                      # boxes, scores, classes and their order


@pytest.mark.parametrize("name", ["post_yolox_a", "post_yolox_b", "post_yolox_c"])
def test_yolox_post_process_equals_reference(name):
    z = load(name)
    nc, conf, thr = z["cfg"].tolist()
    ref = split_rows(z["out"], z["counts"])
    got = RX.yolox_post_process([T(z["f0"]).clone(), T(z["f1"]).clone(), T(z["f2"]).clone()], tuple(z["strides"].tolist()), int(nc), conf, thr)
    assert ref[1] is None and got[1] is None
    for a, r in zip(got, ref):
        if r is None:
            assert a is None
        else:
            assert torch.equal(a, r)


@pytest.mark.parametrize("name", ["plain", "split", "agnostic"])
def test_batched_nms_equals_reference(name):
    z = load("post_batched_nms_" + name)
    thr, split, agn = z["iou"].tolist()
    cfg = dict(type="nms", iou_threshold=thr, split_thr=int(split))
    if agn:
        cfg["class_agnostic"] = True
    dets, keep = RN.batched_nms(T(z["boxes"]), T(z["scores"]), T(z["idxs"]), cfg)
    assert torch.equal(keep, T(z["keep"])) and torch.equal(dets, T(z["dets"]))


@pytest.mark.parametrize("name", ["shared", "perclass", "factors", "agnostic", "empty"])
def test_multiclass_nms_equals_reference(name):
    z = load("post_multiclass_nms_" + name)
    thr, iou, agn, max_num = z["cfg"].tolist()
    cfg = dict(type="nms", iou_threshold=iou)
    if agn:
        cfg["class_agnostic"] = True
    sf = T(z["score_factors"]) if z["score_factors"].size else None
    dets, labels = RN.multiclass_nms(T(z["multi_bboxes"]), T(z["multi_scores"]), thr, cfg, max_num=int(max_num), score_factors=sf)
    assert torch.equal(labels, T(z["labels"])) and torch.equal(dets.reshape(-1, 5), T(z["dets"]))
    if name == "empty":
        assert dets.shape[0] == 0


def _check_running(mod, z):
    sd = mod.state_dict()
    keys = [str(k) for k in z["running_keys"].tolist()]
    cs = np.array([[float(sd[k].double().sum()), float((sd[k].double() ** 2).sum())] for k in keys])
    assert np.allclose(cs, z["running_checksums"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,subtype,stages", [("os8_cfg", "resnet50v1c", (1, 4)), ("os16_3stages", "resnet50v1c", (2, 3, 4)),
                                                 ("plain_stem", "resnet50", (1, 4))])
def test_seg_resnet_wrapper_equals_reference(name, subtype, stages):
    """the reference's wrapper as BUILT: deep stem for *v1c, out_stages, and NO dilation for ResNet-50 whatever output_stride says
    (backbones/seg/resnet.py:102-118 only matches resnet18/34) — the oracle's `output_stride=32`"""
    z = load("seg_resnet_wrapper_" + name)
    geo = z["conv2_geometry"]
    assert geo[:, 1].tolist() == [1] * 9 and geo[:, 2].tolist() == [1] * 9 and geo[0, 0] == 2 and geo[6, 0] == 2   # strides kept, no dilation
    m = R.ResNet50(subtype, out_stages=stages, output_stride=32).train()
    sig = seed_state(m, int(z["state_seed"]))
    assert sig == [str(s) for s in z["state_sig"].tolist()]           # same state_dict keys and shapes as the reference's module
    assert list(z["out_channels"]) == [{1: 256, 2: 512, 3: 1024, 4: 2048}[s] for s in stages]
    x = T(z["x"]).clone().requires_grad_(True)
    m.eval()
    with torch.no_grad():
        ev = m(x.detach())
    for i, o in enumerate(ev if isinstance(ev, (list, tuple)) else [ev]):
        close(o, z["eval_out%d" % i])
    m.train()
    outs = m(x)
    outs = outs if isinstance(outs, (list, tuple)) else [outs]
    loss = sum((o.float() ** 2).mean() for o in outs)
    loss.backward()
    for i, o in enumerate(outs):
        close(o, z["out%d" % i])
    close(loss, float(z["loss"]))
    close(x.grad, z["dx"], rtol=2e-4)
    named = dict(m.named_parameters())
    for k in [k for k in z.files if k.startswith("grad.")]:
        close(named[k[5:]].grad, z[k], rtol=2e-4)
    _check_running(m, z)


def test_encoder_decoder_equals_reference():
    """backbone -> Deeplabv3PlusHead -> bilinear resize to the label size -> CrossEntropyLoss2d (ignore 255), train and val modes"""
    z = load("seg_encoder_decoder")
    m = R.EncoderDecoder(19, output_stride=32).train()
    sig = seed_state(m, int(z["state_seed"]))
    assert sig == [str(s) for s in z["state_sig"].tolist()]
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout2d, torch.nn.Dropout)):
            mod.p = 0.0
    x = T(z["x"]).clone().requires_grad_(True)
    tgt = T(z["target"])
    m.eval()
    with torch.no_grad():
        close(m.head(m.backbone(x.detach())), z["eval_logits"])
    m.train()
    losses = m(x, tgt, mode="train")
    keys = [str(k) for k in z["loss_keys"].tolist()]
    assert sorted(losses.keys()) == keys
    for k, v in zip(keys, z["loss_values"].tolist()):
        close(losses[k], v)
    losses["loss"].backward()
    close(x.grad, z["dx"], rtol=5e-4)
    named = dict(m.named_parameters())
    for k in [k for k in z.files if k.startswith("grad.")]:
        close(named[k[5:]].grad, z[k], rtol=5e-4)
    _check_running(m, z)
    m.eval()
    with torch.no_grad():
        am = m(x.detach(), tgt, mode="val")
    ref = T(z["val_argmax"].astype(np.int64))
    assert float((am != ref).float().mean()) <= 1e-3    # arg-max of fp32 logits: identical up to exact ties
