"""Reference-run fixtures for the STDC segmentation TRAIN path (VERDICT r05 missing item 3; conf/seg/stdc/cityscapes_stdc1.yml:55-68).

Build container only (needs /root/reference; never runs on the GPU box). The reference's OWN classes are imported and executed:

  src.models.heads.seg.fcn_head.FCNHead / stdc_head.STDCHead        (fcn_head.py:14-63, stdc_head.py:16-18, base_seg_head.py:12-41)
  src.losses.seg.cross_entropy_loss.OhemCrossEntropyLoss2d           (cross_entropy_loss.py:51-69)
  src.losses.seg.detail_loss.DetailAggregateLoss                     (detail_loss.py:23-88)
  src.models.segmentors.encoder_decoder.EncoderDecoder               (encoder_decoder.py:21-150: STDCNet -> STDCNeck -> FCNHead + three
                                                                      auxiliary heads / losses, the branch at :136-148)

Two of them move tensors to a GPU in their constructors (`.cuda()`, `.type(torch.cuda.FloatTensor)`): on this CPU-only box
`torch.Tensor.cuda` is patched to the identity and `torch.cuda.FloatTensor` to `torch.FloatTensor` — placement only, no arithmetic.

    python tools/gen_golden_stdc_train.py        # writes tests/golden/stdctrain_*.npz
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
import gen_golden as GG  # noqa: E402
import gen_golden_more as GM  # noqa: E402  (module_case)


def cpu_placement():
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor


def no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, (torch.nn.Dropout2d, torch.nn.Dropout)):
            mod.p = 0.0   # (dropout draws from the global generator: off, so that the fixture is a function of its inputs)
    return m


def gen_heads():
    from src.models.heads.seg.fcn_head import FCNHead
    from src.models.heads.seg.stdc_head import STDCHead
    torch.manual_seed(31)
    GM.module_case("stdctrain_fcn_head_concat", no_dropout(FCNHead(num_classes=5, in_channels=16, channels=24, num_convs=2, is_concat=True)),
                   [torch.randn(3, 16, 9, 11)])
    torch.manual_seed(32)
    GM.module_case("stdctrain_fcn_head_plain", no_dropout(FCNHead(num_classes=19, in_channels=32, channels=16, num_convs=1, is_concat=False)),
                   [torch.randn(2, 32, 8, 12)])
    torch.manual_seed(33)
    GM.module_case("stdctrain_stdc_head", no_dropout(STDCHead(num_classes=1, in_channels=32, channels=16, num_convs=1, is_concat=False)),
                   [torch.randn(2, 32, 8, 12)])


def gen_ohem():
    from src.losses.seg.cross_entropy_loss import OhemCrossEntropyLoss2d
    for name, seed, scale, min_kept, thresh in (("hard", 41, 1.0, 300, 0.7),      # loss[min_kept] > thresh: the mean of everything above the threshold
                                                ("easy", 42, 9.0, 300, 0.7),      # confident predictions: the mean of the min_kept largest
                                                ("ignored", 43, 6.0, 900, 0.7)):  # min_kept reaches into the zero losses of ignored pixels (ties at 0)
        g = torch.Generator().manual_seed(seed)
        tgt = torch.randint(0, 6, (2, 24, 32), generator=g)
        pred = torch.randn(2, 6, 24, 32, generator=g)
        if scale > 1.0:   # push the logit of the true class up: most pixels become easy
            pred = pred + scale * torch.nn.functional.one_hot(tgt, 6).permute(0, 3, 1, 2).float() * (torch.rand(2, 1, 24, 32, generator=g) > 0.1)
        tgt[:, :5] = 255
        pred.requires_grad_(True)
        l = OhemCrossEntropyLoss2d(thresh=thresh, min_kept=min_kept)
        loss = l(pred, tgt)
        loss.backward()
        with torch.no_grad():
            per = torch.nn.functional.cross_entropy(pred, tgt, ignore_index=255, reduction="none").view(-1)
            v = torch.sort(per, descending=True).values[min_kept]
        GG.save("stdctrain_ohem_" + name, pred=pred.detach(), target=tgt, loss=loss.detach(), dpred=pred.grad,
                cfg=np.array([thresh, min_kept], np.float64), branch=np.array(int(v > l.thresh)))


def gen_detail():
    from src.losses.seg.detail_loss import DetailAggregateLoss
    for name, seed, lh, lw in (("same", 51, 32, 48), ("resized", 52, 16, 24)):
        g = torch.Generator().manual_seed(seed)
        # blocky label map (regions with straight and diagonal borders), some ignored pixels
        tgt = torch.zeros(2, 32, 48, dtype=torch.int64)
        for i in range(6):
            y0, x0 = int(torch.randint(0, 24, (1,), generator=g)), int(torch.randint(0, 36, (1,), generator=g))
            tgt[:, y0:y0 + int(torch.randint(4, 14, (1,), generator=g)), x0:x0 + int(torch.randint(4, 20, (1,), generator=g))] = i + 1
        tgt[1] = torch.roll(tgt[1], 5, 1)
        tgt[0, :3, :7] = 255
        logits = torch.randn(2, 1, lh, lw, generator=g).requires_grad_(True)
        l = DetailAggregateLoss()
        loss = l(logits, tgt)
        loss.backward()
        GG.save("stdctrain_detail_" + name, logits=logits.detach(), target=tgt, loss=loss.detach(), dlogits=logits.grad)


def gen_encoder_decoder():
    from src.models.segmentors.encoder_decoder import EncoderDecoder
    from src.utils.config import CommonConfiguration
    cfg = {"BACKBONE": {"name": "STDCNet", "subtype": "stdc1", "out_channels": [8, 16, 64, 128, 256], "layers": [2, 2, 2], "out_stages": [2, 3, 4],
                        "pretrained": False},
           "NECK": {"name": "STDCNeck", "in_channels": [64, 128, 256], "out_channels": 64, "aux_out_channels": 32},
           "HEAD": {"name": "FCNHead", "num_classes": 19, "in_channels": 64, "channels": 64, "num_convs": 1, "is_concat": False},
           "AUX_HEAD": [{"name": "STDCHead", "num_classes": 1, "in_channels": 64, "channels": 16, "num_convs": 1, "is_concat": False},
                        {"name": "FCNHead", "num_classes": 19, "in_channels": 32, "channels": 16, "num_convs": 1, "is_concat": False},
                        {"name": "FCNHead", "num_classes": 19, "in_channels": 32, "channels": 16, "num_convs": 1, "is_concat": False}],
           "LOSS": {"name": "OhemCrossEntropyLoss2d", "min_kept": 2000},
           "AUX_LOSS": [{"name": "DetailAggregateLoss"}, {"name": "OhemCrossEntropyLoss2d", "min_kept": 2000},
                        {"name": "OhemCrossEntropyLoss2d", "min_kept": 2000}]}
    try:
        model_cfg = CommonConfiguration.from_dict(cfg) if hasattr(CommonConfiguration, "from_dict") else CommonConfiguration(cfg)
    except Exception:
        model_cfg = types.SimpleNamespace(**cfg)
    torch.manual_seed(61)
    m = no_dropout(EncoderDecoder([{"c%d" % i: 1.0} for i in range(19)], model_cfg))
    m.train()
    state0 = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 64, 128, generator=g).requires_grad_(True)
    tgt = torch.zeros(4, 64, 128, dtype=torch.int64)
    for i in range(10):
        y0, x0 = int(torch.randint(0, 50, (1,), generator=g)), int(torch.randint(0, 100, (1,), generator=g))
        tgt[:, y0:y0 + int(torch.randint(6, 30, (1,), generator=g)), x0:x0 + int(torch.randint(8, 50, (1,), generator=g))] = int(torch.randint(0, 19, (1,), generator=g))
    for b in range(4):
        tgt[b] = torch.roll(tgt[b], 7 * b, 1)
    tgt[:, :3] = 255
    losses = m(x, tgt, mode="train")
    losses["loss"].backward()
    keys = sorted(losses.keys())
    named = dict(m.named_parameters())
    arrs = {"x": x.detach(), "target": tgt, "state": state0, "dx": x.grad, "loss_keys": np.array(keys),
            "loss_values": np.array([float(losses[k].detach()) for k in keys], np.float64),
            "gparam_norms": {k: v.grad.norm() for k, v in named.items() if v.grad is not None},
            "no_grad_params": np.array(sorted(k for k, v in named.items() if v.grad is None))}
    for k in ("head.cls_seg.weight", "head.cls_seg.bias", "auxiliary_head.0.cls_seg.weight", "auxiliary_head.1.cls_seg.weight",
              "auxiliary_head.2.convs.0.conv.weight", "head.convs.0.bn.weight"):
        arrs["grad." + k] = named[k].grad
    m.eval()
    with torch.no_grad():
        arrs["val_argmax"] = GG.npy(m(x.detach(), tgt, mode="val")).astype(np.int16)
    GG.save("stdctrain_encoder_decoder", **arrs)


def main():
    GG.install()
    cpu_placement()
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["heads", "ohem", "detail", "encdec"]
    if "heads" in which:
        gen_heads()
    if "ohem" in which:
        gen_ohem()
    if "detail" in which:
        gen_detail()
    if "encdec" in which:
        gen_encoder_decoder()


if __name__ == "__main__":
    main()
