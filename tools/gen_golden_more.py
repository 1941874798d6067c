"""Golden vectors for the widened §8(a) rows (YOLOX #8/#12/#15, YOLOv7 blocks #19, STDC #10) — same method as
tools/gen_golden.py: import the reference's own classes in THIS container (never on the GPU box), run them on seeded
inputs, store inputs / state / outputs / gradients as small .npz fixtures under tests/golden/.

    python tools/gen_golden_more.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import install, run_module, save  # noqa: E402


class _ListArg(torch.nn.Module):
    """adapter: module(list_of_tensors) called as wrapper(*tensors), parameter names unchanged"""

    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, *xs):
        return self.m(list(xs))


def module_case(name, mod, inputs, extra=None):
    state0 = {k: v.clone() for k, v in mod.state_dict().items()}
    if len(inputs) == 1 and isinstance(inputs[0], (list, tuple)):
        inputs = list(inputs[0])
        outs, cots, gin, gpar = run_module(_ListArg(mod), inputs)
        gpar = {k[2:]: v for k, v in gpar.items()}
    else:
        outs, cots, gin, gpar = run_module(mod, inputs)
    save(name, x=inputs, state=state0, out=outs, cot=cots, gx=gin, gparam=gpar, **(extra or {}))


def yolox(bn):
    from src.models.backbones import build_backbone
    from src.models.heads.det.yolox_head import YOLOXHead
    from src.losses.det.yolox_loss import YOLOXLoss

    torch.manual_seed(21)
    bb = build_backbone({"name": "YOLOXCSPDarknet", "subtype": "cspdark_n", "out_stages": [2, 3, 4]})
    x = torch.randn(2, 3, 64, 96)
    state0 = {k: v.clone() for k, v in bb.state_dict().items()}
    outs, cots, gin, gpar = run_module(bb, [x])
    save("yolox_backbone_n", x=x, state=state0, out=outs, cot=cots, gparam_norms={k: v.norm() for k, v in gpar.items()},
         g_stem=gpar["stem.conv.conv.weight"])

    torch.manual_seed(22)
    head = YOLOXHead("yolox_n", num_classes=80, in_channels=256, channels=256, stacked_convs=2,
                     norm_cfg=dict(type="BN", momentum=0.03, eps=0.001), act_cfg=dict(type="Swish"))
    module_case("yolox_head_n", head, [[torch.randn(2, 64, 8, 12), torch.randn(2, 64, 4, 6), torch.randn(2, 64, 2, 3)]])

    # loss: raw head maps at three levels, pixel-unit targets; record the per-image SimOTA assignment
    for trial, (bs, size, nmax) in enumerate([(2, 64, 5), (3, 96, 9), (2, 64, 0)]):
        g = torch.Generator().manual_seed(70 + trial)
        preds = [torch.randn(bs, 85, size // s, size // s, generator=g) * 0.5 for s in (8, 16, 32)]
        for p in preds:  # plausible logits: objness/cls around the prior, sizes around 1-3 cells
            p[:, 4:] -= 2.0
        tg = torch.zeros(bs, max(nmax, 1), 5)
        for i in range(bs):
            n = int(torch.randint(1, nmax + 1, (1,), generator=g)) if nmax else 0
            if trial == 1 and i == 1:
                n = 0  # an image without labels inside a batch that has some
            tg[i, :n, 0] = torch.randint(0, 80, (n,), generator=g).float()
            tg[i, :n, 1:3] = (torch.rand(n, 2, generator=g) * 0.8 + 0.1) * size
            tg[i, :n, 3:5] = (torch.rand(n, 2, generator=g) * 0.45 + 0.05) * size
        loss = YOLOXLoss(80)
        rec = []
        orig = loss.get_assignments

        def spy(*a, **k):
            r = orig(*a, **k)
            rec.append((r[1].clone(), r[3].clone(), r[2].clone(), r[0].clone()))
            return r

        loss.get_assignments = spy
        pr = [q.clone().requires_grad_(True) for q in preds]
        out = loss(pr, tg)
        grads = torch.autograd.grad(out["loss"], pr)
        save("yolox_loss_%d" % trial, p=preds, targets=tg, loss=out["loss"], conf_loss=out["conf_loss"], cls_loss=out["cls_loss"],
             iou_loss=out["iou_loss"], num_fg=out["num_fg"], grads=grads,
             fg=[r[0] for r in rec], matched_gt=[r[1] for r in rec], matched_iou=[r[2] for r in rec], matched_cls=[r[3] for r in rec])


def yolov7():
    from src.models.modules.yolov7_modules import EELAN, DownA, DownB, SPPCSPC, UpSampling, FeatureFusion, RepConv
    from src.models.necks.yolov7_neck import YOLOv7Neck
    from src.models.heads.yolov7_head import YOLOv7Head
    from src.models.detects.yolov7_detect import YOLOv7Detect
    import src.models.yolov7 as ref_v7

    def fix_bn(m):
        for mm in m.modules():
            if isinstance(mm, torch.nn.BatchNorm2d):
                mm.eps, mm.momentum = 1e-3, 0.03
        return m

    torch.manual_seed(31)
    module_case("v7_eelan", fix_bn(EELAN(16, 8, 32)), [torch.randn(2, 16, 10, 12)])
    torch.manual_seed(32)
    module_case("v7_downa", fix_bn(DownA(16, 8)), [torch.randn(2, 16, 10, 12)])
    torch.manual_seed(33)
    module_case("v7_downb", fix_bn(DownB(16, 16)), [torch.randn(2, 16, 10, 12), torch.randn(2, 24, 5, 6)])
    torch.manual_seed(34)
    module_case("v7_sppcspc", fix_bn(SPPCSPC(32, 16)), [torch.randn(2, 32, 6, 7)])
    torch.manual_seed(35)
    module_case("v7_upsampling", fix_bn(UpSampling(16, 24, 8)), [torch.randn(2, 16, 5, 6), torch.randn(2, 24, 10, 12)])
    torch.manual_seed(36)
    module_case("v7_featurefusion", fix_bn(FeatureFusion(16, 8)), [torch.randn(2, 16, 10, 12)])
    torch.manual_seed(37)
    module_case("v7_repconv_id", fix_bn(RepConv(16, 16)), [torch.randn(2, 16, 10, 12)])
    torch.manual_seed(38)
    module_case("v7_repconv", fix_bn(RepConv(16, 24)), [torch.randn(2, 16, 10, 12)])
    torch.manual_seed(39)
    neck = YOLOv7Neck(in_channels=[512, 1024, 1024], out_channels=[128, 256, 512], width_mul=0.0625)
    module_case("v7_neck", neck, [[torch.randn(2, 32, 8, 12), torch.randn(2, 64, 4, 6), torch.randn(2, 64, 2, 3)]])
    torch.manual_seed(40)
    head = YOLOv7Head(in_channels=[128, 256, 512], out_channels=[256, 512, 1024], width_mul=0.0625)
    module_case("v7_head", head, [[torch.randn(2, 8, 8, 12), torch.randn(2, 16, 4, 6), torch.randn(2, 32, 2, 3)]])
    torch.manual_seed(41)
    det = YOLOv7Detect(num_classes=80, in_channels=[256, 512, 1024], anchors=ref_v7.YOLOv7.anchors, width_mul=0.0625)
    fe = [torch.randn(2, 16, 8, 8), torch.randn(2, 32, 4, 4), torch.randn(2, 64, 2, 2)]
    det.train()
    _, tr = det([f.clone() for f in fe])
    det.eval()
    z, _ = det([f.clone() for f in fe])
    save("v7_detect", x=fe, state=det.state_dict(), train_out=tr, z=z)


def v7_ota():
    from src.losses.yolov7_loss import YOLOv7Loss
    import src.models.yolov7 as ref_v7
    for trial, (bs, size, nmax) in enumerate([(2, 64, 5), (3, 96, 8), (2, 64, 0)]):
        g = torch.Generator().manual_seed(90 + trial)
        p = [torch.randn(bs, 3, size // s, size // s, 85, generator=g) for s in (8, 16, 32)]
        rows = []
        for i in range(bs):
            n = int(torch.randint(1, nmax + 1, (1,), generator=g)) if nmax else 0
            if trial == 1 and i == 1:
                n = 0
            t = torch.zeros(n, 6)
            t[:, 0] = i
            t[:, 1] = torch.randint(0, 80, (n,), generator=g).float()
            t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
            t[:, 4:6] = torch.rand(n, 2, generator=g) * 0.4 + 0.05
            rows.append(t)
        targets = torch.cat(rows, 0) if rows else torch.zeros(0, 6)
        imgs = torch.zeros(bs, 3, size, size)
        loss = YOLOv7Loss(80, anchors=ref_v7.YOLOv7.anchors, device="cpu")
        pr = [q.clone().requires_grad_(True) for q in p]
        if targets.shape[0] == 0:
            continue  # the reference crashes on a batch without any label (torch.cat of empty lists): not a fixture
        total, stats = loss(pr, targets, imgs)
        grads = torch.autograd.grad(total, pr)
        bs_, as_, gjs, gis, tg, anch = loss.build_targets(p, targets, imgs)
        save("v7_ota_loss_%d" % trial, p=p, targets=targets, size=np.array([size]), total=total, stats=stats, grads=grads,
             b=bs_, a=as_, gj=gjs, gi=gis, tcls=[t[:, 1] for t in tg], tbox=[t[:, 2:6] for t in tg])


def stdc():
    from src.models.backbones.seg.stdcnet import STDCNet, CatBottleneck, AddBottleneck
    from src.models.necks.seg.stdc_neck import AttentionRefinementModule, FeatureFusionModule

    torch.manual_seed(51)
    module_case("stdc_cat_s2", CatBottleneck(16, 32, 4, 2), [torch.randn(2, 16, 10, 12)])
    torch.manual_seed(52)
    module_case("stdc_cat_s1", CatBottleneck(32, 32, 4, 1), [torch.randn(2, 32, 6, 7)])
    torch.manual_seed(53)
    module_case("stdc_add_s2", AddBottleneck(16, 32, 4, 2), [torch.randn(2, 16, 10, 12)])
    torch.manual_seed(54)
    net = STDCNet("stdc1", out_channels=[8, 16, 64, 128, 256], layers=[2, 2, 2], block_num=4, out_stages=[2, 3, 4])
    x = torch.randn(2, 3, 64, 96)
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    outs, cots, gin, gpar = run_module(net, [x])
    save("stdc_net_small", x=x, state=state0, out=outs, cot=cots, gparam_norms={k: v.norm() for k, v in gpar.items()},
         g_stem=gpar["stem.conv.weight"])
    torch.manual_seed(55)
    module_case("stdc_arm", AttentionRefinementModule(32, 16), [torch.randn(6, 32, 6, 7)])  # 1x1-spatial BN: needs a batch > 2
    torch.manual_seed(56)
    module_case("stdc_ffm", FeatureFusionModule(48, 32), [torch.randn(6, 32, 6, 7), torch.randn(6, 16, 6, 7)])
    from src.models.necks.seg.stdc_neck import STDCNeck
    torch.manual_seed(58)
    neck = STDCNeck(in_channels=[32, 64, 128], out_channels=32, aux_out_channels=16)

    class _Flat(torch.nn.Module):  # (feat, [aux...]) -> flat tuple of tensors
        def __init__(self, m):
            super().__init__()
            self.m = m

        def forward(self, *xs):
            f, aux = self.m(list(xs))
            return [f] + list(aux[1:])

    state0 = {k: v.clone() for k, v in neck.state_dict().items()}
    xs = [torch.randn(6, 32, 8, 12), torch.randn(6, 64, 4, 6), torch.randn(6, 128, 2, 3)]
    outs, cots, gin, gpar = run_module(_Flat(neck), xs)
    save("stdc_neck", x=xs, state=state0, out=outs, cot=cots, gx=gin, gparam={k[2:]: v for k, v in gpar.items()})
    # full-size structure facts for STDC1 (parameter names / count / output shapes)
    torch.manual_seed(57)
    full = STDCNet("stdc1")
    sd = full.state_dict()
    full.train()
    feats = full(torch.randn(1, 3, 64, 128))
    save("stdc1_structure", state_keys=np.array(sorted(sd.keys())), n_params=np.array([sum(p.numel() for p in full.parameters())]),
         shapes=np.array([list(f.shape) for f in feats]))


def main():
    install()
    which = sys.argv[1:] or ["yolox", "yolov7", "stdc", "v7_ota"]
    bn = dict(type="BN", momentum=0.03, eps=0.001)
    if "yolox" in which:
        yolox(bn)
    if "yolov7" in which:
        yolov7()
    if "stdc" in which:
        stdc()
    if "v7_ota" in which:
        v7_ota()


if __name__ == "__main__":
    main()
