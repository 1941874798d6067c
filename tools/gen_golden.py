"""Generate golden vectors from the REFERENCE ITSELF (build container only; never runs on the GPU box).

Imports the reference's own Python modules from /root/reference (read-only), runs them on seeded inputs
on CPU/fp32 and stores inputs, parameters/buffers, outputs and gradients as small .npz fixtures under
tests/golden/. The fixtures are data; no reference source is copied.

Import recipe: absent third-party roots (torchvision, cv2, pycocotools, ...) are
satisfied by an import-time stub that is never *called* on these paths; YOLOv5Loss needs the integer
`clamp_` shim that restores torch<=1.9 semantics for src/losses/yolov5_loss.py:273.

    python tools/gen_golden.py            # writes tests/golden/*.npz
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
STUB_ROOTS = ("torchvision", "cv2", "pycocotools", "termcolor", "tensorboardX", "timm", "thop", "mmcv", "lxml", "skimage",
              "albumentations", "seaborn", "onnx", "prefetch_generator", "glob2", "matplotlib", "PIL", "scipy")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return mock.MagicMock(name="%s.%s" % (self.__name__, name))


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            try:
                # prefer the real module when it exists
                for f in sys.meta_path:
                    if f is self:
                        continue
                    spec = f.find_spec(fullname, path, target) if hasattr(f, "find_spec") else None
                    if spec is not None:
                        return spec
            except Exception:
                pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install():
    sys.path.insert(0, REF)
    sys.meta_path.insert(0, _StubFinder())
    orig = torch.Tensor.clamp_

    def clamp_(self, min=None, max=None):
        if not self.dtype.is_floating_point:
            if torch.is_tensor(min):
                min = int(min.item())
            if torch.is_tensor(max):
                max = int(max.item())
        return orig(self, min, max)

    torch.Tensor.clamp_ = clamp_


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    flat = {}
    for k, v in arrs.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                flat["%s/%s" % (k, kk)] = npy(vv) if torch.is_tensor(vv) else np.asarray(vv)
        elif isinstance(v, (list, tuple)):
            for i, vv in enumerate(v):
                flat["%s/%d" % (k, i)] = npy(vv) if torch.is_tensor(vv) else np.asarray(vv)
        else:
            flat[k] = npy(v) if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **flat)
    print("wrote", name, sum(a.size for a in flat.values()), "elems")


def run_module(mod, inputs, train=True):
    """forward + backward with a fixed cotangent; returns outputs, input grads, param grads."""
    mod.train(train)
    inputs = [x.clone().requires_grad_(True) for x in inputs]
    out = mod(*inputs)
    outs = list(out) if isinstance(out, (tuple, list)) else [out]
    outs = [o for o in outs if torch.is_tensor(o)]
    g = torch.Generator().manual_seed(7)
    cots = [torch.randn(o.shape, generator=g) for o in outs]
    loss = sum((o * c).sum() for o, c in zip(outs, cots))
    params = [p for p in mod.parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, inputs + params, allow_unused=True)
    gin = grads[:len(inputs)]
    gpar = {n: (g if g is not None else torch.zeros_like(p)) for (n, p), g in
            zip([(n, p) for n, p in mod.named_parameters() if p.requires_grad], grads[len(inputs):])}
    return outs, cots, gin, gpar


def main():
    install()
    from src.models.bricks import ConvModule, DepthwiseSeparableConvModule
    from src.models.modules.yolo_modules import CSPLayer, SPPF, Focus, UpsamplingModule, DownsamplingModule, DarknetBottleneck
    from src.models.detects.yolov5_detect import YOLOv5Detect
    from src.losses.yolov5_loss import bbox_iou, YOLOv5Loss
    from src.losses.det.iou_losses import bbox_overlaps
    from src.models.backbones import build_backbone
    import src.models.yolov5 as ref_yolov5

    bn = dict(type="BN", momentum=0.03, eps=0.001)
    silu = dict(type="SiLU", inplace=True)

    # ---- ConvModule variants -------------------------------------------------------------------------
    cases = {
        "k1": dict(cin=16, cout=24, k=1, s=1, p=0, d=1, g=1, norm=bn, act=silu, hw=(10, 12)),
        "k3": dict(cin=16, cout=16, k=3, s=1, p=1, d=1, g=1, norm=bn, act=silu, hw=(9, 11)),
        "k3s2": dict(cin=8, cout=32, k=3, s=2, p=1, d=1, g=1, norm=bn, act=dict(type="Swish"), hw=(12, 10)),
        "k3s2odd": dict(cin=8, cout=16, k=3, s=2, p=1, d=1, g=1, norm=bn, act=silu, hw=(11, 13)),
        "k6s2": dict(cin=3, cout=16, k=6, s=2, p=2, d=1, g=1, norm=bn, act=silu, hw=(20, 24)),
        "k3d2": dict(cin=16, cout=8, k=3, s=1, p=2, d=2, g=1, norm=dict(type="BN"), act=dict(type="ReLU"), hw=(9, 9)),
        "k1bias": dict(cin=16, cout=21, k=1, s=1, p=0, d=1, g=1, norm=None, act=None, hw=(6, 7)),
        "dw3d3": dict(cin=16, cout=16, k=3, s=1, p=3, d=3, g=16, norm=dict(type="BN"), act=dict(type="ReLU"), hw=(10, 10)),
        "k1s2": dict(cin=16, cout=32, k=1, s=2, p=0, d=1, g=1, norm=dict(type="BN"), act=None, hw=(8, 10)),
    }
    for name, c in cases.items():
        torch.manual_seed(100 + len(name))
        m = ConvModule(c["cin"], c["cout"], c["k"], stride=c["s"], padding=c["p"], dilation=c["d"], groups=c["g"],
                       norm_cfg=c["norm"], act_cfg=c["act"])
        if c["norm"] is not None:
            with torch.no_grad():
                m.bn.weight.uniform_(0.5, 1.5)
                m.bn.bias.uniform_(-0.5, 0.5)
        x = torch.randn(2, c["cin"], *c["hw"])
        state0 = {k: v.clone() for k, v in m.state_dict().items()}
        outs, cots, gin, gpar = run_module(m, [x])
        save("convmodule_" + name, x=x, state=state0, out=outs[0], cot=cots[0], gx=gin[0], gparam=gpar,
             state_after={k: v for k, v in m.state_dict().items() if "running" in k},
             meta=np.array([c["cin"], c["cout"], c["k"], c["s"], c["p"], c["d"], c["g"]]))

    # ---- blocks ------------------------------------------------------------------------------------------
    def block_case(name, mod, inputs):
        for mm in mod.modules():
            if isinstance(mm, torch.nn.BatchNorm2d):
                mm.eps, mm.momentum = 1e-3, 0.03
                with torch.no_grad():
                    mm.weight.uniform_(0.5, 1.5)
                    mm.bias.uniform_(-0.3, 0.3)
        state0 = {k: v.clone() for k, v in mod.state_dict().items()}
        outs, cots, gin, gpar = run_module(mod, inputs)
        save("block_" + name, x=inputs, state=state0, out=outs, cot=cots, gx=gin, gparam=gpar)

    torch.manual_seed(1)
    block_case("bottleneck", DarknetBottleneck(16, 16, 1.0, True, norm_cfg=bn, act_cfg=silu), [torch.randn(2, 16, 8, 8)])
    torch.manual_seed(2)
    block_case("csp", CSPLayer(32, 32, n=2, shortcut=True, norm_cfg=bn, act_cfg=silu), [torch.randn(2, 32, 8, 10)])
    torch.manual_seed(3)
    block_case("sppf", SPPF(32, 32, kernel_sizes=5, norm_cfg=bn, act_cfg=silu), [torch.randn(2, 32, 9, 9)])
    torch.manual_seed(4)
    block_case("spp", SPPF(32, 32, kernel_sizes=(5, 9, 13), norm_cfg=bn, act_cfg=dict(type="Swish")), [torch.randn(2, 32, 10, 10)])
    torch.manual_seed(5)
    block_case("focus", Focus(3, 16, 3, norm_cfg=bn, act_cfg=dict(type="Swish")), [torch.randn(2, 3, 16, 20)])
    torch.manual_seed(6)
    block_case("up", UpsamplingModule(32, 16, 1, norm_cfg=bn, act_cfg=silu), [torch.randn(2, 32, 5, 6), torch.randn(2, 16, 10, 12)])
    torch.manual_seed(7)
    block_case("down", DownsamplingModule(16, 32, 1, norm_cfg=bn, act_cfg=silu), [torch.randn(2, 16, 10, 12), torch.randn(2, 16, 5, 6)])

    # ---- backbone (full YOLOv5-s CSPDarknet on a small image) ---------------------------------------------
    torch.manual_seed(8)
    bb = build_backbone({"name": "YOLOv5CSPDarknet", "subtype": "cspdark_s", "out_stages": [2, 3, 4]})
    x = torch.randn(1, 3, 64, 64)
    bb.train()
    feats = bb(x)
    sd = bb.state_dict()
    save("backbone_v5s", x=x, feats=feats,
         state_checksum=np.array([float(sum(v.double().sum() for v in sd.values() if v.dtype.is_floating_point))]),
         state_keys=np.array(sorted(sd.keys())), n_params=np.array([sum(p.numel() for p in bb.parameters())]))
    # small-width full backbone with all parameters stored (n subtype)
    torch.manual_seed(9)
    bbn = build_backbone({"name": "YOLOv5CSPDarknet", "subtype": "cspdark_n", "out_stages": [2, 3, 4]})
    xn = torch.randn(2, 3, 64, 96)
    state0 = {k: v.clone() for k, v in bbn.state_dict().items()}
    outs, cots, gin, gpar = run_module(bbn, [xn])
    save("backbone_v5n_full", x=xn, state=state0, out=outs, cot=cots, gparam_norms={k: v.norm() for k, v in gpar.items()},
         g_stem=gpar["stem.conv.weight"])

    # ---- detect head -------------------------------------------------------------------------------------------
    torch.manual_seed(10)
    det = YOLOv5Detect(num_classes=80, in_channels=[256, 512, 1024], anchors=ref_yolov5.YOLOv5.anchors, width_mul=0.125)
    fe = [torch.randn(2, 32, 8, 8), torch.randn(2, 64, 4, 4), torch.randn(2, 128, 2, 2)]
    det.train()
    _, tr = det([f.clone() for f in fe])
    det.eval()
    z, _ = det([f.clone() for f in fe])
    save("detect_v5", x=fe, state=det.state_dict(), train_out=tr, z=z)

    # ---- DeepLabv3+ head (reference class, reduced width, dropout disabled for determinism) + CrossEntropyLoss2d --------
    from src.models.heads.seg.deeplabv3plus_head import Deeplabv3PlusHead
    from src.losses.seg.cross_entropy_loss import CrossEntropyLoss2d
    torch.manual_seed(12)
    head = Deeplabv3PlusHead(low_in_channels=16, low_channels=8, num_classes=19, in_channels=64, channels=32, dilations=(1, 2, 3, 4),
                             dropout_ratio=0)
    for mm in head.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            with torch.no_grad():
                mm.weight.uniform_(0.5, 1.5)
                mm.bias.uniform_(-0.3, 0.3)
    xs = [torch.randn(2, 16, 16, 24), torch.randn(2, 64, 4, 6)]
    state0 = {k: v.clone() for k, v in head.state_dict().items()}
    head.train()
    xin = [x.clone().requires_grad_(True) for x in xs]
    logits = head(xin)
    tgt = torch.randint(0, 19, (2, 32, 48))
    tgt[torch.rand(2, 32, 48) < 0.1] = 255
    up = torch.nn.functional.interpolate(logits, size=tgt.shape[-2:], mode="bilinear", align_corners=False)
    loss = CrossEntropyLoss2d()(up, tgt)
    named = [(n, q) for n, q in head.named_parameters()]
    grads = torch.autograd.grad(loss, xin + [q for _, q in named])
    save("deeplabv3plus_head", x=xs, state=state0, logits=logits, target=tgt, loss=loss, gx=grads[:2],
         gparam={n: g for (n, _), g in zip(named, grads[2:])})

    # ---- IoU family + known-answer vector ------------------------------------------------------------------------
    torch.manual_seed(11)
    b1 = torch.rand(4, 50) * torch.tensor([[10.], [10.], [5.], [5.]]) + 0.1
    b2 = (torch.rand(50, 4) * torch.tensor([10., 10., 5., 5.]) + 0.1)
    save("bbox_iou", b1=b1, b2=b2, iou=bbox_iou(b1, b2, x1y1x2y2=False), giou=bbox_iou(b1, b2, x1y1x2y2=False, GIoU=True),
         diou=bbox_iou(b1, b2, x1y1x2y2=False, DIoU=True), ciou=bbox_iou(b1, b2, x1y1x2y2=False, CIoU=True))
    kb1 = torch.FloatTensor([[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]])
    kb2 = torch.FloatTensor([[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]])
    save("bbox_overlaps_kat", b1=kb1, b2=kb2, iou=bbox_overlaps(kb1, kb2), giou=bbox_overlaps(kb1, kb2, mode="giou", eps=1e-7))
    bx = torch.rand(40, 4) * 100
    bx[:, 2:] += bx[:, :2]
    by = torch.rand(30, 4) * 100
    by[:, 2:] += by[:, :2]
    save("box_iou", a=bx, b=by, iou=ref_yolov5.box_iou(bx, by), xyxy=ref_yolov5.xywh2xyxy(bx))

    # ---- YOLOv5 loss (with build_targets indices) ---------------------------------------------------------------------
    for trial, (bs, sizes, nmax) in enumerate([(2, (16, 8, 4), 6), (3, (20, 10, 5), 12), (2, (16, 8, 4), 0)]):
        g = torch.Generator().manual_seed(50 + trial)
        p = [torch.randn(bs, 3, s, s, 85, generator=g) for s in sizes]
        rows = []
        for i in range(bs):
            n = int(torch.randint(1, nmax + 1, (1,), generator=g)) if nmax else 0
            t = torch.zeros(n, 6)
            t[:, 0] = i
            t[:, 1] = torch.randint(0, 80, (n,), generator=g).float()
            t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
            t[:, 4:6] = torch.rand(n, 2, generator=g) * 0.48 + 0.02
            rows.append(t)
        targets = torch.cat(rows, 0)
        if trial == 1:
            targets[0, 2:4] = torch.tensor([1.0, 0.999])  # border target: exercises the in-place index clamp
        loss = YOLOv5Loss(80, anchors=ref_yolov5.YOLOv5.anchors, device="cpu")
        pr = [q.clone().requires_grad_(True) for q in p]
        total, stats = loss(pr, targets)
        grads = torch.autograd.grad(total, pr)
        tcls, tbox, indices, anch = loss.build_targets(p, targets)
        save("yolov5_loss_%d" % trial, p=p, targets=targets, total=total, stats=stats, grads=grads, tcls=tcls, tbox=tbox,
             b=[i[0] for i in indices], a=[i[1] for i in indices], gj=[i[2] for i in indices], gi=[i[3] for i in indices], anch=anch)


if __name__ == "__main__":
    main()
