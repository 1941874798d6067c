"""GPU parity of the fused on-device YOLOv5 loss (cvhip_yolov5_loss_*) against the fixed-shape torch
formulation (cvpytorch_amd.yolov5.YOLOv5Loss — itself equal to the reference's loss on its golden vectors,
tests/test_yolov5_loss.py) evaluated on the SAME bf16 head maps, and against the reference's golden vectors directly.
Loss values: rtol 1e-5 (fp32 arithmetic, different summation order). Gradients are stored in bf16 by the fused path:
|d| <= 2^-8 * |ref| + 1e-6 * max|ref| element-wise."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvpytorch_amd import ops, yolov5
from test_gpu_modules import cosine, dev

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _torch_path(raws, targets, nc=80):
    """reference-layout loss on the same values: raw (N, 3*(5+nc), H, W) -> p (N, 3, H, W, 5+nc) fp32"""
    leaves = [r.float().detach().requires_grad_(True) for r in raws]
    p = [l.view(l.shape[0], 3, nc + 5, l.shape[2], l.shape[3]).permute(0, 1, 3, 4, 2) for l in leaves]
    total, stats = yolov5.YOLOv5Loss(nc).to(raws[0].device)(p, targets)
    grads = torch.autograd.grad(total, leaves)
    return total, stats, grads


def _check(raws, targets, nc=80):
    total_t, stats_t, grads_t = _torch_path(raws, targets, nc)
    rr = [r.clone().requires_grad_(True) for r in raws]
    total_f, stats_f = yolov5.YOLOv5LossFused(nc)(rr, targets)
    grads_f = torch.autograd.grad(total_f, rr)
    assert torch.allclose(total_f, total_t, rtol=1e-5, atol=1e-6), (float(total_f), float(total_t))
    assert torch.allclose(stats_f, stats_t, rtol=1e-5, atol=1e-7), (stats_f, stats_t)
    for gf, gt in zip(grads_f, grads_t):
        gf = gf.float()
        tol = gt.abs() * 2.0 ** -8 + 1e-6 * float(gt.abs().max()) + 1e-12
        bad = (gf - gt).abs() > tol
        assert not bool(bad.any()), (int(bad.sum()), float((gf - gt).abs().max()), float(gt.abs().max()))


def _maps(bs, sizes, seed, nc=80):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(bs, 3 * (nc + 5), s, s, generator=g).to(torch.bfloat16).to(dev()).contiguous(memory_format=torch.channels_last)
            for s in sizes]


def _targets(bs, nmax, seed, pad_to):
    g = torch.Generator().manual_seed(100 + seed)
    rows = []
    for i in range(bs):
        n = int(torch.randint(1, nmax + 1, (1,), generator=g)) if nmax else 0
        t = torch.zeros(n, 6)
        t[:, 0] = i
        t[:, 1] = torch.randint(0, 80, (n,), generator=g).float()
        t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
        t[:, 4:6] = torch.rand(n, 2, generator=g) * 0.48 + 0.02
        rows.append(t)
    t = torch.cat(rows, 0) if rows else torch.zeros(0, 6)
    pad = torch.zeros(pad_to - t.shape[0], 6)
    pad[:, 0] = -1
    pad[:, 2:] = 0.5
    return torch.cat([t, pad], 0).to(dev())


@pytest.mark.parametrize("bs,sizes,nmax,seed", [(2, (16, 8, 4), 6, 0), (3, (20, 10, 5), 12, 1), (4, (8, 4, 2), 20, 2), (2, (16, 8, 4), 0, 3),
                                                 (8, (40, 20, 10), 20, 4)])
def test_fused_loss_equals_torch_formulation(bs, sizes, nmax, seed):
    _check(_maps(bs, sizes, seed), _targets(bs, nmax, seed, bs * 24))


def test_fused_loss_duplicate_cells_and_borders():
    """many targets in the same cell (objectness 'last writer wins', box/class gradients summed over duplicates) and
    targets on the image border (index clamp)"""
    raws = _maps(2, (8, 4, 2), 7)
    t = torch.zeros(96, 6)
    t[:, 0] = -1
    t[:, 2:] = 0.5
    g = torch.Generator().manual_seed(5)
    for k in range(80):  # 80 boxes of image 0 centred in (almost) the same spot => long per-cell lists (> 64 at the coarse level)
        t[k] = torch.tensor([0, k % 80, 0.52 + 0.001 * (k % 7), 0.47 + 0.001 * (k % 5), 0.2 + 0.002 * k, 0.25 + 0.001 * k])
    t[80] = torch.tensor([1, 3, 1.0, 0.999, 0.3, 0.3])
    t[81] = torch.tensor([1, 4, 0.0, 0.0, 0.2, 0.2])
    t[82] = torch.tensor([1, 5, 0.999, 0.001, 0.4, 0.1])
    _check(raws, t.to(dev()))


@pytest.mark.parametrize("bs,sizes,nmax,seed", [(2, (16, 8, 4), 6, 0), (8, (40, 20, 10), 20, 4), (2, (16, 8, 4), 0, 3)])
def test_fused_loss_offers_the_bias_gradient(bs, sizes, nmax, seed):
    """cvhip_yolov5_loss_level_bwd_bias: the column sums of every level's gradient map (the detect convolutions' bias gradient), summed
    from the loss's compact state, equal the sums of the torch formulation's fp32 gradient — and the sums of the map the kernel wrote, up
    to that map's 16-bit rounding; no targets at all (n = 0) gives finite sums (objectness only)."""
    import ctypes as C
    from cvpytorch_amd import lib as L, ops
    raws, tg = _maps(bs, sizes, seed), _targets(bs, nmax, seed, bs * 24)
    _, _, grads_t = _torch_path(raws, tg)
    rr = [r.clone().requires_grad_(True) for r in raws]
    total_f, _ = yolov5.YOLOv5LossFused(80)(rr, tg)
    grads_f = torch.autograd.grad(total_f, rr)
    tab = dict(ops._COLSUMS)
    assert len(tab) == 3
    for gf, gt in zip(grads_f, grads_t):
        hit = [v for k, v in tab.items() if k == gf.data_ptr()]
        assert len(hit) == 1, "the sums are keyed by the address of channel 0 of the map"
        partial, rows, K, _keep = hit[0]
        assert rows == L.YOLO_BIAS_ROWS and K == 255
        out = torch.empty(K, dtype=torch.float32, device=gf.device)
        L.call("cvhip_colsum_finalize", partial.data_ptr(), rows, K, out.data_ptr(), 0, None)
        torch.cuda.synchronize()
        ref = gt.double().sum((0, 2, 3)).cpu()
        got = out.double().cpu()
        assert torch.isfinite(got).all()
        scale = float(ref.abs().max()) + 1e-12
        assert float((got - ref).abs().max()) <= 2e-4 * scale + 1e-9, (float((got - ref).abs().max()), scale)
        wrote = gf.double().sum((0, 2, 3)).cpu()   # the 16-bit map: every entry rounded, so only close
        assert float((got - wrote).abs().max()) <= 2e-2 * scale + 1e-9
    ops.clear_colsums()


def test_detect_bias_gradient_through_the_offer_equals_the_map_sum():
    """end to end: a bias-carrying 1x1 convolution in front of the fused loss takes its bias gradient from the offered sums (no pass over
    the map) and gets the same gradient as with the table emptied (the colsum pass), up to the map's rounding"""
    from cvpytorch_amd import ops, bricks
    torch.manual_seed(3)
    tg = _targets(4, 10, 9, 96)
    xs = [torch.randn(4, 64, s, s).to(torch.bfloat16).to(dev()).contiguous(memory_format=torch.channels_last) for s in (16, 8, 4)]
    convs = [bricks.HipConv2d(64, 255, 1, bias=True).to(dev()) for _ in range(3)]
    loss = yolov5.YOLOv5LossFused(80)

    def run(drop_offer):
        for c in convs:
            c.zero_grad(set_to_none=True)
        raws = [c(x) for c, x in zip(convs, xs)]
        total, _ = loss(raws, tg)
        if drop_offer:
            orig = ops.offer_colsum
            ops.offer_colsum = lambda *a, **k: None
            try:
                total.backward()
            finally:
                ops.offer_colsum = orig
        else:
            total.backward()
        torch.cuda.synchronize()
        return [c.bias.grad.detach().float().cpu().clone() for c in convs]

    a, b = run(False), run(True)
    assert not ops._COLSUMS, "every offered sum was taken by its convolution"
    for ga, gb in zip(a, b):
        scale = float(gb.abs().max()) + 1e-12
        assert float((ga - gb).abs().max()) <= 2e-2 * scale, (float((ga - gb).abs().max()), scale)
        assert float(ga.abs().max()) > 0


@pytest.mark.parametrize("trial", [0, 1, 2])
def test_fused_loss_equals_reference_vectors(trial):
    """the reference's own golden vectors (p in (N,3,H,W,85) fp32): rounded to bf16 maps first, so the expectation is the
    torch formulation on the rounded values; the unrounded reference total must agree to bf16 accuracy."""
    z = np.load(os.path.join(GOLD, "yolov5_loss_%d.npz" % trial))
    p = [torch.from_numpy(z["p/%d" % i]) for i in range(3)]
    raws = [q.permute(0, 1, 4, 2, 3).reshape(q.shape[0], 255, q.shape[2], q.shape[3]).to(torch.bfloat16).to(dev())
            .contiguous(memory_format=torch.channels_last) for q in p]
    t = torch.from_numpy(z["targets"])
    pad = torch.zeros(40 - t.shape[0], 6)
    pad[:, 0] = -1
    pad[:, 2:] = 0.5
    tg = torch.cat([t, pad], 0).to(dev())
    _check(raws, tg)
    total_f, _ = yolov5.YOLOv5LossFused(80)(raws, tg)
    assert abs(float(total_f) - float(z["total"])) <= 1e-2 * abs(float(z["total"]))


def test_model_fused_loss_matches_unfused_and_captures_one_graph():
    from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
    from cvpytorch_amd.data import synthetic_detection_batch
    torch.manual_seed(0)
    a = yolov5.YOLOv5(80, "s", max_targets=64).to(dev()).train()
    b = yolov5.YOLOv5(80, "s", max_targets=64, fused_loss=True).to(dev()).train()
    b.load_state_dict(a.state_dict(), strict=False)
    imgs, targets = synthetic_detection_batch(4, 128, device=dev())
    gts = yolov5.targets_to_tensor(targets, 64, dev())
    la = a(imgs, gts, "train")
    la["loss"].backward()
    lb = b(imgs, gts, "train")
    lb["loss"].backward()
    for k in ("loss", "box_loss", "obj_loss", "cls_loss"):
        assert abs(float(la[k]) - float(lb[k])) <= 1e-4 * abs(float(la[k])) + 1e-6, (k, float(la[k]), float(lb[k]))
    pa = dict(a.named_parameters())
    cs = [cosine(p.grad.float(), pa[n].grad.float()) for n, p in b.named_parameters() if p.grad is not None]
    assert min(cs) > 0.995, sorted(cs)[:5]
    # eval / val path with the fused loss
    b.eval()
    with torch.no_grad():
        losses, outs = b(imgs, gts, "val")
    assert len(outs) == 4 and torch.isfinite(losses["loss"]).all()
    # one-graph capture: replay reproduces the eager trajectory within run-to-run (atomic-order) noise
    torch.manual_seed(1)
    m1 = yolov5.YOLOv5(80, "s", max_targets=64, fused_loss=True).to(dev()).train()
    m2 = yolov5.YOLOv5(80, "s", max_targets=64, fused_loss=True).to(dev()).train()
    m2.load_state_dict(m1.state_dict())
    s1, s2 = FlatTrainState(m1), FlatTrainState(m2)
    e, gph = FlatTrainStep(m1, s1), FlatTrainStep(m2, s2)
    gph.capture(imgs, gts, warmup=2)   # 2 eager warm-up steps, then the capture (which executes nothing)
    assert gph.g1 is not None and gph.g2 is None
    for _ in range(2):
        e(imgs, gts)
    le = lg = None
    for _ in range(3):
        le = float(e(imgs, gts)["loss"])
        lg = float(gph(imgs, gts)["loss"])
    torch.cuda.synchronize()
    assert abs(le - lg) <= 2e-2 * abs(le), (le, lg)
    assert s1.steps == s2.steps == 5
