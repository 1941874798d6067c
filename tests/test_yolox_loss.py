"""The product's dense fixed-shape SimOTA loss (cvpytorch_amd/yolox.py) must reproduce the reference's per-image loop:
checked against the oracle restatement of src/losses/det/yolox_loss.py (pinned to the reference by
tests/test_oracle_golden.py) AND against the reference's golden vectors directly. Assignments (fg mask, matched gt) are
integer results and must be identical; loss values / gradients to fp32 round-off (the dense form sums the class cost in a
different order: rtol 1e-5)."""
import os

import numpy as np
import pytest
import torch

from cvpytorch_amd import yolox as X
from oracle import yolox_ref as RX

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _to_product(p_nchw):
    """(B, C, H, W) raw maps -> product layout (B, HW, C) + hw list"""
    hw = [(int(q.shape[2]), int(q.shape[3])) for q in p_nchw]
    return [q.permute(0, 2, 3, 1).reshape(q.shape[0], -1, q.shape[1]) for q in p_nchw], hw


def _check(p_nchw, targets, expect, expect_assign, gpad=0):
    pp = [q.detach().clone().requires_grad_(True) for q in p_nchw]
    feats, hw = _to_product(pp)
    if gpad:
        targets = torch.cat([targets, torch.zeros(targets.shape[0], gpad, 5)], 1)
    out, (fg, matched, miou) = X.YOLOXLoss(80)(feats, targets, hw=hw, return_assign=True)
    for k in ("loss", "conf_loss", "cls_loss", "iou_loss", "num_fg"):
        assert torch.allclose(out[k], torch.as_tensor(expect[k]).reshape(out[k].shape), rtol=1e-5, atol=1e-6), k
    j = 0
    nlabel = (targets.sum(2) > 0).sum(1)
    for b in range(targets.shape[0]):
        if int(nlabel[b]) == 0:
            assert not bool(fg[b].any())
            continue
        efg, egt, eiou = expect_assign[j]
        j += 1
        assert torch.equal(fg[b], efg)
        assert torch.equal(matched[b][efg], egt)
        assert torch.allclose(miou[b][efg], eiou, rtol=1e-5, atol=1e-7)
    return torch.autograd.grad(out["loss"], pp)


@pytest.mark.parametrize("trial", [0, 1, 2])
@pytest.mark.parametrize("gpad", [0, 7])
def test_dense_simota_equals_reference_vectors(trial, gpad):
    z = np.load(os.path.join(GOLD, "yolox_loss_%d.npz" % trial))
    p = [torch.from_numpy(z["p/%d" % i]) for i in range(3)]
    n_assign = len([k for k in z.files if k.startswith("fg/")])
    assigns = [(torch.from_numpy(z["fg/%d" % i]), torch.from_numpy(z["matched_gt/%d" % i]), torch.from_numpy(z["matched_iou/%d" % i]))
               for i in range(n_assign)]
    grads = _check(p, torch.from_numpy(z["targets"]), {k: z[k] for k in ("loss", "conf_loss", "cls_loss", "iou_loss", "num_fg")}, assigns, gpad)
    for i, g in enumerate(grads):
        assert torch.allclose(g, torch.from_numpy(z["grads/%d" % i]), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("seed,bs,size,nmax", [(0, 2, 64, 6), (1, 4, 96, 20), (2, 3, 128, 12)])
def test_dense_simota_equals_oracle(seed, bs, size, nmax):
    g = torch.Generator().manual_seed(seed)
    p = [torch.randn(bs, 85, size // s, size // s, generator=g) * 0.6 for s in (8, 16, 32)]
    for q in p:
        q[:, 4:] -= 1.5
    _, tg = RX.synthetic_batch(bs, size, seed=seed, max_boxes=nmax)
    targets = RX.targets_to_padded(tg)
    pr = [q.clone().requires_grad_(True) for q in p]
    out, assigns = RX.YOLOXLoss(80)(pr, targets, return_assign=True)
    go = torch.autograd.grad(out["loss"], pr)
    grads = _check(p, targets, out, [a for a in assigns if a is not None], gpad=3)
    for a, b in zip(grads, go):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-7)


def test_targets_to_padded_static_shape():
    _, tg = RX.synthetic_batch(3, 64, seed=5, max_boxes=4)
    t = X.targets_to_padded(tg, max_labels=9)
    assert tuple(t.shape) == (3, 9, 5)
    assert torch.equal(t[:, :RX.targets_to_padded(tg).shape[1]], RX.targets_to_padded(tg))
    with pytest.raises(ValueError):
        X.targets_to_padded(tg, max_labels=1) if max(x["labels"].shape[0] for x in tg) > 1 else (_ for _ in ()).throw(ValueError())
