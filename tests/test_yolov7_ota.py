"""The product's fixed-shape YOLOv7 OTA loss (cvpytorch_amd/yolov7.py: YOLOv7OTALoss) must give the reference's results:
checked against the reference's golden vectors (tests/golden/v7_ota_loss_*.npz, captured from src/losses/yolov7_loss.py) and
against the oracle restatement of the reference loop on seeded inputs. Pure torch — runs on CPU."""
import os

import numpy as np
import pytest
import torch

from cvpytorch_amd import yolov7 as V
from oracle import yolov7_ref as R7

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pad(t, rows):
    pad = torch.zeros((rows - t.shape[0], 6))
    pad[:, 0] = -1
    pad[:, 2:] = 0.5
    return torch.cat([t, pad], 0)


@pytest.mark.parametrize("trial", [0, 1])
def test_dense_ota_equals_reference_vectors(trial):
    z = np.load(os.path.join(GOLD, "v7_ota_loss_%d.npz" % trial))
    p = [torch.from_numpy(z["p/%d" % i]).requires_grad_(True) for i in range(3)]
    t = torch.from_numpy(z["targets"])
    total, stats = V.YOLOv7OTALoss(80, max_per_image=12)(p, _pad(t, 40), int(z["size"][0]))
    assert torch.allclose(total, torch.from_numpy(z["total"]), rtol=1e-5), (float(total), float(z["total"]))
    assert torch.allclose(stats, torch.from_numpy(z["stats"]), rtol=1e-5, atol=1e-7)
    grads = torch.autograd.grad(total, p)
    for i, g in enumerate(grads):
        assert torch.allclose(g, torch.from_numpy(z["grads/%d" % i]), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("seed,bs,size,nmax", [(0, 2, 64, 6), (1, 4, 96, 10), (2, 3, 128, 16)])
def test_dense_ota_equals_oracle(seed, bs, size, nmax):
    g = torch.Generator().manual_seed(seed)
    p = [torch.randn(bs, 3, size // s, size // s, 85, generator=g) for s in (8, 16, 32)]
    rows = []
    for i in range(bs):
        n = int(torch.randint(1, nmax + 1, (1,), generator=g))
        t = torch.zeros(n, 6)
        t[:, 0] = i
        t[:, 1] = torch.randint(0, 80, (n,), generator=g).float()
        t[:, 2:4] = torch.rand(n, 2, generator=g) * 0.8 + 0.1
        t[:, 4:6] = torch.rand(n, 2, generator=g) * 0.4 + 0.05
        rows.append(t)
    targets = torch.cat(rows, 0)
    pr = [q.clone().requires_grad_(True) for q in p]
    lo, so = R7.YOLOv7OTALoss(80)(pr, targets, torch.zeros(bs, 3, size, size))
    go = torch.autograd.grad(lo, pr)
    pd = [q.clone().requires_grad_(True) for q in p]
    ld, sd = V.YOLOv7OTALoss(80, max_per_image=nmax + 2)(pd, _pad(targets, bs * nmax + 5), size)
    gd = torch.autograd.grad(ld, pd)
    assert torch.allclose(ld, lo, rtol=1e-5, atol=1e-6), (float(ld), float(lo))
    assert torch.allclose(sd, so, rtol=1e-5, atol=1e-7)
    for a, b in zip(gd, go):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-7)


def test_flat_to_padded():
    t = torch.tensor([[0, 1, .1, .1, .2, .2], [0, 2, .3, .3, .2, .2], [2, 3, .5, .5, .1, .1], [-1, 0, .5, .5, .5, .5]])
    out, valid = V.flat_to_padded(t, 3, 2)
    assert valid.tolist() == [[True, True], [False, False], [True, False]]
    assert torch.equal(out[0, 1], t[1]) and torch.equal(out[2, 0], t[2])
