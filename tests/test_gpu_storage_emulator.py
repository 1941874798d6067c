"""Proof of the gradient-noise claim: the engine against tests/storage_emulator.py — the oracle with the
engine's 16-bit rounding points restated on the CPU (and nothing else changed).

  * emulator(fp32 rounding = identity) == oracle                      (the emulator's hand-written backward formulas are right; CPU)
  * emulator(bf16) vs fp32 oracle: median per-parameter cosine ~0.9   (storage rounding ALONE produces the model-level gap; CPU)
  * ENGINE vs emulator(bf16), same weights / batch: loss to 1e-3, per-parameter gradient cosine >= 0.995 median, >= 0.97 worst —
    i.e. an order of magnitude closer than either is to the fp32 oracle: what separates the engine from the fp32 oracle is
    where 16-bit values are stored, not what the kernels compute. Same for fp16 storage."""
import numpy as np
import pytest
import torch

import storage_emulator as E
from oracle import torch_ref as R


def _cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def _setup(variant="s", batch=4, size=128):
    torch.manual_seed(0)
    ref = R.YOLOv5(80, variant).train()
    imgs, targets = R.synthetic_batch(batch, size, seed=1029, max_boxes=8)
    return ref, imgs, targets


def test_emulator_without_rounding_is_the_oracle():
    ref, imgs, targets = _setup("n", 2, 96)
    emu = R.YOLOv5(80, "n").train()
    emu.load_state_dict(ref.state_dict())
    E.emulate_storage(emu, torch.float32)
    lr = ref(imgs, targets, "train")["loss"]
    lr.backward()
    le = emu(imgs, targets, "train")["loss"]
    le.backward()
    assert abs(float(lr) - float(le)) <= 1e-5 * abs(float(lr))
    rp = dict(ref.named_parameters())
    assert min(_cos(p.grad, rp[n].grad) for n, p in emu.named_parameters()) > 0.99999


def test_storage_rounding_alone_explains_the_gap_to_the_fp32_oracle():
    ref, imgs, targets = _setup("s", 4, 128)
    emu = R.YOLOv5(80, "s").train()
    emu.load_state_dict(ref.state_dict())
    E.emulate_storage(emu, torch.bfloat16)
    ref(imgs, targets, "train")["loss"].backward()
    emu(imgs, targets, "train")["loss"].backward()
    rp = dict(ref.named_parameters())
    cs = [_cos(p.grad, rp[n].grad) for n, p in emu.named_parameters()]
    assert 0.8 < np.median(cs) < 0.97, np.median(cs)      # far from 0.999 with no kernel in the picture at all


def engine_vs_emulator(precision, variant="s", batch=4, size=128):
    """-> dict(loss_engine, loss_emulator, loss_oracle, cos_engine_emulator [per parameter], cos_engine_oracle, cos_emulator_oracle)"""
    from cvpytorch_amd import ops, yolov5
    from cvpytorch_amd.arena import FlatTrainState
    dev = torch.device("cuda:0")
    dt = torch.float16 if precision == "fp16" else torch.bfloat16
    ref, imgs, targets = _setup(variant, batch, size)
    emu = R.YOLOv5(80, variant).train()
    emu.load_state_dict(ref.state_dict())
    E.emulate_storage(emu, dt)
    lo = ref(imgs, targets, "train")["loss"]
    lo.backward()
    le = emu(imgs, targets, "train")["loss"]
    le.backward()
    ops.set_precision(precision)
    try:
        hip = yolov5.YOLOv5(80, variant, max_targets=64, fused_loss=True)
        hip.load_state_dict(ref.state_dict(), strict=False)
        hip.to(dev).train()
        state = FlatTrainState(hip, use_ema=False, loss_scaling=False)   # flat arenas: sibling pairs + fused 1x1 backward are live
        gts = yolov5.targets_to_tensor([{k: v.to(dev) for k, v in t.items()} for t in targets], 64, dev)
        lh = hip(imgs.to(dev), gts, "train")["loss"]
        lh.backward()
        torch.cuda.synchronize()
        ep, rp = dict(emu.named_parameters()), dict(ref.named_parameters())
        names = [n for n, p in hip.named_parameters() if n in ep and p.grad is not None]
        out = dict(loss_engine=float(lh), loss_emulator=float(le), loss_oracle=float(lo), names=names,
                   cos_engine_emulator=[_cos(dict(hip.named_parameters())[n].grad, ep[n].grad) for n in names],
                   cos_engine_oracle=[_cos(dict(hip.named_parameters())[n].grad, rp[n].grad) for n in names],
                   cos_emulator_oracle=[_cos(ep[n].grad, rp[n].grad) for n in names])
    finally:
        ops.set_precision("bf16")
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_engine_matches_the_storage_emulator(precision):
    r = engine_vs_emulator(precision)
    ee, eo, mo = np.array(r["cos_engine_emulator"]), np.array(r["cos_engine_oracle"]), np.array(r["cos_emulator_oracle"])
    assert abs(r["loss_engine"] - r["loss_emulator"]) <= 1e-3 * abs(r["loss_emulator"]), r
    assert len(ee) > 150
    worst = [(round(c, 4), n) for c, n in sorted(zip(ee, r["names"]))[:5]]
    assert np.median(ee) >= 0.995 and ee.min() >= 0.97, (np.median(ee), worst)
    # the engine is (much) closer to the emulator than either is to the fp32 oracle
    assert (1 - np.median(ee)) * 5 < (1 - np.median(eo)), (np.median(ee), np.median(eo), np.median(mo))
