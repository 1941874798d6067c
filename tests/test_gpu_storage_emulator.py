"""Proof of the gradient-noise claim: the engine against tests/storage_emulator.py — the oracle with the
engine's 16-bit rounding points restated on the CPU (and nothing else changed).

  * emulator(fp32 rounding = identity) == oracle                      (the emulator's hand-written backward formulas are right; CPU)
  * emulator(bf16) vs fp32 oracle: median per-parameter cosine ~0.91  (storage rounding ALONE produces the model-level gap; CPU)
  * two emulators with IDENTICAL rounding points whose fp32 convolution sums differ only in summation order (fp32 vs
    exactly-rounded) agree with each other to a median cosine of only ~0.955 in bf16: a 1e-7 change before a rounding flips the
    rounding of ~1 % of the elements by a whole ulp, and that alone is ~1/3 of the storage noise. This is the CEILING for any two
    correct implementations of the same storage format — bit-level agreement of gradients is not a meaningful criterion here. (CPU)
  * TEACHER-FORCED (GPU): every block of the engine (18 blocks = every conv / BN parameter of YOLOv5-s), fed the activations and the
    output gradient the emulator's block saw, reproduces the emulator's output, input gradient and parameter gradients: single
    layers to cosine 1.000000 / 1e-5 relative, 9-layer CSP blocks to >= 0.99997. The engine computes the storage-rounding model.
  * FREE-RUNNING engine vs emulator vs oracle, same weights / batch (GPU): (a) the engine's deviation from the fp32 oracle has the magnitude
    storage rounding alone produces (median cosines within 0.02 of each other); (b) it is largely the SAME deviation: the engine
    agrees with the emulator as well as the emulator agrees with a second realisation of itself (ceiling - 0.02), far better than
    two independent noise realisations of that size would (product of the two cosines to the oracle); (c) losses agree to 2e-3.
    fp16 runs with a loss scale (as the product does: arena.FlatTrainState(loss_scaling)) in the engine AND the emulator."""
import numpy as np
import pytest
import torch

import storage_emulator as E
from oracle import torch_ref as R


def _cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    if float(a.norm()) == 0.0 and float(b.norm()) == 0.0:      # a branch the loss does not reach (no positives on a level): 0 == 0
        return 1.0
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


def _grads(model, imgs, targets, loss_scale=1.0):
    loss = model(imgs, targets, "train")["loss"]
    (loss * loss_scale).backward()
    return float(loss.detach())


FP16_LOSS_SCALE = 4096.0


def two_realisations(dt, variant="s", batch=4, size=128):
    """-> (cosines emulator~emulator(acc64), cosines emulator~oracle): the agreement ceiling for one storage format"""
    ref, imgs, targets = _setup(variant, batch, size)
    _grads(ref, imgs, targets)
    ms = []
    for a64 in (False, True):
        m = R.YOLOv5(80, variant).train()
        m.load_state_dict(ref.state_dict())
        E.emulate_storage(m, dt, acc64=a64)
        _grads(m, imgs, targets, FP16_LOSS_SCALE if dt == torch.float16 else 1.0)
        ms.append(dict(m.named_parameters()))
    rp = dict(ref.named_parameters())
    return (np.array([_cos(ms[0][n].grad, ms[1][n].grad) for n in ms[0]]), np.array([_cos(ms[0][n].grad, rp[n].grad) for n in ms[0]]))


def test_two_exact_realisations_of_bf16_storage_only_agree_to_the_rounding_flip_ceiling():
    ab, ao = two_realisations(torch.bfloat16)
    assert 0.92 < np.median(ab) < 0.985, np.median(ab)          # NOT 0.999: rounding decisions flip under 1e-7 perturbations
    assert np.median(ab) > np.median(ao) + 0.015, (np.median(ab), np.median(ao))   # ... but the two share most of their noise


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
    ls = FP16_LOSS_SCALE if precision == "fp16" else 1.0
    lo = _grads(ref, imgs, targets)
    le = _grads(emu, imgs, targets, ls)
    emu2 = R.YOLOv5(80, variant).train()
    emu2.load_state_dict(ref.state_dict())
    E.emulate_storage(emu2, dt, acc64=True)
    _grads(emu2, imgs, targets, ls)
    ops.set_precision(precision)
    try:
        hip = yolov5.YOLOv5(80, variant, max_targets=64, fused_loss=True)
        hip.load_state_dict(ref.state_dict(), strict=False)
        hip.to(dev).train()
        # flat arenas: sibling pairs + fused 1x1 backward are live
        state = FlatTrainState(hip, use_ema=False, loss_scaling=precision == "fp16", init_scale=ls)
        gts = yolov5.targets_to_tensor([{k: v.to(dev) for k, v in t.items()} for t in targets], 64, dev)
        lh = hip(imgs.to(dev), gts, "train")["loss"]
        state.scale_loss(lh).backward()
        torch.cuda.synchronize()
        ep, rp, e2 = dict(emu.named_parameters()), dict(ref.named_parameters()), dict(emu2.named_parameters())
        names = [n for n, p in hip.named_parameters() if n in ep and p.grad is not None]
        out = dict(loss_engine=float(lh), loss_emulator=le, loss_oracle=lo, names=names,
                   cos_emulator_emulator=[_cos(ep[n].grad, e2[n].grad) for n in names],
                   cos_engine_emulator=[_cos(dict(hip.named_parameters())[n].grad, ep[n].grad) for n in names],
                   cos_engine_oracle=[_cos(dict(hip.named_parameters())[n].grad, rp[n].grad) for n in names],
                   cos_emulator_oracle=[_cos(ep[n].grad, rp[n].grad) for n in names])
    finally:
        ops.set_precision("bf16")
    return out


def _family(kind):
    """-> (oracle factory, engine factory, block classes, training-loss callable, batch, size) for 'yolov5s' | 'yoloxs' | 'yolov7l'"""
    if kind == "yolov5s":
        from cvpytorch_amd import yolov5
        return (lambda: R.YOLOv5(80, "s"), lambda: yolov5.YOLOv5(80, "s", max_targets=64, fused_loss=True), (R.ConvModule, R.CSPLayer, R.SPPF),
                lambda m, imgs, targets: m(imgs, targets, "train")["loss"], 4, 128)
    if kind == "yoloxs":
        from cvpytorch_amd import yolox
        from oracle import yolox_ref as RX

        def loss(m, imgs, targets):
            return m.loss(m.head(m.neck(m.backbone(imgs))), RX.targets_to_padded(targets))["loss"]
        return (lambda: RX.YOLOX(80, "s"), lambda: yolox.YOLOX(80, "s", max_labels=12), (R.ConvModule, R.CSPLayer, R.SPPF), loss, 4, 128)
    from cvpytorch_amd import yolov7
    from oracle import yolov7_ref as R7
    return (lambda: R7.YOLOv7(80, width_mul=1.0), lambda: yolov7.YOLOv7(80, width_mul=1.0, max_targets=64), (R7.Conv,),
            lambda m, imgs, targets: m(imgs, targets, "train")["loss"], 2, 128)


def blocks_teacher_forced(precision, kind="yolov5s"):
    """Every single-input block of the network (ConvModule / CSPLayer / SPPF with no such ancestor; every Conv of YOLOv7-l),
    TEACHER-FORCED: the engine's block is fed the activations and the output gradient the emulator's block saw in a full training
    step, so no upstream divergence enters. -> list of dict(name, kind, out_rel, dx_cos, param_cos_min, n_params)"""
    from cvpytorch_amd import ops
    from cvpytorch_amd.arena import FlatTrainState
    dev = torch.device("cuda:0")
    dt = torch.float16 if precision == "fp16" else torch.bfloat16
    ls = FP16_LOSS_SCALE if precision == "fp16" else 1.0
    make_ref, make_hip, classes, loss_of, batch, size = _family(kind)
    torch.manual_seed(0)
    ref = make_ref().train()
    imgs, targets = R.synthetic_batch(batch, size, seed=1029, max_boxes=8)
    emu = make_ref().train()
    emu.load_state_dict(ref.state_dict())
    E.emulate_storage(emu, dt)
    blocks, picked = {}, []
    for name, m in emu.named_modules():
        if isinstance(m, classes) and not any(name.startswith(p + ".") for p in picked):
            picked.append(name)
            rec = blocks[name] = dict(kind=type(m).__name__)
            m.register_forward_hook(lambda mod, inp, out, rec=rec: rec.update(x=inp[0].detach().clone(), out=out.detach().clone(), calls=rec.get("calls", 0) + 1))
            m.register_full_backward_hook(lambda mod, gi, go, rec=rec: rec.update(dx=None if gi[0] is None else gi[0].detach().clone(), dout=go[0].detach().clone()))
    (loss_of(emu, imgs, targets) * ls).backward()
    ep = dict(emu.named_parameters())
    ops.set_precision(precision)
    rows = []
    try:
        hip = make_hip()
        hip.load_state_dict(ref.state_dict(), strict=False)
        hip.to(dev).train()
        state = FlatTrainState(hip, use_ema=False, loss_scaling=precision == "fp16", init_scale=ls)  # flat arenas: sibling pairs + fused 1x1 backward are live
        hm = dict(hip.named_modules())
        for name in picked:
            rec, mod = blocks[name], hm[name]
            if "dout" not in rec or rec.get("calls", 0) != 1:
                continue                      # a block the loss does not reach / a module applied several times in one forward (YOLOv7's
                                              # FeatureFusion.conv4, yolov7_modules.py:113-120): its recorded input and gradient belong to different calls
            x = rec["x"].to(dev)
            if x.shape[1] != 3:               # image-fed stems take the fp32 image; every other block a 16-bit activation
                x = x.to(dt).contiguous(memory_format=torch.channels_last)
            x.requires_grad_(rec["dx"] is not None)
            out = mod(x)
            out.backward(rec["dout"].to(dev).to(out.dtype).contiguous(memory_format=torch.channels_last))
            torch.cuda.synchronize()
            pc = [_cos(p.grad, ep[name + "." + n].grad) for n, p in mod.named_parameters()]
            rows.append(dict(name=name, kind=rec["kind"], n_params=len(pc), param_cos_min=min(pc),
                             out_rel=float((out.detach().float().cpu() - rec["out"]).norm() / rec["out"].norm()),
                             dx_cos=None if rec["dx"] is None else _cos(x.grad, rec["dx"])))
    finally:
        ops.set_precision("bf16")
    return rows


@pytest.mark.gpu
@pytest.mark.parametrize("precision,kind,min_blocks", [("bf16", "yolov5s", 18), ("fp16", "yolov5s", 18), ("bf16", "yoloxs", 30), ("fp16", "yolov7l", 70)])
def test_every_block_teacher_forced_matches_the_emulator(precision, kind, min_blocks):
    rows = blocks_teacher_forced(precision, kind)
    assert len(rows) >= min_blocks and sum(r["n_params"] for r in rows) > 150       # (nearly) every conv / BN parameter of the network is inside one block
    ulp = 2.0 ** -8 if precision == "bf16" else 2.0 ** -11
    bad = []
    for r in rows:
        single = r["kind"] in ("ConvModule", "Conv")      # one layer: only its own output rounding can flip; a block chains up to 9 layers
        cos_min = 0.99999 if single else 0.9995
        if (r["out_rel"] > (0.3 if single else 4.0) * ulp or r["param_cos_min"] < cos_min or (r["dx_cos"] is not None and r["dx_cos"] < cos_min)):
            bad.append(r)
    assert not bad, bad


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["bf16", "fp16"])
def test_engine_deviation_from_fp32_is_the_storage_format(precision):
    r = engine_vs_emulator(precision)
    ee, eo, mo, mm = (np.median(r[k]) for k in ("cos_engine_emulator", "cos_engine_oracle", "cos_emulator_oracle", "cos_emulator_emulator"))
    msg = dict(engine_emulator=ee, engine_oracle=eo, emulator_oracle=mo, emulator_emulator=mm)
    assert len(r["names"]) > 150
    assert abs(r["loss_engine"] - r["loss_emulator"]) <= 2e-3 * abs(r["loss_emulator"]), r
    # (a) same magnitude of deviation from fp32 as storage rounding alone
    assert abs(eo - mo) <= 0.02, msg
    # (b) and mostly the same deviation: at the two-realisation ceiling, well above independent noise of that size
    assert ee >= mm - 0.02, msg
    if precision == "bf16":
        assert ee >= eo * mo + 0.05, msg
    worst = min(r["cos_engine_emulator"])
    assert worst >= min(r["cos_emulator_emulator"]) - 0.05, (worst, min(r["cos_emulator_emulator"]))
