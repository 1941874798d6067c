"""GPU parity of the one-launch BatchNorm + activation backward (csrc/bn_act.hip bn_bwd_fused_kernel, cvhip_bn_act_bwd_fused_acc: the
layer's (dz, y) slice stays in registers across a device-wide barrier) against the two passes it replaces (cvhip_bn_act_bwd_sums_acc,
then cvhip_bn_act_bwd_apply_acc) and against fp32 torch arithmetic — aten::native_batch_norm_backward + silu_backward of
conv_module.py:211-213 under trainer.py:189.

Tolerances: dgamma / dbeta relative 1e-5 against the two-pass form (same fp32 per-thread sums in another partition, fp64 accumulation);
dy: the two forms differ only through those sums, so at most one 16-bit rounding step on a handful of elements — max |diff| <= 2^-7 of
the tensor's max and relative L2 <= 1e-3; against fp32 torch: relative L2 <= 4e-3 (16-bit output)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

import test_gpu_kernels as K
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

dev, rel_l2 = K.dev, K.rel_l2
SH = L.BN_ACC_SHARDS


def _inputs(M, Cc, seed, ld_extra=0):
    g = torch.Generator().manual_seed(seed)
    ld = Cc + ld_extra
    dzb = torch.randn(M, ld, generator=g).to(K.BF).to(dev())
    yb = (torch.randn(M, ld, generator=g) * 1.5 + 0.3).to(K.BF).to(dev())
    dz, y = dzb[:, ld_extra:], yb[:, ld_extra:]        # channel slices [ld_extra, ld) of wider buffers
    gamma = (torch.rand(Cc, generator=g) + 0.5).to(dev())
    beta = (torch.randn(Cc, generator=g) * 0.2).to(dev())
    yf = y.float()
    mean = yf.mean(0)
    invstd = 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-3)
    scale = gamma * invstd
    shift = beta - mean * scale
    return dzb, yb, dz, y, ld, mean.contiguous(), invstd.contiguous(), scale.contiguous(), shift.contiguous()


def _two_pass(dz, y, ld, M, Cc, st4, act):
    mean, invstd, scale, shift = st4
    acc = torch.zeros(SH * 2 * Cc, dtype=torch.float64, device=dev())
    dy = torch.full((M, Cc), float("nan"), dtype=K.BF, device=dev())
    dg, db = torch.empty(Cc, device=dev()), torch.empty(Cc, device=dev())
    s = ops._stream()
    L.call("cvhip_bn_act_bwd_sums_acc", dz.data_ptr(), ld, y.data_ptr(), ld, M, Cc, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
           act, 0.0, acc.data_ptr(), Cc, s)
    L.call("cvhip_bn_act_bwd_apply_acc", dz.data_ptr(), ld, y.data_ptr(), ld, dy.data_ptr(), Cc, M, Cc, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
           invstd.data_ptr(), acc.data_ptr(), Cc, dg.data_ptr(), db.data_ptr(), 0, act, 0.0, s)
    torch.cuda.synchronize()
    return dy, dg, db


def _one_launch(dz, y, ld, M, Cc, st4, act, bar, accumulate=0, dg=None, db=None):
    mean, invstd, scale, shift = st4
    acc = torch.zeros(SH * 2 * Cc, dtype=torch.float64, device=dev())
    dy = torch.full((M, Cc), float("nan"), dtype=K.BF, device=dev())
    dg = torch.empty(Cc, device=dev()) if dg is None else dg
    db = torch.empty(Cc, device=dev()) if db is None else db
    L.call("cvhip_bn_act_bwd_fused_acc", dz.data_ptr(), ld, y.data_ptr(), ld, dy.data_ptr(), Cc, M, Cc, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(),
           invstd.data_ptr(), acc.data_ptr(), Cc, dg.data_ptr(), db.data_ptr(), accumulate, act, 0.0, bar.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    return dy, dg, db


CASES = [
    # M, C, activation, extra pitch
    (102400, 128, L.ACT_SILU, 0),     # 40 x 40 x 64 images x 128 channels: 13 visits (the 16-visit instance)
    (25600, 256, L.ACT_SILU, 0),      # 20 x 20: 7 visits (the 8-visit instance)
    (25600, 512, L.ACT_SILU, 0),
    (100003, 64, L.ACT_SILU, 64),     # ragged row count, operands are channel slices of 128-wide buffers
    (8192, 256, L.ACT_RELU, 0),       # ResNet bottleneck interior at batch 16
    (30000, 128, L.ACT_NONE, 0),
    (40000, 64, L.ACT_LEAKY, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_one_launch_equals_two_passes_and_torch(case):
    M, Cc, act, extra = case
    assert L.load().cvhip_bn_act_bwd_fused_ok(M, Cc) == 1
    dzb, yb, dz, y, ld, mean, invstd, scale, shift = _inputs(M, Cc, 17 + Cc, extra)
    st4 = (mean, invstd, scale, shift)
    bar = torch.zeros(4, dtype=torch.int32, device=dev())
    ref_dy, ref_dg, ref_db = _two_pass(dz, y, ld, M, Cc, st4, act)
    for rep in range(3):                                   # the barrier words re-arm themselves: back-to-back launches on the same words
        dy, dg, db = _one_launch(dz, y, ld, M, Cc, st4, act, bar)
        assert bar.tolist() == [0, 0, 0, 0], bar.tolist()   # no block gave up waiting; counters back at zero
        assert torch.isfinite(dy.float()).all()
        assert rel_l2(dg, ref_dg) < 1e-5 and rel_l2(db, ref_db) < 1e-5
        d = (dy.float() - ref_dy.float()).abs()
        assert float(d.max()) <= 2 ** -7 * float(ref_dy.float().abs().max()) and rel_l2(dy.float(), ref_dy.float()) < 1e-3
    # fp32 torch arithmetic on the same 16-bit operands
    yf, dzf = y.float(), dz.float()
    u = yf * scale + shift
    if act == L.ACT_SILU:
        sg = torch.sigmoid(u)
        du = dzf * (sg * (1 + u * (1 - sg)))
    elif act == L.ACT_RELU:
        du = dzf * (u > 0).float()
    elif act == L.ACT_LEAKY:
        du = dzf * torch.where(u > 0, torch.ones_like(u), torch.zeros_like(u))   # (slope 0.0 passed above)
    else:
        du = dzf
    xh = (yf - mean) * invstd
    exp = scale * (du - du.mean(0) - xh * (du * xh).mean(0))
    assert rel_l2(dy.float(), exp) < 4e-3
    assert rel_l2(db, du.sum(0)) < 1e-4 and rel_l2(dg, (du * xh).sum(0)) < 1e-4


def test_accumulate_into_existing_parameter_gradients():
    M, Cc, act = 25600, 256, L.ACT_SILU
    dzb, yb, dz, y, ld, mean, invstd, scale, shift = _inputs(M, Cc, 5)
    st4 = (mean, invstd, scale, shift)
    bar = torch.zeros(4, dtype=torch.int32, device=dev())
    _, g0, b0 = _one_launch(dz, y, ld, M, Cc, st4, act, bar)
    base_g, base_b = torch.randn(Cc, device=dev()), torch.randn(Cc, device=dev())
    _, g1, b1 = _one_launch(dz, y, ld, M, Cc, st4, act, bar, accumulate=1, dg=base_g.clone(), db=base_b.clone())
    assert torch.allclose(g1, base_g + g0, rtol=1e-5, atol=1e-4) and torch.allclose(b1, base_b + b0, rtol=1e-5, atol=1e-4)


def test_geometry_refusals():
    ok = L.load().cvhip_bn_act_bwd_fused_ok
    assert ok(409600, 128) == 0          # 80 x 80 x 64 x 128 channels: 105 MB of (dz + y) do not fit the register file
    assert ok(102400, 96) == 0           # 12 channel vectors: not a power of two
    assert ok(102400, 32) == 0           # < 64 channels
    assert ok(2000, 128) == 0            # fewer row passes than CUs: nothing to gain
    assert ok(102400, 128) == 1 and ok(25600, 512) == 1
    dzb, yb, dz, y, ld, mean, invstd, scale, shift = _inputs(409600, 128, 1)
    acc = torch.zeros(SH * 2 * 128, dtype=torch.float64, device=dev())
    dy = torch.empty((409600, 128), dtype=K.BF, device=dev())
    bar = torch.zeros(4, dtype=torch.int32, device=dev())
    rc = L.load().cvhip_bn_act_bwd_fused_acc(dz.data_ptr(), ld, y.data_ptr(), ld, dy.data_ptr(), 128, 409600, 128, scale.data_ptr(), shift.data_ptr(),
                                             mean.data_ptr(), invstd.data_ptr(), acc.data_ptr(), 128, None, None, 0, L.ACT_SILU, 0.0, bar.data_ptr(), ops._stream())
    assert rc == L.ERR_UNSUPPORTED


def test_train_step_with_and_without_the_one_launch_form(monkeypatch):
    """a YOLOv5-s step (batch 8, 256 x 256: its 16 x 16 and 8 x 8 layers qualify) with the fused backward on and off: same loss, gradient
    arenas equal to the summation-order tolerance"""
    import importlib
    from cvpytorch_amd import arena, yolov5
    from cvpytorch_amd.data import synthetic_detection_batch
    outs = []
    for flag in (True, False):
        monkeypatch.setattr(ops, "_BN_FUSED_BWD", flag)
        torch.manual_seed(11)
        m = yolov5.YOLOv5(80, "s", fused_loss=True).to(dev()).train()
        imgs, tg = synthetic_detection_batch(32, 320, seed=3, device=dev())
        gts = yolov5.targets_to_tensor(tg, 32 * 20, dev())
        state = arena.FlatTrainState(m, lr=0.0, use_ema=False)
        names = []
        orig = ops._timed_ew

        def spy(kname, *a, **k):
            names.append(kname)
            return orig(kname, *a, **k)

        monkeypatch.setattr(ops, "_timed_ew", spy)
        state.prepare_weights()
        losses = m(imgs, gts, "train")
        state.backward(losses["loss"])
        torch.cuda.synchronize()
        monkeypatch.setattr(ops, "_timed_ew", orig)
        outs.append((float(losses["loss"]), state.grad.clone(), sum(1 for n in names if n.startswith("bn_act_bwd_fused"))))
    (l1, g1, n1), (l0, g0, n0) = outs
    assert n1 > 0 and n0 == 0, (n1, n0)
    assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
    assert rel_l2(g1, g0) < 2e-2, rel_l2(g1, g0)
