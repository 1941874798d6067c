"""fp16 storage path (BASELINE config 5; reference: torch.cuda.amp.autocast(fp16) + GradScaler, trainer.py:179-201).

Every kernel that touches 16-bit tensors is compiled a second time with fp16 storage (entry points cvhip_*_f16, selected by
ops.set_precision("fp16")). This file re-runs the kernel parity tests of test_gpu_kernels.py / test_gpu_bwd1x1.py under fp16
(same references, same tolerances: fp16 carries 3 more mantissa bits than bf16), checks the device-side GradScaler against the
reference rule, and runs YOLOv5-s and full-width YOLOv7-l train steps against the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import test_gpu_bwd1x1 as B1
import test_gpu_kernels as K
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops


def dev():
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def fp16_mode(monkeypatch):
    ops.set_precision("fp16")
    monkeypatch.setattr(K, "BF", torch.float16)
    monkeypatch.setattr(B1, "BF", torch.float16)
    yield
    ops.set_precision("bf16")


def test_precision_switch_routes_to_f16_symbols():
    assert L.PRECISION == "fp16" and ops.ACT_DTYPE == torch.float16
    assert L.fn("cvhip_conv2d_fprop") is getattr(L.load(), "cvhip_conv2d_fprop_f16")
    assert L.fn("cvhip_comm_available") is getattr(L.load(), "cvhip_comm_available")   # no 16-bit operands: one copy
    assert len(L._F16) >= 70


@pytest.mark.parametrize("case", K.CONV_CASES)
def test_conv_fprop_fp16(case):
    K.test_conv_fprop(case)


@pytest.mark.parametrize("case", K.CONV_CASES)
def test_conv_fprop_bn_stats_fp16(case):
    K.test_conv_fprop_bn_stats(case)


@pytest.mark.parametrize("case", K.CONV_CASES)
def test_conv_dgrad_fp16(case):
    K.test_conv_dgrad(case)


@pytest.mark.parametrize("case", K.CONV_CASES)
def test_conv_wgrad_fp16(case):
    K.test_conv_wgrad(case)


@pytest.mark.parametrize("act", [L.ACT_SILU, L.ACT_RELU, L.ACT_LEAKY, L.ACT_NONE])
@pytest.mark.parametrize("Cc,M_hw", [(32, (9, 11)), (256, (5, 5)), (20, (4, 4))])
def test_bn_act_fwd_bwd_fp16(act, Cc, M_hw):
    K.test_bn_act_fwd_bwd(act, "fp16", Cc, M_hw)


@pytest.mark.parametrize("Cc,H,W,s,p,d", [(32, 12, 14, 1, 1, 1), (48, 11, 9, 2, 1, 1)])
def test_depthwise_fp16(Cc, H, W, s, p, d):
    K.test_depthwise(Cc, H, W, s, p, d)


def test_glue_ops_fp16():
    K.test_maxpool_exact(5, 1, 2, 32, 13, 11)
    K.test_upsample_cat_exact()
    K.test_cat_add_exact()
    K.test_images_and_focus_layout()
    K.test_head_permute_exact()
    K.test_bilinear(False, 4, 8, 16, 32)
    for shape in ((3, 40, 7, 9), (2, 128, 32, 64), (3, 72, 17, 31)):
        K.test_global_avg_pool(shape)


@pytest.mark.parametrize("case", B1.CASES)
def test_bwd1x1_fused_fp16(case):
    B1.test_bwd1x1_fused_vs_cpu(case)


# ---- dynamic loss scaling ---------------------------------------------------------------------------------------------------
def test_loss_scaler_kernels_follow_gradscaler_rule():
    """cvhip_loss_scale_check / _update / cvhip_sgd_nesterov_ema_scaled over 12 steps with overflows injected at steps 2, 3 and 8,
    growth interval 3: scale, skip decisions, parameters and momentum must equal a host restatement of torch.cuda.amp.GradScaler
    (grow x2 after 3 clean steps in a row, back off x0.5 and skip on inf/NaN) driving torch.optim.SGD."""
    d = dev()
    n = 1000
    torch.manual_seed(0)
    p0 = torch.randn(n)
    param, mom, ema = p0.clone().to(d), torch.zeros(n, device=d), p0.clone().to(d)
    seg = torch.tensor([[0, 600], [600, n]], dtype=torch.int64, device=d)
    seg_lr = torch.tensor([0.1, 0.05], device=d)
    seg_wd = torch.tensor([0.0, 0.01], device=d)
    dyn = torch.tensor([0.9, 1.0], device=d)
    state = torch.tensor([1024.0, 0.0, 0.0, 0.0], device=d)
    sc2 = torch.zeros(2, device=d)
    # host reference
    rp = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([{"params": [rp]}], lr=1.0, momentum=0.9, nesterov=True)
    r_ema = p0.clone()
    scale, tracker, skipped = 1024.0, 0, 0
    lr_vec = torch.cat([torch.full((600,), 0.1), torch.full((n - 600,), 0.05)])
    wd_vec = torch.cat([torch.zeros(600), torch.full((n - 600,), 0.01)])
    rmom = torch.zeros(n)
    for it in range(12):
        g_true = torch.randn(n, generator=torch.Generator().manual_seed(it))
        g = g_true * scale
        if it in (2, 8):
            g[17] = float("inf")
        if it == 3:
            g[999] = float("nan")
        gd = g.to(d)
        st = torch.cuda.current_stream().cuda_stream
        L.call("cvhip_loss_scale_check", gd.data_ptr(), n, state.data_ptr(), st)
        L.call("cvhip_loss_scale_update", state.data_ptr(), sc2.data_ptr(), 2.0, 0.5, 3, st)
        L.call("cvhip_sgd_nesterov_ema_scaled", param.data_ptr(), gd.data_ptr(), mom.data_ptr(), ema.data_ptr(), n, seg.data_ptr(),
               seg_lr.data_ptr(), seg_wd.data_ptr(), 2, 0.9, 1, 0, 0.0, 1.0, dyn.data_ptr(), sc2.data_ptr(), st)
        torch.cuda.synchronize()
        bad = not bool(torch.isfinite(g).all())
        if bad:
            scale, tracker, skipped = scale * 0.5, 0, skipped + 1
        else:
            with torch.no_grad():
                dd = g / (scale) + wd_vec * rp
                rmom = 0.9 * rmom + dd
                dd = dd + 0.9 * rmom
                rp -= lr_vec * dd
            tracker += 1
            if tracker == 3:
                scale, tracker = scale * 2.0, 0
        r_ema = 0.9 * r_ema + 0.1 * rp.detach()
        s_dev = state.tolist()
        assert s_dev[0] == scale and int(s_dev[1]) == tracker and s_dev[2] == 0.0 and int(s_dev[3]) == skipped, (it, s_dev, scale, tracker)
        assert torch.allclose(param.cpu(), rp.detach(), rtol=1e-5, atol=1e-6), it
        assert torch.allclose(mom.cpu(), rmom, rtol=1e-5, atol=1e-6), it
        assert torch.allclose(ema.cpu(), r_ema, rtol=1e-5, atol=1e-6), it
    assert skipped == 3


def _v5(variant="n", batch=2, size=128):
    from cvpytorch_amd import yolov5
    from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
    from cvpytorch_amd.data import synthetic_detection_batch
    from oracle import torch_ref as R
    d = dev()
    torch.manual_seed(0)
    ref = R.YOLOv5(80, variant)
    hip = yolov5.YOLOv5(80, variant, max_targets=64, fused_loss=True)
    hip.load_state_dict(ref.state_dict(), strict=False)
    imgs, targets = synthetic_detection_batch(batch, size, seed=1029, max_boxes=8)
    hip.to(d).train()
    gts = yolov5.targets_to_tensor([{k: v.to(d) for k, v in t.items()} for t in targets], 64, d)
    return ref, hip, imgs, targets, gts, FlatTrainState, FlatTrainStep


def test_overflow_skips_the_step_and_backs_the_scale_off():
    ref, hip, imgs, targets, gts, FlatTrainState, FlatTrainStep = _v5()
    state = FlatTrainState(hip, use_ema=True, growth_interval=2)
    assert state.loss_scaling                                  # on by default under fp16 storage
    step = FlatTrainStep(hip, state)
    d = dev()
    with torch.no_grad():
        state.ls_state[0] = 3.0e38                              # the scaled loss gradient overflows fp16 at once
    p_before, m_before = state.param.clone(), state.mom.clone()
    step(imgs.to(d), gts)
    torch.cuda.synchronize()
    sc, skipped = state.loss_scale()
    assert skipped == 1 and abs(sc / 1.5e38 - 1.0) < 1e-6
    assert torch.equal(state.param, p_before) and torch.equal(state.mom, m_before)     # optimizer.step() was skipped
    assert float(state.grad.abs().max()) == 0.0                                         # ... and the gradients cleared
    with torch.no_grad():
        state.ls_state[0] = 1024.0
    step(imgs.to(d), gts)
    step(imgs.to(d), gts)
    torch.cuda.synchronize()
    sc, skipped = state.loss_scale()
    assert skipped == 1 and sc == 2048.0                                                # two clean steps: grown once
    assert not torch.equal(state.param, p_before)
    assert torch.isfinite(state.param).all()


@pytest.mark.parametrize("capture", [False, True])
def test_yolov5s_fp16_train_step_vs_oracle(capture):
    """fp16 storage + dynamic loss scaling, eager and as ONE hipGraph: the loss equals the fp32 oracle's within 2e-2 (the bf16 bar of
    __graft_entry__.smoke), the update direction equals the bf16 engine's (same weights, same batch)."""
    ref, hip, imgs, targets, gts, FlatTrainState, FlatTrainStep = _v5("s")
    d = dev()
    ref.train()
    lr = float(ref(imgs, targets, "train")["loss"])
    # captured variant: lr = 0, so the eager warm-up steps capture() runs leave the weights where the oracle's are
    state = FlatTrainState(hip, use_ema=False, init_scale=1024.0, lr=0.0 if capture else 0.01)
    step = FlatTrainStep(hip, state)
    x = imgs.to(d)
    if capture:
        step.capture(x, gts)
        x, gts = step.static_imgs, step.static_targets
    p0 = state.param.clone()
    lh = float(step(x, gts)["loss"].detach())
    torch.cuda.synchronize()
    assert abs(lh - lr) <= 2e-2 * abs(lr) + 1e-4, (lh, lr)
    sc, skipped = state.loss_scale()
    assert skipped == 0 and torch.isfinite(state.param).all()
    if capture:
        assert float(state.mom.abs().max()) > 0.0     # the replayed graph produced (unscaled, finite) gradients
        return
    # gradient quality: the first-step momentum buffer is g + wd*p per parameter. Against the fp32 oracle's gradient the fp16 engine
    # must be at least as good as the bf16 engine on the same weights and batch (fp16 stores 3 more mantissa bits)
    from cvpytorch_amd.arena import _dense_view
    ref.zero_grad()
    ref(imgs, targets, "train")["loss"].backward()
    rg = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}

    def grad_cos(model, st):
        a, b = [], []
        for n, p in model.named_parameters():
            if n in rg and id(p) in st.index:
                off = st.offsets[st.index[id(p)]]
                a.append(_dense_view(st.mom, off, p).float().cpu().reshape(-1))
                b.append(rg[n].reshape(-1))
        a, b = torch.cat(a), torch.cat(b)
        return float((a * b).sum() / (a.norm() * b.norm()))

    c16 = grad_cos(hip, state)
    ops.set_precision("bf16")
    ref2, hip2, imgs2, targets2, gts2, _, _ = _v5("s")
    st2 = FlatTrainState(hip2, use_ema=False)
    assert not st2.loss_scaling
    FlatTrainStep(hip2, st2)(imgs2.to(d), gts2)
    torch.cuda.synchronize()
    cbf = grad_cos(hip2, st2)
    assert c16 > 0.8 and c16 >= cbf - 0.02, (c16, cbf)


def test_yolov7l_full_width_fp16_end_to_end_vs_oracle():
    """BASELINE config 5's model at FULL width (37.6 M parameters), reduced resolution (128x128, batch 2) so the CPU oracle finishes
    in seconds: fp16 storage, loss within 2e-2 of the fp32 oracle, finite gradients for every parameter, BN running statistics
    within 3e-2."""
    from cvpytorch_amd import yolov7
    from oracle import torch_ref as R
    from oracle import yolov7_ref as R7
    torch.manual_seed(0)
    ref = R7.YOLOv7(80, width_mul=1.0)
    hip = yolov7.YOLOv7(80, width_mul=1.0, max_targets=64)
    sd = ref.state_dict()
    missing, unexpected = hip.load_state_dict(sd, strict=False)
    assert all(k.startswith("loss.") for k in missing) and not unexpected
    imgs, targets = R.synthetic_batch(2, 128, seed=1029, max_boxes=10)
    ref.train()
    lr = ref(imgs, targets, "train")
    d = dev()
    hip.to(d).train()
    tg = [{k: v.to(d) for k, v in t.items()} for t in targets]
    lh = hip(imgs.to(d), tg, "train")
    (lh["loss"] * 256.0).backward()
    torch.cuda.synchronize()
    for k in ("loss", "box_loss", "obj_loss", "cls_loss"):
        a, b = float(lh[k]), float(lr[k])
        assert abs(a - b) <= 2e-2 * abs(b) + 1e-4, (k, a, b)
    n_grad = 0
    for n, p in hip.named_parameters():
        if p.grad is not None:
            assert torch.isfinite(p.grad).all(), n
            n_grad += 1
    assert n_grad > 200
    rb = dict(ref.named_buffers())
    for n, b in hip.named_buffers():
        if "running_var" in n and "conv5" not in n and "conv6" not in n:
            assert K.rel_l2(b.float(), rb[n]) < 3e-2, n
