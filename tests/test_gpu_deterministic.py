"""Deterministic mode (ops.set_deterministic): two runs of the same train step from the same state give BIT-IDENTICAL gradient arenas,
parameters and losses — no floating-point atomics on the path (dense weight gradients by per-split slabs + an ordered fold,
BatchNorm sums by the partial-row kernels). The default mode (fp32 atomic split-K epilogues) is allowed to differ in the last bits;
both modes agree to rounding. Reference behaviour: torch.use_deterministic_algorithms on the reference's cuDNN path."""
import copy

import pytest
import torch

from cvpytorch_amd import ops, yolov5
from cvpytorch_amd.arena import FlatTrainState
from cvpytorch_amd.data import synthetic_detection_batch

pytestmark = pytest.mark.gpu


def _grads(det, seed=0):
    ops.set_deterministic(det)
    try:
        torch.manual_seed(seed)
        dev = torch.device("cuda:0")
        B = 4
        model = yolov5.YOLOv5(80, "s", max_targets=B * 20, fused_loss=True).to(dev).train()
        state = FlatTrainState(model, use_ema=False)
        imgs, targets = synthetic_detection_batch(B, 256, seed=3, device=dev)
        gts = yolov5.targets_to_tensor(targets, B * 20, dev)
        out = []
        for _ in range(2):
            state.zero_grad()
            state.zero_stats()
            state.prepare_weights()
            losses = model(imgs, gts, "train")
            state.backward(losses["loss"])
            torch.cuda.synchronize()
            out.append((state.grad.clone(), losses["loss"].detach().clone()))
        _grads.names = [(n, p.grad.data_ptr(), p.numel()) for n, p in model.named_parameters() if p.grad is not None]
        _grads.base = state.grad.data_ptr()
        return out
    finally:
        ops.set_deterministic(False)


def test_two_runs_are_bit_identical_in_deterministic_mode():
    a = _grads(True)
    b = _grads(True)
    for (ga, la), (gb, lb) in zip(a, b):
        assert torch.equal(la, lb)
        if not torch.equal(ga, gb):
            bad = [n for n, ptr, cnt in _grads.names
                   if not torch.equal(ga[(ptr - _grads.base) // 4:(ptr - _grads.base) // 4 + cnt], gb[(ptr - _grads.base) // 4:(ptr - _grads.base) // 4 + cnt])]
            raise AssertionError("gradients differ between two deterministic runs: %d tensors, first %s" % (len(bad), bad[:8]))
    # and within one process the same step twice (same weights, same batch) as well
    assert torch.equal(a[0][0], a[1][0])


def test_deterministic_and_default_mode_agree_to_rounding(monkeypatch):
    # The two modes are compared on the SAME convolution kernels: the row-band kernel (which the default mode takes for these small maps
    # since round 6) does not run in deterministic mode (it folds BatchNorm sums with atomics), and on a randomly initialised network
    # with train-mode BatchNorm over 4 x 8 x 8 samples a different fp32 accumulation order in the forward convolutions alone moves the
    # gradients by several percent (measured: per-layer activations drift from 1 - cos = 3e-9 at stage 1 to 5e-3 at the last neck
    # block, gradients to cos 0.94, with bit-exact, run-to-run reproducible kernels on both sides — DESIGN.md §5).
    monkeypatch.setenv("CVHIP_BAND", "0")
    a = _grads(True)[0][0]
    b = _grads(False)[0][0]
    cos = torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=0)
    assert float(cos) > 0.999, float(cos)
    assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max()) + 1e-6
