"""Fused AdamW + EMA over the flat arenas (csrc/weights_optim.hip cvhip_adamw_ema) against torch.optim.AdamW — the optimizer
conf/mini-imagenet.yml:91-99 (BASELINE config 1) is written with (src/optimizers/__init__.py:71-73): decoupled weight decay,
per-segment lr / weight decay, bias corrections from a DEVICE-side step counter (so a captured step replays correctly), the
GradScaler skip, and the whole config-1 train step through arena.FlatTrainState(optimizer="adamw")."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvpytorch_amd import lib as L
from cvpytorch_amd import ops


def dev():
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


def test_adamw_kernel_matches_torch_over_10_steps():
    torch.manual_seed(0)
    sizes = [4096, 40, 1003, 8, 77777]             # segments with their own lr / weight decay; odd sizes cross the 16-byte vectors
    lrs = [1e-3, 3e-3, 1e-3, 5e-4, 2e-3]
    wds = [0.01, 0.0, 0.05, 0.01, 0.0]
    offs, total = [], 0
    for n in sizes:
        offs.append(total)
        total += (n + 7) // 8 * 8
    P = torch.zeros(total, device=dev())
    refs = []
    for n, o in zip(sizes, offs):
        r = torch.randn(n)
        P[o:o + n] = r.to(dev())
        refs.append(r.clone().requires_grad_(True))
    opt = torch.optim.AdamW([dict(params=[r], lr=lr, weight_decay=wd) for r, lr, wd in zip(refs, lrs, wds)], betas=(0.9, 0.999), eps=1e-8)
    G = torch.zeros(total, device=dev())
    M1, M2, E = torch.zeros_like(P), torch.zeros_like(P), P.clone()
    seg = torch.tensor([[o, o + (n + 7) // 8 * 8] for n, o in zip(sizes, offs)], dtype=torch.int64, device=dev())
    slr, swd = torch.tensor(lrs, device=dev()), torch.tensor(wds, device=dev())
    step = torch.zeros(1, device=dev())
    dyn = torch.tensor([0.99, 1.0], device=dev())
    ema_ref = [r.detach().clone() for r in refs]
    for it in range(10):
        for r, n, o in zip(refs, sizes, offs):
            g = torch.randn(n) * (0.1 + it)
            r.grad = g.clone()
            G[o:o + n] = g.to(dev())
        opt.step()
        for e, r in zip(ema_ref, refs):
            e.mul_(0.99).add_(r.detach(), alpha=0.01)
        L.call("cvhip_adamw_ema", P.data_ptr(), G.data_ptr(), M1.data_ptr(), M2.data_ptr(), E.data_ptr(), total, seg.data_ptr(),
               slr.data_ptr(), swd.data_ptr(), len(sizes), 0.9, 0.999, 1e-8, step.data_ptr(), 0.0, 1.0, dyn.data_ptr(), None, ops._stream())
    torch.cuda.synchronize()
    assert float(step) == 10.0
    for r, e, n, o in zip(refs, ema_ref, sizes, offs):
        assert rel(P[o:o + n], r.detach()) < 2e-6, (n, rel(P[o:o + n], r.detach()))
        assert rel(E[o:o + n], e) < 2e-6
        st = opt.state[r]
        # (the hyper-parameters cross the C ABI as fp32: 1 - beta2 then carries beta2 = 0.999f's representation error, 4.7e-5
        # relative, where torch uses the double 0.001 — visible in exp_avg_sq only, far below it in the parameters)
        assert rel(M1[o:o + n], st["exp_avg"]) < 2e-6 and rel(M2[o:o + n], st["exp_avg_sq"]) < 1e-4
    # GradScaler skip: parameters, moments and the step count stay; the EMA still moves towards the (unchanged) parameters
    before = (P.clone(), M1.clone(), M2.clone(), E.clone())
    skip = torch.tensor([1.0, 1.0], device=dev())
    L.call("cvhip_adamw_ema", P.data_ptr(), G.data_ptr(), M1.data_ptr(), M2.data_ptr(), E.data_ptr(), total, seg.data_ptr(),
           slr.data_ptr(), swd.data_ptr(), len(sizes), 0.9, 0.999, 1e-8, step.data_ptr(), 0.0, 1.0, dyn.data_ptr(), skip.data_ptr(), ops._stream())
    torch.cuda.synchronize()
    assert float(step) == 10.0 and torch.equal(P, before[0]) and torch.equal(M1, before[1]) and torch.equal(M2, before[2])
    assert torch.allclose(E, 0.99 * before[3] + 0.01 * P, rtol=1e-6, atol=1e-7)


def test_config1_step_with_fused_adamw_matches_stock_adamw():
    """ResNet-50 classification (conf/mini-imagenet.yml) for 3 steps: FlatTrainState(optimizer='adamw') vs torch.optim.AdamW on a twin
    model running the same engine kernels"""
    from cvpytorch_amd import classification
    from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
    torch.manual_seed(3)
    dictionary = [{"c%d" % i: 1.0} for i in range(100)]
    a = classification.Classification(dictionary).to(dev()).train()
    b = classification.Classification(dictionary).to(dev()).train()
    b.load_state_dict(a.state_dict())
    imgs = torch.randn(8, 3, 96, 96, device=dev())
    tg = torch.randint(0, 100, (8,), device=dev())
    opt = torch.optim.AdamW(a.parameters(), lr=1e-3, weight_decay=0.01)
    state = FlatTrainState(b, lr=1e-3, weight_decay=0.01, use_ema=False, optimizer="adamw")
    # torch.optim.AdamW over model.parameters() decays every parameter: give the flat state the same single group
    state.seg_wd.fill_(0.01)
    state.seg_lr.fill_(1e-3)
    step = FlatTrainStep(b, state)
    before = {n: p.detach().clone() for n, p in a.named_parameters()}
    opt.zero_grad(set_to_none=True)
    la = a(imgs, tg, "train")["loss"]
    la.backward()
    opt.step()
    lb = step(imgs, tg)["loss"]
    torch.cuda.synchronize()
    assert abs(float(la) - float(lb)) <= 3e-2 * abs(float(la)) + 1e-3, (float(la), float(lb))
    # AdamW's first update is ~lr * sign(g) per weight: elements whose gradient is rounding noise flip freely between two runs of the
    # same atomics-based backward, so compare the UPDATE VECTORS' direction, not the values
    pa, pb = dict(a.named_parameters()), dict(b.named_parameters())
    ua = torch.cat([(pa[n].detach() - before[n]).flatten() for n in before]).double()
    ub = torch.cat([(pb[n].detach() - before[n]).flatten() for n in before]).double()
    cos = float((ua * ub).sum() / (ua.norm() * ub.norm()))
    assert cos > 0.9, cos
    assert abs(float(ub.abs().max()) - float(ua.abs().max())) <= 0.05 * float(ua.abs().max())
    first = float(lb)
    for it in range(4):
        lb = step(imgs, tg)["loss"]
    torch.cuda.synchronize()
    assert float(state.adam_step) == 5.0 and torch.isfinite(lb) and float(lb) < first   # it trains (same batch: the loss falls)
