"""GPU parity of the fused 1x1 backward kernel (csrc/conv1x1_bwd.hip, cvhip_conv1x1_bwd_fused) called through the C ABI, against
fp32 CPU arithmetic on the SAME bf16 operands:
    du = dz * act'(scale*y + shift); dy = scale*(du - dbeta/M - xhat*dgamma/M) rounded to bf16 (the kernel feeds MFMA with bf16);
    dx = dy @ W (+ addend) rounded to bf16;  dw += dy^T @ x in fp32.
Tolerances: dx one bf16 rounding of an fp32-accumulated value (max |err| <= 2^-7 * max|ref|: dy itself can differ by one bf16
ulp where the fp32 affine form rounds differently; rel-L2 <= 6e-3); dw fp32 rel-L2 <= 2e-3 (same reason + summation order).
Also checks that the autograd ops take the fused path and agree with the three-pass path."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

BF = torch.bfloat16


def dev():
    return torch.device("cuda:0")


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


def act_bwd_ref(u, act, ap):
    if act == L.ACT_SILU:
        s = torch.sigmoid(u)
        return s * (1 + u * (1 - s))
    if act == L.ACT_RELU:
        return (u > 0).float()
    if act == L.ACT_LEAKY:
        return torch.where(u > 0, torch.ones_like(u), torch.full_like(u, ap))
    return torch.ones_like(u)


CASES = [
    # N, H, W, C, K, k_split, act, with_mean, with_res
    (2, 48, 48, 64, 64, 64, L.ACT_SILU, True, False),
    (2, 48, 48, 64, 64, 32, L.ACT_SILU, True, False),      # sibling pair: two gradient tensors
    (1, 70, 61, 32, 32, 32, L.ACT_SILU, True, True),       # M % 64 != 0, GradLink addend
    (2, 40, 40, 128, 128, 128, L.ACT_SILU, True, False),
    (2, 40, 40, 256, 128, 128, L.ACT_RELU, True, False),   # two 128-wide input-channel slices
    (2, 40, 40, 64, 128, 64, L.ACT_LEAKY, True, True),
    (1, 64, 64, 96, 32, 32, L.ACT_NONE, True, False),      # C = 96 -> three 32-wide slices
    (1, 64, 64, 128, 64, 64, L.ACT_SILU, False, False),    # eval-mode BN: scale/shift only
    (2, 40, 40, 64, 256, 256, L.ACT_NONE, True, True),     # K = 256 (round 4): ResNet expansion conv3 (BN, no activation) + addend
    (1, 70, 61, 128, 256, 128, L.ACT_RELU, True, False),   # K = 256, C = 128, two gradient tensors, ragged M
]


@pytest.mark.parametrize("case", CASES)
def test_bwd1x1_fused_vs_cpu(case):
    N, H, W, Cc, K, ks, act, with_mean, with_res = case
    ap = 0.1
    M = N * H * W
    torch.manual_seed(K * 7 + Cc)
    x = torch.randn(M, Cc).to(BF)
    y = (torch.randn(M, K) * 1.5 + 0.3).to(BF)
    dz = (torch.randn(M, K) * 0.1).to(BF)
    w = (torch.randn(K, Cc) / Cc ** 0.5).to(BF)          # logical [K][C]
    mean = y.float().mean(0)
    var = y.float().var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-3)
    gamma = torch.rand(K) + 0.5
    beta = torch.randn(K) * 0.2
    scale = gamma * invstd
    shift = beta - mean * scale
    u = y.float() * scale + shift
    du = dz.float() * act_bwd_ref(u, act, ap)
    xhat = (y.float() - mean) * invstd
    dbeta = du.sum(0)
    dgamma = (du * xhat).sum(0)
    if with_mean:
        dy = scale * (du - dbeta / M - xhat * dgamma / M)
    else:
        dy = scale * du
    dyb = dy.to(BF).float()
    res = (torch.randn(M, Cc) * 0.1).to(BF) if with_res else None
    dx_ref = dyb @ w.float()
    if res is not None:
        dx_ref = dx_ref + res.float()
    dw_ref = dyb.t() @ x.float()
    d = dev()
    desc = L.ConvDesc(N, Cc, H, W, K, 1, 1, 1, 1, 0, 0, 1, 1, 1, Cc, K, 0, 0)
    xd, yd, wd = x.to(d), y.to(d), w.t().contiguous().to(d)   # dgrad image [C][K]
    if ks < K:
        # two separately allocated gradient tensors with different pitches (a channel slice of a wider buffer for the second)
        dz0 = dz[:, :ks].contiguous().to(d)
        wide = torch.zeros(M, (K - ks) + 16, dtype=BF, device=d)
        wide[:, 8:8 + K - ks] = dz[:, ks:].to(d)
        dz1 = wide[:, 8:]
        dz1_ld = wide.shape[1]
        dz0_ld = ks
    else:
        dz0, dz0_ld, dz1, dz1_ld = dz.to(d), K, None, 0
    f = lambda t: t.float().to(d).contiguous()  # noqa: E731
    sc, sh, mu, isd, dg, db = f(scale), f(shift), f(mean), f(invstd), f(dgamma), f(dbeta)
    dx = torch.empty(M, Cc, dtype=BF, device=d)
    dw = torch.full((K, Cc), 0.5, dtype=torch.float32, device=d)   # accumulate semantics: starts non-zero
    resd = res.to(d) if res is not None else None
    L.call("cvhip_conv1x1_bwd_fused", C.byref(desc), dz0.data_ptr(), dz0_ld, dz1.data_ptr() if dz1 is not None else None, dz1_ld, ks,
           yd.data_ptr(), xd.data_ptr(), wd.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr() if with_mean else None,
           isd.data_ptr() if with_mean else None, dg.data_ptr() if with_mean else None, db.data_ptr() if with_mean else None,
           act, ap, resd.data_ptr() if resd is not None else None, Cc, dx.data_ptr(), Cc, dw.data_ptr(), None)
    torch.cuda.synchronize()
    e_dx = rel_l2(dx.float(), dx_ref)
    assert e_dx <= 6e-3, ("dx", case, e_dx)
    assert float((dx.float().cpu() - dx_ref).abs().max()) <= 2 ** -7 * float(dx_ref.abs().max()) + 1e-6
    e_dw = rel_l2(dw - 0.5, dw_ref)
    assert e_dw <= 2e-3, ("dw", case, e_dw)


def test_bwd1x1_policy_and_refusals():
    lib = L.load()
    d = dev()
    buf = torch.zeros(1 << 20, dtype=BF, device=d)
    f32 = torch.zeros(1 << 18, dtype=torch.float32, device=d)
    for (Cc, K, R, s) in [(256, 256, 1, 1), (64, 512, 1, 1), (64, 64, 3, 1), (64, 64, 1, 2), (24, 64, 1, 1), (64, 48, 1, 1)]:
        desc = L.ConvDesc(2, Cc, 16, 16, K, R, R, s, s, R // 2, R // 2, 1, 1, 1, Cc, K, 0, 0)
        assert lib.cvhip_conv1x1_bwd_fused_ok(C.byref(desc)) == 0
        st = lib.cvhip_conv1x1_bwd_fused(C.byref(desc), buf.data_ptr(), K, None, 0, K, buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), None, None,
                                         None, None, None, None, 0, 0.0, None, 0, buf.data_ptr(), Cc, f32.data_ptr(), None)
        assert st == L.ERR_UNSUPPORTED, (Cc, K, R, s, st)
    small = L.ConvDesc(1, 64, 16, 16, 64, 1, 1, 1, 1, 0, 0, 1, 1, 1, 64, 64, 0, 0)      # fits, but 4 trips: the policy says three-pass
    big = L.ConvDesc(64, 64, 80, 80, 64, 1, 1, 1, 1, 0, 0, 1, 1, 1, 64, 64, 0, 0)       # 6400 trips
    import os
    if os.environ.get("CVHIP_BWD1X1", "1") == "1":
        assert lib.cvhip_conv1x1_bwd_fused_ok(C.byref(small)) == 0
        assert lib.cvhip_conv1x1_bwd_fused_ok(C.byref(big)) == 1


@pytest.mark.parametrize("Cc,K,res", [(64, 64, False), (32, 32, True), (128, 64, False)])
def test_conv_module_backward_takes_fused_path_and_matches_three_pass(Cc, K, res, monkeypatch):
    """HipConvModule 1x1 (BN train + SiLU): gradients through the fused kernel == gradients through the three-pass kernels."""
    from cvpytorch_amd import bricks
    torch.manual_seed(3)
    d = dev()
    m = bricks.HipConvModule(Cc, K, 1, norm_cfg=dict(type="HipBN"), act_cfg=dict(type="HipSiLU")).to(d).train()
    x0 = torch.randn(8, Cc, 160, 160, device=d).to(BF).contiguous(memory_format=torch.channels_last)   # 3200 64-row trips
    gout = (torch.randn(8, K, 160, 160, device=d) * 0.1).to(BF).contiguous(memory_format=torch.channels_last)

    calls = []
    real_call = L.call

    def spy(name, *a):
        calls.append(name)
        return real_call(name, *a)

    def run(fused):
        monkeypatch.setattr(ops, "_bwd1x1_ok", (lambda *a, **k: False) if not fused else _orig_ok)
        for p in m.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        z = m(x)
        if res:
            z = z + x[:, :K] if K <= Cc else z
        z.backward(gout)
        torch.cuda.synchronize()
        return x.grad.float().clone(), m.conv.weight.grad.float().clone(), m.bn.weight.grad.float().clone(), m.bn.bias.grad.float().clone()

    _orig_ok = ops._bwd1x1_ok
    monkeypatch.setattr(L, "call", spy)
    a = run(True)
    fused_names = ("cvhip_conv1x1_bwd_fused", "cvhip_conv1x1_bwd_fused_acc")   # the _acc form folds the BN sums itself
    apply_names = ("cvhip_bn_act_bwd_apply", "cvhip_bn_act_bwd_apply_acc")
    assert any(n in calls for n in fused_names) and not any(n in calls for n in apply_names)
    calls.clear()
    b = run(False)
    assert not any(n in calls for n in fused_names) and "cvhip_conv2d_wgrad" in calls
    for name, u, v in zip(("dx", "dw", "dgamma", "dbeta"), a, b):
        e = rel_l2(u, v)
        assert e <= (8e-3 if name == "dx" else 3e-3), (name, e)
