"""GPU parity of the LAZY-ACTIVATION training path (round 5): `ConvModule.forward` = act(norm(conv(x)))
(src/models/bricks/conv_module.py:201-214) whose BN-apply + activation pass is deferred into the consumers' loads.

Every lazy form must equal the two-pass form it replaces:
  * cvhip_conv2d_fprop_fused with a prologue on the streaming 1x1 kernel  == cvhip_bn_act_fwd then cvhip_conv2d_fprop_acc:
    BIT-identical outputs (the prologue rounds act(scale*y+shift) to 16 bits exactly as the pass stores it) and equal BN sums;
  * cvhip_conv1x1_bwd_fused_lazy == cvhip_conv1x1_bwd_fused_acc on the materialised input: dx bit-identical (it does not depend on
    x), dW equal up to the order of the fp32 atomics;
  * cvhip_bn_act_fwd_acc_lazyres == cvhip_bn_act_fwd_acc with the materialised residual: bit-identical;
  * cvhip_bn_finalize_acc == the statistics block cvhip_bn_act_fwd_acc writes (bit-identical, running statistics included);
  * module level: CSPLayer / YOLOv5CSPDarknet stages and one whole YOLOv5-s train step with ops.set_lazy(True) vs (False):
    identical forward values, gradients equal to atomics-order tolerance, and the lazy run really skips the passes.
Tolerances are written at each assert."""
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


def _bn_consts(K, d, seed):
    g = torch.Generator().manual_seed(seed)
    scale = (torch.rand(K, generator=g) + 0.5).to(d)
    shift = (torch.randn(K, generator=g) * 0.3).to(d)
    return scale.contiguous(), shift.contiguous()


def _pack_w(w_kc, d):
    """[K][C] fp32 -> the fprop operand image [K][1*1*C] in the engine's storage precision"""
    return w_kc.to(BF).contiguous().to(d)


PRO_CASES = [
    # N, H, W, C, K, act, (lo, hi) or None      (>= 192 128-row tiles: the streaming kernel's own policy threshold)
    (4, 80, 80, 64, 64, L.ACT_SILU, None),
    (4, 80, 80, 32, 32, L.ACT_SILU, None),
    (3, 111, 97, 64, 128, L.ACT_SILU, None),       # ragged M (not a multiple of 128)
    (4, 80, 80, 128, 64, L.ACT_RELU, None),
    (4, 80, 80, 256, 128, L.ACT_LEAKY, None),      # two 128-channel pipeline stages
    (4, 80, 80, 96, 32, L.ACT_SILU, None),         # C not a multiple of 32 (padded reduction)
    (4, 80, 80, 128, 128, L.ACT_SILU, (64, 128)),  # concat: only the second slice is lazy
    (4, 80, 80, 64, 64, L.ACT_SILU, (0, 32)),
    (4, 80, 80, 64, 64, L.ACT_NONE, None),         # BN without activation
]


@pytest.mark.parametrize("case", PRO_CASES)
@pytest.mark.parametrize("with_stats", [True, False])
def test_stream1x1_prologue_equals_two_pass(case, with_stats):
    N, H, W, Cc, K, act, rng = case
    ap = 0.1
    d = dev()
    M = N * H * W
    torch.manual_seed(Cc * 3 + K)
    yraw = (torch.randn(M, Cc) * 1.5 + 0.2).to(BF).to(d)
    w = _pack_w(torch.randn(K, Cc) / Cc ** 0.5, d)
    scale, shift = _bn_consts(Cc, d, 5)
    lo, hi = rng if rng is not None else (0, Cc)
    desc = L.ConvDesc(N, Cc, H, W, K, 1, 1, 1, 1, 0, 0, 1, 1, 1, Cc, K, 0, 0)
    lib = L.load()
    if not lib.cvhip_conv1x1_stream_prologue_ok(C.byref(desc), int(with_stats)):
        raise AssertionError("the streaming kernel should run this descriptor (policy changed?)")
    # two-pass reference: materialise the lazy slice, then the plain convolution
    z = yraw.clone()
    zs = torch.empty(M, hi - lo, dtype=BF, device=d)
    L.call("cvhip_bn_act_fwd", yraw.data_ptr() + 2 * lo, Cc, zs.data_ptr(), hi - lo, M, hi - lo, scale.data_ptr() + 4 * lo, shift.data_ptr() + 4 * lo,
           act, ap, None, 0, None)
    z[:, lo:hi] = zs
    out_ref = torch.empty(M, K, dtype=BF, device=d)
    acc_ref = torch.zeros(L.BN_ACC_SHARDS, 2, K, dtype=torch.float64, device=d)
    if with_stats:
        L.call("cvhip_conv2d_fprop_acc", C.byref(desc), z.data_ptr(), w.data_ptr(), out_ref.data_ptr(), acc_ref.data_ptr(), None)
    else:
        L.call("cvhip_conv2d_fprop", C.byref(desc), z.data_ptr(), w.data_ptr(), None, out_ref.data_ptr(), None, None)
    # lazy: one launch on the raw tensor
    out = torch.empty(M, K, dtype=BF, device=d)
    acc = torch.zeros(L.BN_ACC_SHARDS, 2, K, dtype=torch.float64, device=d)
    f = L.ConvFuse()
    f.pro_scale, f.pro_shift, f.pro_act, f.pro_act_param = scale.data_ptr(), shift.data_ptr(), act, ap
    if rng is not None:
        f.pro_lo, f.pro_hi = lo, hi
    if with_stats:
        f.bn_acc = acc.data_ptr()
    L.call("cvhip_conv2d_fprop_fused", C.byref(desc), yraw.data_ptr(), w.data_ptr(), out.data_ptr(), C.byref(f), None)
    torch.cuda.synchronize()
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)), (case, float((out.float() - out_ref.float()).abs().max()))
    if with_stats:
        a, b = acc.sum(0), acc_ref.sum(0)
        # identical fp32 tile sums, fp64 accumulation in a different arrival order: ~1e-16 relative
        assert float((a - b).abs().max()) <= 1e-9 * max(1.0, float(b.abs().max()))


def test_stream1x1_prologue_refusals():
    lib = L.load()
    d = dev()
    buf = torch.zeros(1 << 22, dtype=BF, device=d)
    one = torch.ones(1024, dtype=torch.float32, device=d)
    # 3x3 with an odd channel count for the patch kernel, a strided 1x1: no kernel with a prologue -> UNSUPPORTED, never a silent plain conv
    for (Cc, K, R, s, H) in [(24, 64, 3, 1, 40), (64, 64, 1, 2, 80)]:
        desc = L.ConvDesc(2, Cc, H, H, K, R, R, s, s, R // 2, R // 2, 1, 1, 1, Cc, K, 0, 0)
        assert lib.cvhip_conv1x1_stream_prologue_ok(C.byref(desc), 1) == 0
        f = L.ConvFuse()
        f.pro_scale, f.pro_shift, f.pro_act = one.data_ptr(), one.data_ptr(), L.ACT_SILU
        st = lib.cvhip_conv2d_fprop_fused(C.byref(desc), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), C.byref(f), None)
        assert st == L.ERR_UNSUPPORTED, (Cc, K, R, s, st)
    # a misaligned channel range is an invalid argument
    desc = L.ConvDesc(4, 64, 80, 80, 64, 1, 1, 1, 1, 0, 0, 1, 1, 1, 64, 64, 0, 0)
    assert lib.cvhip_conv1x1_stream_prologue_ok(C.byref(desc), 1) == 1
    f = L.ConvFuse()
    f.pro_scale, f.pro_shift, f.pro_act, f.pro_lo, f.pro_hi = one.data_ptr(), one.data_ptr(), L.ACT_SILU, 4, 60
    st = lib.cvhip_conv2d_fprop_fused(C.byref(desc), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), C.byref(f), None)
    assert st == L.ERR_INVALID


BWD_CASES = [
    # N, H, W, C, K, k_split, act(layer), act(input), with addend
    (2, 48, 48, 64, 64, 64, L.ACT_SILU, L.ACT_SILU, False),
    (2, 48, 48, 64, 64, 32, L.ACT_SILU, L.ACT_SILU, False),      # sibling pair
    (1, 70, 61, 32, 32, 32, L.ACT_SILU, L.ACT_SILU, True),       # ragged M + GradLink addend
    (2, 40, 40, 128, 128, 128, L.ACT_SILU, L.ACT_SILU, False),   # the register-tight instance
    (2, 40, 40, 256, 128, 64, L.ACT_RELU, L.ACT_RELU, False),    # two 128-wide input-channel slices
    (1, 64, 64, 96, 32, 32, L.ACT_LEAKY, L.ACT_NONE, False),     # three 32-wide slices, input BN without activation
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_bwd1x1_lazy_input_equals_materialised(case):
    N, H, W, Cc, K, ks, act, xact, with_res = case
    ap = 0.1
    d = dev()
    M = N * H * W
    torch.manual_seed(K * 5 + Cc)
    xraw = (torch.randn(M, Cc) * 1.2 - 0.1).to(BF).to(d)
    y = (torch.randn(M, K) * 1.5 + 0.3).to(BF).to(d)
    dz = (torch.randn(M, K) * 0.1).to(BF).to(d)
    wd = (torch.randn(Cc, K) / Cc ** 0.5).to(BF).to(d)            # dgrad image [C][K]
    xs, xh = _bn_consts(Cc, d, 9)
    mean = y.float().mean(0)
    invstd = 1.0 / torch.sqrt(y.float().var(0, unbiased=False) + 1e-3)
    gamma, beta = _bn_consts(K, d, 11)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    res = (torch.randn(M, Cc) * 0.1).to(BF).to(d) if with_res else None
    desc = L.ConvDesc(N, Cc, H, W, K, 1, 1, 1, 1, 0, 0, 1, 1, 1, Cc, K, 0, 0)
    # the layer's backward sums (both runs consume the same accumulator contents)
    acc0 = torch.zeros(L.BN_ACC_SHARDS, 2, K, dtype=torch.float64, device=d)
    L.call("cvhip_bn_act_bwd_sums_acc", dz.data_ptr(), K, y.data_ptr(), K, M, K, scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
           act, ap, acc0.data_ptr(), K, None)
    if ks < K:
        dz0, dz1 = dz[:, :ks].contiguous(), dz[:, ks:].contiguous()
        seg = (dz0.data_ptr(), ks, dz1.data_ptr(), K - ks)
    else:
        seg = (dz.data_ptr(), K, None, 0)
    z = torch.empty(M, Cc, dtype=BF, device=d)
    L.call("cvhip_bn_act_fwd", xraw.data_ptr(), Cc, z.data_ptr(), Cc, M, Cc, xs.data_ptr(), xh.data_ptr(), xact, ap, None, 0, None)
    outs = []
    for lazy in (False, True):
        dx = torch.empty(M, Cc, dtype=BF, device=d)
        dw = torch.full((K, Cc), 0.25, dtype=torch.float32, device=d)
        dg = torch.zeros(K, dtype=torch.float32, device=d)
        db = torch.zeros(K, dtype=torch.float32, device=d)
        common = (C.byref(desc), seg[0], seg[1], seg[2], seg[3], ks, y.data_ptr())
        tailargs = (wd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), acc0.data_ptr(), K, dg.data_ptr(),
                    db.data_ptr(), 0, act, ap, res.data_ptr() if res is not None else None, Cc, dx.data_ptr(), Cc, dw.data_ptr())
        if lazy:
            li = L.LazyIn(xs.data_ptr(), xh.data_ptr(), xact, ap, 0, 0)
            L.call("cvhip_conv1x1_bwd_fused_lazy", *common, xraw.data_ptr(), *tailargs, C.byref(li), None)
        else:
            L.call("cvhip_conv1x1_bwd_fused_acc", *common, z.data_ptr(), *tailargs, None, None)
        torch.cuda.synchronize()
        outs.append((dx, dw, dg, db))
    (dx0, dw0, dg0, db0), (dx1, dw1, dg1, db1) = outs
    assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16))
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    # same bf16 operands, fp32 atomics in a different order
    assert rel_l2(dw1, dw0) <= 1e-5, rel_l2(dw1, dw0)


@pytest.mark.parametrize("act", [L.ACT_SILU, L.ACT_RELU, L.ACT_LEAKY])
@pytest.mark.parametrize("K,M", [(64, 40 * 40 * 3), (32, 12345), (128, 6400)])
def test_lazy_residual_pass_and_finalize_acc(act, K, M):
    ap = 0.1
    d = dev()
    torch.manual_seed(K + M)
    y = (torch.randn(M, K) * 1.3 + 0.1).to(BF).to(d)
    rraw = (torch.randn(M, K) * 1.1 - 0.2).to(BF).to(d)
    rs, rh = _bn_consts(K, d, 21)
    gamma, beta = _bn_consts(K, d, 23)
    acc = torch.zeros(L.BN_ACC_SHARDS, 2, K, dtype=torch.float64, device=d)
    acc[3, 0] = y.double().sum(0)
    acc[7, 1] = (y.double() ** 2).sum(0)
    res = torch.empty(M, K, dtype=BF, device=d)
    L.call("cvhip_bn_act_fwd", rraw.data_ptr(), K, res.data_ptr(), K, M, K, rs.data_ptr(), rh.data_ptr(), act, ap, None, 0, None)
    outs = []
    for lazy in (False, True):
        z = torch.empty(M, K, dtype=BF, device=d)
        st4 = torch.zeros(4, K, dtype=torch.float32, device=d)
        rm = torch.full((K,), 0.5, dtype=torch.float32, device=d)
        rv = torch.full((K,), 2.0, dtype=torch.float32, device=d)
        head = (y.data_ptr(), K, z.data_ptr(), K, M, K, acc.data_ptr(), K, M, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.03,
                1e-3, st4[0].data_ptr(), st4[1].data_ptr(), st4[2].data_ptr(), st4[3].data_ptr(), act, ap)
        if lazy:
            L.call("cvhip_bn_act_fwd_acc_lazyres", *head, rraw.data_ptr(), K, rs.data_ptr(), rh.data_ptr(), None)
        else:
            L.call("cvhip_bn_act_fwd_acc", *head, res.data_ptr(), K, 0, None)
        torch.cuda.synchronize()
        outs.append((z, st4, rm, rv))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a.view(torch.int16) if a.dtype == BF else a, b.view(torch.int16) if b.dtype == BF else b)
    # the stand-alone finalize writes the same statistics block and running statistics
    st4 = torch.zeros(4, K, dtype=torch.float32, device=d)
    rm = torch.full((K,), 0.5, dtype=torch.float32, device=d)
    rv = torch.full((K,), 2.0, dtype=torch.float32, device=d)
    L.call("cvhip_bn_finalize_acc", acc.data_ptr(), K, K, M, gamma.data_ptr(), beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.03, 1e-3,
           st4[0].data_ptr(), st4[1].data_ptr(), st4[2].data_ptr(), st4[3].data_ptr(), None)
    torch.cuda.synchronize()
    assert torch.equal(st4, outs[0][1]) and torch.equal(rm, outs[0][2]) and torch.equal(rv, outs[0][3])


SPLIT_CASES = [
    # N, H, W, C, K, k_split, lazy input range or None
    (4, 80, 80, 64, 64, 32, None),          # the backbone's first CSP pair
    (4, 80, 80, 128, 128, 64, None),
    (3, 111, 97, 64, 128, 64, None),        # ragged M
    (4, 80, 80, 64, 64, 32, (0, 64)),       # split store AND a prologue (lazy input)
    (4, 80, 80, 128, 64, 48, (64, 128)),    # uneven halves (8-aligned), partial prologue
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_stream1x1_split_store_equals_plain(case):
    """cvhip_conv_fuse.y2 / y_split: channels [k_split, K) of the raw output go to a second buffer (a concat slice: pitch > its width);
    both destinations and the BN sums equal the plain launch's"""
    N, H, W, Cc, K, ks, rng = case
    d = dev()
    M = N * H * W
    torch.manual_seed(Cc + 7 * K + ks)
    x = (torch.randn(M, Cc) * 1.1).to(BF).to(d)
    w = _pack_w(torch.randn(K, Cc) / Cc ** 0.5, d)
    scale, shift = _bn_consts(Cc, d, 6)
    desc = L.ConvDesc(N, Cc, H, W, K, 1, 1, 1, 1, 0, 0, 1, 1, 1, Cc, K, 0, 0)
    assert L.load().cvhip_conv1x1_stream_prologue_ok(C.byref(desc), 1) == 1
    outs = []
    k2 = K - ks
    cat_ld = 2 * k2 + 8                                       # the second half lands at channel offset 8 of a wider buffer
    for split in (False, True):
        y = torch.full((M, K), 7.0, dtype=BF, device=d)
        cat = torch.full((M, cat_ld), 3.0, dtype=BF, device=d)
        acc = torch.zeros(L.BN_ACC_SHARDS, 2, K, dtype=torch.float64, device=d)
        f = L.ConvFuse()
        f.bn_acc = acc.data_ptr()
        if rng is not None:
            f.pro_scale, f.pro_shift, f.pro_act, f.pro_act_param = scale.data_ptr(), shift.data_ptr(), L.ACT_SILU, 0.0
            f.pro_lo, f.pro_hi = rng
        if split:
            f.y2, f.y2_ld, f.y_split = cat.data_ptr() + 16, cat_ld, ks
        L.call("cvhip_conv2d_fprop_fused", C.byref(desc), x.data_ptr(), w.data_ptr(), y.data_ptr(), C.byref(f), None)
        torch.cuda.synchronize()
        outs.append((y, cat, acc.sum(0)))
    (y0, _, a0), (y1, cat1, a1) = outs
    assert torch.equal(y0[:, :ks].view(torch.int16), y1[:, :ks].view(torch.int16))
    assert torch.equal(y0[:, ks:].contiguous().view(torch.int16), cat1[:, 8:8 + k2].contiguous().view(torch.int16))
    # nothing else was touched: y's second half and the rest of the concat buffer keep their fill values
    assert bool((y1[:, ks:] == 7.0).all()) and bool((cat1[:, :8] == 3.0).all()) and bool((cat1[:, 8 + k2:] == 3.0).all())
    assert float((a0 - a1).abs().max()) <= 1e-9 * max(1.0, float(a0.abs().max()))


def test_stream1x1_split_store_refusals():
    lib = L.load()
    d = dev()
    buf = torch.zeros(1 << 22, dtype=BF, device=d)
    desc = L.ConvDesc(4, 64, 80, 80, 64, 1, 1, 1, 1, 0, 0, 1, 1, 1, 64, 64, 0, 0)
    for ks, ld, off in [(30, 64, 0), (32, 60, 0), (32, 64, 2), (0, 64, 0), (64, 64, 0)]:   # misaligned split / pitch / address, empty halves
        f = L.ConvFuse()
        f.y2, f.y2_ld, f.y_split = buf.data_ptr() + (1 << 20) + off, ld, ks
        st = lib.cvhip_conv2d_fprop_fused(C.byref(desc), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), C.byref(f), None)
        assert st == L.ERR_INVALID, (ks, ld, off, st)
    # a problem the streaming kernel does not run (3x3): no other kernel has the split store -> UNSUPPORTED, never a silent single store
    desc = L.ConvDesc(2, 64, 40, 40, 64, 3, 3, 1, 1, 1, 1, 1, 1, 1, 64, 64, 0, 0)
    f = L.ConvFuse()
    f.y2, f.y2_ld, f.y_split = buf.data_ptr() + (1 << 20), 64, 32
    st = lib.cvhip_conv2d_fprop_fused(C.byref(desc), buf.data_ptr(), buf.data_ptr(), buf.data_ptr(), C.byref(f), None)
    assert st == L.ERR_UNSUPPORTED, st


@pytest.mark.parametrize("case", [(2, 48, 48, 64, 64, 32, False), (1, 70, 61, 64, 128, 64, False), (2, 40, 40, 128, 128, 64, True)])
def test_bwd1x1_split_raw_output_equals_joint(case):
    """cvhip_conv1x1_bwd_fused_split: the second sibling's raw output read from its concat slice == the joint-buffer form"""
    N, H, W, Cc, K, ks, lazy_in = case
    ap = 0.0
    act = L.ACT_SILU
    d = dev()
    M = N * H * W
    torch.manual_seed(K * 3 + Cc + ks)
    xraw = (torch.randn(M, Cc) * 1.2 - 0.1).to(BF).to(d)
    y = (torch.randn(M, K) * 1.5 + 0.3).to(BF).to(d)
    k2 = K - ks
    cat_ld = 2 * k2
    cat = torch.zeros(M, cat_ld, dtype=BF, device=d)
    cat[:, k2:] = y[:, ks:]                                       # the second half lives in channels [k2, 2 k2) of a concat buffer
    ylow = y.clone()
    ylow[:, ks:] = 99.0                                           # ... and NOT in the joint buffer (never written by the split store)
    dz0 = (torch.randn(M, ks) * 0.1).to(BF).to(d)
    dz1 = (torch.randn(M, k2) * 0.1).to(BF).to(d)
    wd = (torch.randn(Cc, K) / Cc ** 0.5).to(BF).to(d)
    xs, xh = _bn_consts(Cc, d, 9)
    mean = y.float().mean(0)
    invstd = 1.0 / torch.sqrt(y.float().var(0, unbiased=False) + 1e-3)
    gamma, beta = _bn_consts(K, d, 11)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    desc = L.ConvDesc(N, Cc, H, W, K, 1, 1, 1, 1, 0, 0, 1, 1, 1, Cc, K, 0, 0)
    acc0 = torch.zeros(L.BN_ACC_SHARDS, 2, K, dtype=torch.float64, device=d)
    for dzh, kh, off in ((dz0, ks, 0), (dz1, k2, ks)):
        L.call("cvhip_bn_act_bwd_sums_acc", dzh.data_ptr(), kh, y.data_ptr() + 2 * off, K, M, kh, scale.data_ptr() + 4 * off, shift.data_ptr() + 4 * off,
               mean.data_ptr() + 4 * off, invstd.data_ptr() + 4 * off, act, ap, acc0.data_ptr() + 8 * off, K, None)
    z = torch.empty(M, Cc, dtype=BF, device=d)
    L.call("cvhip_bn_act_fwd", xraw.data_ptr(), Cc, z.data_ptr(), Cc, M, Cc, xs.data_ptr(), xh.data_ptr(), act, ap, None, 0, None)
    outs = []
    for split in (False, True):
        dx = torch.empty(M, Cc, dtype=BF, device=d)
        dw = torch.zeros((K, Cc), dtype=torch.float32, device=d)
        dg = torch.zeros(K, dtype=torch.float32, device=d)
        db = torch.zeros(K, dtype=torch.float32, device=d)
        head = (C.byref(desc), dz0.data_ptr(), ks, dz1.data_ptr(), k2, ks)
        tailargs = (wd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), acc0.data_ptr(), K, dg.data_ptr(),
                    db.data_ptr(), 0, act, ap, None, 0, dx.data_ptr(), Cc, dw.data_ptr())
        li = L.LazyIn(xs.data_ptr(), xh.data_ptr(), act, ap, 0, 0)
        if split:
            L.call("cvhip_conv1x1_bwd_fused_split", *head, ylow.data_ptr(), cat.data_ptr() + 2 * k2, cat_ld, (xraw if lazy_in else z).data_ptr(),
                   *tailargs, C.byref(li) if lazy_in else None, None)
        elif lazy_in:
            L.call("cvhip_conv1x1_bwd_fused_lazy", *head, y.data_ptr(), xraw.data_ptr(), *tailargs, C.byref(li), None)
        else:
            L.call("cvhip_conv1x1_bwd_fused_acc", *head, y.data_ptr(), z.data_ptr(), *tailargs, None, None)
        torch.cuda.synchronize()
        outs.append((dx, dw, dg, db))
    (dx0, dw0, dg0, db0), (dx1, dw1, dg1, db1) = outs
    assert torch.equal(dx0.view(torch.int16), dx1.view(torch.int16))
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert rel_l2(dw1, dw0) <= 1e-5, rel_l2(dw1, dw0)


def test_cat_lazy_slice_rules():
    """ops.cat keeps ONE in-place, whole, 8-aligned lazy slice raw (range tag on the result); any other lazy input is activated on its
    way into the destination; partial-range tensors materialise correctly"""
    d = dev()
    N, H, W, k = 2, 16, 16, 16
    torch.manual_seed(3)
    buf = ops.empty_nhwc(N, 2 * k, H, W, d)
    a = torch.randn(N, k, H, W, device=d).to(BF)
    b = torch.randn(N, k, H, W, device=d).to(BF)
    buf[:, :k] = a
    buf[:, k:] = b
    sc, sh = _bn_consts(k, d, 1)
    ref_b = torch.nn.functional.silu(b.float() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    x1, x2 = buf[:, :k], buf[:, k:]
    x2._hip_lazy = ops.LazyAct(sc, sh, L.ACT_SILU, 0.0)
    out = ops.cat([x1, x2])
    lz = ops.lazy_of(out)
    assert lz is not None and (lz.lo, lz.hi) == (k, 2 * k) and out.data_ptr() == buf.data_ptr()
    zm = ops.materialize(out)                     # partial range: copy + activation of the slice
    torch.cuda.synchronize()
    assert torch.equal(zm[:, :k].float(), a.float())
    assert float((zm[:, k:].float() - ref_b).abs().max()) <= 3e-2
    assert torch.equal(buf[:, k:].float(), b.float())          # the raw data is still there (its producer's backward reads it)
    # a lazy input that is NOT in place: activated on the copy, result not lazy
    y = b.clone(memory_format=torch.channels_last)
    y._hip_lazy = ops.LazyAct(sc, sh, L.ACT_SILU, 0.0)
    out2 = ops.cat([a.contiguous(memory_format=torch.channels_last), y])
    torch.cuda.synchronize()
    assert ops.lazy_of(out2) is None
    assert torch.equal(out2[:, :k].float(), a.float()) and float((out2[:, k:].float() - ref_b).abs().max()) <= 3e-2
    # two in-place lazy slices: a fresh buffer, both activated, raw data untouched
    x1b, x2b = buf[:, :k], buf[:, k:]
    x1b._hip_lazy = ops.LazyAct(sc, sh, L.ACT_SILU, 0.0)
    x2b._hip_lazy = ops.LazyAct(sc, sh, L.ACT_SILU, 0.0)
    out3 = ops.cat([x1b, x2b])
    torch.cuda.synchronize()
    assert ops.lazy_of(out3) is None and out3.data_ptr() != buf.data_ptr()
    assert float((out3[:, k:].float() - ref_b).abs().max()) <= 3e-2 and torch.equal(buf[:, :k].float(), a.float())


# ---- module level ---------------------------------------------------------------------------------------------------------------------
def _spy(monkeypatch):
    calls = []
    real = L.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)

    monkeypatch.setattr(L, "call", spy)
    return calls


def _grads(model):
    return {n: p.grad.detach().float().clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("shortcut", [True, False])
def test_stage_lazy_equals_eager(shortcut, monkeypatch):
    """stride-2 ConvModule -> CSPLayer (a YOLOv5CSPDarknet stage) through a flat train state (sibling pair active): the lazy run skips
    the stride-2 layer's and the first sibling's apply passes and equals the eager run."""
    from cvpytorch_amd import arena
    from cvpytorch_amd.bricks import HipConvModule
    from cvpytorch_amd.yolo_blocks import CSPLayer
    d = dev()

    class Stage(torch.nn.Module):
        def __init__(self):
            super().__init__()
            cfg = dict(norm_cfg=dict(type="BN", momentum=0.03, eps=0.001), act_cfg=dict(type="SiLU"))
            self.down = HipConvModule(64, 128, 3, 2, 1, **cfg)
            self.csp = CSPLayer(128, 128, n=2, shortcut=shortcut, **cfg)

        def forward(self, x):
            return self.csp(self.down(x, lazy=True))

    torch.manual_seed(11)
    m = Stage().to(d).train()
    state = arena.FlatTrainState(m, lr=0.0, use_ema=False)   # parameters into the flat arenas: the sibling pair and the accumulators become active
    # 25 x 80 x 80 output pixels = 2500 64-row trips: above the fused 1x1 backward's policy threshold, as the real layers are; 128- and
    # 64-channel edges: the ones the measured policy (ops.lazy_edge_ok) admits for SiLU
    x0 = torch.randn(25, 64, 160, 160, device=d).to(BF).contiguous(memory_format=torch.channels_last)
    gout = (torch.randn(25, 128, 80, 80, device=d) * 0.1).to(BF).contiguous(memory_format=torch.channels_last)
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    calls = _spy(monkeypatch)
    res = {}
    for lazy in (False, True):
        ops.set_lazy(lazy)
        m.load_state_dict(sd0)
        ops.bump_weights_epoch()
        state.zero_grad()
        state.zero_stats()
        calls.clear()
        x = x0.clone().requires_grad_(True)
        z = m(x)
        z.backward(gout)
        torch.cuda.synchronize()
        res[lazy] = (z.detach().clone(), x.grad.detach().clone(), state.grad.clone(),
                     {k: v.clone() for k, v in m.state_dict().items() if "running" in k}, list(calls))
    ops.set_lazy(True)
    z0, dx0, g0, rs0, c0 = res[False]
    z1, dx1, g1, rs1, c1 = res[True]
    assert torch.equal(z0.view(torch.int16), z1.view(torch.int16))
    for k in rs0:
        assert torch.equal(rs0[k], rs1[k]), k
    assert rel_l2(dx1, dx0) <= 2e-3 and rel_l2(g1, g0) <= 1e-4, (rel_l2(dx1, dx0), rel_l2(g1, g0))
    assert "cvhip_bn_finalize_acc" in c1 and "cvhip_bn_finalize_acc" not in c0
    assert "cvhip_conv1x1_bwd_fused_lazy" in c1
    # the second sibling's concat slice stays raw too (split store; conv3 transforms channels [64, 128) on load)
    assert "cvhip_conv1x1_bwd_fused_split" in c1 and "cvhip_conv1x1_bwd_fused_split" not in c0
    n_apply = lambda c: sum(1 for n in c if n.startswith("cvhip_bn_act_fwd"))  # noqa: E731
    assert n_apply(c1) <= n_apply(c0) - 3, (n_apply(c0), n_apply(c1))


def test_yolov5s_step_lazy_equals_eager():
    """One YOLOv5-s train step (fused loss, flat state) at BASELINE config 2's own size (batch 64, 640x640: the lazy edges need the
    real layers' row counts): lazy and eager agree on the loss (bit-identical forward) and on the gradient arena (atomics-order
    tolerance)."""
    from cvpytorch_amd import arena
    from cvpytorch_amd.yolov5 import YOLOv5
    d = dev()
    torch.manual_seed(5)
    model = YOLOv5(num_classes=80, subtype="s", max_targets=64, fused_loss=True).to(d).train()
    state = arena.FlatTrainState(model, lr=0.0, use_ema=False)
    B = 64
    imgs = torch.rand(B, 3, 640, 640, device=d)
    tg = torch.zeros(64, 6, device=d)
    tg[:, 0] = -1
    tg[:, 2:] = 0.5
    g = torch.Generator().manual_seed(3)
    for i in range(40):
        tg[i, 0] = i % B
        tg[i, 1] = int(torch.randint(0, 80, (1,), generator=g))
        tg[i, 2:4] = torch.rand(2, generator=g) * 0.8 + 0.1
        tg[i, 4:6] = torch.rand(2, generator=g) * 0.3 + 0.05
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    out = {}
    for lazy in (False, True):
        ops.set_lazy(lazy)
        model.load_state_dict(sd0)
        ops.bump_weights_epoch()
        state.zero_grad()
        state.zero_stats()
        loss = model(imgs, tg, mode="train")["loss"]
        loss.backward()
        torch.cuda.synchronize()
        out[lazy] = (float(loss), state.grad.clone())
    ops.set_lazy(True)
    assert out[False][0] == out[True][0], (out[False][0], out[True][0])
    e = rel_l2(out[True][1], out[False][1])
    assert e <= 2e-3, e
