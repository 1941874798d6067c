"""GPU parity of the Hip MODULES (the drop-in boundary) against the REFERENCE's golden vectors
(tests/golden, captured from the reference's own classes by tools/gen_golden.py) and against the
oracle's full YOLOv5-s on seeded synthetic batches.

The HIP path stores activations in bf16 (fp32 accumulate, fp32 BN statistics); the golden vectors are
fp32. Stated tolerance: per-layer / per-block outputs relative L2 <= 2e-2, gradients
cosine >= 0.999 (single ConvModule) / >= 0.995 (multi-layer blocks), end-to-end loss |d|/|loss| <= 2e-2.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvpytorch_amd import bricks, ops, yolo_blocks, yolov5

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BN_YOLO = dict(type="BN", momentum=0.03, eps=0.001)


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        parts = k.split("/", 1)
        if len(parts) == 1:
            out[k] = z[k]
        else:
            out.setdefault(parts[0], {})[parts[1]] = z[k]
    return out


def T(a):
    return torch.from_numpy(np.asarray(a))


def lst(d):
    return [T(d[str(i)]) for i in range(len(d))]


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


def cosine(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    if float(b.norm()) == 0:
        return 1.0 if float(a.norm()) < 1e-6 else 0.0
    return float((a @ b) / (a.norm() * b.norm()).clamp(min=1e-30))


def run(mod, inputs, cots):
    inputs = [x.to(dev()).requires_grad_(True) for x in inputs]
    out = mod(*inputs)
    outs = [o for o in (list(out) if isinstance(out, (tuple, list)) else [out]) if torch.is_tensor(o)]
    loss = sum((o.float() * c.to(dev())).sum() for o, c in zip(outs, cots))
    named = [(n, p) for n, p in mod.named_parameters() if p.requires_grad]
    grads = torch.autograd.grad(loss, inputs + [p for _, p in named], allow_unused=True)
    gpar = {n: (g if g is not None else torch.zeros_like(p)) for (n, p), g in zip(named, grads[len(inputs):])}
    return outs, grads[:len(inputs)], gpar


CONV_ACT = {"k1": dict(type="SiLU"), "k3": dict(type="SiLU"), "k3s2": dict(type="Swish"), "k3s2odd": dict(type="SiLU"),
            "k6s2": dict(type="SiLU"), "k3d2": dict(type="ReLU"), "k1bias": None, "dw3d3": dict(type="ReLU"), "k1s2": None}
CONV_NORM = {"k3d2": dict(type="BN"), "k1bias": None, "dw3d3": dict(type="BN"), "k1s2": dict(type="BN")}


@pytest.mark.parametrize("name", sorted(CONV_ACT))
def test_hip_convmodule_vs_reference_vectors(name):
    g = load("convmodule_" + name)
    cin, cout, k, s, p, d, grp = [int(v) for v in g["meta"]]
    m = bricks.HipConvModule(cin, cout, k, stride=s, padding=p, dilation=d, groups=grp, norm_cfg=CONV_NORM.get(name, BN_YOLO),
                             act_cfg=CONV_ACT[name])
    missing, unexpected = m.load_state_dict({kk: T(v) for kk, v in g["state"].items()}, strict=True)
    assert not missing and not unexpected  # same state_dict keys as the reference ConvModule
    m.to(dev()).train()
    x = T(g["x"])
    needs_grad_x = cin % 8 == 0
    xin = x.to(dev())
    if needs_grad_x:
        xin = xin.to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    out = m(xin)
    assert rel_l2(out.float(), T(g["out"])) < 2e-2, rel_l2(out.float(), T(g["out"]))
    loss = (out.float() * T(g["cot"]).to(dev())).sum()
    named = [(n, pp) for n, pp in m.named_parameters()]
    grads = torch.autograd.grad(loss, ([xin] if needs_grad_x else []) + [pp for _, pp in named])
    if needs_grad_x:
        assert cosine(grads[0].float(), T(g["gx"])) > 0.999
        assert rel_l2(grads[0].float(), T(g["gx"])) < 3e-2
        grads = grads[1:]
    for (n, _), gr in zip(named, grads):
        assert cosine(gr.float(), T(g["gparam"][n])) > 0.999, n
        assert rel_l2(gr.float(), T(g["gparam"][n])) < 3e-2, (n, rel_l2(gr.float(), T(g["gparam"][n])))
    for n, v in g.get("state_after", {}).items():
        assert rel_l2(m.state_dict()[n].float(), T(v)) < 1e-2, n


BLOCKS = {
    "bottleneck": lambda: yolo_blocks.DarknetBottleneck(16, 16, 1.0, True, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "csp": lambda: yolo_blocks.CSPLayer(32, 32, n=2, shortcut=True, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "sppf": lambda: yolo_blocks.SPPF(32, 32, kernel_sizes=5, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "spp": lambda: yolo_blocks.SPPF(32, 32, kernel_sizes=(5, 9, 13), norm_cfg=BN_YOLO, act_cfg=dict(type="Swish")),
    "focus": lambda: yolo_blocks.Focus(3, 16, 3, norm_cfg=BN_YOLO, act_cfg=dict(type="Swish")),
    "up": lambda: yolo_blocks.UpsamplingModule(32, 16, 1, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
    "down": lambda: yolo_blocks.DownsamplingModule(16, 32, 1, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU")),
}


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_hip_block_vs_reference_vectors(name):
    g = load("block_" + name)
    m = BLOCKS[name]()
    for mm in m.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.eps, mm.momentum = 1e-3, 0.03
    missing, unexpected = m.load_state_dict({kk: T(v) for kk, v in g["state"].items()}, strict=True)
    assert not missing and not unexpected
    m.to(dev()).train()
    xs = lst(g["x"])
    if name == "focus":
        out = m(xs[0].to(dev()))
        outs = [out]
        for o, e in zip(outs, lst(g["out"])):
            assert rel_l2(o.float(), e) < 2e-2
        return
    xs = [x.to(torch.bfloat16).to(dev()).contiguous(memory_format=torch.channels_last) for x in xs]
    outs, gx, gpar = run(m, xs, lst(g["cot"]))
    for o, e in zip(outs, lst(g["out"])):
        assert rel_l2(o.float(), e) < 2e-2, rel_l2(o.float(), e)
    for a, e in zip(gx, lst(g["gx"])):
        assert cosine(a.float(), e) > 0.995, cosine(a.float(), e)
    for n, v in g["gparam"].items():
        assert cosine(gpar[n].float(), T(v)) > 0.99, (n, cosine(gpar[n].float(), T(v)))


def test_hip_backbone_v5n_vs_reference_vectors():
    g = load("backbone_v5n_full")
    m = yolov5.YOLOv5CSPDarknet("cspdark_n")
    missing, unexpected = m.load_state_dict({kk: T(v) for kk, v in g["state"].items()}, strict=True)
    assert not missing and not unexpected
    # noise floor: the oracle (same weights) under CPU bf16 autocast vs the fp32 reference vectors
    from oracle import torch_ref as R
    om = R.YOLOv5CSPDarknet("cspdark_n")
    om.load_state_dict({kk: T(v) for kk, v in g["state"].items()}, strict=True)
    om.train()
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        floor = [rel_l2(f.float(), e) for f, e in zip(om(T(g["x"])), lst(g["out"]))]
    m.to(dev()).train()
    feats = m(T(g["x"]).to(dev()))
    for f, e, fl in zip(feats, lst(g["out"]), floor):
        assert tuple(f.shape) == tuple(e.shape)
        assert rel_l2(f.float(), e) < max(3e-2, 1.5 * fl), (rel_l2(f.float(), e), fl)
    loss = sum((f.float() * c.to(dev())).sum() for f, c in zip(feats, lst(g["cot"])))
    loss.backward()
    # gradient criteria against the STORAGE EMULATOR (tests/storage_emulator.py: the oracle with the engine's bf16 rounding points
    # and nothing else; test_gpu_storage_emulator.py shows the engine reproduces it block by block): the engine may be as far from
    # the fp32 reference vectors as 16-bit storage alone puts the emulator, plus a small margin for rounding-flip divergence
    import storage_emulator as E
    em = R.YOLOv5CSPDarknet("cspdark_n")
    em.load_state_dict({kk: T(v) for kk, v in g["state"].items()}, strict=True)
    em.train()
    E.emulate_storage(em, torch.bfloat16)
    sum((f * c).sum() for f, c in zip(em(T(g["x"])), lst(g["cot"]))).backward()
    gs = m.stem.conv.weight.grad
    floor_stem = cosine(em.stem.conv.weight.grad, T(g["g_stem"]))
    assert cosine(gs.float(), T(g["g_stem"])) > floor_stem - 0.05, (cosine(gs.float(), T(g["g_stem"])), floor_stem)
    ep = dict(em.named_parameters())

    def off(grad, n):
        ref = float(g["gparam_norms"][n])
        return abs(float(grad.float().norm()) - ref) > 0.25 * max(ref, 1e-3)
    bad = [n for n, p in m.named_parameters() if off(p.grad, n)]
    bad_emulator = [n for n in ep if off(ep[n].grad, n)]
    assert len(bad) <= len(bad_emulator) + 3, (bad[:8], bad_emulator[:8])


def test_convert_to_hip_keeps_state_dict_and_matches():
    """Module-swap pass on an oracle-built (plain torch) module: same keys, same tensors, HIP forward."""
    from oracle import torch_ref as R
    torch.manual_seed(0)
    ref = R.CSPLayer(32, 32, n=1, norm_cfg=BN_YOLO, act_cfg=dict(type="SiLU"))
    x = torch.randn(2, 32, 8, 8)
    ref.train()
    yr = ref(x)
    import copy
    hip = bricks.convert_to_hip(copy.deepcopy(ref))
    assert list(hip.state_dict().keys()) == list(ref.state_dict().keys())
    assert any(isinstance(mm, bricks.HipConv2d) for mm in hip.modules())
    hip.to(dev()).train()
    y = hip(x.to(dev()))
    assert rel_l2(y.float(), yr) < 3e-2


def test_yolov5s_end_to_end_vs_oracle():
    """Full YOLOv5-s: same weights, same synthetic batch (SURVEY §8d config 2 at reduced size) ->
    loss and its three terms within 1e-2 relative. End-to-end gradients of a randomly initialised 60-conv network are
    noise-limited at bf16 storage precision; the gradient criterion is stated against the storage emulator (see below and
    tests/test_gpu_storage_emulator.py, which proves block by block that the engine computes exactly that rounding model)."""
    from oracle import torch_ref as R
    torch.manual_seed(0)
    ref = R.YOLOv5(80, "s")
    hip = yolov5.YOLOv5(80, "s", max_targets=64)
    sd = ref.state_dict()
    missing, unexpected = hip.load_state_dict(sd, strict=False)
    assert all(k.startswith("loss.") for k in missing), missing
    assert not unexpected, unexpected
    imgs, targets = R.synthetic_batch(4, 128, seed=1029, max_boxes=10)
    ref.train()
    lr = ref(imgs, targets, "train")
    lr["loss"].backward()
    hip.to(dev()).train()
    tg = [{k: v.to(dev()) for k, v in t.items()} for t in targets]
    lh = hip(imgs.to(dev()), tg, "train")
    lh["loss"].backward()
    torch.cuda.synchronize()
    for k in ("loss", "box_loss", "obj_loss", "cls_loss"):
        a, b = float(lh[k]), float(lr[k])
        assert abs(a - b) <= 1e-2 * abs(b) + 1e-4, (k, a, b)
    rp = dict(ref.named_parameters())
    cos = []
    for n, p in hip.named_parameters():
        if p.grad is None or n not in rp:
            continue
        cos.append((cosine(p.grad.float(), rp[n].grad), n))
    cos.sort()
    # the floor is the storage emulator (the oracle + the engine's bf16 rounding points): storage rounding alone puts it at ~0.91
    # median cosine against fp32; the engine must be within 0.02 of that and its worst parameter within 0.06 of the emulator's worst
    import storage_emulator as E
    emu = R.YOLOv5(80, "s")
    emu.load_state_dict(sd)
    emu.train()
    E.emulate_storage(emu, torch.bfloat16)
    emu(imgs, targets, "train")["loss"].backward()
    floor = sorted(cosine(p.grad.float(), rp[n].grad) for n, p in emu.named_parameters())
    assert np.median([c for c, _ in cos]) > np.median(floor) - 0.02, (np.median([c for c, _ in cos]), np.median(floor), cos[:5])
    assert cos[0][0] > floor[0] - 0.06, (cos[:5], floor[:5])
    # running statistics followed the reference's BatchNorm update
    rb = dict(ref.named_buffers())
    for n, b in hip.named_buffers():
        if "running_var" in n:
            assert rel_l2(b.float(), rb[n]) < 3e-2, n


def test_yolov5s_val_mode_nms():
    from oracle import torch_ref as R
    torch.manual_seed(0)
    hip = yolov5.YOLOv5(80, "s", max_targets=64).to(dev())
    imgs, targets = R.synthetic_batch(2, 128, seed=3, max_boxes=5)
    tg = [{k: v.to(dev()) for k, v in t.items()} for t in targets]
    hip.eval()
    with torch.no_grad():
        losses, outs = hip(imgs.to(dev()), tg, "val")
    assert len(outs) == 2 and all(o["boxes"].shape[1] == 4 for o in outs)
    # the post-processing path itself (decode output -> NMS) is bit-identical to the oracle's on the same tensor
    with torch.no_grad():
        z, _ = hip.forward_features(imgs.to(dev()))
    a = yolov5.non_max_suppression(z.clone(), 0.001, 0.6, multi_label=True)
    b = R.non_max_suppression(z.cpu().clone(), 0.001, 0.6, multi_label=True)
    for u, v in zip(a, b):
        assert torch.equal(u.cpu(), v)


def test_hip_ops_refuse_cpu_tensors():
    from cvpytorch_amd import lib as L
    with pytest.raises(L.CvhipError):
        ops.max_pool2d(torch.zeros(1, 8, 4, 4, dtype=torch.bfloat16), 2)


def test_out_slices_and_copy_free_cat():
    """Concat elimination: a ConvModule writing into its channel slice of a concat buffer (`out=`) gives bit-identical results to the
    module writing a fresh tensor, `ops.cat` of in-place slices aliases the buffer (no copy), `cat(into=)` copies only what is not
    already in place, and gradients flow as with the copying concat."""
    from cvpytorch_amd import bricks, ops
    torch.manual_seed(11)
    m1 = bricks.HipConvModule(32, 24, 3, padding=1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")).to(dev()).train()
    m2 = bricks.HipConvModule(32, 40, 1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")).to(dev()).train()
    x = torch.randn(3, 32, 18, 14, device=dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ref = ops.cat([m1(xa), m2(xa)])
    buf = ops.empty_nhwc(3, 64, 18, 14, dev())
    a, b = m1(xb, out=buf[:, :24]), m2(xb, out=buf[:, 24:])
    got = ops.cat([a, b])
    assert got.data_ptr() == buf.data_ptr() and tuple(got.shape) == (3, 64, 18, 14)
    assert torch.equal(got.float(), ref.float())
    cot = torch.randn(ref.shape, device=dev()).to(ref.dtype)
    (ref.float() * cot.float()).sum().backward()
    g_ref = [p.grad.clone() for p in list(m1.parameters()) + list(m2.parameters())]
    for p in list(m1.parameters()) + list(m2.parameters()):
        p.grad = None
    (got.float() * cot.float()).sum().backward()
    assert torch.equal(xa.grad.float(), xb.grad.float())
    for p, g in zip(list(m1.parameters()) + list(m2.parameters()), g_ref):
        assert rel_l2(p.grad, g) < 1e-5          # wgrad atomics: order-dependent rounding only
    # partial in-place: first input produced in place, second copied
    buf2 = ops.empty_nhwc(3, 64, 18, 14, dev())
    with torch.no_grad():
        a2 = m1(x, out=buf2[:, :24])
        other = m2(x)
        got2 = ops.cat([a2, other], into=buf2)
        ref2 = ops.cat([m1(x), m2(x)])
    assert got2.data_ptr() == buf2.data_ptr() and torch.equal(got2.float(), ref2.float())
    with pytest.raises(Exception):
        m1(x, out=buf2[:, :16])               # wrong slice width


@pytest.mark.parametrize("kind", ["darknet", "resnet"])
def test_grad_link_equals_autograd_accumulation(kind):
    """Skip-connection gradient folded into conv1's dgrad epilogue (ops.GradLink / cvhip_conv2d_dgrad_add) == the gradient autograd
    accumulates with a separate add (only the rounding point moves: the sum is formed in fp32 before the bf16 store)."""
    from cvpytorch_amd import deeplab, ops, yolo_blocks
    from cvpytorch_amd import lib as L
    res = {}
    for link in (True, False):
        torch.manual_seed(5)
        if kind == "darknet":
            m = yolo_blocks.DarknetBottleneck(64, 64, 1.0, True, act_cfg=dict(type="SiLU")).to(dev()).train()
            shape = (4, 64, 40, 36)
        else:
            m = deeplab.Bottleneck(256, 64).to(dev()).train()
            shape = (2, 256, 33, 65)
        yolo_blocks._GRAD_LINK = deeplab._GRAD_LINK = link
        calls = []
        orig = L.call
        try:
            L.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
            ops.L.call = L.call
            x = torch.randn(shape, generator=torch.Generator().manual_seed(6)).to(dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            pre = ops.add(x, x)                  # non-leaf input, as inside a network
            out = m(pre)
            cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(7)).to(dev()).to(out.dtype)
            (out.float() * cot.float()).sum().backward()
            torch.cuda.synchronize()
        finally:
            L.call = orig
            ops.L.call = orig
            yolo_blocks._GRAD_LINK = deeplab._GRAD_LINK = True
        assert ("cvhip_conv2d_dgrad_add" in calls) == link
        res[link] = (out.float().cpu(), x.grad.float().cpu(), [p.grad.float().cpu().clone() for p in m.parameters()])
    assert torch.equal(res[True][0], res[False][0])
    assert rel_l2(res[True][1], res[False][1]) < 5e-3
    for a, b in zip(res[True][2], res[False][2]):
        assert rel_l2(a, b) < 5e-3


def test_resnet_tail_fused_into_bn_pass_equals_separate_add():
    """relu(bn3(conv3(x)) + identity) inside conv3's BN pass (cfg.res_pre / cvhip_bn_add_act_fwd) vs the separate add_act pass:
    same values up to one bf16 rounding of the intermediate, same gradients (identity branch through the GradLink in both)."""
    from cvpytorch_amd import deeplab
    res = {}
    for fused in (True, False):
        torch.manual_seed(8)
        m = deeplab.Bottleneck(256, 64).to(dev()).train()
        deeplab._FUSE_TAIL = fused
        try:
            x = torch.randn((2, 256, 33, 65), generator=torch.Generator().manual_seed(9)).to(dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            out = m(ops.add(x, x))
            cot = torch.randn(out.shape, generator=torch.Generator().manual_seed(10)).to(dev()).to(out.dtype)
            (out.float() * cot.float()).sum().backward()
            torch.cuda.synchronize()
        finally:
            deeplab._FUSE_TAIL = True
        res[fused] = (out.float().cpu(), x.grad.float().cpu(), [p.grad.float().cpu().clone() for p in m.parameters()],
                      [b.float().cpu().clone() for b in m.buffers() if b.dtype == torch.float32])
    assert rel_l2(res[True][0], res[False][0]) < 5e-3
    assert float(((res[True][0] > 0) != (res[False][0] > 0)).float().mean()) < 2e-3      # the ReLU mask moves only at rounding ties
    # a fraction f of flipped ReLU-mask elements moves the gradient by ~sqrt(f) in relative L2 (f < 2e-3 => < 4.5e-2)
    assert rel_l2(res[True][1], res[False][1]) < 6e-2
    for a, b in zip(res[True][2], res[False][2]):
        assert rel_l2(a, b) < 6e-2
    for a, b in zip(res[True][3], res[False][3]):
        assert torch.equal(a, b)          # BN statistics come from the conv epilogue: identical


@pytest.mark.parametrize("k,s", [(3, 2), (1, 1), (3, 1), (1, 2)])
def test_fanout_link_folds_the_side_gradient_into_the_main_dgrad(k, s):
    """Round 5: ops.fanout_linked — the side consumer's gradient is parked and added by the main consumer's dgrad epilogue
    (cvhip_conv2d_dgrad_add; stride-2 3x3 = four interleaved parity classes) instead of by an add pass. The fused form adds in fp32
    before the one 16-bit rounding (the add pass rounds twice): dx equal to one bf16 ulp of the sum (rel-L2 <= 4e-3), every parameter
    gradient untouched."""
    import torch
    from cvpytorch_amd import bricks, ops
    from cvpytorch_amd import lib as L
    d = torch.device("cuda:0")
    torch.manual_seed(8)
    main = bricks.HipConvModule(64, 64, k, s, k // 2, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")).to(d).train()
    side = bricks.HipConv2d(64, 24, 1).to(d)
    x0 = torch.randn(4, 64, 40, 40, device=d).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    out = {}
    calls = []
    real = L.call

    def spy(name, *a):
        calls.append(name)
        return real(name, *a)

    try:
        L.call = spy
        for flag in (False, True):
            ops._FANOUT_LINK = flag
            calls.clear()
            for p_ in list(main.parameters()) + list(side.parameters()):
                p_.grad = None
            x = x0.clone().requires_grad_(True)
            h = bricks.HipSiLU()(x)                      # a producer in front, so that x's consumers are engine ops on an engine tensor
            a, b, link = ops.fanout_linked(h)
            za = main(a, dx_link=link)
            zb = side(ops.fanout_side(h, b, link))
            (za.float().square().mean() + zb.float().square().mean()).backward()
            torch.cuda.synchronize()
            out[flag] = (x.grad.float().clone(), [p_.grad.float().clone() for p_ in list(main.parameters()) + list(side.parameters())], list(calls))
    finally:
        L.call = real
        ops._FANOUT_LINK = True
    assert "cvhip_add2d" in out[False][2] and "cvhip_add2d" not in out[True][2]
    assert any(n in out[True][2] for n in ("cvhip_conv2d_dgrad_add", "cvhip_conv1x1_bwd_fused_acc", "cvhip_conv1x1_bwd_fused"))
    e = float((out[True][0] - out[False][0]).norm() / out[False][0].norm())
    assert e <= 4e-3, e
    for u, v in zip(out[True][1], out[False][1]):
        assert float((u - v).norm() / max(float(v.norm()), 1e-12)) <= 1e-3
