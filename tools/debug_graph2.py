"""Dev tool: which component misbehaves under hipGraph replay? Each component is captured alone with static
inputs; then inputs AND parameters are changed, the graph is replayed and compared with an eager run."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import yolov5, ops, bricks, yolo_blocks
from cvpytorch_amd.data import synthetic_detection_batch

dev = torch.device("cuda:0")
BN = dict(type="BN", momentum=0.03, eps=0.001)

def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))

def check(name, build, make_inputs, needs_x_grad=True):
    torch.manual_seed(0)
    m = build().to(dev).train()
    xs = make_inputs(0)
    static = [x.clone().requires_grad_(needs_x_grad and x.dtype == torch.bfloat16) for x in xs]
    def run(inputs):
        out = m(*inputs)
        outs = [o for o in (out if isinstance(out, (list, tuple)) else [out]) if torch.is_tensor(o)]
        loss = sum((o.float() ** 2).mean() for o in outs)
        for p in m.parameters():
            p.grad = None
        for x in inputs:
            if x.requires_grad:
                x.grad = None
        loss.backward()
        return loss
    for _ in range(2):
        run(static)
    torch.cuda.synchronize()
    # persistent grads so the graph accumulates into fixed storage
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    def run_acc(inputs):
        out = m(*inputs)
        outs = [o for o in (out if isinstance(out, (list, tuple)) else [out]) if torch.is_tensor(o)]
        loss = sum((o.float() ** 2).mean() for o in outs)
        loss.backward()
        return loss
    g = torch.cuda.CUDAGraph()
    ops.bump_weights_epoch()
    with torch.cuda.graph(g):
        gl = run_acc(static)
    res = []
    for it in range(2):
        with torch.no_grad():
            for p in m.parameters():
                p.mul_(1.0 + 0.05 * (it + 1))
                p.grad.zero_()
            new = make_inputs(it + 1)
            for s, n in zip(static, new):
                s.copy_(n)
                if s.grad is not None:
                    s.grad.zero_()
        ops.bump_weights_epoch()
        g.replay()
        torch.cuda.synchronize()
        gp = [p.grad.clone() for p in m.parameters()]
        gx = [s.grad.clone() for s in static if s.grad is not None]
        glv = float(gl)
        with torch.no_grad():
            for p in m.parameters():
                p.grad.zero_()
            for s in static:
                if s.grad is not None:
                    s.grad.zero_()
        ops.bump_weights_epoch()
        el = run_acc(static)
        torch.cuda.synchronize()
        e = max([rel(a, p.grad) for a, p in zip(gp, m.parameters())] + [0.0])
        ex = max([rel(a, s.grad) for a, s in zip(gx, [s for s in static if s.grad is not None])] + [0.0])
        res.append((abs(glv - float(el)) / max(abs(float(el)), 1e-30), e, ex))
    print("%-28s loss_rel %.2e/%.2e  param_grad_rel %.2e/%.2e  x_grad_rel %.2e/%.2e" % (name, res[0][0], res[1][0], res[0][1], res[1][1], res[0][2], res[1][2]), flush=True)

def nhwc(*shape):
    def f(seed):
        g = torch.Generator().manual_seed(seed)
        return [torch.randn(*shape, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)]
    return f

check("convmodule 3x3", lambda: bricks.HipConvModule(32, 64, 3, padding=1, norm_cfg=BN, act_cfg=dict(type="SiLU")), nhwc(4, 32, 24, 24))
check("convmodule 3x3 s2", lambda: bricks.HipConvModule(32, 64, 3, stride=2, padding=1, norm_cfg=BN, act_cfg=dict(type="SiLU")), nhwc(4, 32, 24, 24))
check("conv 1x1 K=255 bias", lambda: bricks.HipConv2d(64, 255, 1), nhwc(4, 64, 12, 12))
def img(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(4, 3, 64, 64, generator=g).to(dev)]
check("stem k6 s2 (C=3 image)", lambda: bricks.HipConvModule(3, 32, 6, stride=2, padding=2, norm_cfg=BN, act_cfg=dict(type="SiLU")), img, needs_x_grad=False)
check("bottleneck (residual)", lambda: yolo_blocks.DarknetBottleneck(32, 32, 1.0, True, norm_cfg=BN, act_cfg=dict(type="SiLU")), nhwc(4, 32, 16, 16))
check("csp (cat)", lambda: yolo_blocks.CSPLayer(32, 32, n=1, norm_cfg=BN, act_cfg=dict(type="SiLU")), nhwc(4, 32, 16, 16))
check("sppf (maxpool)", lambda: yolo_blocks.SPPF(32, 32, 5, norm_cfg=BN, act_cfg=dict(type="SiLU")), nhwc(4, 32, 12, 12))
def two(seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    return [mk(4, 32, 6, 6), mk(4, 16, 12, 12)]
check("upsampling module", lambda: yolo_blocks.UpsamplingModule(32, 16, 1, norm_cfg=BN, act_cfg=dict(type="SiLU")), two)

# loss alone (pure torch ops) under graph
torch.manual_seed(0)
loss = yolov5.YOLOv5Loss(80).to(dev)
imgs, targets = synthetic_detection_batch(4, 96, seed=7, max_boxes=8, device=dev)
def mkp(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(4, 3, s, s, 85, generator=g).to(dev) for s in (12, 6, 3)]
ps = [p.requires_grad_(True) for p in mkp(0)]
gts = yolov5.targets_to_tensor(targets, 64, dev)
for _ in range(2):
    l, _ = loss(ps, gts); l.backward()
for p in ps:
    p.grad = torch.zeros_like(p)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    gl, _ = loss(ps, gts)
    gl.backward()
for it in range(2):
    imgs2, t2 = synthetic_detection_batch(4, 96, seed=20 + it, max_boxes=8, device=dev)
    with torch.no_grad():
        gts.copy_(yolov5.targets_to_tensor(t2, 64, dev))
        for p, n in zip(ps, mkp(it + 1)):
            p.copy_(n); p.grad.zero_()
    g.replay(); torch.cuda.synchronize()
    gg = [p.grad.clone() for p in ps]; glv = float(gl)
    for p in ps:
        p.grad.zero_()
    el, _ = loss(ps, gts); el.backward(); torch.cuda.synchronize()
    print("loss only it%d: loss_rel %.2e grad_rel %.2e" % (it, abs(glv - float(el)) / abs(float(el)), max(rel(a, p.grad) for a, p in zip(gg, ps))), flush=True)
