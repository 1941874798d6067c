"""Dev tool: bisect a hipGraph-replay divergence (fwd+bwd graph vs optimizer graph)."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import yolov5, ops
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_detection_batch

dev = torch.device("cuda:0")
torch.manual_seed(1)
base = yolov5.YOLOv5(80, "n", max_targets=64).to(dev).train()
imgs, targets = synthetic_detection_batch(4, 96, seed=7, max_boxes=8, device=dev)
gts = yolov5.targets_to_tensor(targets, 64, dev)

def rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))

def fwdbwd(model):
    l = model(imgs, gts, "train")
    l["loss"].backward()
    return l

# eager twin
a = copy.deepcopy(base); sa = FlatTrainState(a, use_ema=True)
b = copy.deepcopy(base); sb = FlatTrainState(b, use_ema=True)
for _ in range(2):
    sa.pre_step(); fwdbwd(a); sa.step_kernels(); sa.post_step()
    sb.pre_step(); fwdbwd(b); sb.step_kernels(); sb.post_step()
torch.cuda.synchronize()
print("after warmup param rel", rel(sb.param, sa.param))
# graph of fwd+bwd only on b
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lb = fwdbwd(b)
torch.cuda.synchronize()
fwdbwd(a)
torch.cuda.synchronize()
print("capture-run grads rel", rel(sb.grad, sa.grad), "absmax", float(sb.grad.abs().max()), float(sa.grad.abs().max()))
for it in range(3):
    sa.grad.zero_(); sb.grad.zero_()
    g.replay(); fwdbwd(a)
    torch.cuda.synchronize()
    print("replay", it, "grads rel", rel(sb.grad, sa.grad), "absmax", float(sb.grad.abs().max()), float(sa.grad.abs().max()), "loss", float(lb["loss"]))
    # per-parameter worst
    worst = []
    for (n, p), (_, q) in zip(b.named_parameters(), a.named_parameters()):
        worst.append((rel(p.grad, q.grad), n))
    worst.sort(reverse=True)
    print("   worst:", worst[:4])
    # now optimizer eagerly on both
    sa.pre_step(); sa.step_kernels(); sa.post_step()
    sb.pre_step(); sb.step_kernels(); sb.post_step()
    torch.cuda.synchronize()
    print("   params rel", rel(sb.param, sa.param))

# ---- finer: compare forward outputs and head gradients between a fresh graph and eager ----
print("---- forward/backward tensor comparison")
c = copy.deepcopy(base); d = copy.deepcopy(base)
outs = {}
def hook(name, store):
    def f(mod, inp, out):
        store[name] = out
    return f
sc, sd = {}, {}
for n, m in c.named_modules():
    if n in ("backbone.stem", "backbone.stage1", "backbone.stage4", "neck"):
        m.register_forward_hook(hook(n, sc))
for n, m in d.named_modules():
    if n in ("backbone.stem", "backbone.stage1", "backbone.stage4", "neck"):
        m.register_forward_hook(hook(n, sd))
for mdl in (c, d):
    with torch.no_grad():
        pass
fwdbwd(c); c.zero_grad(); fwdbwd(d); d.zero_grad()
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    ld = fwdbwd(d)
for it in range(2):
    for p in d.parameters():
        if p.grad is not None: p.grad.zero_()
    c.zero_grad(set_to_none=True)
    g2.replay(); lc = fwdbwd(c)
    torch.cuda.synchronize()
    print("it", it, "loss graph", float(ld["loss"]), "eager", float(lc["loss"]))
    for k in sc:
        a_, b_ = sc[k], sd[k]
        if isinstance(a_, (list, tuple)):
            a_, b_ = a_[0], b_[0]
        print("   ", k, rel(b_.float(), a_.float()))
    w = []
    for (n, p), (_, q) in zip(d.named_parameters(), c.named_parameters()):
        w.append((rel(p.grad, q.grad), n))
    w.sort(reverse=True)
    print("    worst grads", w[:5])
