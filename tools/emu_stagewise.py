"""Dev tool: WHERE along the network does the engine part from the storage emulator? Forward values and activation gradients at
the backbone / neck outputs: engine vs emulator, emulator vs its second realisation (acc64), emulator vs fp32 oracle."""
import os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import storage_emulator as E
from oracle import torch_ref as R
from test_gpu_storage_emulator import _setup, FP16_LOSS_SCALE


def tap(model, store):
    for part in ("backbone", "neck"):
        mod = getattr(model, part)
        orig = mod.forward

        def fwd(self, *a, _orig=orig, _part=part, **k):
            outs = _orig(*a, **k)
            for i, o in enumerate(outs):
                store["f_%s%d" % (_part, i)] = o.detach().float().cpu()
                if o.requires_grad:
                    o.register_hook(lambda g, key="g_%s%d" % (_part, i): store.__setitem__(key, g.detach().float().cpu()))
            return outs
        mod.forward = types.MethodType(fwd, mod)


def rel(a, b):
    return float((a - b).norm() / (b.norm() + 1e-30))


def cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-30))


def main():
    from cvpytorch_amd import ops, yolov5
    from cvpytorch_amd.arena import FlatTrainState
    dev = torch.device("cuda:0")
    for precision in sys.argv[1:] or ("bf16", "fp16"):
        dt = torch.float16 if precision == "fp16" else torch.bfloat16
        ls = FP16_LOSS_SCALE if precision == "fp16" else 1.0
        ref, imgs, targets = _setup("s", 4, 128)
        stores = {k: {} for k in ("oracle", "emu", "emu64", "engine")}
        tap(ref, stores["oracle"])
        ref(imgs, targets, "train")["loss"].backward()
        for key, a64 in (("emu", False), ("emu64", True)):
            m = R.YOLOv5(80, "s").train(); m.load_state_dict(ref.state_dict()); E.emulate_storage(m, dt, acc64=a64)
            tap(m, stores[key])
            (m(imgs, targets, "train")["loss"] * ls).backward()
        ops.set_precision(precision)
        hip = yolov5.YOLOv5(80, "s", max_targets=64, fused_loss=True)
        hip.load_state_dict(ref.state_dict(), strict=False); hip.to(dev).train()
        state = FlatTrainState(hip, use_ema=False, loss_scaling=precision == "fp16", init_scale=ls)
        tap(hip, stores["engine"])
        gts = yolov5.targets_to_tensor([{k: v.to(dev) for k, v in t.items()} for t in targets], 64, dev)
        state.scale_loss(hip(imgs.to(dev), gts, "train")["loss"]).backward()
        torch.cuda.synchronize()
        # teacher-forced head + loss: the emulator's own neck outputs (16-bit representable) through the engine's detect + loss
        xs = [stores["emu"]["f_neck%d" % i].to(dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True) for i in range(3)]
        raw = hip.detect.forward_raw(xs)
        state.scale_loss(hip.loss_from_features(raw, gts)["loss"]).backward()
        torch.cuda.synchronize()
        tf = [x.grad.float().cpu() for x in xs]
        # ... and teacher-forced neck + head + loss from the emulator's backbone outputs
        xb = [stores["emu"]["f_backbone%d" % i].to(dev).to(dt).contiguous(memory_format=torch.channels_last).requires_grad_(True) for i in range(3)]
        stores["engine_tf"] = {}
        raw = hip.detect.forward_raw(hip.neck(xb))
        state.scale_loss(hip.loss_from_features(raw, gts)["loss"]).backward()
        torch.cuda.synchronize()
        tfb = [x.grad.float().cpu() for x in xb]
        ops.set_precision("bf16")
        print("==", precision)
        for i in range(3):
            print("  teacher-forced head+loss   g_neck%d      engine~emu %.6f   (free-running emu64~emu %.5f)" % (i, cos(tf[i], stores["emu"]["g_neck%d" % i]), cos(stores["emu64"]["g_neck%d" % i], stores["emu"]["g_neck%d" % i])))
        for i in range(3):
            print("  teacher-forced neck+head   g_backbone%d  engine~emu %.6f   (free-running emu64~emu %.5f)" % (i, cos(tfb[i], stores["emu"]["g_backbone%d" % i]), cos(stores["emu64"]["g_backbone%d" % i], stores["emu"]["g_backbone%d" % i])))
        for k in sorted(stores["emu"]):
            e, m, m2, o = (stores[n].get(k) for n in ("engine", "emu", "emu64", "oracle"))
            if e is None:
                print("  %-12s (engine tensor not tapped)" % k); continue
            if k.startswith("f_"):
                print("  %-12s rel  engine~emu %.2e   emu64~emu %.2e   emu~oracle %.2e" % (k, rel(e, m), rel(m2, m), rel(m, o)))
            else:
                print("  %-12s cos  engine~emu %.5f  emu64~emu %.5f  emu~oracle %.5f  engine~oracle %.5f" % (k, cos(e, m), cos(m2, m), cos(m, o), cos(e, o)))


main()
