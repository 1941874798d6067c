"""Dev tool: per-shape timing of every conv launch of one train step (eager, HIP events): YOLOv5-s 640x640 bs64, or with
MODEL=deeplab DeepLabv3+ R50 1024x512 bs16, MODEL=yolox YOLOX-s bs64, MODEL=yolov7 YOLOv7-l 1280x1280 fp16 bs16."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import yolov5, ops
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_detection_batch
dev = torch.device("cuda:0")
if os.environ.get("MODEL") == "deeplab":
    from cvpytorch_amd import deeplab
    from cvpytorch_amd.data import synthetic_segmentation_batch
    B = 16
    model = deeplab.EncoderDecoder(19, output_stride=32).to(dev).train()
    state = FlatTrainState(model, lr=0.01, momentum=0.9, nesterov=True, weight_decay=5e-4, backbone_lr=0.001, use_ema=False)
    step = FlatTrainStep(model, state)
    imgs, gts = synthetic_segmentation_batch(B, (512, 1024), device=dev)
elif os.environ.get("MODEL") == "yolox":
    from cvpytorch_amd import yolox
    B = 64
    model = yolox.YOLOX(80, "s", max_labels=20, fused_loss=True).to(dev).train()
    state = FlatTrainState(model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4, use_ema=False)
    step = FlatTrainStep(model, state)
    imgs, targets = synthetic_detection_batch(B, 640, device=dev)
    for t in targets:
        t["boxes"] = t["boxes"] * 640.0
    gts = yolox.targets_to_padded(targets, 20, dev)
elif os.environ.get("MODEL") == "yolov7":
    from cvpytorch_amd import yolov7
    B = 16
    ops.set_precision("fp16")
    model = yolov7.YOLOv7(80, 1.0, max_targets=B * 20, fused_loss=True).to(dev).train()
    state = FlatTrainState(model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4, use_ema=False)
    step = FlatTrainStep(model, state)
    imgs, targets = synthetic_detection_batch(B, 1280, device=dev)
    gts = yolov5.targets_to_tensor(targets, B * 20, dev)
else:
    B = 64
    model = yolov5.YOLOv5(80, "s", max_targets=B * 20, fused_loss=True).to(dev).train()
    state = FlatTrainState(model, use_ema=False)
    step = FlatTrainStep(model, state)
    imgs, targets = synthetic_detection_batch(B, 640, device=dev)
    gts = yolov5.targets_to_tensor(targets, B * 20, dev)
for _ in range(3):
    step(imgs, gts)
ops.TIMER.enabled = True
ops.TIMER.reset()
for _ in range(3):
    step(imgs, gts)
torch.cuda.synchronize()
ops.TIMER.enabled = False
agg = collections.OrderedDict()
ew = collections.OrderedDict()
for (name, fl, by, e0, e1), (fn, geom) in zip(ops.TIMER.records, ops.TIMER.detail):
    if geom is None:  # BN / activation passes: aggregated per kernel
        d = ew.setdefault(name, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += e0.elapsed_time(e1)
        d[2] += by
        continue
    k = (fn.replace("cvhip_conv2d_", ""), geom, name)
    d = agg.setdefault(k, [0, 0.0, fl, by])
    d[0] += 1
    d[1] += e0.elapsed_time(e1)
rows = []
for (fn, g, name), (n, ms, fl, by) in agg.items():
    N, C, H, W, K, R, S, P, Q = g
    us = 1e3 * ms / n
    rows.append((ms / 3, fn, "%dx%d %d->%d k%d @%dx%d->%dx%d" % (N, 1, C, K, R, H, W, P, Q), n // 3, us, fl / us / 1e6, by / us / 1e3, name))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("total conv ms/step %.3f   wgrad %.3f  fprop %.3f  dgrad %.3f" % (tot, sum(r[0] for r in rows if r[1]=="wgrad"), sum(r[0] for r in rows if r[1]=="fprop"), sum(r[0] for r in rows if r[1]=="dgrad")))
for name, (n, ms, by) in ew.items():
    print("%6.3f ms/step  %-44s x%d  %8.1f us avg  %7.1f GB/s algorithmic" % (ms / 3, name, n // 3, 1e3 * ms / n, by / ms / 1e6))
NROWS = int(os.environ.get("TABLE_ROWS", "45"))
for r in rows[:NROWS]:
    print("%6.3f ms/step  %-6s %-34s x%d  %8.1f us  %7.1f TF  %7.1f GB/s  %s" % r)
