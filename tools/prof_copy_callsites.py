"""Dev tool: which python call sites launch strided-copy / add kernels during one eager YOLOv5-s train step (torch profiler with stacks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cvpytorch_amd import yolov5
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_detection_batch
dev = torch.device("cuda:0")
B = 64
model = yolov5.YOLOv5(80, "s", max_targets=B * 20, fused_loss=True).to(dev).train()
state = FlatTrainState(model, use_ema=False)
step = FlatTrainStep(model, state)
imgs, targets = synthetic_detection_batch(B, 640, device=dev)
gts = yolov5.targets_to_tensor(targets, B * 20, dev)
for _ in range(3):
    step(imgs, gts)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(imgs, gts)
    torch.cuda.synchronize()
import collections
small = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy") and ev.device_time_total <= 8:
        stack = [s for s in ev.stack if "cvpytorch_amd" in s][:2]
        small[(ev.name, str(ev.input_shapes)[:60], " <- ".join(stack))] += 1
for k, v in small.most_common(25):
    print("%4d x %s %s\n        %s" % ((v,) + k))
rows = []
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::add", "aten::add_", "aten::contiguous", "aten::clone") and ev.device_time_total > 8:
        stack = [s for s in ev.stack if "cvpytorch_amd" in s or "autograd" in s][:4]
        rows.append((ev.device_time_total, ev.name, str(ev.input_shapes)[:80], " <- ".join(stack)))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%8.1f us  %-14s %s\n      %s" % r)
