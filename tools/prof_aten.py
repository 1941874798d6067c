"""Dev tool: where do the small aten launches of one train step come from? (torch.profiler with stacks)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cvpytorch_amd import yolov5
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_detection_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
B = 16
model = yolov5.YOLOv5(80, "s", max_targets=B * 20).to(dev).train()
state = FlatTrainState(model)
step = FlatTrainStep(model, state)
imgs, targets = synthetic_detection_batch(B, 320, device=dev)
gts = yolov5.targets_to_tensor(targets, B * 20, dev)
for _ in range(2):
    step(imgs, gts)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(imgs, gts)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=8)
rows = sorted(ka, key=lambda e: -e.count)
seen = 0
for e in rows:
    if not e.key.startswith("aten::"):
        continue
    if e.key in ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zeros", "aten::cat", "aten::clone", "aten::to", "aten::_to_copy", "aten::mul", "aten::zero_") and e.count >= 20:
        print("== %s count=%d cpu_total=%.1fus" % (e.key, e.count, e.cpu_time_total))
        for s in e.stack[:8]:
            print("     ", s)
        seen += 1
    if seen > 14:
        break
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=25))
