"""Workload for the PMC (HBM traffic) passes: a calibration copy with a known byte count + eager YOLOv5-s train steps.
Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and again with `--pmc WRITE_SIZE` (separate passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L, yolov5
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_detection_batch
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
# calibration: copy2d of 64 x 320 x 320 x 32 bf16 = 419,430,400 B read and written (>> 256 MiB Infinity Cache)
M, C = 64 * 320 * 320, 32
a = torch.randn(M, C, device=dev).to(torch.bfloat16)
b = torch.empty_like(a)
for _ in range(3):
    L.call("cvhip_copy2d", a.data_ptr(), C, b.data_ptr(), C, M, C, st)
torch.cuda.synchronize()
B = 64
model = yolov5.YOLOv5(80, "s", max_targets=B * 20, fused_loss=True).to(dev).train()
state = FlatTrainState(model, use_ema=True)
step = FlatTrainStep(model, state)
imgs, targets = synthetic_detection_batch(B, 640, device=dev)
gts = yolov5.targets_to_tensor(targets, B * 20, dev)
for _ in range(int(os.environ.get("PMC_STEPS", "2"))):
    step(imgs, gts)
torch.cuda.synchronize()
print("done")
