"""Dev tool: DeepLabv3+ R50 1024x512 bs16 train steps (config 3) for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import deeplab
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_segmentation_batch
dev = torch.device("cuda:0")
torch.manual_seed(1029)
model = deeplab.EncoderDecoder(19, output_stride=32).to(dev).train()
state = FlatTrainState(model, lr=0.01, momentum=0.9, nesterov=True, weight_decay=5e-4, backbone_lr=0.001, use_ema=False)
step = FlatTrainStep(model, state)
imgs, tgt = synthetic_segmentation_batch(16, (512, 1024), device=dev)
for _ in range(2):
    step(imgs, tgt)
if os.environ.get("NO_GRAPH") != "1":
    step.capture(imgs, tgt)
    imgs, tgt = step.static_imgs, step.static_targets
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    l = step(imgs, tgt)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("deeplab %.1f img/s  %.2f ms/step  graphs=%s loss %.4f" % (16 * n / el, 1e3 * el / n, "1" if step.g2 is None else "2", float(l["loss"])))
