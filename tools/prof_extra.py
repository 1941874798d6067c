"""Dev tool: BASELINE configs 4 / 5 at N = 1 for rocprofv3 --kernel-trace --stats (VERDICT r04 task 10):
    python tools/prof_extra.py yolox      YOLOX-s 640x640 bf16 batch 64, fused SimOTA loss
    python tools/prof_extra.py yolov7     YOLOv7-l 1280x1280 fp16 batch 16, dynamic loss scaling
The same steps bench.py times as config4_yolox_s / config5_yolov7l_fp16 (hipGraph replay after two eager steps)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpytorch_amd import ops, yolov5  # noqa: E402
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep  # noqa: E402
from cvpytorch_amd.data import synthetic_detection_batch  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "yolox"
dev = torch.device("cuda:0")
torch.manual_seed(1029)
if which == "yolox":
    from cvpytorch_amd import yolox
    batch = 64
    m = yolox.YOLOX(80, "s", max_labels=20, fused_loss=True).to(dev).train()
    imgs, targets = synthetic_detection_batch(batch, 640, device=dev)
    for t in targets:
        t["boxes"] = t["boxes"] * 640.0
    gts = yolox.targets_to_padded(targets, 20, dev)
else:
    from cvpytorch_amd import yolov7
    batch = 16
    ops.set_precision("fp16")
    m = yolov7.YOLOv7(80, 1.0, max_targets=batch * 20, fused_loss=True).to(dev).train()
    imgs, targets = synthetic_detection_batch(batch, 1280, device=dev)
    gts = yolov5.targets_to_tensor(targets, batch * 20, dev)
state = FlatTrainState(m, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4, use_ema=True)
step = FlatTrainStep(m, state)
for _ in range(2):
    step(imgs, gts)
step.capture(imgs, gts)
imgs, gts = step.static_imgs, step.static_targets
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    l = step(imgs, gts)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("%s %.1f img/s  %.2f ms/step  loss %.4f" % (which, batch * n / el, 1e3 * el / n, float(l["loss"])))
