"""Dev tool: which call sites copy what (cvhip_copy2d) in one eager train step: MODEL=yolox (default) | yolov5 | deeplab | stdc."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import ops, lib as L
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_detection_batch, synthetic_segmentation_batch
dev = torch.device("cuda:0")
which = os.environ.get("MODEL", "yolox")
if which == "yolox":
    from cvpytorch_amd import yolox
    B = 64
    model = yolox.YOLOX(80, "s", max_labels=20, fused_loss=True).to(dev).train()
    imgs, targets = synthetic_detection_batch(B, 640, device=dev)
    for t in targets:
        t["boxes"] = t["boxes"] * 640.0
    gts = yolox.targets_to_padded(targets, 20, dev)
elif which == "deeplab":
    from cvpytorch_amd import deeplab
    model = deeplab.EncoderDecoder(19, output_stride=32).to(dev).train()
    imgs, gts = synthetic_segmentation_batch(16, (512, 1024), device=dev)
elif which == "stdc":
    from cvpytorch_amd import segmentors
    model = segmentors.STDCEncoderDecoder().to(dev).train()
    imgs, gts = synthetic_segmentation_batch(16, (512, 1024), device=dev)
else:
    from cvpytorch_amd import yolov5
    B = 64
    model = yolov5.YOLOv5(80, "s", max_targets=B * 20, fused_loss=True).to(dev).train()
    imgs, targets = synthetic_detection_batch(B, 640, device=dev)
    gts = yolov5.targets_to_tensor(targets, B * 20, dev)
state = FlatTrainState(model, use_ema=False)
step = FlatTrainStep(model, state)
for _ in range(2):
    step(imgs, gts)
sites = collections.OrderedDict()
real = L.call


def spy(name, *a):
    if name in ("cvhip_copy2d", "cvhip_add2d"):
        fr = traceback.extract_stack(limit=6)
        key = name + " " + " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[:-1]))
        M, Cc = (a[4], a[5]) if name == "cvhip_copy2d" else (a[6], a[7])
        d = sites.setdefault((key, M, Cc), [0])
        d[0] += 1
    return real(name, *a)


L.call = spy
step(imgs, gts)
torch.cuda.synchronize()
L.call = real
for (key, M, Cc), (n,) in sites.items():
    print("%3d x  M=%-8d C=%-4d %6.1f MB  %s" % (n, M, Cc, M * Cc * 2 / 1e6, key))
