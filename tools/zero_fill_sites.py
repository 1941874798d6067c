"""Dev tool: which call sites zero-fill what in one eager YOLOv5-s train step (MODEL=deeplab: DeepLabv3+) — bytes and caller."""
import sys, os, collections, traceback
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
else:
    B = 64
    model = yolov5.YOLOv5(80, "s", max_targets=B * 20, fused_loss=True).to(dev).train()
    state = FlatTrainState(model, use_ema=False)
    step = FlatTrainStep(model, state)
    imgs, targets = synthetic_detection_batch(B, 640, device=dev)
    gts = yolov5.targets_to_tensor(targets, B * 20, dev)
for _ in range(2):
    step(imgs, gts)
sites = collections.OrderedDict()
orig = ops.zero_fill


def spy(t):
    fr = traceback.extract_stack(limit=4)
    key = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in reversed(fr[:-1]))
    d = sites.setdefault((key, tuple(t.shape), str(t.dtype)), [0, 0])
    d[0] += 1
    d[1] += t.numel() * t.element_size()
    return orig(t)


ops.zero_fill = spy
import cvpytorch_amd.arena as A
step(imgs, gts)
torch.cuda.synchronize()
for (key, shape, dt), (n, by) in sites.items():
    print("%3d x %8.2f MB  %-28s %-14s %s" % (n, by / n / 1e6, shape, dt, key))
