"""Dev tool: eval-mode forward passes (YOLOv5-s 640x640 batch 64, then DeepLabv3+ R50 1024x512 batch 16; BatchNorm folded by
deploy.fuse_model) for `rocprofv3 --kernel-trace --stats`: the kernel list is the evidence that a ConvModule is ONE launch in
inference — no ew_kernel / colreduce launch appears (VERDICT r03 task 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import deeplab, deploy, ops, yolov5
from cvpytorch_amd.data import synthetic_detection_batch, synthetic_segmentation_batch
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(os.environ.get("STEPS", "5"))
with torch.no_grad():
    if which in ("both", "yolov5s"):
        m = yolov5.YOLOv5(80, "s", fused_loss=True).to(dev).eval()
        deploy.fuse_model(m)
        imgs, _ = synthetic_detection_batch(64, 640, seed=7, device=dev)
        x = ops.images_to_nhwc(imgs, cpad=8)
        for _ in range(n):
            out = m.forward_features(x)[0]
        torch.cuda.synchronize()
        print("yolov5s", n, "forward passes", bool(torch.isfinite(out.float()).all()))
    if which in ("both", "deeplab"):
        m = deeplab.EncoderDecoder(19, output_stride=32).to(dev).eval()
        deploy.fuse_model(m)
        x, _ = synthetic_segmentation_batch(16, (512, 1024), device=dev)
        for _ in range(n):
            out = m.forward_features(x)[1][0]
        torch.cuda.synchronize()
        print("deeplab", n, "forward passes", bool(torch.isfinite(out.float()).all()))
