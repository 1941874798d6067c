"""Side workloads: BASELINE configs 4 and 5 at N = 1 (the 8-GPU DDP runs are the driver's): YOLOX-s 640x640 bs 64 and
YOLOv7-l 1280x1280 bs 16 train steps (synthetic, SGD-nesterov + EMA in the fused arena step, hipGraph replay). YOLOX-s runs in
bf16; YOLOv7-l runs in fp16 storage with dynamic loss scaling as BASELINE config 5 names it (`--bf16` for the bf16 number).
    python tools/bench_extra.py [yolox] [yolov7] [--steps K] [--bf16]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import ops, yolov5, yolov7, yolox
from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
from cvpytorch_amd.data import synthetic_detection_batch

dev = torch.device("cuda:0")
steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 10
which = [a for a in sys.argv[1:] if a in ("yolox", "yolov7")] or ["yolox", "yolov7"]


def run(name, model, imgs, gts, flops_per_img, bytes_per_img):
    state = FlatTrainState(model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4, use_ema=True)
    step = FlatTrainStep(model, state)
    for _ in range(3):
        step(imgs, gts)
    step.capture(imgs, gts)
    imgs, gts = step.static_imgs, step.static_targets
    for _ in range(2):
        step(imgs, gts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = step(imgs, gts)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ips = imgs.shape[0] * steps / el
    sc = state.loss_scale() if state.loss_scaling else None
    print(json.dumps({"workload": name, "dtype": ops.precision(), "loss_scale": sc, "value": round(ips, 1), "unit": "images/sec", "ms_per_step": round(1e3 * el / steps, 2), "batch": imgs.shape[0],
                      "graphs": 1 if step.g2 is None else 2, "final_loss": round(float(losses["loss"]), 4),
                      "step_roofline": {"mfma_frac": round(ips * flops_per_img / 2.5e15, 4), "hbm_frac": round(ips * bytes_per_img / 8e12, 4)}}), flush=True)


if "yolox" in which:
    torch.manual_seed(1029)
    B = 64
    m = yolox.YOLOX(80, "s", max_labels=20, fused_loss="--torch-loss" not in sys.argv).to(dev).train()
    imgs, targets = synthetic_detection_batch(B, 640, device=dev)
    for t in targets:  # YOLOX targets are pixel-unit cxcywh (models/yolox.py:112-139)
        t["boxes"] = t["boxes"] * 640.0
    gts = yolox.targets_to_padded(targets, 20, dev)
    run("coco_yolox_s.yml YOLOX-s 640x640 bf16 bs64 (config 4 at N=1)", m, imgs, gts, 80.06e9, 444e6)
if "yolov7" in which:
    torch.manual_seed(1029)
    B = 16
    prec = "bf16" if "--bf16" in sys.argv else "fp16"
    ops.set_precision(prec)
    m = yolov7.YOLOv7(80, 1.0, max_targets=B * 20, fused_loss=True).to(dev).train()
    imgs, targets = synthetic_detection_batch(B, 1280, device=dev)
    gts = yolov5.targets_to_tensor(targets, B * 20, dev)
    run("coco_yolov7.yml YOLOv7-l 1280x1280 %s bs16 (config 5 at N=1; YOLOv5-style loss%s)" % (prec, ", dynamic loss scaling" if prec == "fp16" else ""),
        m, imgs, gts, 1269.2e9, 5037e6)
    ops.set_precision("bf16")
