"""Dev tool: time a few conv fprop shapes (HipConv2d forward, no grad) — run under CVHIP_IGEMM_ABLATE=0/1/2 and
CVHIP_IGEMM_V1=1 to split staging vs LDS-read+MFMA time of the implicit-GEMM kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import bricks
dev = torch.device("cuda:0")
NB = int(os.environ.get("ABL_BATCH", "256"))
SHAPES = [(64, 128, 128, 3, 1, 40), (64, 64, 64, 3, 1, 80), (64, 256, 128, 1, 1, 40), (64, 128, 256, 3, 2, 80), (64, 256, 256, 3, 1, 20),
          (64, 512, 256, 1, 1, 20), (64, 64, 128, 3, 2, 160), (64, 32, 32, 3, 1, 160), (64, 64, 32, 1, 1, 160), (64, 32, 64, 3, 2, 320)]
if os.environ.get("ABL_ONLY"):
    SHAPES = [SHAPES[int(i)] for i in os.environ["ABL_ONLY"].split(",")]
tag = "abl=%s v1=%s" % (os.environ.get("CVHIP_IGEMM_ABLATE", "0"), os.environ.get("CVHIP_IGEMM_V1", "0"))
for (n, c, k, r, s, h) in SHAPES:
    n = NB
    conv = bricks.HipConv2d(c, k, r, s, r // 2, bias=False).to(dev)
    x = torch.randn(n, c, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(5):
            y = conv(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = conv(x)
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    p = (h + 2 * (r // 2) - r) // s + 1
    fl = 2.0 * n * p * p * k * r * r * c
    print("%s  %3d->%3d k%d s%d @%3d  %8.1f us  %7.1f TF" % (tag, c, k, r, s, h, us, fl / us / 1e6))
