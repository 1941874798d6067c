"""Dev tool: time cvhip_conv2d_wgrad on the YOLOv5-s / DeepLabv3+ layer shapes (batch 64 / 16), e.g. under CVHIP_WGRAD_ABLATE=1/4/5
(no atomic epilogue / staging only / no global loads) to split a launch into its parts."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpytorch_amd import lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
# (N, C, H, W, K, R, stride)
SHAPES = [(64, 128, 40, 40, 128, 3, 1), (64, 64, 80, 80, 64, 3, 1), (64, 256, 20, 20, 256, 3, 1), (64, 32, 160, 160, 32, 3, 1),
          (64, 64, 160, 160, 128, 3, 2), (64, 128, 80, 80, 256, 3, 2), (64, 256, 40, 40, 512, 3, 2),
          (64, 256, 40, 40, 256, 1, 1), (64, 512, 20, 20, 512, 1, 1), (64, 128, 40, 40, 128, 1, 1), (64, 512, 40, 40, 256, 1, 1),
          (16, 64, 128, 256, 64, 3, 1), (16, 128, 64, 128, 128, 3, 1), (16, 256, 32, 64, 256, 3, 1), (16, 512, 16, 32, 512, 3, 1),
          (16, 256, 128, 256, 64, 1, 1), (16, 1024, 32, 64, 256, 1, 1), (16, 512, 16, 32, 2048, 1, 1)]
if os.environ.get("WG_ONLY") == "k1":
    SHAPES = [sh for sh in SHAPES if sh[5] == 1]
elif os.environ.get("WG_ONLY") == "k3":
    SHAPES = [sh for sh in SHAPES if sh[5] == 3]
tag = " ".join("%s=%s" % (k[6:], v) for k, v in sorted(os.environ.items()) if k.startswith("CVHIP_WGRAD"))
st = torch.cuda.current_stream().cuda_stream
tot = 0.0
for (N, Cc, H, W, K, R, s) in SHAPES:
    p = R // 2
    P, Q = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    x = torch.randn(N, H, W, Cc, device=dev).to(torch.bfloat16)
    dy = torch.randn(N, P, Q, K, device=dev).to(torch.bfloat16)
    dw = torch.zeros(K, R, R, Cc, device=dev)
    desc = ops.conv_desc(N, Cc, H, W, K, R, R, (s, s), (p, p), (1, 1), 1, Cc, K)
    fn = lambda: L.call("cvhip_conv2d_wgrad", C.byref(desc), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, st)  # noqa: E731
    if os.environ.get("WG_DET") == "1":   # the slab + ordered-fold form (no floating-point atomics): cvhip_conv2d_wgrad_det
        nb = int(L.load().cvhip_conv2d_wgrad_det_workspace_bytes(C.byref(desc)))
        ws = torch.empty((max(nb, 16),), dtype=torch.uint8, device=dev)
        fn = lambda: L.call("cvhip_conv2d_wgrad_det", C.byref(desc), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 1, ws.data_ptr(), nb, st)  # noqa: E731
        tag = "DET"
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us
    fl = 2.0 * N * P * Q * K * R * R * Cc
    by = (x.numel() + dy.numel()) * 2.0
    print("%-22s %3dx %4d->%4d k%d s%d @%3dx%3d  %8.1f us  %6.1f TF  %5.2f TB/s(operands once)" % (tag, N, Cc, K, R, s, H, W, us, fl / us / 1e6, by / us / 1e6))
print("%-22s total %.1f us" % (tag, tot))
