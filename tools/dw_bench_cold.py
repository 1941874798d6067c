"""Dev tool: depthwise 3x3 fprop / dgrad / wgrad on the DeepLabv3+ decoder shapes with ROTATING operand sets (every launch meets cold
caches, as inside the train step; tools/dw_bench.py re-launches on one hot set)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import ops, lib as L
dev = torch.device("cuda:0")
BF = torch.bfloat16
NSET = int(os.environ.get("NSET", "4"))
SHAPES = [(16, 560, 128, 256), (16, 512, 128, 256), (16, 304, 128, 256), (16, 256, 128, 256)]
if os.environ.get("MORE"):
    SHAPES += [(16, 128, 64, 128), (16, 512, 32, 64), (64, 256, 40, 40)]
for (N, Cc, H, W) in SHAPES:
    xs = [torch.randn(N, H, W, Cc, device=dev).to(BF) for _ in range(NSET)]
    ys = [torch.empty_like(xs[0]) for _ in range(NSET)]
    dys = [torch.randn(N, H, W, Cc, device=dev).to(BF) for _ in range(NSET)]
    w = torch.randn(Cc, 3, 3, device=dev)
    dw = torch.zeros(Cc, 3, 3, device=dev)
    desc = ops.conv_desc(N, Cc, H, W, Cc, 3, 3, (1, 1), (1, 1), (1, 1), Cc, Cc, Cc)
    s = ops._stream()
    res = []
    for name, fn in (("fprop", lambda i: L.call("cvhip_dwconv2d_fprop", C.byref(desc), xs[i].data_ptr(), w.data_ptr(), None, ys[i].data_ptr(), s)),
                     ("dgrad", lambda i: L.call("cvhip_dwconv2d_dgrad", C.byref(desc), dys[i].data_ptr(), w.data_ptr(), ys[i].data_ptr(), s)),
                     ("wgrad", lambda i: L.call("cvhip_dwconv2d_wgrad", C.byref(desc), xs[i].data_ptr(), dys[i].data_ptr(), dw.data_ptr(), 0, s))):
        for i in range(NSET):
            fn(i)
        reps = 2 * NSET
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(reps):
            fn(k % NSET)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        res.append("%s %7.1f us %5.2f TB/s" % (name, us, 2 * xs[0].numel() * 2 / us / 1e6))
    print("cold(%d sets)  N=%d C=%d %dx%d  %s" % (NSET, N, Cc, H, W, "   ".join(res)), flush=True)
    del xs, ys, dys
