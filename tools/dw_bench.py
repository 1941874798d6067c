"""Dev tool: depthwise 3x3 fprop / dgrad / wgrad on the DeepLabv3+ decoder shapes (env CVHIP_DW3_SEG / CVHIP_DW3_RPT sweep the walk geometry)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import ops, lib as L
dev = torch.device("cuda:0")
BF = torch.bfloat16
tag = "seg=%s rpt=%s" % (os.environ.get("CVHIP_DW3_SEG", "-"), os.environ.get("CVHIP_DW3_RPT", "-"))
for (N, Cc, H, W) in [(16, 304, 128, 256), (16, 256, 128, 256)]:
    x = torch.randn(N, H, W, Cc, device=dev).to(BF)
    y = torch.empty_like(x)
    dy = torch.randn(N, H, W, Cc, device=dev).to(BF)
    w = torch.randn(Cc, 3, 3, device=dev)
    dw = torch.zeros(Cc, 3, 3, device=dev)
    desc = ops.conv_desc(N, Cc, H, W, Cc, 3, 3, (1, 1), (1, 1), (1, 1), Cc, Cc, Cc)
    s = ops._stream()
    res = []
    for name, fn in (("fprop", lambda: L.call("cvhip_dwconv2d_fprop", C.byref(desc), x.data_ptr(), w.data_ptr(), None, y.data_ptr(), s)),
                     ("dgrad", lambda: L.call("cvhip_dwconv2d_dgrad", C.byref(desc), dy.data_ptr(), w.data_ptr(), y.data_ptr(), s)),
                     ("wgrad", lambda: L.call("cvhip_dwconv2d_wgrad", C.byref(desc), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), 0, s))):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        res.append("%s %7.1f us %5.2f TB/s" % (name, us, 2 * x.numel() * 2 / us / 1e6))
    print("%s  C=%d  %s" % (tag, Cc, "   ".join(res)), flush=True)
