"""Dev tool: back-to-back timing of 1x1 conv passes through the C ABI (fprop plain / +BN sums / +bias, dgrad), rotating over
enough buffers that nothing stays cache-resident. Run under CVHIP_S1X1=0 and =1 to A/B the streaming kernel."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import ops, lib as L
dev = torch.device("cuda:0")
BF = torch.bfloat16
SHAPES = [(64, 32, 160), (64, 64, 160), (256, 64, 80), (128, 128, 80), (128, 64, 80), (64, 64, 80), (256, 128, 40), (128, 128, 40),
          (128, 256, 80), (256, 256, 40), (512, 128, 40)]
NB = 64
tag = "s1x1=%s" % os.environ.get("CVHIP_S1X1", "1")
lib = L.load()
for (c, k, h) in SHAPES:
    M = NB * h * h
    nbuf = max(2, int(600e6 // (M * (c + k) * 2)) + 1)
    xs = [torch.randn(M, c, device=dev).to(BF) for _ in range(nbuf)]
    ys = [torch.empty(M, k, device=dev, dtype=BF) for _ in range(nbuf)]
    w = torch.randn(k, 1, 1, c, device=dev)
    st = ops.ConvState()
    desc = ops.conv_desc(NB, c, h, h, k, 1, 1, (1, 1), (0, 0), (1, 1), 1, c, k)
    st.prepare(w.permute(0, 3, 1, 2), desc, True, ("b", c, k, h))
    rows = lib.cvhip_conv2d_fprop_stats_rows(C.byref(desc))
    part = torch.empty(rows + 8, 2, k, device=dev)
    bias = torch.randn(k, device=dev)
    s = ops._stream()
    res = []
    for mode in ("plain", "stats", "bias", "dgrad"):
        def run(i):
            x, y = xs[i % nbuf], ys[i % nbuf]
            if mode == "dgrad":
                L.call("cvhip_conv2d_dgrad", C.byref(desc), y.data_ptr(), st.w_dgrad.data_ptr(), x.data_ptr(), s)
            else:
                L.call("cvhip_conv2d_fprop", C.byref(desc), x.data_ptr(), st.w_fprop.data_ptr(), bias.data_ptr() if mode == "bias" else None,
                       y.data_ptr(), part.data_ptr() if mode == "stats" else None, s)
        for i in range(3):
            run(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            run(i)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        res.append("%s %6.1f us %5.0f GB/s" % (mode, us, M * (c + k) * 2 / us / 1e3))
    print("%s  %4d->%4d @%3d  %s" % (tag, c, k, h, "   ".join(res)), flush=True)
