"""Dev tool: SPPF's 5x5 stride-1 max pools (forward + arg-max, backward accumulate form) on rotating concat buffers, as ops.SppfChain
issues them: slice j -> slice j + 1 of a [N, H, W, 4c] buffer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import ops, lib as L
dev = torch.device("cuda:0")
st = ops._stream()
for (N, c, H, W) in [(64, 256, 20, 20), (64, 128, 20, 20), (16, 512, 40, 40), (64, 256, 32, 32)]:
    NS = 6
    bufs = [torch.randn(N, H, W, 4 * c, device=dev).to(torch.bfloat16) for _ in range(NS)]
    idx = [torch.empty(N, H, W, c, dtype=torch.uint8, device=dev) for _ in range(NS)]
    ld = 4 * c

    def fwd(i):
        b = bufs[i]
        L.call("cvhip_maxpool2d_fwd", b.data_ptr(), ld, b.data_ptr() + c * 2, ld, idx[i].data_ptr(), N, c, H, W, 5, 1, 2, st)

    def bwd(i):
        b = bufs[i]
        L.call("cvhip_maxpool2d_bwd", b.data_ptr() + c * 2, ld, idx[i].data_ptr(), b.data_ptr(), ld, N, c, H, W, 5, 1, 2, 1, st)

    out = []
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for i in range(NS):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(4 * NS):
            fn(k % NS)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / (4 * NS) * 1e3
        out.append("%s %6.1f us" % (name, us))
    print("N=%d c=%d %dx%d (%.1f MB per slice)  %s" % (N, c, H, W, N * H * W * c * 2 / 1e6, "   ".join(out)), flush=True)
