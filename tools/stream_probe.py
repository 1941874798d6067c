"""Dev tool: achieved HBM traffic of the streaming (BN / activation / copy) kernels on a large NHWC tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


SHAPES = [(64, 160, 64), (64, 320, 32), (64, 80, 128), (64, 40, 256), (64, 20, 512), (64, 80, 64), (64, 40, 128), (64, 20, 256)]
for (N, H, C) in SHAPES:
    M = N * H * H
    y = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dz = torch.randn(M, C, device=dev).to(torch.bfloat16)
    z = torch.empty_like(y)
    sc = torch.rand(C, device=dev) + 0.5
    sh = torch.randn(C, device=dev)
    mean = torch.randn(C, device=dev) * 0.1
    inv = torch.rand(C, device=dev) + 0.5
    rows = L.load().cvhip_colreduce_rows(M, C)
    partial = torch.empty(rows + L.REDUCE_SCRATCH_ROWS, 2, C, device=dev)
    dg, db = torch.randn(C, device=dev), torch.randn(C, device=dev)
    E = M * C * 2 / 1e6  # MB
    res = {}
    res["copy2d (r+w)"] = (2 * E, timeit(lambda: L.call("cvhip_copy2d", y.data_ptr(), C, z.data_ptr(), C, M, C, st)))
    for act, nm in ((L.ACT_NONE, "none"), (L.ACT_RELU, "relu"), (L.ACT_SILU, "silu")):
        res["bn_act_fwd %s (r+w)" % nm] = (2 * E, timeit(lambda: L.call("cvhip_bn_act_fwd", y.data_ptr(), C, z.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), act, 0.0, None, 0, st)))
        res["bwd_partial %s (2r)" % nm] = (2 * E, timeit(lambda: L.call("cvhip_bn_act_bwd_partial", dz.data_ptr(), C, y.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), act, 0.0, partial.data_ptr(), st)))
        res["bwd_apply %s (2r+w)" % nm] = (3 * E, timeit(lambda: L.call("cvhip_bn_act_bwd_apply", dz.data_ptr(), C, y.data_ptr(), C, z.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), dg.data_ptr(), db.data_ptr(), act, 0.0, st)))
    res["bn_stats_partial (r)"] = (E, timeit(lambda: L.call("cvhip_bn_stats_partial", y.data_ptr(), M, C, C, partial.data_ptr(), st)))
    res["torch copy_ (r+w)"] = (2 * E, timeit(lambda: z.copy_(y)))
    print("--- M=%d C=%d  tensor %.0f MB" % (M, C, E))
    for k, (mb, us) in res.items():
        print("  %-28s %8.1f us  %6.2f TB/s" % (k, us, mb / us))
