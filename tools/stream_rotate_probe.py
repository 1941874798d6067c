"""Dev tool: the BN / activation passes on ROTATING buffer sets. tools/stream_probe.py re-reads the same tensors every call, so on
tensors smaller than the 256 MB Infinity Cache (MALL) its rates are cache rates; here every call works on the next of K tensor sets
(K x set size >> 256 MB), which is what a train step does (each activation is read once or twice, a step moves > 20 GB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timeit(fns, reps=3):
    for f in fns:
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns:
            f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3  # us


for (N, H, C) in [(64, 80, 128), (64, 40, 256), (64, 20, 512), (64, 40, 128)]:
    M = N * H * H
    E = M * C * 2 / 1e6
    for K in (1, max(2, int(1200 / (3 * E)))):
        sets = []
        for k in range(K):
            y = torch.randn(M, C, device=dev).to(torch.bfloat16)
            dz = torch.randn(M, C, device=dev).to(torch.bfloat16)
            z = torch.empty_like(y)
            sets.append((y, dz, z))
        sc = torch.rand(C, device=dev) + 0.5
        sh = torch.randn(C, device=dev)
        mean = torch.randn(C, device=dev) * 0.1
        inv = torch.rand(C, device=dev) + 0.5
        rows = L.load().cvhip_colreduce_rows(M, C)
        partial = torch.empty(rows + L.REDUCE_SCRATCH_ROWS, 2, C, device=dev)
        dg, db = torch.randn(C, device=dev), torch.randn(C, device=dev)
        act = L.ACT_SILU
        res = {}
        res["copy2d (r+w)"] = (2 * E, timeit([(lambda y=y, z=z: L.call("cvhip_copy2d", y.data_ptr(), C, z.data_ptr(), C, M, C, st)) for (y, dz, z) in sets]))
        res["bn_act_fwd silu (r+w)"] = (2 * E, timeit([(lambda y=y, z=z: L.call("cvhip_bn_act_fwd", y.data_ptr(), C, z.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), act, 0.0, None, 0, st)) for (y, dz, z) in sets]))
        res["bwd_partial silu (2r)"] = (2 * E, timeit([(lambda y=y, dz=dz: L.call("cvhip_bn_act_bwd_partial", dz.data_ptr(), C, y.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), act, 0.0, partial.data_ptr(), st)) for (y, dz, z) in sets]))
        res["bwd_apply silu (2r+w)"] = (3 * E, timeit([(lambda y=y, dz=dz, z=z: L.call("cvhip_bn_act_bwd_apply", dz.data_ptr(), C, y.data_ptr(), C, z.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), mean.data_ptr(), inv.data_ptr(), dg.data_ptr(), db.data_ptr(), act, 0.0, st)) for (y, dz, z) in sets]))
        # producer -> consumer adjacency of a train step: the pass reads what the previous launch just wrote
        res["copy -> bn_act_fwd (pair)"] = (4 * E, 2 * timeit([f for (y, dz, z) in sets for f in (
            (lambda y=y, dz=dz: L.call("cvhip_copy2d", dz.data_ptr(), C, y.data_ptr(), C, M, C, st)),
            (lambda y=y, z=z: L.call("cvhip_bn_act_fwd", y.data_ptr(), C, z.data_ptr(), C, M, C, sc.data_ptr(), sh.data_ptr(), act, 0.0, None, 0, st)))]))
        print("--- M=%d C=%d  tensor %.0f MB, %d rotating sets (%.0f MB)" % (M, C, E, K, 3 * E * K))
        for k, (mb, us) in res.items():
            print("  %-28s %8.1f us  %6.2f TB/s" % (k, us, mb / us))
        del sets
        torch.cuda.empty_cache()
