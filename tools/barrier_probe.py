"""Dev tool: cost and correctness of a device-wide barrier inside one persistent kernel (cvhip_probe_grid_barrier)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
for blocks in (256, 512):
    for mode in (0, 1):
        res = []
        for iters in (1, 21):
            scratch = torch.zeros(2 * blocks, device=dev)
            counter = torch.zeros(1, dtype=torch.int32, device=dev)
            out = torch.zeros(blocks, device=dev)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            L.call("cvhip_probe_grid_barrier", mode, iters, blocks, scratch.data_ptr(), counter.data_ptr(), out.data_ptr(), st)
            e1.record()
            torch.cuda.synchronize()
            want = sum(blocks * (blocks + 1) / 2 + it * blocks for it in range(iters))
            ok = bool((out == want).all())
            res.append((iters, e0.elapsed_time(e1) * 1e3, ok))
        per = (res[1][1] - res[0][1]) / 20
        print("blocks %d mode %d: 1 barrier %.1f us, 21 barriers %.1f us => %.2f us per barrier, correct=%s" % (blocks, mode, res[0][1], res[1][1], per, all(r[2] for r in res)), flush=True)
