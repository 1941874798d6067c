"""Dev tool: LDS read bandwidth of the igemm fragment pattern (cvhip_probe_lds_read_bw)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L
dev = torch.device("cuda:0")
out = torch.zeros(4096, device=dev)
st = torch.cuda.current_stream().cuda_stream
for blocks in (256, 512, 768):
    for mode in (0, 1, 2, 3):
        iters = 4000
        L.call("cvhip_probe_lds_read_bw", mode, 10, blocks, out.data_ptr(), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("cvhip_probe_lds_read_bw", mode, iters, blocks, out.data_ptr(), st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        nbytes = blocks * 4 * 8 * 1024.0 * iters
        per_cu = nbytes / 256 / (ms * 1e-3) / 2.4e9
        print("blocks %4d mode %d: %.3f ms  %.1f TB/s aggregate  %.1f B/clk/CU (at 2.4 GHz)" % (blocks, mode, ms, nbytes / ms / 1e9, per_cu))
