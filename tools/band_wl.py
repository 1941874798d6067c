"""Dev tool: workload of tools/band_sq.sh (rocprofv3 counter passes): the row-band kernel's forms, the patch-resident and the per-tap
kernel on one stride-1 3x3 shape (SHAPE=N,C,H,W,K; default YOLOv5-s 128 -> 128 @40x40 batch 64), 30 fprop launches each on rotating operands."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L, ops
dev = torch.device("cuda:0")
N, Cc, H, W, K = [int(v) for v in os.environ.get("SHAPE", "64,128,40,40,128").split(",")]
w = (torch.randn(K, Cc, 3, 3, device=dev) / (Cc * 9) ** 0.5).contiguous(memory_format=torch.channels_last)
st = ops.ConvState()
pdesc = ops.conv_desc(N, Cc, H, W, K, 3, 3, (1, 1), (1, 1), (1, 1), 1, Cc, K)
st.prepare(w, pdesc, True, ("wl",))
nsets = 6
xs = [torch.randn(N, H, W, Cc, device=dev).to(ops.ACT_DTYPE) for _ in range(nsets)]
ys = [torch.empty(N, H, W, K, device=dev, dtype=ops.ACT_DTYPE) for _ in range(nsets)]
stream = ops._stream()
for name, env in (("band, default plan (two 4-wave blocks per CU)", {"CVHIP_BAND": "2"}),
                  ("band, one 8-wave block per CU", {"CVHIP_BAND": "2", "CVHIP_BAND_NW": "8"}),
                  ("patch", {"CVHIP_BAND": "0", "CVHIP_PATCH": "1"}), ("tap", {"CVHIP_BAND": "0", "CVHIP_PATCH": "0"})):
    for k in ("CVHIP_BAND", "CVHIP_BAND_NW", "CVHIP_PATCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for i in range(30):
        L.call("cvhip_conv2d_fprop", C.byref(pdesc), xs[i % nsets].data_ptr(), st.w_fprop.data_ptr(), None, ys[i % nsets].data_ptr(), None, stream)
    torch.cuda.synchronize()
print("done")
