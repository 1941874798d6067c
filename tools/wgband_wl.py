"""Dev tool: workload of tools/wgband_sq.sh (rocprofv3 counter passes): the tap-resident 3x3 weight-gradient kernel and the general
kernel on one stride-1 3x3 shape (SHAPE=N,C,H,W,K; default YOLOv5-s 128 -> 128 @40x40 batch 64), 20 launches each on rotating operands."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L, ops
dev = torch.device("cuda:0")
N, Cc, H, W, K = [int(v) for v in os.environ.get("SHAPE", "64,128,40,40,128").split(",")]
desc = ops.conv_desc(N, Cc, H, W, K, 3, 3, (1, 1), (1, 1), (1, 1), 1, Cc, K)
nsets = 6
xs = [torch.randn(N, H, W, Cc, device=dev).to(ops.ACT_DTYPE) for _ in range(nsets)]
dys = [torch.randn(N, H, W, K, device=dev).to(ops.ACT_DTYPE) for _ in range(nsets)]
dw = torch.zeros(K, 3, 3, Cc, device=dev)
stream = ops._stream()
for mode in (os.environ.get("WL_MODES", "2,0").split(",")):
    os.environ["CVHIP_WGRAD_BAND"] = mode
    for i in range(20):
        L.call("cvhip_conv2d_wgrad", C.byref(desc), xs[i % nsets].data_ptr(), dys[i % nsets].data_ptr(), dw.data_ptr(), 1, stream)
    torch.cuda.synchronize()
print("done")
