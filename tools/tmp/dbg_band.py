import ctypes as C, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_kernels as K
from cvpytorch_amd import lib as L, ops
import torch.nn.functional as F
mode = sys.argv[1]
case = tuple(int(v) for v in sys.argv[2].split(","))
N, Cc, H, W, Kk, R, S, s, p, d = case
dev = K.dev()
x, w = K._mk(case, 1)
st, Kp = K._prep(case, w, True)
xd = K.to_nhwc_dev(x)
P, Q = ops.conv_out_hw(H, W, R, S, (s, s), (p, p), (d, d))
y = ops.empty_nhwc(N, Kk, P, Q, dev)
desc = ops.conv_desc(N, Cc, H, W, Kk, R, S, (s, s), (p, p), (d, d), 1, Cc, Kk)
if mode == "acc":
    acc = torch.zeros(L.BN_ACC_SHARDS, 2, Kk, dtype=torch.float64, device=dev)
    L.call("cvhip_conv2d_fprop_acc", C.byref(desc), xd.data_ptr(), st.w_fprop.data_ptr(), y.data_ptr(), acc.data_ptr(), ops._stream())
elif mode == "fprop":
    L.call("cvhip_conv2d_fprop", C.byref(desc), xd.data_ptr(), st.w_fprop.data_ptr(), None, y.data_ptr(), None, ops._stream())
elif mode == "dgrad":
    dy = torch.randn(N, P, Q, Kk, device=dev).to(K.BF)
    dx = ops.empty_nhwc(N, Cc, H, W, dev)
    L.call("cvhip_conv2d_dgrad", C.byref(desc), dy.data_ptr(), st.w_dgrad.data_ptr(), dx.data_ptr(), ops._stream())
torch.cuda.synchronize()
print("ok", mode, case, os.environ.get("CVHIP_BAND"))
if mode == "acc":
    s = acc.sum(0).cpu()
    yy = y.float().cpu().double()
    r1 = yy.sum((0, 2, 3)); r2 = (yy * yy).sum((0, 2, 3))
    print("s1 finite", bool(torch.isfinite(s[0]).all()), "s2 finite", bool(torch.isfinite(s[1]).all()))
    print("s1[:8]", s[0][:8].tolist()); print("r1[:8]", r1[:8].tolist())
    bad = (~torch.isfinite(s[0])).nonzero().flatten().tolist(); print("bad ch", bad[:40])
    print("shards nonfinite:", [(int(i), int((~torch.isfinite(acc[i])).sum())) for i in range(acc.shape[0])])
    print("max rel err s1", float(((s[0]-r1).abs()/(r1.abs()+1)).max()), "s2", float(((s[1]-r2).abs()/(r2.abs()+1)).max()))
