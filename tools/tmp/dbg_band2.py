import ctypes as C, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_kernels as K
from cvpytorch_amd import lib as L, ops
case = tuple(int(v) for v in sys.argv[1].split(","))
N, Cc, H, W, Kk, R, S, s, p, d = case
dev = K.dev()
x, w = K._mk(case, 1)
st, Kp = K._prep(case, w, True)
xd = K.to_nhwc_dev(x)
P, Q = ops.conv_out_hw(H, W, R, S, (s, s), (p, p), (d, d))
ny = N * P * Q * Kk
G = 1 << 22
pool = torch.zeros(G + ny + G, dtype=K.BF, device=dev)
ypt = pool.data_ptr() + 2 * G
accpool = torch.zeros(3, L.BN_ACC_SHARDS, 2, Kk, dtype=torch.float64, device=dev)
desc = ops.conv_desc(N, Cc, H, W, Kk, R, S, (s, s), (p, p), (d, d), 1, Cc, Kk)
torch.cuda.synchronize()
L.call("cvhip_conv2d_fprop_acc", C.byref(desc), xd.data_ptr(), st.w_fprop.data_ptr(), ypt, accpool[1].data_ptr(), ops._stream())
torch.cuda.synchronize()
pre = pool[:G]; post = pool[G + ny:]
print("nonzero before y:", int((pre != 0).sum()), " after y:", int((post != 0).sum()))
nz = (pre.view(torch.int16) != 0).nonzero().flatten()
if len(nz): print("  first/last idx before (from y start):", int(nz[0]) - G, int(nz[-1]) - G)
nz = (post.view(torch.int16) != 0).nonzero().flatten()
if len(nz): print("  first/last idx after (from y end):", int(nz[0]), int(nz[-1]))
print("acc guards nonzero:", int((accpool[0] != 0).sum()), int((accpool[2] != 0).sum()))
a = accpool[1]
print("shards touched:", [int(i) for i in range(16) if bool((a[i] != 0).any())])
print("nonfinite per shard:", [int((~torch.isfinite(a[i])).sum()) for i in range(16)])
bad = (~torch.isfinite(a)).nonzero()
print("bad (shard, which, ch):", bad[:24].tolist())
yy = pool[G:G + ny].view(N, P, Q, Kk).float().double()
r1 = yy.sum((0, 1, 2)).cpu(); r2 = (yy * yy).sum((0, 1, 2)).cpu()
s = a.sum(0).cpu()
e1 = (s[0] - r1).abs() / (r1.abs() + 1); e2 = (s[1] - r2).abs() / (r2.abs() + 1)
print("s1 bad channels:", [(int(i), float(s[0][i]), float(r1[i])) for i in (e1 > 1e-2).nonzero().flatten()[:12]])
print("s2 bad channels:", [(int(i), float(s[1][i]), float(r2[i])) for i in (~(e2 < 1e-2)).nonzero().flatten()[:12]])
for sh, wch, ch in bad[:6].tolist():
    print("  raw", sh, wch, ch, float(a[sh, wch, ch]), " s1 same slot:", float(a[sh, 0, ch]))
