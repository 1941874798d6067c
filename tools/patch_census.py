"""Dev tool: where do the blocks of the patch-resident convolution kernel run, and when? Every block records {HW_ID, XCC_ID, start, end}
(s_memtime); this prints blocks per CU over time (are two 72-KB blocks ever co-resident on one CU?) for a few tile counts."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import torch
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

dev = torch.device("cuda:0")
raw = C.CDLL(L.LIB_PATH)
raw.cvhip_patch_debug_buffer.argtypes = [C.c_void_p]
raw.cvhip_patch_debug_buffer.restype = None
name = "cvhip_patch_debug_buffer"
os.environ["CVHIP_PATCH"] = "2"
for N in (40, 64, 80):
    Cc, H, W, K = 128, 40, 40, 128
    desc = ops.conv_desc(N, Cc, H, W, K, 3, 3, (1, 1), (1, 1), (1, 1), 1, Cc, K)
    w = (torch.randn(K, Cc, 3, 3, device=dev) / 34).contiguous(memory_format=torch.channels_last)
    st = ops.ConvState()
    st.prepare(w, desc, False, ("census", N))
    x = torch.randn(N, H, W, Cc, device=dev).to(ops.ACT_DTYPE)
    y = torch.empty(N, H, W, K, device=dev, dtype=ops.ACT_DTYPE)
    buf = (C.c_int32 * (4 * L.PATCH_CLASS_INTS))()
    assert L.load().cvhip_conv2d_patch_plan(C.byref(desc), 0, buf, 4) == 1
    tiles = buf[24]
    dbg = torch.zeros((tiles, 4), dtype=torch.int64, device=dev)
    for it in range(3):
        raw.cvhip_patch_debug_buffer(dbg.data_ptr() if it == 2 else None)
        L.call("cvhip_conv2d_fprop", C.byref(desc), x.data_ptr(), st.w_fprop.data_ptr(), None, y.data_ptr(), None, ops._stream())
        torch.cuda.synchronize()
    raw.cvhip_patch_debug_buffer(None)
    d = dbg.cpu().numpy()
    t0 = d[:, 2].min()
    span = d[:, 3].max() - t0
    cu = collections.defaultdict(list)
    for hw, xcc, a, b in d:
        key = (int(xcc) & 0xf, (int(hw) >> 13) & 7, (int(hw) >> 12) & 1, (int(hw) >> 8) & 0xf)   # xcc, se, sh, cu
        cu[key].append((int(a - t0), int(b - t0)))
    per = collections.Counter(len(v) for v in cu.values())
    overl = 0
    for v in cu.values():
        v.sort()
        for i in range(1, len(v)):
            if v[i][0] < v[i - 1][1] - 1000:
                overl += 1
    dur = (d[:, 3] - d[:, 2])
    print("N=%d tiles=%d distinct CUs=%d blocks-per-CU histogram=%s  overlapping pairs=%d  kernel span=%d ticks, block duration mean %d min %d max %d" % (
        N, tiles, len(cu), dict(per), overl, span, dur.mean(), dur.min(), dur.max()))
    late = sorted(int(a - t0) for a in d[:, 2])
    print("   start times (ticks) percentiles: 50%%=%d 75%%=%d 90%%=%d 100%%=%d" % (late[len(late) // 2], late[len(late) * 3 // 4], late[len(late) * 9 // 10], late[-1]))
