"""Dev tool: what the skip-connection addend costs in the input-gradient kernels (cvhip_conv2d_dgrad against cvhip_conv2d_dgrad_add) on
the stride-2 3x3 layers of YOLOv5-s (batch 64, 640x640) and the residual-block projections of DeepLabv3+ R50 (batch 16, 1024x512),
plus fprop of the same layer for scale. HIP events around REPS back-to-back launches on rotating operand sets (>= 300 MB in flight).
    ONLY=<substring of a label>   REPS / ROUNDS   VARIANTS="name:ENV=V,ENV=V;..." (environment switches read per launch)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

dev = torch.device("cuda:0")
SHAPES = [
    # N, C, H, W, K, R, S, stride, pad, label
    (64, 32, 320, 320, 64, 3, 3, 2, 1, "y5s 32->64 s2 @320"),
    (64, 64, 160, 160, 128, 3, 3, 2, 1, "y5s 64->128 s2 @160"),
    (64, 128, 80, 80, 256, 3, 3, 2, 1, "y5s 128->256 s2 @80"),
    (64, 256, 40, 40, 512, 3, 3, 2, 1, "y5s 256->512 s2 @40"),
    (64, 128, 80, 80, 128, 3, 3, 2, 1, "y5s pan 128->128 s2 @80"),
    (64, 256, 40, 40, 256, 3, 3, 2, 1, "y5s pan 256->256 s2 @40"),
    (16, 512, 32, 64, 512, 3, 3, 2, 1, "r50 layer4 512->512 s2 @32x64"),
    (16, 256, 64, 128, 256, 3, 3, 2, 1, "r50 layer3 256->256 s2 @64x128"),
    (16, 128, 128, 256, 128, 3, 3, 2, 1, "r50 layer2 128->128 s2 @128x256"),
    (64, 512, 20, 20, 512, 3, 3, 2, 1, "y5 512->512 s2 @20"),
    (16, 256, 64, 128, 512, 3, 3, 2, 1, "stdc 256->512 s2 @64x128"),
    (16, 32, 256, 512, 64, 3, 3, 1, 1, "dl stem 32->64 k3 @256x512"),
    (64, 32, 160, 160, 32, 3, 3, 1, 1, "y5s 32->32 k3 @160 (per-tap: CVHIP_BAND=0 CVHIP_PATCH=0)"),
    (64, 256, 40, 40, 256, 1, 1, 1, 0, "y5s 256->256 k1 @40"),
    (64, 512, 20, 20, 512, 1, 1, 1, 0, "y5s 512->512 k1 @20"),
    (16, 256, 128, 256, 512, 1, 1, 2, 0, "dl 256->512 k1 s2 @128x256"),
    (16, 1024, 32, 64, 256, 1, 1, 1, 0, "dl 1024->256 k1 @32x64"),
    (16, 2048, 16, 32, 512, 1, 1, 1, 0, "dl 2048->512 k1 @16x32"),
    (16, 512, 64, 128, 128, 1, 1, 1, 0, "dl 512->128 k1 @64x128"),
    (16, 256, 32, 64, 1024, 1, 1, 1, 0, "dl 256->1024 k1 @32x64"),
    (16, 512, 16, 32, 2048, 1, 1, 1, 0, "dl 512->2048 k1 @16x32"),
    (16, 560, 128, 256, 512, 1, 1, 1, 0, "dl 560->512 k1 @128x256"),
    (16, 512, 128, 256, 512, 1, 1, 1, 0, "dl 512->512 k1 @128x256"),
]
REPS = int(os.environ.get("REPS", "20"))
ROUNDS = int(os.environ.get("ROUNDS", "3"))
only = os.environ.get("ONLY")


def timed(fn, nsets):
    for i in range(3):
        fn(i % nsets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(REPS):
        fn(i % nsets)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / REPS


def main():
    L.load()
    variants = [("default", {})]
    for extra in os.environ.get("VARIANTS", "").split(";"):
        if extra:
            name, kv = extra.split(":")
            variants.append((name, dict(x.split("=") for x in kv.split(","))))
    for (N, Cc, H, W, K, R, S, s, p, label) in SHAPES:
        if only and only not in label:
            continue
        P, Q = ops.conv_out_hw(H, W, R, S, (s, s), (p, p), (1, 1))
        bytes_set = 2 * N * (2 * H * W * Cc + P * Q * K)
        nsets = max(2, min(8, int(3e8 // bytes_set) + 1))
        w = (torch.randn(K, Cc, R, S, device=dev) / (Cc * R * S) ** 0.5).contiguous(memory_format=torch.channels_last)
        st = ops.ConvState()
        pdesc = ops.conv_desc(N, Cc, H, W, K, R, S, (s, s), (p, p), (1, 1), 1, Cc, K)
        st.prepare(w, pdesc, True, ("bench", label))
        xs = [torch.randn(N, H, W, Cc, device=dev).to(ops.ACT_DTYPE) for _ in range(nsets)]
        ys = [torch.empty(N, P, Q, K, device=dev, dtype=ops.ACT_DTYPE) for _ in range(nsets)]
        dys = [torch.randn(N, P, Q, K, device=dev).to(ops.ACT_DTYPE) for _ in range(nsets)]
        dxs = [torch.empty(N, H, W, Cc, device=dev, dtype=ops.ACT_DTYPE) for _ in range(nsets)]
        stream = ops._stream()
        flops = 2.0 * N * P * Q * K * R * S * Cc
        mb = 2e-6 * N * (H * W * Cc + P * Q * K)

        def fprop(i):
            L.call("cvhip_conv2d_fprop", C.byref(pdesc), xs[i].data_ptr(), st.w_fprop.data_ptr(), None, ys[i].data_ptr(), None, stream)

        acc = torch.zeros(L.BN_ACC_SHARDS, 2, K, dtype=torch.float64, device=dev)

        def fprop_acc(i):
            L.call("cvhip_conv2d_fprop_acc", C.byref(pdesc), xs[i].data_ptr(), st.w_fprop.data_ptr(), ys[i].data_ptr(), acc.data_ptr(), stream)

        def dgrad(i):
            L.call("cvhip_conv2d_dgrad", C.byref(pdesc), dys[i].data_ptr(), st.w_dgrad.data_ptr(), dxs[i].data_ptr(), stream)

        def dgrad_add(i):
            L.call("cvhip_conv2d_dgrad_add", C.byref(pdesc), dys[i].data_ptr(), st.w_dgrad.data_ptr(), xs[i].data_ptr(), Cc, dxs[i].data_ptr(), stream)

        res = {}
        for rnd in range(ROUNDS):
            for name, env in variants:
                os.environ.update(env)
                res.setdefault((name, "fprop"), []).append(timed(fprop, nsets))
                if os.environ.get("ACC"):
                    res.setdefault((name, "fprop_acc"), []).append(timed(fprop_acc, nsets))
                res.setdefault((name, "dgrad"), []).append(timed(dgrad, nsets))
                res.setdefault((name, "dgrad_add"), []).append(timed(dgrad_add, nsets))
                for k in env:
                    os.environ.pop(k, None)
        print("%-32s (%d operand sets, %.0f MB of operands, addend %.0f MB)" % (label, nsets, mb, 2e-6 * N * H * W * Cc))
        for (name, what), v in res.items():
            us = sorted(v)[len(v) // 2]
            print("    %-10s %-10s %8.1f us   %7.1f TF/s   (min %.1f)" % (name, what, us, flops / us / 1e6, min(v)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
