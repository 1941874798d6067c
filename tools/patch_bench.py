"""Dev tool: the patch-resident convolution kernel (csrc/conv_patch.hip) against the per-tap implicit GEMM on the multi-tap layers of
YOLOv5-s (batch 64, 640x640) and DeepLabv3+ R50 (batch 16, 1024x512): fprop, dgrad, and fprop with the BN+activation prologue.
HIP events around REPS back-to-back launches on rotating operand sets (>= 256 MB in flight so the Infinity Cache does not flatter).
    CVHIP_PATCH is read per launch, so both kernels are timed in one process (interleaved rounds)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cvpytorch_amd import lib as L
from cvpytorch_amd import ops

dev = torch.device("cuda:0")
SHAPES = [
    # N, C, H, W, K, R, S, stride, pad, dil, label
    (64, 128, 40, 40, 128, 3, 3, 1, 1, 1, "y5s 128->128 @40"),
    (64, 64, 80, 80, 64, 3, 3, 1, 1, 1, "y5s 64->64 @80"),
    (64, 32, 160, 160, 32, 3, 3, 1, 1, 1, "y5s 32->32 @160"),
    (64, 256, 20, 20, 256, 3, 3, 1, 1, 1, "y5s 256->256 @20"),
    (64, 128, 80, 80, 256, 3, 3, 2, 1, 1, "y5s 128->256 s2 (dgrad only)"),
    (64, 64, 160, 160, 128, 3, 3, 2, 1, 1, "y5s 64->128 s2 (dgrad only)"),
    (16, 64, 128, 256, 64, 3, 3, 1, 1, 1, "dl 64->64 @128x256"),
    (16, 128, 64, 128, 128, 3, 3, 1, 1, 1, "dl 128->128 @64x128"),
    (16, 256, 32, 64, 256, 3, 3, 1, 1, 1, "dl 256->256 @32x64"),
    (16, 512, 32, 64, 512, 3, 3, 1, 2, 2, "dl 512->512 dil2 @32x64"),
]
if os.environ.get("V7"):   # YOLOv7-l (batch 16, 1280x1280): the wide stride-1 3x3 layers of the first stages
    SHAPES = [(16, 64, 320, 320, 64, 3, 3, 1, 1, 1, "v7 64->64 @320"), (16, 64, 640, 640, 64, 3, 3, 1, 1, 1, "v7 64->64 @640"),
              (16, 128, 160, 160, 128, 3, 3, 1, 1, 1, "v7 128->128 @160"), (16, 256, 80, 80, 256, 3, 3, 1, 1, 1, "v7 256->256 @80")]
if os.environ.get("STDC"):   # STDC1-Seg (batch 16, 1024x512): the 3x3 stride-1 layers of the STDC blocks, the neck and the heads
    SHAPES = [
        (16, 128, 64, 128, 64, 3, 3, 1, 1, 1, "stdc 128->64 @64x128"), (16, 64, 64, 128, 32, 3, 3, 1, 1, 1, "stdc 64->32 @64x128"),
        (16, 32, 64, 128, 32, 3, 3, 1, 1, 1, "stdc 32->32 @64x128"), (16, 256, 32, 64, 128, 3, 3, 1, 1, 1, "stdc 256->128 @32x64"),
        (16, 128, 32, 64, 64, 3, 3, 1, 1, 1, "stdc 128->64 @32x64"), (16, 64, 32, 64, 64, 3, 3, 1, 1, 1, "stdc 64->64 @32x64"),
        (16, 512, 16, 32, 256, 3, 3, 1, 1, 1, "stdc 512->256 @16x32"), (16, 256, 16, 32, 128, 3, 3, 1, 1, 1, "stdc 256->128 @16x32"),
        (16, 128, 16, 32, 128, 3, 3, 1, 1, 1, "stdc 128->128 @16x32"), (16, 256, 64, 128, 256, 3, 3, 1, 1, 1, "stdc head 256->256 @64x128"),
        (16, 256, 64, 128, 64, 3, 3, 1, 1, 1, "stdc detail head 256->64 @64x128"), (16, 1024, 16, 32, 128, 3, 3, 1, 1, 1, "stdc arm 1024->128 @16x32"),
        (16, 512, 32, 64, 128, 3, 3, 1, 1, 1, "stdc arm 512->128 @32x64"),
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
    lib = L.load()
    variants = [("per-tap", {"CVHIP_PATCH": "0", "CVHIP_BAND": "0"}), ("patch", {"CVHIP_PATCH": "1", "CVHIP_BAND": "0"}),
                ("band", {"CVHIP_PATCH": "1", "CVHIP_BAND": "2"}), ("default", {"CVHIP_PATCH": "1", "CVHIP_BAND": "1"})]   # (conv_band.hip: stride-1 3x3 only; elsewhere it falls through to "patch")
    for extra in os.environ.get("VARIANTS", "").split(";"):
        if extra:
            name, kv = extra.split(":")
            variants.append((name, dict(x.split("=") for x in kv.split(","))))
    for (N, Cc, H, W, K, R, S, s, p, d, label) in SHAPES:
        if only and only not in label:
            continue
        P, Q = ops.conv_out_hw(H, W, R, S, (s, s), (p, p), (d, d))
        bytes_set = 2 * N * (H * W * Cc + P * Q * K)
        nsets = max(2, min(8, int(3e8 // bytes_set) + 1))
        w = (torch.randn(K, Cc, R, S, device=dev) / (Cc * R * S) ** 0.5).contiguous(memory_format=torch.channels_last)
        st = ops.ConvState()
        pdesc = ops.conv_desc(N, Cc, H, W, K, R, S, (s, s), (p, p), (d, d), 1, Cc, K)
        st.prepare(w, pdesc, True, ("bench", label))
        xs = [torch.randn(N, H, W, Cc, device=dev).to(ops.ACT_DTYPE) for _ in range(nsets)]
        ys = [torch.empty(N, P, Q, K, device=dev, dtype=ops.ACT_DTYPE) for _ in range(nsets)]
        dys = [torch.randn(N, P, Q, K, device=dev).to(ops.ACT_DTYPE) for _ in range(nsets)]
        dxs = [torch.empty(N, H, W, Cc, device=dev, dtype=ops.ACT_DTYPE) for _ in range(nsets)]
        zs = [torch.empty(N, H, W, Cc, device=dev, dtype=ops.ACT_DTYPE) for _ in range(nsets)]
        sc = torch.rand(Cc, device=dev) + 0.5
        sh = torch.randn(Cc, device=dev)
        stream = ops._stream()
        flops = 2.0 * N * P * Q * K * R * S * Cc
        fuse = L.ConvFuse()
        fuse.pro_scale, fuse.pro_shift, fuse.pro_act = sc.data_ptr(), sh.data_ptr(), L.ACT_SILU
        fz = L.ConvFuse()
        fz.pro_scale, fz.pro_shift, fz.pro_act, fz.z_ld = sc.data_ptr(), sh.data_ptr(), L.ACT_SILU, Cc

        def fprop(i):
            L.call("cvhip_conv2d_fprop", C.byref(pdesc), xs[i].data_ptr(), st.w_fprop.data_ptr(), None, ys[i].data_ptr(), None, stream)

        def dgrad(i):
            L.call("cvhip_conv2d_dgrad", C.byref(pdesc), dys[i].data_ptr(), st.w_dgrad.data_ptr(), dxs[i].data_ptr(), stream)

        def fprop_pro(i):
            L.call("cvhip_conv2d_fprop_fused", C.byref(pdesc), xs[i].data_ptr(), st.w_fprop.data_ptr(), ys[i].data_ptr(), C.byref(fuse), stream)

        def fprop_pro_z(i):
            fz.z_out = zs[i].data_ptr()
            L.call("cvhip_conv2d_fprop_fused", C.byref(pdesc), xs[i].data_ptr(), st.w_fprop.data_ptr(), ys[i].data_ptr(), C.byref(fz), stream)

        def bn_apply(i):
            L.call("cvhip_bn_act_fwd", xs[i].data_ptr(), Cc, zs[i].data_ptr(), Cc, N * H * W, Cc, sc.data_ptr(), sh.data_ptr(), L.ACT_SILU, 0.0, None, 0, stream)

        res = {}
        for rnd in range(ROUNDS):
            for name, env in variants:
                os.environ.update(env)
                if s == 1:
                    res.setdefault((name, "fprop"), []).append(timed(fprop, nsets))
                res.setdefault((name, "dgrad"), []).append(timed(dgrad, nsets))
                if name != "per-tap" and s == 1 and not os.environ.get("NO_PRO") and lib.cvhip_conv2d_fprop_prologue_ok(C.byref(pdesc), 1):
                    res.setdefault((name, "fprop+pro"), []).append(timed(fprop_pro, nsets))
                    res.setdefault((name, "fprop+pro+z"), []).append(timed(fprop_pro_z, nsets))
                for k in env:
                    os.environ.pop(k, None)
            if s == 1:
                res.setdefault(("-", "bn_act pass"), []).append(timed(bn_apply, nsets))
        buf = (C.c_int32 * (4 * L.PATCH_CLASS_INTS))()
        ncl = lib.cvhip_conv2d_patch_plan(C.byref(pdesc), 0, buf, 4)
        geo = "TH x TW = %d x %d, patch %d x %d, tiles %d, BN %d, CK %d" % (buf[12], buf[13], buf[14], buf[15], buf[24], buf[25], buf[26]) if ncl > 0 else "fprop: per-tap"
        print("%-32s %s  (%d operand sets)" % (label, geo, nsets))
        for name, env in variants:   # which form of the band kernel each variant's fprop runs (conv_band.hip; 0 = another kernel)
            os.environ.update(env)
            bb = (C.c_int32 * L.BAND_PLAN_INTS)()
            if hasattr(lib, "cvhip_conv2d_band_plan") and lib.cvhip_conv2d_band_plan(C.byref(pdesc), 0, bb) > 0:
                print("    %-10s band plan: %d ch/wave, %d x %d waves x %d fragments, PPS %d, read-ahead %d, TH %d, %d blocks, LDS %d KB" %
                      (name, 16 * bb[0], bb[1], 8 // bb[1], bb[2], bb[3], bb[4], bb[5], bb[8], bb[9] // 1024))
            for k in env:
                os.environ.pop(k, None)
        for (name, what), v in res.items():
            us = sorted(v)[len(v) // 2]
            print("    %-10s %-12s %8.1f us   %7.1f TF/s   (min %.1f)" % (name, what, us, flops / us / 1e6 if "bn_act" not in what else 0.0, min(v)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
