"""Dev tool: the machine ceilings the kernel designs are argued against (csrc/probes.hip), measured on the box in front of us.
Prints one table per question; the output of one run is kept as profiles/r03_ceilings_probe.log.

    python tools/ceilings_probe.py [lds] [mfma] [load] [atomic]      (default: all)

Clock: B/clk/CU figures divide by 256 CUs and the NOMINAL 2.4 GHz; the chip clocks lower under load (DVFS), so a figure of
~80 % of a documented per-clock rate can still be the hardware limit. TB/s and TFLOP/s columns are clock-free.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvpytorch_amd import lib as L  # noqa: E402

dev = torch.device("cuda:0")
out = torch.zeros(8192, device=dev)
st = torch.cuda.current_stream().cuda_stream
CUS, GHZ = 256, 2.4


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best  # ms


def lds():
    print("== LDS read rate: 16 DS reads per s_waitcnt lgkmcnt(0) (guide: 256 B/clk/CU for ds_read_b128 / b64 from 4 waves per CU) ==")
    names = {0: "ds_read_b128 lane-linear", 1: "ds_read_b128 igemm fragment pattern (64-B rows, XOR swizzle)", 2: "ds_read_b64 lane-linear"}
    for mode in (0, 1, 2):
        for threads, blocks in ((256, 256), (256, 512), (512, 256), (1024, 256), (1024, 512)):
            iters = 20000
            ms = timed(lambda: L.call("cvhip_probe_lds_read2", mode, iters, blocks, threads, out.data_ptr(), st))
            nbytes = blocks * (threads // 64) * iters * (8192.0 if mode == 2 else 16384.0)
            print("  %-62s %4d thr x %3d blocks (%2d waves/CU): %7.3f ms  %6.1f TB/s  %6.1f B/clk/CU" % (
                names[mode], threads, blocks, blocks * threads // 64 // CUS, ms, nbytes / ms / 1e9, nbytes / CUS / (ms * 1e-3) / (GHZ * 1e9)))


def mfma():
    print("== MFMA issue rate (guide: 2495 TF 32x32x16 on zero operands; >= 2382 TF ubench ceiling; DVFS lowers it on real data) ==")
    for shape, nm, per in ((0, "v_mfma_f32_32x32x16_bf16 x8 acc", 8 * 32768.0), (1, "v_mfma_f32_16x16x32_bf16 x16 acc", 16 * 16384.0)):
        for data, dn in ((0, "zeros"), (1, "small ints"), (2, "full-range")):
            for threads, blocks in ((256, 256), (256, 1024), (512, 512)):
                iters = 4000
                ms = timed(lambda: L.call("cvhip_probe_mfma_peak2", shape, data, iters, blocks, threads, out.data_ptr(), st))
                fl = blocks * (threads // 64) * iters * per
                print("  %-34s %-10s %4d thr x %4d blocks: %7.3f ms  %7.1f TFLOP/s" % (nm, dn, threads, blocks, ms, fl / ms / 1e9))


def load():
    print("== global -> on-chip load path, batches of D KiB per wave (16 B per lane, full 128-B lines) ==")
    big = torch.empty((1 << 32,), dtype=torch.uint8, device=dev)   # 4 GiB
    big.random_(0, 255)
    names = {0: "global_load_lds_dwordx4 (LDS DMA)", 1: "global_load_dwordx4 -> VGPR", 2: "global_load_dwordx4 -> VGPR -> ds_write_b128"}
    for src_name, span, stride in (("L2-resident 1 MiB window shared by all blocks", 1 << 20, 0), ("HBM stream, 8 MiB per block", 1 << 23, 1 << 23)):
        print(" source: %s" % src_name)
        for mode in (0, 1, 2):
            for depth in (4, 8):
                for threads, blocks in ((256, 256), (256, 512), (512, 256), (1024, 256)):
                    iters = 2048 // (threads // 64) * 4 // depth * 2
                    if stride and blocks * stride > big.numel():
                        continue
                    ms = timed(lambda: L.call("cvhip_probe_load_path", mode, depth, big.data_ptr(), span, stride, iters, blocks, threads, out.data_ptr(), st))
                    nbytes = blocks * (threads // 64) * iters * depth * 1024.0
                    print("  %-46s D=%d %4d thr x %3d blocks (%2d waves/CU): %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU" % (
                        names[mode], depth, threads, blocks, blocks * threads // 64 // CUS, ms, nbytes / ms / 1e9, nbytes / CUS / (ms * 1e-3) / (GHZ * 1e9)))


def gather():
    print("== the implicit GEMM's A-tile staging pattern alone (global_load_lds_dwordx4; 256 rows per block per K step) ==")
    big = torch.empty((1 << 30,), dtype=torch.uint8, device=dev)
    big.random_(0, 255)
    for span_name, span in (("32 MiB tensor (26 MB activations of a 40x40x128 layer at batch 64)", 1 << 25), ("1 GiB", 1 << 30)):
        print(" source span: %s" % span_name)
        for rowb in (64, 128):
            for stride, kb in ((128, 128), (256, 256), (512, 512), (1024, 1024), (256, 128), (272, 256)):
                if kb < rowb:
                    continue
                for threads, blocks, depth in ((256, 512, 2), (256, 512, 4), (512, 512, 2), (256, 1024, 2)):
                    iters = 288
                    ms = timed(lambda: L.call("cvhip_probe_gather", rowb, big.data_ptr(), span, stride, kb, iters, depth, blocks, threads, out.data_ptr(), st))
                    nbytes = blocks * iters * 256.0 * rowb
                    print("  rows of %3d B, pitch %4d B, sweep %4d B  %3d thr x %4d blocks, %d K steps in flight: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU" % (
                        rowb, stride, kb, threads, blocks, depth, ms, nbytes / ms / 1e9, nbytes / CUS / (ms * 1e-3) / (GHZ * 1e9)))


def stage():
    print("== the implicit GEMM's staging STRUCTURE around the same loads (64-B rows, pitch 256 B = 128 channels, 256 threads) ==")
    big = torch.empty((1 << 27,), dtype=torch.uint8, device=dev)     # 128 MiB: the activations of a 40x40x128 layer at batch 256
    big.random_(0, 255)
    w = torch.empty((128 * 256 * 9,), dtype=torch.uint8, device=dev)
    w.random_(0, 255)
    names = {0: "batch of 2 steps + drain", 1: "+ barrier", 2: "ring, counted waits", 3: "ring + barrier", 7: "ring + barrier + weight tile",
             11: "ring + barrier + tap shifts", 15: "ring + barrier + weights + tap shifts (the kernel)"}
    for span_name, span in (("32 MiB", 1 << 25), ("128 MiB", 1 << 27)):
        for blocks in (400, 1600):
            for flags in (0, 1, 2, 3, 7, 11, 15):
                iters = 36 * 8
                ms = timed(lambda: L.call("cvhip_probe_stage", flags, big.data_ptr(), span, w.data_ptr(), 256, 256, iters, blocks, out.data_ptr(), st))
                nbytes = blocks * iters * (16384.0 + (8192.0 if flags & 4 else 0.0))
                print("  span %-8s %4d blocks  %-52s %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU" % (
                    span_name, blocks, names[flags], ms, nbytes / ms / 1e9, nbytes / CUS / (ms * 1e-3) / (GHZ * 1e9)))


def atomic():
    print("== per-block atomics into sharded accumulators: `blocks` blocks x n adds (a conv epilogue folding 2 x K BatchNorm sums) ==")
    acc = torch.zeros((64 * 256,), dtype=torch.float64, device=dev)
    for f32 in (0, 1):
        for blocks, n in ((6400, 128), (1600, 256), (400, 256)):
            for shards in (1, 4, 16, 64):
                acc.zero_()
                ms = timed(lambda: L.call("cvhip_probe_atomic_add", f32, acc.data_ptr(), shards, n, blocks, st), reps=5)
                print("  %s  %5d blocks x %3d adds, %2d shards: %8.2f us  (%.1f ns per same-address add, serialised)" % (
                    "fp32" if f32 else "fp64", blocks, n, shards, ms * 1e3, ms * 1e6 / (blocks / shards)))
    # correctness of the fp64 form: 4 launches x 6400 blocks into 16 shards
    acc.zero_()
    for _ in range(4):
        L.call("cvhip_probe_atomic_add", 0, acc.data_ptr(), 16, 128, 6400, st)
    torch.cuda.synchronize()
    got = acc[:16 * 128].view(16, 128).sum(0).cpu()
    want = (torch.arange(128, dtype=torch.float64) + 1.0) * 4 * 6400
    print("  fp64 sums exact:", bool(torch.equal(got, want)))


if __name__ == "__main__":
    which = sys.argv[1:] or ["lds", "mfma", "load", "gather", "stage", "atomic"]
    print("device:", torch.cuda.get_device_name(0))
    for w in which:
        {"lds": lds, "mfma": mfma, "load": load, "gather": gather, "stage": stage, "atomic": atomic}[w]()
