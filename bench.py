"""bench.py — images/sec of one full YOLOv5-s train step (BASELINE.json metric) on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Step (mirrors trainer_det_yolov5.py:145-207 between timer.tic :379 and timer.toc :382): forward ->
loss -> backward (+ bucketed RCCL gradient all-reduce) -> SGD(nesterov) -> zero_grad -> EMA, on a
synthetic COCO-shape batch already resident in HBM (640x640, bf16 activations,
per-GPU batch 64, weak scaling). One JSON line on rank 0.

Extra objects:
  roofline     : the dominant kernel of the step (by summed launch time; HIP events on the launch stream around EVERY conv,
                 fused-backward and BN/activation launch) against the MI355X roofline that binds it (MI355X_MICROARCH.md:
                 2.5 PFLOP/s dense bf16 MFMA, 8 TB/s HBM3E), `conv_roofline` the same for the dominant MFMA conv kernel.
                 Algorithmic flops/bytes per launch: DESIGN.md §4. `peaks_measured`: what a plain device copy and a bare
                 MFMA loop reach on THIS box (the attainable ceilings beside the nominal ones).
  with_h2d     : the same K steps fed through the on-device input pipeline (cvpytorch_amd.data.GraphFeed: pinned uint8 NHWC
                 host batch -> copy stream -> fused normalise kernel -> the graph's static input): SURVEY.md §8(d)'s step,
                 H2D inside the timed region and overlapped with the previous step. `value` stays the HBM-resident rate.
  config3_deeplabv3plus_r50 : the metric's second workload (DeepLabv3+ R50 1024x512 bs16), K timed steps as well.
  cpu_baseline : the oracle (pure-PyTorch CPU restatement of the reference, fp32) timed on this box's
                 host cores on a bounded sample, 32 threads and 1 thread, rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_MFMA_TFLOPS = 2500.0  # dense bf16, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch (conf/coco_yolov5_s.yml:17)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8)
    ap.add_argument("--no-deeplab", action="store_true", help="skip the DeepLabv3+ (config 3) workload")
    ap.add_argument("--no-h2d", action="store_true", help="skip the host-fed leg (H2D inside the timed region)")
    ap.add_argument("--no-graph", action="store_true", help="run the timed steps eagerly instead of replaying a hipGraph")
    ap.add_argument("--stock-optimizer", action="store_true", help="torch.optim.SGD + ModelEMA instead of the fused arena step")
    ap.add_argument("--sync-bn", action="store_true", help="N > 1 only: HipSyncBN (trainer.py:126-127 converts BN to SyncBN under DDP); its statistics exchange is captured with the native RCCL transport")
    ap.add_argument("--torch-loss", action="store_true", help="fixed-shape torch-op YOLOv5 loss instead of the fused libcvhip loss kernels")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _cpu_sample(size, cpu_batch, budget_s, threads):
    from oracle import torch_ref as R
    torch.manual_seed(1029)
    torch.set_num_threads(threads)
    model = R.YOLOv5(80, "s").train()
    opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)
    imgs, targets = R.synthetic_batch(cpu_batch, size, seed=1029)

    def step(b):
        loss = model(imgs[:b], targets[:b], "train")["loss"]
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)

    t0 = time.perf_counter()
    step(1)  # warm-up on one image (allocator, thread pool, lazy inits)
    warm = time.perf_counter() - t0
    # size the timed sample from the warm-up: whole batches if they fit the budget, otherwise a smaller batch
    b = cpu_batch
    while b > 1 and warm * b > budget_s:
        b //= 2
    n, t0 = 0, time.perf_counter()
    while True:
        step(b)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 40:
            break
    return b * n / el, b, n, el


def cpu_baseline(size, cpu_batch, budget_s=12.0, max_threads=32):
    """Oracle train step on the host cores: bounded samples (SURVEY.md §8d 'CPU baseline') with `max_threads` intra-op threads
    and with ONE thread. Threads are capped at 32: with all 256 logical cores of the GPU box torch's intra-op pool
    oversubscribes and one batch-8 step took 192 s (0.04 img/s); the samples are sized so that warm-up + timed steps stay
    within ~30 s together."""
    threads = max(1, min(max_threads, os.cpu_count() or 1))
    v, b, n, el = _cpu_sample(size, cpu_batch, budget_s, threads)
    v1, b1, n1, el1 = _cpu_sample(size, 1, min(8.0, budget_s), 1)
    return {"value": round(v, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "value_1_thread": round(v1, 3), "cpu_model": _cpu_model(), "logical_cores": os.cpu_count() or 1,
            "torch": torch.__version__,
            "sample": "oracle (oracle/torch_ref.py) YOLOv5-s fp32 train step (fwd+loss+bwd+SGD-nesterov) @%dx%d: batch %d, %d timed step(s) after a "
                      "1-image warm-up, %.1f s, %d intra-op threads of %d logical cores; 1-thread figure: batch %d, %d step(s), %.1f s"
                      % (size, size, b, n, el, threads, os.cpu_count() or 1, b1, n1, el1)}


def deeplab_workload(dev, a, batch=16, size=(512, 1024), steps=20, warmup=3):
    """Second headline workload of BASELINE.json's metric (config 3): DeepLabv3+ ResNet-50-v1c, 1024x512, bf16, batch 16,
    OS-32 as the reference builds it. Same step definition; reported next to the YOLOv5-s line."""
    from cvpytorch_amd import deeplab
    from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
    from cvpytorch_amd.data import synthetic_segmentation_batch
    torch.manual_seed(1029)
    model = deeplab.EncoderDecoder(19, output_stride=32).to(dev).train()
    state = FlatTrainState(model, lr=0.01, momentum=0.9, nesterov=True, weight_decay=5e-4, backbone_lr=0.001, use_ema=False)
    step = FlatTrainStep(model, state)
    imgs, tgt = synthetic_segmentation_batch(batch, size, device=dev)
    for _ in range(warmup):
        step(imgs, tgt)
    graph = not a.no_graph
    if graph:
        step.capture(imgs, tgt)
        imgs, tgt = step.static_imgs, step.static_targets
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        losses = step(imgs, tgt)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ips = batch * steps / el
    # 434.7 GFLOP and ~2558 MB per image (train, OS-32 as written)
    return {"value": round(ips, 2), "unit": "images/sec", "ms_per_step": round(1e3 * el / steps, 2), "steps": steps, "warmup": warmup,
            "workload": "DeepLabv3+ R50-v1c %dx%d bf16 batch %d OS-32 (as written), SGD-nesterov, synthetic" % (size[1], size[0], batch),
            "launch": "hipGraph replay" if graph else "eager", "final_loss": round(float(losses["loss"]), 4),
            "step_roofline": {"mfma_frac": round(ips * 434.7e9 / (PEAK_MFMA_TFLOPS * 1e12), 4),
                              "hbm_frac": round(ips * 2558e6 / (PEAK_HBM_GBS * 1e9), 4)}}



def pmc_traffic(kernel_label):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/r02_pmc_traffic_raw.json, falling back to
    r01's; collected by `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and, separately, `--pmc WRITE_SIZE --kernel-trace` over
    tools/pmc_workload.py = the same eager train step). Corrections per MI355X_MICROARCH.md (HBM section): the counters are KiB;
    on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B => x2 (confirmed in the same pass on a kernel with a known byte count:
    a 419.4 MB bf16 copy reads FETCH_SIZE 204,8xx KiB; its 419.4 MB of writes read WRITE_SIZE 409,600 KiB => WRITE_SIZE x1).
    null if no file / no matching kernel."""
    raw = None
    for name in ("r02_pmc_traffic_raw.json", "r01_pmc_traffic_raw.json"):
        try:
            raw = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", name)))
            break
        except Exception:
            continue
    if raw is None:
        return None
    # bench label -> substrings the demangled kernel name must contain
    if kernel_label.startswith("bn_act_bwd_sums"):
        need = ["colreduce_kernel<1,"]
    elif kernel_label.startswith("bn_act_fwd"):
        need = ["ew_kernel<0,"]
    elif kernel_label.startswith("bn_act_bwd_apply"):
        need = ["ew_kernel<1,"]
    elif kernel_label.startswith("bwd1x1"):
        need = ["bwd1x1_kernel"]
    elif kernel_label.startswith("stem_wgrad"):
        need = ["stem_wgrad_kernel"]
    elif kernel_label.startswith("stem"):
        need = ["stem_fprop_kernel"]
    elif kernel_label.startswith("conv1x1_stream"):  # label carries the output-tile width, the template its fragment count
        need = ["conv1x1_stream_kernel<%d, " % (int(kernel_label[kernel_label.index("<") + 1:kernel_label.index(">")]) // 16)]
    else:
        tile = kernel_label[kernel_label.index("<") + 1:kernel_label.index(">")].replace(",", ", ")
        need = [("igemm_dma_kernel<" if kernel_label.startswith("igemm") else "wgrad_kernel<") + tile]
    n = tot = 0.0
    for k, v in raw.items():  # a configuration may exist in several template variants (ring depth, group count): launch-weighted mean
        if all(x in k for x in need) and v.get("fetch_size_raw_kb_per_launch") is not None and v.get("write_size_raw_kb_per_launch") is not None:
            n += v["launches"]
            tot += v["launches"] * (2.0 * v["fetch_size_raw_kb_per_launch"] + v["write_size_raw_kb_per_launch"]) * 1024.0
    return round(tot / n) if n else None


def measured_peaks(dev):
    """What THIS box attains: a plain device copy (cvhip_copy2d over 512 MiB: bytes read + written per second) and a bare MFMA loop
    (cvhip_probe_mfma_peak: 8 independent v_mfma_f32_32x32x16_bf16 per round, operands in registers). The nominal peaks
    (8 TB/s, 2.5 PFLOP/s dense bf16) stay the denominators of `frac`; these are printed beside them."""
    from cvpytorch_amd import lib as L
    out = {}
    M, C = 1 << 18, 1024                       # 256 Ki rows x 1024 bf16 = 512 MiB
    a = torch.empty((M, C), dtype=torch.bfloat16, device=dev).normal_()
    b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        L.call("cvhip_copy2d", a.data_ptr(), C, b.data_ptr(), C, M, C, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        L.call("cvhip_copy2d", a.data_ptr(), C, b.data_ptr(), C, M, C, st)
    e1.record()
    torch.cuda.synchronize()
    out["hbm_copy_gbs"] = round(reps * 2.0 * M * C * 2 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
    del a, b
    blocks, iters = 256 * 4, 4096
    scratch = torch.zeros(blocks, dtype=torch.float32, device=dev)
    L.call("cvhip_probe_mfma_peak", 64, blocks, scratch.data_ptr(), st)
    e0.record()
    L.call("cvhip_probe_mfma_peak", iters, blocks, scratch.data_ptr(), st)
    e1.record()
    torch.cuda.synchronize()
    out["mfma_bf16_tflops"] = round(blocks * 4.0 * iters * 8 * 2 * 32 * 32 * 16 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    out["how"] = "device copy of 512 MiB bf16 (read+write bytes / time); 1024 blocks x 4 waves x 4096 rounds of 8 independent v_mfma_f32_32x32x16_bf16"
    return out


def _roof(name, d, timing_source):
    sec = d["ms"] * 1e-3
    tflops = d["flops"] / sec / 1e12
    gbs = d["bytes"] / sec / 1e9
    ai = d["flops"] / d["bytes"] if d["bytes"] else 0.0
    balance = PEAK_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
    if ai < balance:
        roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
    else:
        roof = {"bound": "mfma", "achieved": round(tflops, 1), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(tflops / PEAK_MFMA_TFLOPS, 4)}
    roof.update({"traffic": pmc_traffic(name), "timing_source": timing_source, "kernel": name, "launches": d["launches"],
                 "avg_launch_us": round(1e3 * d["ms"] / d["launches"], 2), "arith_intensity_flop_per_byte": round(ai, 1),
                 "mfma_tflops": round(tflops, 1), "mfma_frac": round(tflops / PEAK_MFMA_TFLOPS, 4), "hbm_gbs": round(gbs, 1),
                 "hbm_frac": round(gbs / PEAK_HBM_GBS, 4)})
    return roof


def h2d_leg(model, state, a, dev, imgs_f32, gts, steps):
    """SURVEY.md §8(d)'s step: the batch starts on the HOST (pinned uint8 NHWC, what the CPU augmentation pipeline hands over)
    and crosses PCIe inside the timed region, overlapped with the previous step (data.GraphFeed). Returns the extra JSON object."""
    from cvpytorch_amd.arena import FlatTrainStep
    from cvpytorch_amd.data import GraphFeed
    from cvpytorch_amd import ops
    B = imgs_f32.shape[0]
    x0 = ops.images_to_nhwc(imgs_f32, cpad=8)                      # (B, 8, H, W) bf16 NHWC view: the format the feed produces
    step = FlatTrainStep(model, state)
    step.capture(x0, gts)
    feed = GraphFeed(step.static_imgs, step.static_targets)
    g = torch.Generator().manual_seed(1029)
    host = [torch.randint(0, 256, (B, a.size, a.size, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(2)]
    tgt_host = gts.detach().cpu().pin_memory()
    feed.stage(host[0], tgt_host)
    for i in range(3):                                             # warm the pipeline
        feed.commit()
        feed.stage(host[(i + 1) & 1], tgt_host)
        step(step.static_imgs, step.static_targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        feed.commit()
        feed.stage(host[i & 1], tgt_host)
        losses = step(step.static_imgs, step.static_targets)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return {"value": round(B * steps / el, 2), "unit": "images/sec", "ms_per_step": round(1e3 * el / steps, 3), "steps": steps,
            "host_batch": "pinned uint8 NHWC %dx%dx%dx3 (%.1f MB) + pinned fp32 target tensor per step" % (B, a.size, a.size, B * a.size * a.size * 3 / 1e6),
            "pipeline": "copy stream H2D into one of two device staging buffers, overlapped with the previous step's replay; main stream: "
                        "cvhip_u8_nhwc_to_bf16_norm into the graph's static input, then the hipGraph replay",
            "final_loss": round(float(losses["loss"]), 4)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP engine has no CPU fallback)")
    # transport: the RCCL communicator behind the C ABI (cvpytorch_amd/comm.py -> csrc/comm.hip); no torch process group is
    # created. CVHIP_DIST_BACKEND=gloo is a control-flow smoke test only (N ranks sharing one GPU over torch.distributed/gloo).
    backend = os.environ.get("CVHIP_DIST_BACKEND", "rccl")
    dev_index = local_rank if backend == "rccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    from cvpytorch_amd import comm as CM
    comm = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("LOCAL_WORLD_SIZE", str(world)) == str(world):
            # one node: RCCL's bootstrap sockets on loopback (the container's hostname / outward interface may not be usable)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        if backend == "rccl":
            try:
                comm = CM.init_from_env(dev)
            except Exception as e:  # never lose the scaling line to the transport: torch's process group carries the same bucket protocol
                sys.stderr.write("[bench] native RCCL communicator failed on rank %d (%r): falling back to torch.distributed\n" % (rank, e))
                dist.init_process_group("nccl")
                comm = CM.TorchDistComm()
        else:
            dist.init_process_group(backend)
            comm = CM.TorchDistComm()
        CM.set_default(comm)

    from cvpytorch_amd import ops, yolov5
    from cvpytorch_amd.data import synthetic_detection_batch
    from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
    from cvpytorch_amd.train import GradBucketer, ModelEMA, TrainStep, build_optimizer

    torch.manual_seed(1029)
    max_boxes = 20
    model = yolov5.YOLOv5(80, "s", max_targets=a.batch * max_boxes, fused_loss=not a.torch_loss).to(dev).train()
    if a.sync_bn and world > 1:
        from cvpytorch_amd.bricks import convert_sync_batchnorm
        model = convert_sync_batchnorm(model)
    if world > 1:  # same initial weights everywhere (DDP broadcasts rank 0's at construction)
        for t in list(model.parameters()) + list(model.buffers()):
            if t.is_floating_point():
                comm.broadcast_(t.data, 0)
    if a.stock_optimizer:  # reference-shaped tail: .grad tensors -> torch.optim.SGD -> ModelEMA (+ GradBucketer)
        opt = build_optimizer(model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4)
        ema = ModelEMA(model) if rank == 0 else None  # trainer.py:293: EMA on the main process only
        bucketer = GradBucketer(model, comm=comm) if world > 1 else None
        step = TrainStep(model, opt, ema, bucketer, sync_buffers=world > 1)
    else:  # flat arenas: direct gradient writes, in-place bucketed all-reduce, ONE fused SGD+EMA kernel
        state = FlatTrainState(model, lr=0.01, momentum=0.937, nesterov=True, weight_decay=5e-4, use_ema=(rank == 0), comm=comm)
        step = FlatTrainStep(model, state, sync_buffers=world > 1)
    imgs, targets = synthetic_detection_batch(a.batch, a.size, seed=1029 + rank, max_boxes=max_boxes, device=dev)
    gts = yolov5.targets_to_tensor(targets, a.batch * max_boxes, dev)

    def barrier():
        if world > 1:
            comm.barrier()
        torch.cuda.synchronize()

    # at world > 1 the bucketed RCCL all-reduces (and SyncBN's exchanges) are captured INSIDE the graph, on a forked stream that
    # runs beside the rest of backward
    use_graph = (not a.no_graph) and (not a.stock_optimizer) and not (a.sync_bn and world > 1 and not comm.capturable)
    for _ in range(a.warmup):
        step(imgs, gts)
    if use_graph:  # the W warm-up steps above ran eagerly; the K timed steps replay ONE hipGraph of the whole step
        ok = 1
        try:
            step.capture(imgs, gts)
        except Exception as e:  # never lose the bench line to a capture problem: fall back to eager steps
            ok = 0
            sys.stderr.write("[bench] hipGraph capture failed on rank %d (%r): running eagerly\n" % (rank, e))
        if world > 1:  # every rank must run the same mode (the modes issue different collectives)
            flag = torch.tensor([ok], device=dev, dtype=torch.int32)
            comm.allreduce_(flag, "min")
            comm.wait()
            ok = int(flag.item())
        if ok:
            imgs, gts = step.static_imgs, step.static_targets
        else:
            step.graph = None
            use_graph = False
    ops.TIMER.enabled = (not a.no_kernel_timing) and rank == 0 and not use_graph
    ops.TIMER.reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step(imgs, gts)
    barrier()
    el = time.perf_counter() - t0
    ops.TIMER.enabled = False
    timing_source = "HIP events around every conv launch inside the timed region (eager)"
    if use_graph and not a.no_kernel_timing:  # every rank takes part (the steps contain collectives); only rank 0 records events
        # per-kernel events cannot ride inside a graph replay: time the same K steps once more, eagerly, right after
        step.graph = None
        ops.TIMER.enabled = rank == 0
        ops.TIMER.reset()
        for _ in range(a.steps):
            step(imgs, gts)
        torch.cuda.synchronize()
        ops.TIMER.enabled = False
        timing_source = "HIP events around every conv launch in an eager re-run of the same K steps right after the timed region (the timed region replays a hipGraph)"
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        comm.allreduce_(t, "max")
        comm.wait()
        el = float(t.item())
    loss_val = float(losses["loss"])

    if rank == 0:
        gb = a.batch * world
        out = {
            "metric": "images/sec/node train step, YOLOv5-s@640 (value) & DeepLabv3+R50@1024x512 (config3_deeplabv3plus_r50.value)", "value": round(gb * a.steps / el, 2), "unit": "images/sec",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * el / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "coco_yolov5_s.yml YOLOv5-s %dx%d bf16 train step (fwd+loss+bwd+SGD-nesterov+EMA), per-GPU batch %d, "
                                   "synthetic COCO-shape tensors resident in HBM" % (a.size, a.size, a.batch),
                       "global_batch": gb, "parallelism": "dp%d" % world, "final_loss": round(loss_val, 4),
                       "launch": ("hipGraph replay of the whole step" if world == 1 else ("hipGraph replay of the whole step incl. bucketed RCCL all-reduces (cvhip_allreduce_bucket) on a forked stream beside backward" if not step.eager_tail else "hipGraph replay of forward+loss+backward, then one all-reduce of the gradient arena + fused optimizer")) if use_graph else "eager"},
        }
        summ = ops.TIMER.summary()
        if summ:
            # the step's dominant kernel over EVERYTHING that was timed (convs, the fused 1x1 backward, BN/activation passes) ...
            name, d = max(summ.items(), key=lambda kv: kv[1]["ms"])
            out["roofline"] = _roof(name, d, timing_source)
            # ... and the dominant MFMA convolution kernel (the north_star's "fraction of conv-MFMA roofline")
            convs = {k: v for k, v in summ.items() if v["flops"] > 0 and not k.startswith("bwd1x1")}
            if convs:
                cname, cd = max(convs.items(), key=lambda kv: kv[1]["ms"])
                out["conv_roofline"] = _roof(cname, cd, timing_source)
            out["kernels"] = {k: {"launches": v["launches"], "ms_per_step": round(v["ms"] / a.steps, 3),
                                  "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1), "alg_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}
                              for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])}
            # whole-step roofline fractions from BASELINE.md §2 (49.30 GFLOP and 366 MB per image, train)
            ips_gpu = a.batch * a.steps / el
            out["step_roofline"] = {"mfma_frac": round(ips_gpu * 49.30e9 / (PEAK_MFMA_TFLOPS * 1e12), 4),
                                    "hbm_frac": round(ips_gpu * 366e6 / (PEAK_HBM_GBS * 1e9), 4)}
        if world == 1 and not a.no_kernel_timing:
            try:
                out["peaks_measured"] = measured_peaks(dev)
            except Exception as e:
                out["peaks_measured"] = {"error": repr(e)[:200]}
        if world == 1 and not a.no_h2d and not a.stock_optimizer and not a.no_graph:
            try:
                out["with_h2d"] = h2d_leg(model, state, a, dev, imgs, gts, a.steps)
            except Exception as e:  # the headline line must still be printed
                out["with_h2d"] = {"error": repr(e)[:300]}
        if world == 1 and not a.no_deeplab:
            try:
                out["config3_deeplabv3plus_r50"] = deeplab_workload(dev, a, steps=max(a.steps, 20), warmup=max(a.warmup, 3))
            except Exception as e:  # the headline line must still be printed
                out["config3_deeplabv3plus_r50"] = {"error": repr(e)[:300]}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.size, a.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        comm.barrier()
        comm.close()
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
