"""The N > 1 data-parallel train step (flat arenas + gradient all-reduce), exercised on ONE GPU: two processes share
cuda:0 and exchange gradients over the gloo backend (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the
driver's). Both modes of arena.FlatTrainStep are covered: eager (bucketed all-reduce overlapped with backward) and hipGraph
replay of forward+loss+backward followed by one eager all-reduce of the gradient arena + the fused optimizer.
Checks: ranks stay bit-identical to each other (same reduced gradients => same parameters), they differ from an un-synced
run, and the reduced gradient equals the mean of the two ranks' local gradients (trainer.py:312-313 DDP semantics)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, use_graph, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cvpytorch_amd import yolov5
        from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
        from cvpytorch_amd.data import synthetic_detection_batch
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        probe = torch.ones(4, device=dev)
        try:
            dist.all_reduce(probe)
        except Exception as e:  # gloo built without device support
            q.put((rank, "skip", repr(e)[:200]))
            return
        torch.manual_seed(0)
        model = yolov5.YOLOv5(80, "n", max_targets=64, fused_loss=True).to(dev).train()
        state = FlatTrainState(model, use_ema=(rank == 0))
        step = FlatTrainStep(model, state, sync_buffers=True)
        imgs, targets = synthetic_detection_batch(4, 96, seed=11 + rank, max_boxes=8, device=dev)
        gts = yolov5.targets_to_tensor(targets, 64, dev)
        # local gradient of step 0 (before any reduction) for the mean check
        model(imgs, gts, "train")["loss"].backward()
        local = state.grad.clone()
        state.finish_allreduce()
        reduced = state.grad.clone() / world       # the optimizer folds 1/world into grad_scale
        both = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(both, local)
        mean = sum(both) / world
        err = float((reduced - mean).abs().max() / mean.abs().max().clamp(min=1e-12))
        state.zero_grad()
        state._reset_buckets()
        if use_graph:
            step.capture(imgs, gts, warmup=1)
        for _ in range(3):
            step(imgs, gts)
        torch.cuda.synchronize()
        psum = state.param.double().sum().item()
        ph = state.param.clone()
        others = [torch.zeros_like(ph) for _ in range(world)]
        dist.all_gather(others, ph)
        same = all(torch.equal(others[0], o) for o in others)
        q.put((rank, "ok", err, same, psum, state.steps))
        dist.destroy_process_group()
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, "error", traceback.format_exc()[-1500:]))


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_ranks_on_one_gpu(use_graph):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, use_graph, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = []
    for _ in procs:
        res.append(q.get(timeout=600))
    for p in procs:
        p.join(timeout=60)
    if any(r[1] == "skip" for r in res):
        pytest.skip("gloo cannot reduce device tensors in this build: %s" % [r for r in res if r[1] == "skip"][0][2])
    assert all(r[1] == "ok" for r in res), res
    for r in res:
        assert r[2] < 1e-5, r          # reduced gradient == mean of the local gradients
        assert r[3], r                  # parameters identical on both ranks after 3 steps
    assert res[0][4] == res[1][4]
    assert all(r[5] == (4 if use_graph else 3) for r in res), res


def _syncbn_worker(rank, world, port, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cvpytorch_amd import bricks, yolo_blocks
        dev = torch.device("cuda:0")
        torch.cuda.set_device(0)
        torch.manual_seed(3)
        net = torch.nn.Sequential(bricks.HipConvModule(16, 32, 3, padding=1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")),
                                  yolo_blocks.CSPLayer(32, 32, n=1, norm_cfg=dict(type="BN"), act_cfg=dict(type="SiLU")),
                                  bricks.HipBN(32))
        g = torch.Generator().manual_seed(9)
        xfull = torch.randn(8, 16, 12, 10, generator=g).to(torch.bfloat16)
        cot = torch.randn(8, 32, 12, 10, generator=g)
        import copy
        full = copy.deepcopy(net).to(dev).train()                       # plain BN on the whole batch
        sync = bricks.convert_sync_batchnorm(copy.deepcopy(net)).to(dev).train()   # SyncBN on this rank's half
        assert list(sync.state_dict().keys()) == list(full.state_dict().keys())
        assert sum(isinstance(m, bricks.HipSyncBN) for m in sync.modules()) == sum(isinstance(m, torch.nn.BatchNorm2d) for m in full.modules())

        def run(m, x, c):
            x = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            z = m(x)
            (z.float() * c.to(dev)).sum().backward()
            return z.detach().float(), x.grad.detach().float()

        zf, gxf = run(full, xfull, cot)
        lo, hi = rank * 4, rank * 4 + 4
        zs, gxs = run(sync, xfull[lo:hi], cot[lo:hi])

        def rel(a, b):
            return float((a - b).norm() / b.norm().clamp(min=1e-12))

        out = {"z": rel(zs, zf[lo:hi]), "gx": rel(gxs, gxf[lo:hi])}
        pf = dict(full.named_parameters())
        worst = 0.0
        for n, p in sync.named_parameters():
            gsum = p.grad.detach().float().clone()
            dist.all_reduce(gsum)                                     # sum-type loss: full-batch grad == sum of the ranks' grads
            worst = max(worst, rel(gsum, pf[n].grad.float()))
        out["gparam"] = worst
        bf = dict(full.named_buffers())
        out["running"] = max(rel(b.float(), bf[n].float()) for n, b in sync.named_buffers() if "running" in n)
        q.put((rank, "ok", out))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put((rank, "error", traceback.format_exc()[-1500:]))


def test_syncbn_two_ranks_equals_full_batch_bn():
    """HipSyncBN over two ranks (half a batch each) == plain BN over the whole batch: outputs, input gradients, parameter
    gradients (summed over ranks) and running statistics, to bf16 storage accuracy."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    for r in res:
        o = r[2]
        assert o["z"] < 1e-2 and o["gx"] < 2e-2 and o["gparam"] < 2e-2 and o["running"] < 1e-3, o
