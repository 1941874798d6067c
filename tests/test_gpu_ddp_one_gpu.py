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
