"""world_size-2 gloo test (CPU) of the DDP replacement: GradBucketer's bucketed, hook-driven,
asynchronous all-reduce must leave every rank with the MEAN gradient (DistributedDataParallel
semantics, trainer.py:312-313) and broadcast_buffers must copy rank 0's BN statistics."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cvpytorch_amd.train import GradBucketer, broadcast_buffers
    torch.manual_seed(rank)   # DIFFERENT initial weights per rank: the bucketer must broadcast rank 0's, as DDP does at construction
    model = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 16, 3, padding=1), nn.BatchNorm2d(16),
                          nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(16, 4))
    model[0].weight.data = model[0].weight.data.contiguous(memory_format=torch.channels_last)   # a non-contiguous parameter
    frozen = model[3].bias
    frozen.requires_grad_(False)
    bucketer = GradBucketer(model, bucket_bytes=2048)  # tiny buckets => several collectives in flight
    assert len(bucketer.buckets) >= 3
    res = []
    same = True
    for t in list(model.parameters()) + list(model.buffers()):
        if t.is_floating_point():
            ref = t.detach().clone().contiguous()
            dist.broadcast(ref, 0)
            same = same and torch.equal(ref, t.detach().contiguous())
    res.append(same)
    for it in range(2):
        g = torch.Generator().manual_seed(100 * it + rank)
        x = torch.randn(4, 3, 8, 8, generator=g)
        model(x).square().mean().backward()
        local = [p.grad.clone() for p in model.parameters() if p.requires_grad]
        bucketer.finish()
        avg = [p.grad.clone() for p in model.parameters() if p.requires_grad]
        # reference: explicit all_reduce of the local grads
        exp = []
        for t in local:
            t = t.clone()
            dist.all_reduce(t)
            exp.append(t / world)
        res.append(all(torch.allclose(a, b, rtol=1e-6, atol=1e-7) for a, b in zip(avg, exp)))
        model.zero_grad(set_to_none=True)
    with torch.no_grad():
        model[1].running_mean.fill_(float(rank + 1))
    broadcast_buffers(model)
    res.append(bool((model[1].running_mean == 1.0).all()))
    q.put((rank, res))
    dist.destroy_process_group()


def test_grad_bucketer_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in out:
        assert all(res), (rank, res)


def test_grad_bucketer_single_process_is_noop():
    from cvpytorch_amd.train import GradBucketer
    m = nn.Linear(4, 4)
    b = GradBucketer(m)
    m(torch.ones(2, 4)).sum().backward()
    g = m.weight.grad.clone()
    b.finish()
    assert torch.equal(m.weight.grad, g)


def _worker_uneven(rank, world, port, q):
    """Rank 1 never touches the second branch: rank 0 has a gradient for it, rank 1 has none. Every rank must still issue the
    same collectives (no hang) and END with the same averaged gradient — including rank 1, whose .grad was None."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cvpytorch_amd.train import GradBucketer
    torch.manual_seed(0)
    a, b = nn.Linear(6, 6), nn.Linear(6, 6)
    model = nn.ModuleList([a, b])
    bucketer = GradBucketer(model, bucket_bytes=64)   # one parameter per bucket
    x = torch.ones(2, 6) * (rank + 1)
    out = a(x).sum()
    if rank == 0:
        out = out + b(x).sum()   # data-dependent branch
    out.backward()
    local_b = b.weight.grad.clone() if b.weight.grad is not None else torch.zeros_like(b.weight)
    bucketer.finish()
    exp = local_b.clone()
    dist.all_reduce(exp)
    ok = b.weight.grad is not None and torch.allclose(b.weight.grad, exp / world)
    g = b.weight.grad.clone()
    dist.broadcast(g, 0)
    ok = ok and torch.equal(g, b.weight.grad)   # replicas agree
    q.put((rank, [bool(ok)]))
    dist.destroy_process_group()


def test_grad_bucketer_ranks_with_different_gradient_sets():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_uneven, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in out:
        assert all(res), (rank, res)
