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


def _worker_schedule(rank, world, port, q):
    """arena.BucketSchedule (the bookkeeping FlatTrainState inherits) driven by two ranks whose gradients complete in DIFFERENT orders
    and over DIFFERENT parameter sets (a data-dependent branch: rank 1 never touches parameters 2 and 5, and uses parameter 7 twice).
    `_launch` records the side-stream protocol FlatTrainState._launch performs (fork: side stream waits for the compute stream; then
    the collective on the side stream) and really runs the collective over gloo — a rank that issued its collectives in another
    order would deadlock or average the wrong buckets."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from cvpytorch_amd.arena import BucketSchedule, build_buckets
    sizes = [40, 8, 300, 16, 16, 64, 8, 120, 4, 30]          # parameter sizes (floats), arena order
    seg, o = [], 0
    for n in sizes:
        seg.append((o, o + n))
        o += n
    buckets = build_buckets(seg, cap=128)

    class Sched(BucketSchedule):
        def __init__(self):
            self.multi, self.defer_allreduce = True, False
            self.grad = torch.zeros(o)
            self.events = []
            self._init_schedule(buckets)

        def _launch(self, bi):
            lo, hi = self.buckets[bi][0], self.buckets[bi][1]
            self.events.append(("fork", bi))                 # side stream waits for the compute stream
            dist.all_reduce(self.grad[lo:hi])                # the bucket's collective, on the side stream
            self.events.append(("allreduce", bi, lo, hi))

        def finish(self):
            self._launch_rest()
            self.events.append(("join",))                    # compute stream waits for the side stream: once, after the last bucket

    s = Sched()
    res = [len(buckets) >= 4, sorted(set(s.bucket_of.values())) == list(range(len(buckets)))]
    for step in range(3):
        s._reset_buckets()
        s.events.clear()
        s.grad.zero_()
        if rank == 0:
            order = [9, 8, 7, 6, 5, 4, 3, 2, 1, 0] if step != 1 else [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]   # reverse / forward completion
            uses = {}
        else:
            order = [3, 9, 0, 7, 8, 1, 7, 6, 4]            # no 2, no 5; parameter 7 reports twice (a layer applied twice)
            uses = {7: 2}
        for i, n in uses.items():
            for _ in range(n):
                s.note_use(i)
        for i in order:
            lo, hi = seg[i]
            if not (rank == 1 and i == 7 and s._uses.get(7, 0) > 1):
                s.grad[lo:hi] += float(rank + 1)             # this rank's gradient contribution
            s.mark_ready(i)
        s.finish()
        mine = [e[:2] for e in s.events]
        seqs = [None] * world
        dist.all_gather_object(seqs, mine)
        res.append(seqs[0] == seqs[1])                       # identical fork / collective / join sequence on both ranks
        launched = [e[1] for e in s.events if e[0] == "allreduce"]
        res.append(launched == list(range(len(buckets))))    # index order, every bucket exactly once
        res.append(s.events[-1] == ("join",) and all(s.events[2 * k][0] == "fork" and s.events[2 * k + 1][0] == "allreduce" for k in range(len(buckets))))
        # the sums are what an all-reduce of the two ranks' arenas gives: rank 0 wrote 1.0 everywhere, rank 1 2.0 except on 2 and 5
        exp = torch.zeros(o)
        for i, (lo, hi) in enumerate(seg):
            exp[lo:hi] = 1.0 + (0.0 if i in (2, 5) else 2.0)
        res.append(torch.equal(s.grad, exp))
    q.put((rank, [bool(r) for r in res]))
    dist.destroy_process_group()


def test_bucket_schedule_order_is_rank_invariant():
    """the first real multi-GPU run must not be able to deadlock on collective ORDER"""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_schedule, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in out:
        assert all(res), (rank, res)


def test_build_buckets_partition():
    from cvpytorch_amd.arena import build_buckets
    seg = [(0, 10), (10, 30), (30, 35), (35, 100), (100, 101)]
    b = build_buckets(seg, 30)
    assert b == [(35, 101, 3, 4), (10, 35, 1, 2), (0, 10, 0, 0)] or b[0][3] == 4     # the oversized parameter 3 shares no more than the cap allows
    cover = sorted((lo, hi) for lo, hi, _, _ in b)
    assert cover[0][0] == 0 and cover[-1][1] == 101 and all(cover[i][1] == cover[i + 1][0] for i in range(len(cover) - 1))   # contiguous partition
    assert [x[3] for x in b] == sorted((x[3] for x in b), reverse=True)            # bucket 0 holds the LAST parameters
