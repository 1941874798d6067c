"""RCCL communicator behind the C ABI on ONE GPU (a 1-rank communicator: every collective is the identity, but every call goes
through librccl on the caller's stream): eager and under hipGraph capture; then a whole FlatTrainStep with the bucketed
all-reduce machinery live (force_collectives) — eager and captured — must equal the plain single-GPU step.
The multi-rank bookkeeping is covered on CPU (tests/test_ddp_gloo.py, tests/test_comm_host.py) and with two gloo ranks sharing
the GPU (tests/test_gpu_ddp_one_gpu.py); the 8-GPU run is the driver's."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from cvpytorch_amd import comm as CM
from cvpytorch_amd import lib as L


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def comm1():
    torch.cuda.set_device(0)
    c = CM.RcclComm(1, 0, CM.RcclComm.unique_id())
    yield c
    c.close()


def test_allreduce_bucket_in_place_eager_and_captured(comm1):
    d = dev()
    arena = torch.arange(1 << 20, dtype=torch.float32, device=d)
    ref = arena.clone()
    lo, hi = 1000, 1000 + (1 << 18)
    comm1.allreduce_(arena[lo:hi])                                  # cvhip_allreduce_bucket on the current stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm1.allreduce_(arena[hi:hi + 4096], stream=side)              # ... and on a side stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(arena, ref)
    # captured: kernel -> all-reduce on a forked stream -> join -> kernel, replayed twice
    static = torch.ones(1 << 16, dtype=torch.float32, device=d)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        static.mul_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        static.mul_(2.0)
        side.wait_stream(torch.cuda.current_stream())
        comm1.allreduce_(static, stream=side)
        torch.cuda.current_stream().wait_stream(side)
        static.add_(1.0)
    static.fill_(1.0)
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert float(static[0]) == 7.0 and float(static[-1]) == 7.0    # ((1*2)+1)*2+1


def test_other_collectives(comm1):
    d = dev()
    t = torch.randn(4096, device=d)
    r = t.clone()
    comm1.broadcast_(t, 0)
    comm1.allreduce_(t, "max")
    h = torch.randn(1024, device=d).to(torch.bfloat16)
    hr = h.clone()
    comm1.allreduce_(h)
    i = torch.arange(64, dtype=torch.int32, device=d)
    comm1.allreduce_(i, "min")
    L.call("cvhip_comm_reduce_scatter_f32", comm1._h, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
    L.call("cvhip_comm_all_gather_f32", comm1._h, t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
    comm1.barrier()
    assert torch.equal(t, r) and torch.equal(h, hr) and int(i[63]) == 63
    assert L.load().cvhip_comm_world(comm1._h) == 1 and L.load().cvhip_comm_rank(comm1._h) == 0


def _steps(comm, capture, n=3, grad_exchange="allreduce"):
    from cvpytorch_amd import yolov5
    from cvpytorch_amd.arena import FlatTrainState, FlatTrainStep
    from cvpytorch_amd.data import synthetic_detection_batch
    d = dev()
    torch.manual_seed(5)
    model = yolov5.YOLOv5(80, "n", max_targets=32, fused_loss=True).to(d).train()
    state = FlatTrainState(model, use_ema=False, comm=comm, force_collectives=comm is not None, bucket_bytes=1 << 20, grad_exchange=grad_exchange)
    step = FlatTrainStep(model, state, sync_buffers=comm is not None)
    imgs, targets = synthetic_detection_batch(4, 128, seed=7, max_boxes=8, device=d)
    gts = yolov5.targets_to_tensor(targets, 32, d)
    if comm is not None:
        assert state.multi and len(state.buckets) >= 2
    if capture:
        step.capture(imgs, gts)
        imgs, gts = step.static_imgs, step.static_targets
        if comm is not None:
            assert not step.eager_tail and not state.defer_allreduce   # the collectives are INSIDE the graph
    losses = [float(step(imgs, gts)["loss"].detach()) for _ in range(n)]
    torch.cuda.synchronize()
    return losses, state.param.clone()


@pytest.mark.parametrize("capture", [False, True])
def test_train_step_with_live_bucket_allreduce_equals_plain_step(comm1, capture):
    l0, p0 = _steps(None, capture)
    l1, p1 = _steps(comm1, capture)
    # wgrad / fused-backward kernels accumulate with fp32 atomics: runs agree to rounding, not bit for bit
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-3 * abs(a), (l0, l1)
    assert float((p0 - p1).norm() / p0.norm()) <= 1e-3


@pytest.mark.parametrize("capture", [False, True])
def test_train_step_with_reduce_scatter_all_gather_exchange_equals_plain_step(comm1, capture):
    """FlatTrainState(grad_exchange="rsag"): every bucket goes through cvhip_comm_reduce_scatter_f32 + cvhip_comm_all_gather_f32 on the
    side stream (captured as a graph branch) instead of one all-reduce; over a 1-rank communicator both are the identity"""
    l0, p0 = _steps(None, capture)
    l1, p1 = _steps(comm1, capture, grad_exchange="rsag")
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-3 * abs(a), (l0, l1)
    assert float((p0 - p1).norm() / p0.norm()) <= 1e-3


def test_reduce_scatter_allgather_entry_points(comm1):
    t = torch.randn(4096, device=dev())
    r = t.clone()
    comm1.reduce_scatter_allgather_(t)
    torch.cuda.synchronize()
    assert torch.equal(t, r)                    # one rank: identity, and the buffer is left intact
    h = torch.randn(1024, device=dev()).to(torch.bfloat16)   # not fp32: routed to the plain all-reduce
    comm1.reduce_scatter_allgather_(h)
    torch.cuda.synchronize()
