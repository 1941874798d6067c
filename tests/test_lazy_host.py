"""Host-side logic of the lazy-activation path (round 5), no GPU needed: the admission policy is pure arithmetic inside libcvhip
(cvhip_conv1x1_stream_prologue_ok / cvhip_conv1x1_bwd_fused_ok) + ops.lazy_edge_ok; argument validation of the prologue range happens
before any launch. Reference call site: ConvModule.forward, src/models/bricks/conv_module.py:201-214 (training mode)."""
import ctypes as C

import pytest
import torch  # noqa: F401  (one HIP runtime per process: torch first)

from cvpytorch_amd import lib as L
from cvpytorch_amd import ops


def desc(N, Cc, H, W, K):
    return L.ConvDesc(N, Cc, H, W, K, 1, 1, 1, 1, 0, 0, 1, 1, 1, Cc, K, 0, 0)


def test_struct_layouts_match_the_header():
    # cvhip_conv_fuse grew at its END only (older members keep their offsets): pro_lo / pro_hi, then the split-store triple;
    # cvhip_lazy_in is 2 pointers + 4 x 32 bit
    assert C.sizeof(L.ConvFuse) == 144 and L.ConvFuse.pro_lo.offset == 116 and L.ConvFuse.pro_hi.offset == 120
    assert L.ConvFuse.y2.offset == 128 and L.ConvFuse.y2_ld.offset == 136 and L.ConvFuse.y_split.offset == 140
    assert L.ConvFuse.x_image_planes.offset == 112
    assert C.sizeof(L.LazyIn) == 32 and L.LazyIn.c_hi.offset == 28


def test_policy_on_the_yolov5s_edges_at_batch_64():
    """which producer -> consumer edges of YOLOv5-s (batch 64, 640 x 640) stay lazy: 64- and 128-channel inputs of 1x1 layers whose forward
    is the streaming kernel and whose backward is the fused kernel (>= 2400 64-row trips); never 32-channel SiLU inputs (measured loss),
    never K > 128 (implicit GEMM / three-pass backward), never the small maps"""
    ok = lambda *a: ops.lazy_edge_ok(*a, L.ACT_SILU)  # noqa: E731
    assert ok(64, 64, 160, 160, 64)          # stride-2 conv -> CSP1 sibling pair
    assert ok(64, 128, 80, 80, 128)          # stride-2 conv -> CSP2 sibling pair
    assert ok(64, 64, 80, 80, 64)            # CSP2 / neck first sibling -> bottleneck conv1
    assert not ok(64, 32, 160, 160, 32)      # CSP1 first sibling -> bottleneck conv1: 32-channel SiLU edge, excluded by measurement
    assert ops.lazy_edge_ok(64, 32, 160, 160, 32, L.ACT_RELU)   # ... the same edge with ReLU (1 VALU instruction) is admitted
    assert not ok(64, 256, 40, 40, 256)      # CSP3 pair: K > 128
    assert not ok(64, 128, 40, 40, 128)      # 1600 trips: the fused backward's policy says three-pass
    assert not ok(4, 64, 32, 32, 64)         # smoke-sized maps: neither kernel's policy takes them
    assert not ok(64, 60, 160, 160, 64)      # channel count not a multiple of 8
    assert not ops.lazy_edge_ok(64, 64, 160, 160, 64, L.ACT_HSWISH)   # activation without an on-load instance


def test_policy_respects_the_global_switches(monkeypatch):
    assert ops.lazy_edge_ok(64, 64, 160, 160, 64, L.ACT_SILU)
    monkeypatch.setattr(ops, "_LAZY", False)
    assert not ops.lazy_edge_ok(64, 64, 160, 160, 64, L.ACT_SILU)
    monkeypatch.setattr(ops, "_LAZY", True)
    monkeypatch.setattr(ops, "_DETERMINISTIC", True)
    assert not ops.lazy_edge_ok(64, 64, 160, 160, 64, L.ACT_SILU)


def test_prologue_argument_validation_happens_before_any_launch():
    lib = L.load()
    buf = (C.c_char * 8192)()
    a = (C.addressof(buf) + 63) // 64 * 64
    d = desc(4, 64, 80, 80, 64)
    assert lib.cvhip_conv1x1_stream_prologue_ok(C.byref(d), 1) == 1 and lib.cvhip_conv1x1_stream_prologue_ok(C.byref(d), 0) == 1
    for lo, hi in ((4, 60), (0, 72), (32, 32), (-8, 8)):
        f = L.ConvFuse()
        f.pro_scale, f.pro_shift, f.pro_act, f.pro_lo, f.pro_hi = a, a, L.ACT_SILU, lo, hi
        assert lib.cvhip_conv2d_fprop_fused(C.byref(d), a, a, a, C.byref(f), None) == L.ERR_INVALID, (lo, hi)
    f = L.ConvFuse()
    f.pro_lo, f.pro_hi = 0, 32                                   # a range without a prologue
    assert lib.cvhip_conv2d_fprop_fused(C.byref(d), a, a, a, C.byref(f), None) == L.ERR_INVALID
    for dd in (L.ConvDesc(2, 64, 40, 40, 64, 1, 1, 2, 2, 0, 0, 1, 1, 1, 64, 64, 0, 0), desc(1, 64, 16, 16, 64), desc(4, 64, 80, 80, 256)):
        assert lib.cvhip_conv1x1_stream_prologue_ok(C.byref(dd), 1) == 0   # strided; too few tiles; 256-wide with BN sums: no on-load form
    li = L.LazyIn(a, a, L.ACT_SIGMOID, 0.0, 0, 0)                  # an activation the fused backward has no on-load instance for
    big = desc(64, 64, 80, 80, 64)
    st = lib.cvhip_conv1x1_bwd_fused_lazy(C.byref(big), a, 64, None, 0, 64, a, a, a, a, a, a, a, a, 64, None, None, 0, L.ACT_SILU, 0.0, None, 0, a, 64,
                                          a, C.byref(li), None)
    assert st == L.ERR_UNSUPPORTED
    k256 = desc(64, 64, 80, 80, 256)
    li = L.LazyIn(a, a, L.ACT_SILU, 0.0, 0, 0)
    st = lib.cvhip_conv1x1_bwd_fused_lazy(C.byref(k256), a, 256, None, 0, 256, a, a, a, a, a, a, a, a, 256, None, None, 0, L.ACT_SILU, 0.0, None, 0, a, 64,
                                          a, C.byref(li), None)
    assert st == L.ERR_UNSUPPORTED                                # K = 256 instances sit at their register budget: no lazy-input form


def test_lazy_tensor_reaching_an_unaware_op_raises():
    """a lazy tensor holds RAW convolution outputs: an op without an on-load transform must never read it silently"""
    t = torch.zeros(1, 8, 2, 2).to(memory_format=torch.channels_last)
    t._hip_lazy = ops.LazyAct(torch.ones(8), torch.zeros(8), L.ACT_SILU, 0.0)
    with pytest.raises(L.CvhipError):
        ops.as_nhwc(t)          # (raises for the CPU tensor first or for the tag: either way nothing reads it)
    assert ops.lazy_of(t) is not None and ops.lazy_of(torch.zeros(1)) is None


def test_lazy_raw_tensor_is_safe_outside_the_engine():
    """the storage of a lazy activation holds RAW convolution outputs. Its type (ops.LazyRaw) hands every stock torch
    function the ACTIVATED tensor (ops.materialize; here the cache is pre-filled, so no kernel runs on this CPU box): views, slices,
    casts, detach, arithmetic, a stock nn.Conv2d, torch.cat — none of them can read the un-normalised data; metadata queries and the
    engine's own tag lookups still see the raw tensor."""
    import torch.nn as nn
    raw = torch.arange(2 * 4 * 3 * 3, dtype=torch.float32).reshape(2, 4, 3, 3)
    act = torch.tanh(raw * 0.01) + 5.0                      # stands for act(scale * y + shift)
    lz = ops.LazyAct(torch.ones(4), torch.zeros(4), 2, 0.0)
    lz.z = act                                              # "already materialised"
    z = ops.tag_lazy(raw, lz)
    assert isinstance(z, ops.LazyRaw) and ops.lazy_of(z) is lz
    assert z.shape == raw.shape and z.data_ptr() == raw.data_ptr() and z.stride() == raw.stride() and z.dim() == 4 and z.numel() == raw.numel()
    assert z.dtype == torch.float32 and not z.is_cuda and z.device == raw.device
    checks = {
        "relu": (torch.relu(z), torch.relu(act)),
        "add": (z + 1.0, act + 1.0),
        "radd": (1.0 + z, act + 1.0),
        "slice": (z[:, 1:3], act[:, 1:3]),
        "view": (z.view(2, -1), act.view(2, -1)),
        "float64": (z.double(), act.double()),
        "detach": (z.detach(), act),
        "clone": (z.clone(), act),
        "cat": (torch.cat([z, z], 1), torch.cat([act, act], 1)),
        "mean": (z.mean(), act.mean()),
        "permute": (z.permute(0, 2, 3, 1), act.permute(0, 2, 3, 1)),
        "F.silu": (torch.nn.functional.silu(z), torch.nn.functional.silu(act)),
    }
    conv = nn.Conv2d(4, 2, 1)
    checks["nn.Conv2d"] = (conv(z), conv(act))
    for name, (got, exp) in checks.items():
        assert type(got) is torch.Tensor, name             # results are ordinary tensors: the tag does not leak
        assert torch.equal(got, exp), name
    # a hook-style consumer that indexes and compares
    assert bool((z > 4.0).all()) and not bool((raw > 4.0).all())
    # an untagged LazyRaw (tag removed) behaves as the plain tensor it is
    z2 = ops.tag_lazy(raw.clone(), lz)
    z2._hip_lazy = None
    assert torch.equal(z2 + 0.0, raw)


def test_cat_lazy_tag_is_thread_local():
    import threading
    ops._cat_lazy[0] = "main"
    seen = []
    th = threading.Thread(target=lambda: seen.append(ops._cat_lazy[0]))
    th.start()
    th.join()
    assert seen == [None] and ops._cat_lazy[0] == "main"
    ops._cat_lazy[0] = None


def test_parked_gradients_are_accounted_for():
    """a gradient parked in a GradLink whose main consumer never runs its backward must not vanish silently"""
    link = ops.GradLink()
    ops.check_parked()                       # nothing parked
    link.g = torch.ones(2)
    with pytest.raises(L.CvhipError):
        ops.check_parked()
    assert link.g is None                    # cleared for the next step
    link.g = torch.ones(2)
    link.g = None                            # the folding layer took it
    ops.check_parked()


def test_offered_column_sums_are_taken_once_by_address_and_width():
    """ops.offer_colsum / _take_colsum (round 6: the fused YOLOv5 loss hands the detect convolutions their bias gradient): keyed by the
    gradient map's address, consumed by the first taker, refused on a width mismatch, emptied by the next producer's backward; the
    entry keeps the map alive so that its address cannot be re-used while the sums wait"""
    import torch
    from cvpytorch_amd import ops
    ops.clear_colsums()
    draw = torch.zeros(2, 4, 4, 8)
    part = torch.zeros(3, 2, 5)
    ops.offer_colsum(draw, part, 3, 5)
    assert ops._take_colsum(draw.data_ptr() + 2, 5) is None          # a channel slice that does not start at channel 0
    assert ops._take_colsum(draw.data_ptr(), 4) is None              # width mismatch: refused (and dropped: the taker computes its own sums)
    assert ops._take_colsum(draw.data_ptr(), 5) is None
    ops.offer_colsum(draw, part, 3, 5)
    ptr = draw.data_ptr()
    del draw
    got = ops._take_colsum(ptr, 5)
    assert got is not None and got[0] is part and got[1] == 3
    assert ops._take_colsum(ptr, 5) is None                          # taken once
    other = torch.zeros(2, 4, 4, 8)
    ops.offer_colsum(other, part, 3, 5)
    ops.clear_colsums()
    assert not ops._COLSUMS
