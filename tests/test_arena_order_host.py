"""Host logic of the flat-arena layout: sibling tensors (ops.ConvBnActPair) must end up back to back, everything else keeps its
registration order, and the returned permutation maps the EMA copy's tensors onto the same slots."""
import torch

from cvpytorch_amd.arena import paired_arena_order


def test_paired_arena_order():
    t = [torch.zeros(1) for _ in range(8)]          # registration order: a.w a.g a.b  b.w b.g b.b  c.w  d.w
    follow = {id(t[0]): t[3], id(t[1]): t[4], id(t[2]): t[5]}
    out, perm = paired_arena_order(t, follow)
    assert [id(x) for x in out] == [id(t[i]) for i in (0, 3, 1, 4, 2, 5, 6, 7)]
    assert perm == [0, 3, 1, 4, 2, 5, 6, 7] and sorted(perm) == list(range(8))
    ema = ["e%d" % i for i in range(8)]            # a deep copy's tensors in registration order
    assert [ema[i] for i in perm] == ["e0", "e3", "e1", "e4", "e2", "e5", "e6", "e7"]


def test_paired_arena_order_partner_missing_or_earlier():
    t = [torch.zeros(1) for _ in range(4)]
    frozen = torch.zeros(1)
    out, perm = paired_arena_order(t, {id(t[0]): frozen, id(t[3]): t[1]})   # partner not in the arena / registered before its head
    assert [id(x) for x in out] == [id(x) for x in t] and perm == [0, 1, 2, 3]
    out, _ = paired_arena_order(t, {})
    assert [id(x) for x in out] == [id(x) for x in t]


def test_yolov5_sibling_pairs_are_declared():
    from cvpytorch_amd import yolov5, yolo_blocks
    m = yolov5.YOLOv5(80, "n", max_targets=8)
    pairs = [p for mod in m.modules() if hasattr(mod, "hip_sibling_pairs") for p in mod.hip_sibling_pairs()]
    assert len(pairs) == 8 and all(a.conv.weight.shape == b.conv.weight.shape for a, b in pairs)
    assert all(isinstance(mod, yolo_blocks.CSPLayer) for mod in m.modules() if hasattr(mod, "hip_sibling_pairs"))
