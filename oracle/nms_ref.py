"""ORACLE (test infrastructure, CPU): restatement of the reference's mmdet-style NMS helpers, src/models/modules/nms.py.
Only tests/ import this. `nms` itself is third-party (torchvision.ops.nms) and is taken from oracle.torch_ref.nms, which is pinned
by hand-derived known-answer vectors (tests/golden/nms_kat.json); the two functions below are pinned by fixtures captured from the
reference's own functions (tools/gen_golden_post.py -> tests/golden/post_batched_nms_*.npz, post_multiclass_nms_*.npz)."""
import torch

from .torch_ref import nms


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """modules/nms.py:70-132: per-class NMS through class offsets (max coordinate + 1), or class by class above `split_thr` boxes
    (then re-ordered by descending score); returns (dets (k, 5) = boxes | score, keep indices)."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop("class_agnostic", class_agnostic)
    if class_agnostic:
        shifted = boxes
    else:
        shifted = boxes + (idxs.to(boxes) * (boxes.max() + 1))[:, None]
    cfg.pop("type", "nms")                      # popped and ignored by the reference: it always runs hard NMS (:105-106)
    split_thr = cfg.pop("split_thr", 10000)
    thr = cfg.pop("iou_threshold")
    if len(shifted) < split_thr:
        keep = nms(shifted, scores, thr)
    else:
        mask_all = torch.zeros_like(scores, dtype=torch.bool)
        for c in torch.unique(idxs):
            sel = (idxs == c).nonzero(as_tuple=False).view(-1)
            mask_all[sel[nms(shifted[sel], scores[sel], thr)]] = True
        keep = mask_all.nonzero(as_tuple=False).view(-1)
        keep = keep[scores[keep].argsort(descending=True)]
    return torch.cat([boxes[keep], scores[keep][:, None]], -1), keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """modules/nms.py:5-67: the last score column is the background and is dropped; (box, class) pairs above `score_thr` go through
    batched_nms with the class as group; at most `max_num` detections. Returns (dets (k, 5), labels (k,))."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    valid = scores > score_thr
    bboxes = bboxes[valid]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid]
    labels = valid.nonzero(as_tuple=False)[:, 1]
    if bboxes.numel() == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep]
