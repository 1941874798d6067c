"""Detection post-processing on the device.

  detect_postprocess      the whole batch in one launch set (cvhip_detect_postprocess): confidence filter -> top-`cap` selection ->
                          class-offset boxes -> greedy NMS -> FIXED-CAPACITY outputs (dets [B, max_det, 6|7], counts [B],
                          overflow [B]); no host synchronisation anywhere. Replaces the per-image loops of
                          src/models/yolov5.py:62-153 (non_max_suppression) and src/models/yolox.py:48-68 (yolox_post_process).
  non_max_suppression     reference-shaped adapter: list of (n_i, 6) tensors, ONE host read of the counts for all images.
  yolox_post_process      same for YOLOX: list of (n_i, 7) tensors / None.
  batched_nms, multiclass_nms   mirrors of src/models/modules/nms.py:5-132 (same signatures and return values), sorting with
                          cvhip_argsort_desc_f32 and suppressing with cvhip_nms_sorted instead of torch.sort / torchvision.ops.nms.
"""
import torch

from . import lib as L
from .ops import _stream


def _pow2_at_least(x, lo=64, hi=8192):
    p = lo
    while p < x and p < hi:
        p <<= 1
    return p


def detect_postprocess(pred, conf_thres, iou_thres, mode=0, multi_label=False, agnostic=False, max_det=300, cap=4096, cand_cap=None,
                       max_wh=4096.0):
    """pred: fp32 (B, n, 5 + nc [+ extra]) decoded rows {cx, cy, w, h, obj, cls...}. Returns (dets, counts, overflow) device
    tensors: dets (B, max_det, 6) {x1,y1,x2,y2,conf,cls} for mode 0 / (B, max_det, 7) {x1,y1,x2,y2,obj,class_conf,cls} for mode 1,
    rows >= counts[b] zero. `cap` (power of two <= 8192) plays the reference's max_nms."""
    if not pred.is_cuda:
        raise L.CvhipError("detect_postprocess needs a device tensor (no CPU fallback)")
    pred = pred.float().contiguous()
    B, n, no = pred.shape
    nc = no - 5 if mode == 0 else None
    if mode == 1:
        raise L.CvhipError("mode 1 needs nc: use yolox_post_process")
    return _run(pred, B, n, no, nc, conf_thres, iou_thres, mode, multi_label, agnostic, max_det, cap, cand_cap, max_wh)


def _run(pred, B, n, no, nc, conf_thres, iou_thres, mode, multi_label, agnostic, max_det, cap, cand_cap, max_wh):
    dev = pred.device
    multi_label = bool(multi_label and nc > 1 and mode == 0)
    cap = _pow2_at_least(min(int(cap), 8192))
    if cand_cap is None:
        cand_cap = n if not multi_label else min(n * nc, max(n, 262144))
    lib = L.load()
    nbytes = lib.cvhip_detect_postprocess_workspace_bytes(B, cand_cap, cap)
    if nbytes < 0:
        raise L.CvhipError("detect_postprocess: bad capacity %d" % cap)
    ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=dev)
    ncol = 6 if mode == 0 else 7
    dets = torch.empty((B, max_det, ncol), dtype=torch.float32, device=dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    overflow = torch.empty((B,), dtype=torch.int32, device=dev)
    L.call("cvhip_detect_postprocess", pred.data_ptr(), B, n, no, nc, float(conf_thres), float(iou_thres),
           0.0 if agnostic else float(max_wh), mode, int(multi_label), cand_cap, cap, max_det, ws.data_ptr(), dets.data_ptr(),
           counts.data_ptr(), overflow.data_ptr(), _stream())
    return dets, counts, overflow


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300,
                        cap=4096):
    """src/models/yolov5.py:62-153 for the whole batch on the device; returns the reference's list of (n_i, 6) tensors
    [xyxy, conf, cls] (one host read of the B counts). `classes` filtering is not on this path (the hot path never uses it)."""
    if classes is not None:
        raise L.CvhipError("non_max_suppression(classes=...) is not supported by the batched device path")
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    dets, counts, overflow = detect_postprocess(prediction, conf_thres, iou_thres, 0, multi_label, agnostic, max_det, cap)
    host = torch.stack((counts, overflow)).tolist()   # ONE host read: counts and overflow flags of every image
    out = [dets[i, :c] for i, c in enumerate(host[0])]
    over = [i for i, o in enumerate(host[1]) if o]
    if over:
        # an image with more candidates than the batched kernels' capacity: the reference keeps up to max_nms = 30000 of them
        # (models/yolov5.py:66,131-132); those images take the per-image loop, whose device sort / NMS kernels have no capacity limit
        from .yolov5 import non_max_suppression as per_image
        for i in over:
            out[i] = per_image(prediction[i:i + 1], conf_thres, iou_thres, None, agnostic, multi_label, max_det, nms_fn=nms)[0]
    return out


def yolox_post_process(pred, num_classes, conf_thre, nms_thre, cap=4096):
    """src/models/yolox.py:48-68 on decoded rows (B, A, 5 + nc [+ reid]) {cx, cy, w, h, obj, cls...}: per image
    (x1, y1, x2, y2, obj_conf, class_conf, class_pred) or None. torchvision.ops.batched_nms semantics (class offsets of
    max coordinate + 1), all images in one launch set."""
    pred = pred.float().contiguous()
    B, n, no = pred.shape
    cap = _pow2_at_least(min(int(cap), 8192))
    dets, counts, overflow = _run(pred, B, n, no, int(num_classes), conf_thre, nms_thre, 1, False, False, cap, cap, None, 0.0)
    host = torch.stack((counts, overflow)).tolist()
    out = [dets[i, :c] if c else None for i, c in enumerate(host[0])]
    for i, (c, o) in enumerate(zip(*host)):
        if o or c >= cap:
            # more candidates (or detections) than the fixed capacity: the reference is unbounded (models/yolox.py:48-68) — redo this
            # image with the per-box kernels (device sort + NMS of any length)
            out[i] = _yolox_one_image(pred[i], int(num_classes), conf_thre, nms_thre)
    return out


def _yolox_one_image(p, num_classes, conf_thre, nms_thre):
    """models/yolox.py:48-68 for one image on the per-box kernels: (x1, y1, x2, y2, obj_conf, class_conf, class_pred) or None"""
    box = torch.empty_like(p[:, :4])
    box[:, 0] = p[:, 0] - p[:, 2] / 2
    box[:, 1] = p[:, 1] - p[:, 3] / 2
    box[:, 2] = p[:, 0] + p[:, 2] / 2
    box[:, 3] = p[:, 1] + p[:, 3] / 2
    class_conf, class_pred = torch.max(p[:, 5:5 + num_classes], 1, keepdim=True)
    mask = (p[:, 4] * class_conf.squeeze(1) >= conf_thre)
    d = torch.cat((box, p[:, 4:5], class_conf, class_pred.float()), 1)[mask]
    if not d.shape[0]:
        return None
    offs = d[:, 6:7] * (d[:, :4].max() + 1)          # torchvision.ops.batched_nms: per-class coordinate offsets
    keep = nms(d[:, :4] + offs, d[:, 4] * d[:, 5], nms_thre)
    return d[keep]


def argsort_desc(scores):
    """Device argsort: descending, ties by ascending index (stable)."""
    scores = scores.float().contiguous()
    n = scores.numel()
    order = torch.empty((n,), dtype=torch.int64, device=scores.device)
    if n == 0:
        return order
    lib = L.load()
    ws = torch.empty((int(lib.cvhip_sort_workspace_bytes(n)),), dtype=torch.uint8, device=scores.device)
    L.call("cvhip_argsort_desc_f32", scores.data_ptr(), n, ws.data_ptr(), order.data_ptr(), _stream())
    return order


def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms contract (indices of kept boxes, decreasing score) on the HIP kernels, device sort included."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64, device=boxes.device)
    if not boxes.is_cuda:
        raise L.CvhipError("cvpytorch_amd.nms needs CUDA/HIP tensors (no CPU fallback)")
    order = argsort_desc(scores)
    b = boxes.float()[order].contiguous()
    n = b.shape[0]
    lib = L.load()
    ws = torch.empty((int(lib.cvhip_nms_workspace_bytes(n)),), dtype=torch.uint8, device=b.device)
    keep = torch.empty((n,), dtype=torch.int32, device=b.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=b.device)
    L.call("cvhip_nms_sorted", b.data_ptr(), n, float(iou_threshold), ws.data_ptr(), keep.data_ptr(), cnt.data_ptr(), _stream())
    k = int(cnt.item())   # the contract returns a tensor of data-dependent length: one host read
    return order[keep[:k].long()]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """src/models/modules/nms.py:70-132: per-class NMS through coordinate offsets; returns (dets (k, 5), keep (k,))."""
    nms_cfg_ = nms_cfg.copy()
    class_agnostic = nms_cfg_.pop("class_agnostic", class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + 1)
        boxes_for_nms = boxes + offsets[:, None]
    nms_cfg_.pop("type", "nms")
    split_thr = nms_cfg_.pop("split_thr", 10000)
    iou_thr = nms_cfg_.pop("iou_threshold", nms_cfg_.pop("iou_thr", 0.5))
    if len(boxes_for_nms) < split_thr:
        keep = nms(boxes_for_nms, scores, iou_thr)
        boxes = boxes[keep]
        scores = scores[keep]
    else:
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        for id_ in torch.unique(idxs):
            mask = (idxs == id_).nonzero(as_tuple=False).view(-1)
            keep = nms(boxes_for_nms[mask], scores[mask], iou_thr)
            total_mask[mask[keep]] = True
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        keep = keep[argsort_desc(scores[keep])]
        boxes = boxes[keep]
        scores = scores[keep]
    return torch.cat([boxes, scores[:, None]], -1), keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """src/models/modules/nms.py:5-67: (n, #class*4 | 4) boxes, (n, #class + 1) scores (last column = background) ->
    (dets (k, 5), labels (k,))."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 4)
    scores = multi_scores[:, :-1]
    valid_mask = scores > score_thr
    bboxes = torch.masked_select(bboxes, torch.stack((valid_mask, valid_mask, valid_mask, valid_mask), -1)).view(-1, 4)
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = torch.masked_select(scores, valid_mask)
    labels = valid_mask.nonzero(as_tuple=False)[:, 1]
    if bboxes.numel() == 0:
        return multi_bboxes.new_zeros((0, 5)), multi_bboxes.new_zeros((0,), dtype=torch.long)
    dets, keep = batched_nms(bboxes, scores, labels, nms_cfg)
    if max_num > 0:
        dets = dets[:max_num]
        keep = keep[:max_num]
    return dets, labels[keep]
