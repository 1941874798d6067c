"""Detection post-processing on the device.

  detect_postprocess      the whole batch in one launch set (cvhip_detect_postprocess): confidence filter -> top-`cap` selection ->
                          class-offset boxes -> greedy NMS -> FIXED-CAPACITY outputs (dets [B, max_det, 6|7], counts [B],
                          overflow [B]); no host synchronisation anywhere. Replaces the per-image loops of
                          src/models/yolov5.py:62-153 (non_max_suppression) and src/models/yolox.py:48-68 (yolox_post_process).
  non_max_suppression     reference-shaped adapter: list of (n_i, 6) tensors, ONE host read of the counts for all images.
  yolox_post_process      same for YOLOX: list of (n_i, 7) tensors / None.
  batched_nms, multiclass_nms   the contracts of src/models/modules/nms.py:5-132 (same signatures and return values) as fixed-shape
                          device passes: sort keys instead of boolean-mask compaction, ONE class-shifted greedy NMS
                          (cvhip_argsort_desc_f32 + cvhip_nms_sorted), one host read for the data-dependent output length.
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


def _first_argmax(v):
    """(max over the last axis, index of its FIRST occurrence): ATen's CPU tie rule, which the reference's `x[:, 5:].max(1)` follows"""
    m = v.max(-1, keepdim=True).values
    idx = torch.arange(v.shape[-1], device=v.device).expand_as(v)
    first = torch.where(v == m, idx, torch.full_like(idx, v.shape[-1])).min(-1).values
    return m.squeeze(-1), first


def _apply_class_filter(prediction, classes, multi_label):
    """`classes=[...]` of the reference (models/yolov5.py:118-119: keep detections whose class is listed) as a mask on the batched
    path's INPUT: a candidate that fails any of the reference's successive filters is dropped whichever comes first, so the class
    test can run before the kernels — multi-label candidates are (row, class) pairs: unlisted class columns are zeroed (0 > conf_thres
    never holds); best-class candidates are rows: a row whose best class (first maximum of cls * obj) is unlisted gets objectness 0."""
    nc = prediction.shape[2] - 5
    sel = torch.zeros((nc,), dtype=torch.bool, device=prediction.device)
    # the reference compares `x[:, 5:6] == torch.tensor(classes)`: an id outside [0, nc) (or a non-integer one) matches NO detection —
    # it must not be folded onto class 0 / nc - 1
    ids = torch.as_tensor(list(classes), device=prediction.device).double().reshape(-1)
    ids = ids[(ids >= 0) & (ids < nc) & (ids == ids.floor())].long()
    sel[ids] = True
    pred = prediction.float().clone()
    if multi_label and nc > 1:
        pred[..., 5:] = torch.where(sel, pred[..., 5:], torch.zeros((), device=pred.device))
    else:
        _, j = _first_argmax(pred[..., 5:] * pred[..., 4:5])
        pred[..., 4] = torch.where(sel[j], pred[..., 4], torch.zeros((), device=pred.device))
    return pred


def nms_one_image_unbounded(p, conf_thres, iou_thres, agnostic=False, multi_label=False, max_det=300, max_nms=30000, max_wh=4096.0):
    """One image of models/yolov5.py:62-153 without the batched kernels' capacity, on sort keys: every (row[, class]) candidate gets
    the key `confidence if it passes the reference's filters else -inf`; ONE stable device sort (descending, ties by candidate index =
    the order the reference's mask / nonzero compaction leaves them in) puts the survivors first, the best `max_nms` of them go
    through one class-shifted greedy NMS (cvhip_nms_sorted) — no boolean-mask indexing, one host read for the survivor count.
    p: (n, 5 + nc) decoded rows. Returns (k, 6) [xyxy, conf, cls]."""
    p = p.float()
    n, nc = p.shape[0], p.shape[1] - 5
    dev = p.device
    prod = p[:, 5:] * p[:, 4:5]                          # conf = obj_conf * cls_conf
    xc = p[:, 4] > conf_thres
    if multi_label and nc > 1:
        valid = ((prod > conf_thres) & xc[:, None]).reshape(-1)
        key = torch.where(valid, prod.reshape(-1), torch.full((), float("-inf"), device=dev))
        width = nc
    else:
        best, j = _first_argmax(prod)
        valid = xc & (best > conf_thres)
        key = torch.where(valid, best, torch.full((), float("-inf"), device=dev))
        width = 1
    order = argsort_desc(key)
    nv = min(int(valid.sum().item()), int(max_nms))     # the one host read
    if nv == 0:
        return torch.zeros((0, 6), device=dev)
    top = order[:nv]
    row = top // width
    lab = (top % width) if width > 1 else j[row]
    box = torch.empty((nv, 4), dtype=torch.float32, device=dev)
    q = p[row]
    box[:, 0] = q[:, 0] - q[:, 2] / 2
    box[:, 1] = q[:, 1] - q[:, 3] / 2
    box[:, 2] = q[:, 0] + q[:, 2] / 2
    box[:, 3] = q[:, 1] + q[:, 3] / 2
    labf = lab.float()
    shifted = box + (labf * (0.0 if agnostic else max_wh))[:, None]
    keep_pos, cnt = nms_device(shifted.contiguous(), iou_thres)
    k = min(int(cnt.item()), int(max_det))
    sel = keep_pos[:k].long()
    return torch.cat((box[sel], key[top][sel][:, None], labf[sel][:, None]), 1)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, max_det=300,
                        cap=4096):
    """src/models/yolov5.py:62-153 for the whole batch on the device; returns the reference's list of (n_i, 6) tensors
    [xyxy, conf, cls] (one host read of the B counts). `classes` is a mask on the input (_apply_class_filter); an image with more
    candidates than the batched kernels' capacity (the reference keeps up to max_nms = 30000, :66,131-132) is redone by
    nms_one_image_unbounded."""
    assert 0 <= conf_thres <= 1 and 0 <= iou_thres <= 1
    if classes is not None:
        prediction = _apply_class_filter(prediction, classes, multi_label)
    dets, counts, overflow = detect_postprocess(prediction, conf_thres, iou_thres, 0, multi_label, agnostic, max_det, cap)
    host = torch.stack((counts, overflow)).tolist()   # ONE host read: counts and overflow flags of every image
    out = [dets[i, :c] for i, c in enumerate(host[0])]
    for i, o in enumerate(host[1]):
        if o:
            out[i] = nms_one_image_unbounded(prediction[i], conf_thres, iou_thres, agnostic, multi_label, max_det)
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


def _nms_cfg(nms_cfg, class_agnostic):
    """(iou threshold, class_agnostic) out of an mmdet-style nms_cfg dict (modules/nms.py:70-93 documents the keys)"""
    cfg = dict(nms_cfg or {})
    cfg.pop("type", "nms")   # popped and ignored, as the reference does (modules/nms.py:105-106: it always runs hard NMS)
    iou = cfg["iou_threshold"] if "iou_threshold" in cfg else cfg.get("iou_thr", 0.5)
    return float(iou), bool(cfg.get("class_agnostic", class_agnostic))


def nms_device(boxes_sorted, iou_threshold):
    """Greedy NMS over boxes ALREADY in decreasing-score order, entirely on the device: returns (keep positions int32 [n] — the
    first `count` entries are the kept positions in increasing order, count int32 [1]). No host synchronisation."""
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    keep = torch.empty((max(n, 1),), dtype=torch.int32, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    if n:
        ws = torch.empty((int(L.load().cvhip_nms_workspace_bytes(n)),), dtype=torch.uint8, device=dev)
        L.call("cvhip_nms_sorted", boxes_sorted.data_ptr(), n, float(iou_threshold), ws.data_ptr(), keep.data_ptr(), cnt.data_ptr(), _stream())
    return keep, cnt


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """Contract of src/models/modules/nms.py:70-132 — NMS that never suppresses across different `idxs` — returning
    (dets (k, 5) = [x1, y1, x2, y2, score] in decreasing score, keep (k,) int64 indices into `boxes`).

    One device pass instead of the reference's two code paths: boxes of class c are shifted by c * (max coordinate + 1) so that
    different classes cannot overlap, the scores are sorted on the device (cvhip_argsort_desc_f32: stable), ONE greedy NMS
    (cvhip_nms_sorted) runs over all of them and ONE host read fetches the number of survivors. The reference's `split_thr` branch
    (a per-class python loop it takes for >= 10000 boxes to bound torchvision's memory) computes the same set in the same order —
    classes are disjoint after the shift — so it needs no counterpart here; the key is accepted and ignored."""
    if not boxes.is_cuda:
        raise L.CvhipError("cvpytorch_amd.nms needs CUDA/HIP tensors (no CPU fallback)")
    iou, agnostic = _nms_cfg(nms_cfg, class_agnostic)
    n = boxes.shape[0]
    if n == 0:
        return boxes.new_zeros((0, 5)), torch.empty((0,), dtype=torch.int64, device=boxes.device)
    b32 = boxes.float()
    shifted = b32 if agnostic else b32 + (idxs.to(b32) * (b32.max() + 1))[:, None]
    order = argsort_desc(scores)
    keep_pos, cnt = nms_device(shifted[order].contiguous(), iou)
    k = int(cnt.item())                       # the contract returns data-dependent lengths: the one host read
    keep = order[keep_pos[:k].long()]
    return torch.cat([boxes[keep], scores[keep][:, None].to(boxes.dtype)], -1), keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, cap=8192):
    """Contract of src/models/modules/nms.py:5-67: boxes (n, 4) shared by all classes or (n, #class*4) per class, scores
    (n, #class + 1) whose last column is the background -> (dets (k, 5), labels (k,)), at most `max_num`, decreasing score.

    Fixed-shape device formulation (no boolean-mask compaction, no nonzero): every (box, class) pair is a candidate whose sort key is
    its score if it passes `score_thr` and -inf otherwise; a stable device sort puts the passing pairs first (ties by candidate index,
    i.e. the reference's row-major order); the `cap` best go through ONE class-shifted greedy NMS. A failing pair can never suppress a
    passing one (it sorts after all of them), so nothing has to be removed before the NMS: the survivors among the first `n_valid`
    positions are the answer. ONE host read returns (survivors, n_valid); only if more than `cap` pairs pass is the pass repeated
    with the capacity that fits (the reference itself is unbounded)."""
    if not multi_bboxes.is_cuda:
        raise L.CvhipError("cvpytorch_amd.nms needs CUDA/HIP tensors (no CPU fallback)")
    iou, agnostic = _nms_cfg(nms_cfg, False)
    n, ncls = multi_scores.shape[0], multi_scores.shape[1] - 1
    dev = multi_bboxes.device
    if n == 0 or ncls <= 0:
        return multi_bboxes.new_zeros((0, 5)), torch.zeros((0,), dtype=torch.long, device=dev)
    per_class = multi_bboxes.shape[1] > 4
    cand_boxes = (multi_bboxes.reshape(n, ncls, 4) if per_class else multi_bboxes[:, None, :].expand(n, ncls, 4)).reshape(n * ncls, 4).float()
    raw = multi_scores[:, :ncls].float()
    valid = (raw > score_thr).reshape(-1)
    sc = (raw * score_factors[:, None].float() if score_factors is not None else raw).reshape(-1)
    key = torch.where(valid, sc, torch.full_like(sc, float("-inf")))
    order = argsort_desc(key)
    n_valid = valid.sum()
    # (nothing above the threshold: the reference returns its empty pair before any NMS, modules/nms.py:52-59 — decided by the host
    # read below, where n_valid arrives anyway; the -inf class shift of that case never reaches an output)
    # the class shift of the reference uses the largest coordinate of the PASSING boxes
    max_coord = torch.where(valid[:, None], cand_boxes, torch.full_like(cand_boxes, float("-inf"))).max()
    cap = int(min(n * ncls, max(1, cap)))
    while True:
        top = order[:cap]
        lab = top % ncls
        btop = cand_boxes[top]
        shifted = btop if agnostic else btop + (lab.to(btop) * (max_coord + 1))[:, None]
        keep_pos, cnt = nms_device(shifted.contiguous(), iou)
        live = (torch.arange(keep_pos.shape[0], device=dev) < cnt) & (keep_pos < n_valid)
        k, nv = torch.stack((live.sum(), n_valid)).tolist()        # ONE host read
        if nv <= cap or cap >= n * ncls:
            break
        cap = int(min(n * ncls, _pow2_at_least(nv, hi=1 << 30)))   # more passing pairs than the capacity: redo with room for all
    if max_num > 0:
        k = min(k, max_num)
    sel = keep_pos[:k].long()       # kept positions are increasing, the passing ones come first: the first k are the survivors
    dets = torch.cat([btop[sel].to(multi_bboxes.dtype), sc[top][sel][:, None].to(multi_bboxes.dtype)], -1)
    return dets, lab[sel]
