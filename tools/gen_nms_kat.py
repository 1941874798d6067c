"""Hand-derived known-answer vectors for torchvision.ops.nms (the third-party op the reference's post-processing calls:
src/models/yolov5.py:141, src/models/yolox.py:64, src/models/modules/nms.py:111; torchvision is not installed in this image and
is not vendored by the reference). The EXPECTED keep lists below were worked out by hand from the published CPU algorithm
(torchvision/csrc/ops/cpu/nms_kernel.cpp):

    areas = (x2 - x1) * (y2 - y1);  order = scores.sort(descending)
    for i in order (not suppressed): keep i; for every later j: inter = max(0, min(x2)-max(x1)) * max(0, min(y2)-max(y1));
        ovr = inter / (area_i + area_j - inter);  if (ovr > iou_threshold) suppressed[j] = 1        (STRICT >, fp32)

Every case uses coordinates whose areas / intersections / unions are exact small integers or powers of two, so the IoU values
quoted in the comments are exact in fp32 — no arithmetic was delegated to the oracle or to the kernels. Where the algorithm
leaves the order of EQUAL scores open (the sort in torchvision 0.7 is not stable), the case lists the admissible answers.

    python tools/gen_nms_kat.py        -> tests/golden/nms_kat.json
"""
import json
import os

CASES = []


def case(name, boxes, scores, thr, expect, why, alt=None):
    CASES.append({"name": name, "boxes": boxes, "scores": scores, "iou_threshold": thr, "keep": expect, "keep_alternatives": alt or [],
                  "derivation": why})


# 1. IoU exactly AT the threshold is NOT suppressed (strict >):
#    A = [0,0,2,1] area 2, B = [0,0,1,1] area 1, inter = 1, union = 2 + 1 - 1 = 2, IoU = 0.5 exactly.
case("iou_equals_threshold_keeps_both", [[0, 0, 2, 1], [0, 0, 1, 1]], [0.9, 0.8], 0.5, [0, 1],
     "IoU(A,B) = 1/2 = 0.5; 0.5 > 0.5 is false -> B survives")
# 2. ... and just below the threshold value it IS suppressed: same boxes, threshold 0.25 (IoU 0.5 > 0.25).
case("iou_above_threshold_suppresses", [[0, 0, 2, 1], [0, 0, 1, 1]], [0.9, 0.8], 0.25, [0],
     "IoU(A,B) = 0.5 > 0.25 -> B suppressed")
# 3. exact quarter: A = [0,0,4,1] area 4, B = [0,0,1,1] area 1, inter 1, union 4, IoU = 0.25; threshold 0.25 -> kept.
case("iou_quarter_at_threshold", [[0, 0, 4, 1], [0, 0, 1, 1]], [0.6, 0.7], 0.25, [1, 0],
     "B has the higher score so it is visited first; IoU = 1/4, 0.25 > 0.25 false -> both kept, output in score order [B, A]")
# 4. greedy chain: a suppressed box does not suppress anybody.
#    a=[0,0,10,10] (100), b=[4,0,14,10] (100), c=[8,0,18,10] (100).
#    IoU(a,b) = 60 / 140 = 3/7 ~ 0.4286 > 0.4 -> b suppressed.  IoU(a,c) = 20 / 180 = 1/9 ~ 0.111 -> kept.
#    IoU(b,c) = 3/7 > 0.4 but b is already suppressed, so c survives.
case("suppressed_box_does_not_suppress", [[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], [0.9, 0.8, 0.7], 0.4, [0, 2],
     "IoU(a,b)=3/7>0.4 kills b; IoU(a,c)=1/9; b (dead) cannot kill c")
# 5. same boxes, b scored highest: b kills both neighbours (IoU 3/7 with each).
case("middle_box_first_kills_both", [[0, 0, 10, 10], [4, 0, 14, 10], [8, 0, 18, 10]], [0.7, 0.9, 0.8], 0.4, [1],
     "order b, c, a; IoU(b,c) = IoU(b,a) = 3/7 > 0.4")
# 6. disjoint boxes: inter = max(0, negative) * ... = 0, IoU 0 -> all kept, in score order.
case("disjoint_all_kept_in_score_order", [[0, 0, 1, 1], [10, 10, 11, 11], [20, 0, 21, 1], [0, 20, 1, 21]], [0.1, 0.4, 0.3, 0.2], 0.0,
     [1, 2, 3, 0], "all intersections empty: IoU = 0, and 0 > 0.0 is false even at threshold 0")
# 7. touching boxes share an edge: width of the intersection is 0 -> IoU 0.
case("touching_edges_not_overlapping", [[0, 0, 2, 2], [2, 0, 4, 2]], [0.5, 0.4], 0.0, [0, 1],
     "xx2 - xx1 = 2 - 2 = 0 -> inter 0")
# 8. identical boxes: IoU = 4 / (4 + 4 - 4) = 1 > any threshold < 1 -> the lower-scored copy dies.
case("identical_boxes", [[1, 1, 3, 3], [1, 1, 3, 3], [1, 1, 3, 3]], [0.3, 0.9, 0.6], 0.99, [1],
     "IoU = 1 > 0.99 for both copies")
# 9. threshold 1.0 never suppresses (IoU <= 1 is never > 1).
case("threshold_one_keeps_identical", [[1, 1, 3, 3], [1, 1, 3, 3]], [0.2, 0.8], 1.0, [1, 0], "1 > 1 is false")
# 10. zero-area boxes: two identical degenerate boxes give inter 0, union 0 -> 0/0 = NaN; NaN > thr is false -> both kept.
case("zero_area_pair_nan_iou", [[5, 5, 5, 9], [5, 5, 5, 9]], [0.9, 0.8], 0.1, [0, 1],
     "areas 0, inter 0: ovr = 0/0 = NaN, comparison false")
# 11. a zero-area box inside a real one: inter = 0 (zero width) -> IoU 0 -> kept.
case("zero_area_inside_real_box", [[0, 0, 10, 10], [5, 2, 5, 8]], [0.9, 0.8], 0.0, [0, 1], "inter width 5 - 5 = 0")
# 12. containment: big [0,0,4,4] (16) contains small [1,1,3,3] (4): IoU = 4 / 16 = 0.25.
case("contained_box_iou_quarter", [[0, 0, 4, 4], [1, 1, 3, 3]], [0.9, 0.8], 0.2, [0], "IoU = 4/16 = 0.25 > 0.2")
case("contained_box_iou_quarter_kept", [[0, 0, 4, 4], [1, 1, 3, 3]], [0.9, 0.8], 0.25, [0, 1], "0.25 > 0.25 false")
# 13. equal scores on boxes that do not interact: both kept; their relative ORDER is not defined by an unstable sort.
case("equal_scores_disjoint", [[0, 0, 1, 1], [5, 5, 6, 6]], [0.5, 0.5], 0.5, [0, 1], "disjoint; tie order open", alt=[[1, 0]])
# 14. equal scores on identical boxes: exactly one survives; WHICH one is not defined by an unstable sort.
case("equal_scores_identical", [[0, 0, 2, 2], [0, 0, 2, 2]], [0.5, 0.5], 0.5, [0], "IoU 1 > 0.5: one copy survives", alt=[[1]])
# 15. single box / empty input.
case("single_box", [[3, 4, 5, 6]], [0.1], 0.5, [0], "nothing to compare")
case("empty", [], [], 0.5, [], "no boxes")
# 16. negative coordinates and a partial overlap with an exact IoU of 1/7:
#     A = [-2,-2,2,2] (16), B = [0,0,4,4] (16): inter 2*2 = 4, union 28, IoU = 1/7 ~ 0.142857.
case("negative_coordinates_one_seventh", [[-2, -2, 2, 2], [0, 0, 4, 4]], [0.9, 0.8], 0.14, [0], "1/7 = 0.142857 > 0.14")
case("negative_coordinates_one_seventh_kept", [[-2, -2, 2, 2], [0, 0, 4, 4]], [0.9, 0.8], 0.15, [0, 1], "1/7 < 0.15")
# 17. class offsets (batched NMS): two identical boxes shifted by 4096 per class no longer interact.
case("class_offset_separates", [[0, 0, 2, 2], [4096, 4096, 4098, 4098], [0, 0, 2, 2]], [0.9, 0.8, 0.7], 0.5, [0, 1],
     "box 2 duplicates box 0 (IoU 1) and dies; box 1 is the same box offset by one class stride: disjoint")

if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "tests", "golden", "nms_kat.json")
    json.dump({"source": "hand-derived from torchvision/csrc/ops/cpu/nms_kernel.cpp (see tools/gen_nms_kat.py)", "cases": CASES},
              open(path, "w"), indent=1)
    print("%d cases -> %s" % (len(CASES), path))
