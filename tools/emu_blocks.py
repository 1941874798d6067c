"""Dev tool: teacher-forced block-by-block agreement of the engine with the storage emulator (tests/test_gpu_storage_emulator.py asserts it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_storage_emulator import blocks_teacher_forced
for spec in sys.argv[1:] or ("bf16:yolov5s", "fp16:yolov5s", "bf16:yoloxs", "fp16:yolov7l"):
    prec, kind = spec.split(":")
    print("==", prec, kind)
    for r in blocks_teacher_forced(prec, kind):
        print("  %-34s %-10s out_rel %.2e  dx_cos %s  param_cos_min %.6f (%d)" % (r["name"], r["kind"], r["out_rel"], "   -    " if r["dx_cos"] is None else "%.6f" % r["dx_cos"], r["param_cos_min"], r["n_params"]))
