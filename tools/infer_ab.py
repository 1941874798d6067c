"""A/B of the inference legs (bench.py infer_leg) under an environment switch, e.g.
    python tools/infer_ab.py CVHIP_BAND=0 CVHIP_BAND=1
Each setting runs in-process (the switches are read per launch); prints images/s per leg."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    a = types.SimpleNamespace(batch=64, size=640)
    for setting in sys.argv[1:] or ["CVHIP_BAND=1"]:
        k, v = setting.split("=")
        os.environ[k] = v
        for kind in ("yolov5s", "deeplab"):
            best = 0.0
            for rep in range(3):
                r = bench.infer_leg(dev, a, 20, 3, kind)
                best = max(best, r["value"])
            print("%s %s: %.1f img/s (ms/batch %.3f) finite=%s" % (setting, kind, best, r["ms_per_batch"], r["finite"]), flush=True)


if __name__ == "__main__":
    main()
