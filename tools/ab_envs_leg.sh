#!/bin/bash
# In-box A/B of environment settings on ONE side leg of bench.py: LEG=deeplab|yolox|yolov7|stdc bash tools/ab_envs_leg.sh "A=1" "A=2" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
LEG=${LEG:-deeplab}
for i in $(seq ${ROUNDS:-2}); do
  for kv in "$@"; do
    env $kv python - <<PY 2>/dev/null
import sys, types, torch
sys.path.insert(0, "$R")
import bench
a = types.SimpleNamespace(no_graph=False, batch=64, size=640)
dev = torch.device("cuda:0")
leg = "$LEG"
if leg == "deeplab":
    r = bench.deeplab_workload(dev, a, steps=20, warmup=3)
elif leg == "yolox":
    r = bench.yolox_workload(dev, a, 20, 3)
elif leg == "yolov7":
    r = bench.yolov7_workload(dev, a, 10, 2)
else:
    r = bench.stdc_workload(dev, a, 20, 3)
print("$kv", r["value"], r["ms_per_step"])
PY
  done
done
