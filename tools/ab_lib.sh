#!/bin/bash
# In-box A/B of two builds of libcvhip.so: tools/ab/libcvhip_prev.so (built from the previous commit, git-ignored) against the in-tree one.
#   bash tools/ab_lib.sh [rounds]      -> img/s and ms/step of the YOLOv5-s leg, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-2}
for i in $(seq $N); do
  for which in prev new; do
    if [ $which = prev ]; then export CVHIP_LIB=$R/tools/ab/libcvhip_prev.so; else unset CVHIP_LIB; fi
    python $R/bench.py --no-extra --no-deeplab --no-cpu-baseline --no-h2d --no-kernel-timing --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$which', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
