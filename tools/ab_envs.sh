#!/bin/bash
# In-box A/B of environment settings on the YOLOv5-s leg: bash tools/ab_envs.sh "A=1 B=2" "A=3" ... (ROUNDS alternating rounds, default 3)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq ${ROUNDS:-3}); do
  for kv in "$@"; do
    env $kv python $R/bench.py --no-extra --no-deeplab --no-cpu-baseline --no-h2d --no-kernel-timing --steps 30 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$kv', d['value'], d['ms_per_step'], d.get('ms_per_step_median'))"
  done
done
