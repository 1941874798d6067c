#!/bin/bash
# A/B of one environment switch on the YOLOv5-s bench step, alternating runs inside ONE gpurun call (box-to-box variance is larger than
# most effects): tools/ab_env.sh VAR [rounds] [extra bench flags...]
VAR=$1; ROUNDS=${2:-3}; shift; shift
for i in $(seq 1 $ROUNDS); do
  for v in 0 1; do
    echo -n "$VAR=$v  "
    env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d --no-extra "$@" 2>/dev/null |
      python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('ms_per_step %.3f  median %.3f  value %.0f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), d['value']))"
  done
done
