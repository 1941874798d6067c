#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_band.py tests/test_gpu_arena.py tests/test_gpu_lazy.py -x -q > $O/band_tests.log 2>&1; echo "tests rc $?" | tee -a $O/band_tests.log; tail -3 $O/band_tests.log
step() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d --no-extra 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('ms_per_step %.3f  median %.3f  value %.0f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), d['value']))"; }
( for i in 1 2 3; do
  echo -n "band off                          "; step CVHIP_BAND=0
  echo -n "default policy (NW 4 preferred)   "; step CVHIP_BAND=1
  echo -n "default policy, NW 8 only         "; step CVHIP_BAND_NW=8
  echo -n "every stride-1 3x3 on the band    "; step CVHIP_BAND=2
done ) > $O/band_policy_step_ab.log 2>&1
cat $O/band_policy_step_ab.log
( for i in 1 2; do for v in 0 1; do
  echo -n "yolox  CVHIP_BAND=$v  "; CVHIP_BAND=$v python tools/prof_extra.py yolox 2>/dev/null | tail -1
  echo -n "yolov7 CVHIP_BAND=$v  "; CVHIP_BAND=$v python tools/prof_extra.py yolov7 2>/dev/null | tail -1
done; done ) > $O/band_policy_extra_ab.log 2>&1
cat $O/band_policy_extra_ab.log
