#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_band.py -x -q > $O/band_tests.log 2>&1; echo "tests rc $?" | tee -a $O/band_tests.log; tail -3 $O/band_tests.log
W="CVHIP_PATCH=1,CVHIP_BAND=2"
NO_PRO=1 ROUNDS=3 ONLY=y5s VARIANTS="nw8:$W,CVHIP_BAND_NW=8;policy:CVHIP_PATCH=1,CVHIP_BAND=1" timeout 400 python tools/patch_bench.py > $O/band_bench3.log 2>&1
grep -v "^$" $O/band_bench3.log | grep -v "s2 (dgrad\|bn_act" | head -80
step() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d --no-extra 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('ms_per_step %.3f  median %.3f  value %.0f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), d['value']))"; }
( for i in 1 2 3; do
  echo -n "band off                          "; step CVHIP_BAND=0
  echo -n "default policy                    "; step CVHIP_BAND=1
  echo -n "default policy, NW 8 only         "; step CVHIP_BAND_NW=8
done ) > $O/band_policy_step_ab2.log 2>&1
cat $O/band_policy_step_ab2.log
