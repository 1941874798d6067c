#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_band.py tests/test_gpu_arena.py tests/test_gpu_patch.py -x -q > $O/band_tests.log 2>&1; echo "tests rc $?" | tee -a $O/band_tests.log; tail -3 $O/band_tests.log
W="CVHIP_PATCH=1,CVHIP_BAND=2"
NO_PRO=1 ROUNDS=3 VARIANTS="narrow_pf:$W,CVHIP_BAND_NF=2,CVHIP_BAND_PF=1;wide:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=0;wide_pf:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=1;nw4:$W,CVHIP_BAND_NF=2,CVHIP_BAND_PF=0,CVHIP_BAND_NW=4;nw4_pf:$W,CVHIP_BAND_NF=2,CVHIP_BAND_PF=1,CVHIP_BAND_NW=4;nw4_wide:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=0,CVHIP_BAND_NW=4;nw4_wide_pf:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=1,CVHIP_BAND_NW=4" timeout 400 python tools/patch_bench.py > $O/band_bench2.log 2>&1
grep -v "^$" $O/band_bench2.log | grep -v "s2 (dgrad\|bn_act\|band plan" | head -150
step() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d --no-extra 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('ms_per_step %.3f  median %.3f  value %.0f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), d['value']))"; }
( for i in 1 2; do
  echo -n "head build                 "; step CVHIP_LIB=$R/tools/tmp/libcvhip_head.so
  echo -n "default                    "; step CVHIP_BAND=1
  echo -n "default, NW=4              "; step CVHIP_BAND_NW=4
  echo -n "all 3x3 s1 band            "; step CVHIP_BAND=2
  echo -n "all 3x3 s1 band, NW=4      "; step CVHIP_BAND=2 CVHIP_BAND_NW=4
  echo -n "all, wide+pf, NW=4         "; step CVHIP_BAND=2 CVHIP_BAND_NW=4 CVHIP_BAND_NF=4 CVHIP_BAND_PF=1
  echo -n "all, narrow+pf, NW=4       "; step CVHIP_BAND=2 CVHIP_BAND_NW=4 CVHIP_BAND_NF=2 CVHIP_BAND_PF=1
done ) > $O/band_step_ab2.log 2>&1
cat $O/band_step_ab2.log
