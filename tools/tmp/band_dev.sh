#!/bin/bash
# dev call of round 5 (second session): the band kernel's forms — parity, isolated timings against the other 3x3 kernels and against the
# previous build of the library, step A/B, SQ counters
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_band.py -x -q > $O/band_tests.log 2>&1; echo "tests rc $?" | tee -a $O/band_tests.log; tail -3 $O/band_tests.log
W="CVHIP_PATCH=1,CVHIP_BAND=2"
NO_PRO=1 ROUNDS=3 VARIANTS="wide:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=0;wide_pf:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=1;narrow_pf:$W,CVHIP_BAND_NF=2,CVHIP_BAND_PF=1" timeout 300 python tools/patch_bench.py > $O/band_bench.log 2>&1
echo "==== previous build (commit 29458d8: narrow form only) ====" >> $O/band_bench.log
NO_PRO=1 ROUNDS=3 ONLY=y5s CVHIP_LIB=$R/tools/tmp/libcvhip_head.so timeout 200 python tools/patch_bench.py >> $O/band_bench.log 2>&1
cat $O/band_bench.log | grep -v "^$" | tail -120
step() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d --no-extra 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readline()); print('ms_per_step %.3f  median %.3f  value %.0f' % (d['ms_per_step'], d.get('ms_per_step_median', 0), d['value']))"; }
( for i in 1 2; do
  echo -n "head build           "; step CVHIP_LIB=$R/tools/tmp/libcvhip_head.so
  echo -n "narrow (default)     "; step CVHIP_BAND=1
  echo -n "wide                 "; step CVHIP_BAND_NF=4 CVHIP_BAND_PF=0
  echo -n "wide + read-ahead    "; step CVHIP_BAND_NF=4 CVHIP_BAND_PF=1
  echo -n "all 3x3 s1: wide+pf  "; step CVHIP_BAND=2 CVHIP_BAND_NF=4 CVHIP_BAND_PF=1
  echo -n "band off             "; step CVHIP_BAND=0
done ) > $O/band_step_ab.log 2>&1
cat $O/band_step_ab.log
timeout 400 bash tools/tmp/band_pmc.sh; tail -60 $O/band_sq.txt
