#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_band.py -x -q > $O/band_tests.log 2>&1; echo "tests rc $?" | tee -a $O/band_tests.log; tail -3 $O/band_tests.log
W="CVHIP_PATCH=1,CVHIP_BAND=2"
NO_PRO=1 ROUNDS=3 ONLY=y5s VARIANTS="nw4:$W,CVHIP_BAND_NF=2,CVHIP_BAND_PF=0,CVHIP_BAND_NW=4;nw4_pf:$W,CVHIP_BAND_NF=2,CVHIP_BAND_PF=1,CVHIP_BAND_NW=4;nw4_wide_pf:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=1,CVHIP_BAND_NW=4;P_nw4:$W,CVHIP_BAND_NF=2,CVHIP_BAND_PF=0,CVHIP_BAND_NW=4,CVHIP_BAND_PROBE_W=1;P_nw4_wide_pf:$W,CVHIP_BAND_NF=4,CVHIP_BAND_PF=1,CVHIP_BAND_NW=4,CVHIP_BAND_PROBE_W=1" timeout 300 python tools/patch_bench.py > $O/band_probe2.log 2>&1
grep -v "^$" $O/band_probe2.log | grep -v "s2 (dgrad\|bn_act\|per-tap\|patch " | head -90
