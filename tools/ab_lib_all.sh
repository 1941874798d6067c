#!/bin/bash
# In-box A/B of two library builds over EVERY leg of the bench line (tools/ab/libcvhip_prev.so against the in-tree build)
R=${GRAFT_REPO_ROOT:-/root/repo}
for which in prev new prev new; do
  if [ $which = prev ]; then export CVHIP_LIB=$R/tools/ab/libcvhip_prev.so; else unset CVHIP_LIB; fi
  python $R/bench.py --no-cpu-baseline --no-h2d --no-kernel-timing --steps 20 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
legs=['config3_deeplabv3plus_r50','config3_os8','config4_yolox_s','stdc1_cityscapes','config5_yolov7l_fp16','infer','infer_deeplabv3plus_r50']
print('$which', 'y5s %.0f'%d['value'], ' '.join('%s %.1f'%(k.replace('config','c').split('_')[0]+k[-4:], d[k]['value']) for k in legs if k in d))"
done
