#!/bin/bash
# round-2 GPU call Z: granularity of the BN/activation streaming kernels for the small layers
mkdir -p gpurun_out
T="timeout 900"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deeplab --no-h2d --no-kernel-timing"
run() { env "$@" $T $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> gpurun_out/z_ab.log; }
run X=default
run CVHIP_EW_ROWS=4
run CVHIP_EW_ROWS=2
run CVHIP_RED_MINROWS=16
run CVHIP_RED_MINROWS=32
run CVHIP_RED_MINROWS=16 CVHIP_EW_ROWS=4
run X=default
cat gpurun_out/z_ab.log
