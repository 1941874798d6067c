#!/bin/bash
# round-2 GPU call P: wgrad with three steps of operands in flight (asm-issued loads, explicit vmcnt): parity + A/B
mkdir -p gpurun_out
T="timeout 900"
$T python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/p_t_kernels.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deeplab --no-h2d --no-kernel-timing"
for rep in 1 2; do
CVHIP_WGRAD_PD=1 $T $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PD=1', d['value'], d['ms_per_step'])" >> gpurun_out/p_ab.log
$T $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('PD=3', d['value'], d['ms_per_step'])" >> gpurun_out/p_ab.log
done
TABLE_ROWS=60 $T python tools/conv_table.py 2>/dev/null | grep wgrad > gpurun_out/p_table_pd3.log
CVHIP_WGRAD_PD=1 TABLE_ROWS=60 $T python tools/conv_table.py 2>/dev/null | grep wgrad > gpurun_out/p_table_pd1.log
cat gpurun_out/p_t_kernels.log gpurun_out/p_ab.log; head -12 gpurun_out/p_table_pd3.log; echo; head -12 gpurun_out/p_table_pd1.log
