#!/bin/bash
# round-2 GPU call I: FAST igemm staging (running source pointers): parity + A/B + conv table
mkdir -p gpurun_out
T="timeout 900"
$T python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fp16.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/i_t_kernels.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deeplab --no-h2d --no-kernel-timing"
for rep in 1 2; do
CVHIP_IGEMM_FAST=0 $T $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FAST=0', d['value'], d['ms_per_step'])" >> gpurun_out/i_ab.log
CVHIP_IGEMM_FAST=1 $T $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('FAST=1', d['value'], d['ms_per_step'])" >> gpurun_out/i_ab.log
done
$T python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-deeplab --no-h2d > gpurun_out/i_bench.log 2>&1
cat gpurun_out/i_t_kernels.log gpurun_out/i_ab.log; tail -1 gpurun_out/i_bench.log | cut -c1-1500
