#!/bin/bash
# round-2 GPU call K: fast divmod + uniform epilogue: parity + per-shape + bench
mkdir -p gpurun_out
T="timeout 900"
$T python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/k_t_kernels.log
export ABL_BATCH=64
( $T python tools/conv_ablate.py ) 2>&1 | grep -v Warn > gpurun_out/k_ablate.log
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deeplab --no-h2d --no-kernel-timing"
for rep in 1 2; do
$T $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'])" >> gpurun_out/k_ab.log
done
cat gpurun_out/k_t_kernels.log gpurun_out/k_ablate.log gpurun_out/k_ab.log
