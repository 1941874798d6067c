#!/bin/bash
# round-2 GPU call D: full gpu suite after the OTA kernels / test fixes; default bench line
mkdir -p gpurun_out
T="timeout 1500"
$T python -m pytest tests -m gpu -q 2>&1 | tail -80 > gpurun_out/d_t_all.log
$T python bench.py --steps 20 --warmup 5 > gpurun_out/d_bench.log 2>&1
tail -30 gpurun_out/d_t_all.log; grep -h '^{' gpurun_out/d_bench.log | cut -c1-400
