#!/bin/bash
# round-2 GPU call C: full gpu suite incl. fp16 / batched post-process / deploy / CSPDarknet; config 5 in fp16; streaming-kernel grid sweep
mkdir -p gpurun_out
T="timeout 1200"
$T python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/c_t_all.log
$T python tools/bench_extra.py yolov7 --steps 10 > gpurun_out/c_extra_v7_fp16.log 2>&1
$T python tools/bench_extra.py yolov7 --steps 10 --bf16 > gpurun_out/c_extra_v7_bf16.log 2>&1
$T python tools/bench_extra.py yolox --steps 10 > gpurun_out/c_extra_yolox.log 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deeplab --no-kernel-timing --no-h2d"
for r in 8 16 32 64; do CVHIP_EW_ROWS=$r $T $B > gpurun_out/c_b_ewrows_$r.log 2>&1; done
tail -25 gpurun_out/c_t_all.log; grep -h '^{' gpurun_out/c_extra_*.log | cut -c1-330; for r in 8 16 32 64; do echo "ew_rows $r: $(grep -h '^{' gpurun_out/c_b_ewrows_$r.log | cut -c60-130)"; done
