#!/bin/bash
# round-2 GPU call A: correctness of the fused 1x1 backward + BK64 igemm, then A/B timing (conv table + bench)
mkdir -p gpurun_out
T="timeout 600"
$T python -m pytest tests/test_gpu_bwd1x1.py -x -q 2>&1 | tail -25 > gpurun_out/a_t_bwd1x1.log
$T python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/a_t_all.log
CVHIP_IGEMM_BK64=1 $T python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -8 > gpurun_out/a_t_bk64_1.log
CVHIP_IGEMM_BK64=2 $T python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -8 > gpurun_out/a_t_bk64_2.log
export TABLE_ROWS=300
$T python tools/conv_table.py > gpurun_out/a_ct_default.log 2>&1
CVHIP_BWD1X1=0 $T python tools/conv_table.py > gpurun_out/a_ct_nobwd.log 2>&1
CVHIP_IGEMM_BK64=1 $T python tools/conv_table.py > gpurun_out/a_ct_bk64_1.log 2>&1
CVHIP_IGEMM_BK64=2 $T python tools/conv_table.py > gpurun_out/a_ct_bk64_2.log 2>&1
CVHIP_BWD1X1_BLOCKS=256 $T python tools/conv_table.py > gpurun_out/a_ct_b256.log 2>&1
CVHIP_BWD1X1_MINTRIPS=16 $T python tools/conv_table.py > gpurun_out/a_ct_t16.log 2>&1
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deeplab --no-kernel-timing"
$T $B > gpurun_out/a_b_default.log 2>&1
CVHIP_BWD1X1=0 $T $B > gpurun_out/a_b_nobwd.log 2>&1
CVHIP_IGEMM_BK64=1 $T $B > gpurun_out/a_b_bk64_1.log 2>&1
CVHIP_IGEMM_BK64=2 $T $B > gpurun_out/a_b_bk64_2.log 2>&1
tail -3 gpurun_out/a_t_*.log; grep -h '"value"' gpurun_out/a_b_*.log | cut -c1-200; head -1 gpurun_out/a_ct_*.log
