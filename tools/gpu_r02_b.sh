#!/bin/bash
# round-2 GPU call B: full gpu suite (fused 1x1 backward, RCCL communicator), full bench line, rocprof kernel stats
mkdir -p gpurun_out
T="timeout 900"
$T python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/b_t_all.log
$T python bench.py --steps 20 --warmup 5 > gpurun_out/b_bench.log 2>&1
TABLE_ROWS=300 $T python tools/conv_table.py > gpurun_out/b_ct.log 2>&1
MODEL=deeplab TABLE_ROWS=300 $T python tools/conv_table.py > gpurun_out/b_ct_deeplab.log 2>&1
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d > $GRAFT_REPO_ROOT/gpurun_out/b_prof.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/b_kernel_stats.csv
cd $GRAFT_REPO_ROOT
tail -4 gpurun_out/b_t_all.log; tail -c 3000 gpurun_out/b_bench.log
