#!/bin/bash
# round-2 GPU call X: rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the one the driver runs)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profx -- python $R/bench.py > $R/gpurun_out/x_bench_under_rocprof.log 2>&1
cp $(find /tmp/profx -name "*kernel_stats.csv" | head -1) $R/gpurun_out/x_default_kernel_stats.csv
cd $R; tail -1 gpurun_out/x_bench_under_rocprof.log | cut -c1-400; head -5 gpurun_out/x_default_kernel_stats.csv | cut -c1-200
