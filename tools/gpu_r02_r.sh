#!/bin/bash
# round-2 GPU call R: full GPU suite + smoke + default bench line + kernel stats of the final state
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -v -x 2>&1 | grep -v "^$" | tail -400 > gpurun_out/r_tests_full.log
tail -6 gpurun_out/r_tests_full.log > gpurun_out/r_tests.log
timeout 900 python bench.py > gpurun_out/r_bench.log 2>&1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d > $R/gpurun_out/r_prof.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r_kernel_stats.csv
cd $R
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/r_tests_full.log | tail -3; cat gpurun_out/r_tests.log; tail -1 gpurun_out/r_bench.log | cut -c1-3000
