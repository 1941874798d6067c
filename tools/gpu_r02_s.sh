#!/bin/bash
# round-2 GPU call S: tightened model-level tests + DeepLabv3+ kernel stats of the current state
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_storage_emulator.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/s_tests.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profd -- python $R/tools/prof_deeplab.py > $R/gpurun_out/s_prof_deeplab.log 2>&1
cp $(find /tmp/profd -name "*kernel_stats.csv" | head -1) $R/gpurun_out/s_deeplab_kernel_stats.csv
cd $R
cat gpurun_out/s_tests.log; tail -3 gpurun_out/s_prof_deeplab.log
