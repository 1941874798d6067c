#!/bin/bash
# round-2 GPU call E: PMC traffic passes (FETCH_SIZE / WRITE_SIZE in separate runs), kernel stats of the final state, OTA test re-run
mkdir -p gpurun_out
T="timeout 900"
$T python -m pytest tests/test_gpu_yolov7.py -m gpu -q 2>&1 | tail -8 > gpurun_out/e_t_v7.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/pmc_workload.py > $R/gpurun_out/e_pmc_f.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/pmc_workload.py > $R/gpurun_out/e_pmc_w.log 2>&1
python $R/tools/pmc_parse.py /tmp/pmc_f /tmp/pmc_w $R/gpurun_out/e_pmc_traffic_raw.json > $R/gpurun_out/e_pmc_summary.txt 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d > $R/gpurun_out/e_prof.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $R/gpurun_out/e_kernel_stats.csv
cd $R
tail -5 gpurun_out/e_t_v7.log; head -30 gpurun_out/e_pmc_summary.txt
