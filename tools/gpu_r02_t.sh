#!/bin/bash
# round-2 GPU call T: PMC traffic passes of the final state (FETCH_SIZE / WRITE_SIZE in separate runs) + bench_extra (configs 4, 5)
mkdir -p gpurun_out
T="timeout 900"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python $R/tools/pmc_workload.py > $R/gpurun_out/t_pmc_f.log 2>&1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python $R/tools/pmc_workload.py > $R/gpurun_out/t_pmc_w.log 2>&1
python $R/tools/pmc_parse.py /tmp/pmc_f /tmp/pmc_w $R/gpurun_out/t_pmc_traffic_raw.json > $R/gpurun_out/t_pmc_summary.txt 2>&1
cd $R
$T python tools/bench_extra.py > gpurun_out/t_bench_extra.log 2>&1
head -12 gpurun_out/t_pmc_summary.txt; grep "^{" gpurun_out/t_bench_extra.log | cut -c1-330
