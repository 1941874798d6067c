#!/bin/bash
# Collects the per-round profile set on the GPU box (run through gpurun; writes under gpurun_out/<tag>_*, to be copied into profiles/).
#   bash tools/profile_round.sh r04          # everything
#   PARTS="bench stats" bash tools/profile_round.sh r04
# rocprofv3 runs from /tmp with TMPDIR=/tmp; counter passes (--pmc) are separate runs with --kernel-trace only.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
PARTS=${PARTS:-"bench stats y5s deeplab yolox yolov7 infer pmc sq cache rotate"}
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
stats_csv() { find $1 -name "*kernel_stats.csv" | head -1; }
if has bench; then
  timeout 900 python $R/bench.py 2>$O/${TAG}_bench_final.err | tail -1 > $O/${TAG}_bench_final.json.log
fi
if has stats; then
  rm -rf /tmp/p_def; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_def -- python $R/bench.py 2>/dev/null | tail -1 > $O/${TAG}_default_bench_under_rocprof.json.log
  cp $(stats_csv /tmp/p_def) $O/${TAG}_default_bench_kernel_stats.csv
fi
if has y5s; then
  rm -rf /tmp/p_y5; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_y5 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-h2d --no-extra --no-sync-bn-leg --no-deeplab > /dev/null 2>&1
  cp $(stats_csv /tmp/p_y5) $O/${TAG}_yolov5s_bs64_kernel_stats.csv
  python $R/tools/trace_gaps.py /tmp/p_y5 > $O/${TAG}_step_trace_one_replay.txt 2>&1
fi
if has deeplab; then
  rm -rf /tmp/p_dl; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_dl -- python $R/tools/prof_deeplab.py > /dev/null 2>&1
  cp $(stats_csv /tmp/p_dl) $O/${TAG}_deeplabv3plus_bs16_kernel_stats.csv
  python $R/tools/trace_gaps.py /tmp/p_dl > $O/${TAG}_step_trace_deeplab_one_replay.txt 2>&1
fi
if has yolox; then
  rm -rf /tmp/p_yx; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_yx -- python $R/tools/prof_extra.py yolox > $O/${TAG}_yoloxs_run.log 2>&1
  cp $(stats_csv /tmp/p_yx) $O/${TAG}_yoloxs_bs64_kernel_stats.csv
  python $R/tools/trace_gaps.py /tmp/p_yx > $O/${TAG}_step_trace_yoloxs_one_replay.txt 2>&1
fi
if has yolov7; then
  rm -rf /tmp/p_y7; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_y7 -- python $R/tools/prof_extra.py yolov7 > $O/${TAG}_yolov7l_run.log 2>&1
  cp $(stats_csv /tmp/p_y7) $O/${TAG}_yolov7l_fp16_bs16_kernel_stats.csv
  python $R/tools/trace_gaps.py /tmp/p_y7 > $O/${TAG}_step_trace_yolov7l_one_replay.txt 2>&1
fi
if has infer; then
  rm -rf /tmp/p_inf; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_inf -- python $R/tools/prof_infer.py > $O/${TAG}_infer_run.log 2>&1
  cp $(stats_csv /tmp/p_inf) $O/${TAG}_infer_kernel_stats.csv
fi
if has pmc; then
  rm -rf /tmp/p_f /tmp/p_w
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -- python $R/tools/pmc_workload.py > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -- python $R/tools/pmc_workload.py > /dev/null 2>&1
  python $R/tools/pmc_parse.py /tmp/p_f /tmp/p_w $O/${TAG}_pmc_traffic_raw.json > $O/${TAG}_pmc_summary.txt 2>&1
fi
if has sq; then
  rm -rf /tmp/p_sq
  timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_sq -- python $R/tools/pmc_workload.py > /dev/null 2>&1
  python $R/tools/pmc_step_summary.py /tmp/p_sq > $O/${TAG}_sq_step_summary.txt 2>&1
fi
if has cache; then
  rm -rf /tmp/p_ca /tmp/p_cb
  timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/p_ca -- python $R/tools/pmc_workload.py > /dev/null 2>&1
  timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d /tmp/p_cb -- python $R/tools/pmc_workload.py > /dev/null 2>&1
  python $R/tools/cache_residency.py /tmp/p_ca /tmp/p_cb > $O/${TAG}_cache_residency_counters.txt 2>&1
fi
if has rotate; then
  timeout 600 python $R/tools/stream_rotate_probe.py > $O/${TAG}_stream_rotate_probe.log 2>&1
  timeout 600 python $R/tools/mall_probe.py > $O/${TAG}_mall_probe.log 2>&1
fi
ls -la $O | grep ${TAG}_ | awk '{print $5, $9}'
