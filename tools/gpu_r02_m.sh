#!/bin/bash
# round-2 GPU call M: SQ counters of every kernel of one eager YOLOv5-s step
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export PMC_STEPS=2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sqstep -- python $R/tools/pmc_workload.py > $R/gpurun_out/m_sq.log 2>&1
python $R/tools/pmc_step_summary.py /tmp/sqstep > $R/gpurun_out/m_sq_step_summary.txt 2>&1
cd $R; cat gpurun_out/m_sq_step_summary.txt
