#!/bin/bash
# Dev tool: SQ counters of the tap-resident 3x3 weight-gradient kernel beside the general kernel on one shape: three rocprofv3 --pmc passes
# over tools/wgband_wl.py -> gpurun_out/wgband_sq.txt.   gpurun -- 'bash tools/wgband_sq.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/b1 /tmp/b2 /tmp/b3
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/b1 -- python $R/tools/wgband_wl.py > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/b2 -- python $R/tools/wgband_wl.py > /dev/null 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/b3 -- python $R/tools/wgband_wl.py > /dev/null 2>&1
python $R/tools/pmc_step_summary.py /tmp/b1 > $O/wgband_sq.txt 2>&1
python $R/tools/pmc_sq.py /tmp/b2 /tmp/b3 --match wgrad >> $O/wgband_sq.txt 2>&1
