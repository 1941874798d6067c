#!/bin/bash
# round-2 GPU call L2: SQ counters of the implicit-GEMM kernel under the ablations (64->64 k3 @80x80, batch 64)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export ABL_BATCH=64 ABL_ONLY=1
T="timeout 300"
for abl in 2 1 3; do
CVHIP_IGEMM_ABLATE=$abl $T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sqa$abl -- python $R/tools/conv_ablate.py > $R/gpurun_out/l_sqa$abl.log 2>&1
echo "=== ABLATE=$abl" >> $R/gpurun_out/l_sq_abl_summary.txt
python $R/tools/pmc_sq.py /tmp/sqa$abl --match igemm >> $R/gpurun_out/l_sq_abl_summary.txt 2>&1
done
cd $R; cat gpurun_out/l_sq_abl_summary.txt
