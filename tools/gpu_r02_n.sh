#!/bin/bash
# round-2 GPU call N: kernel trace of graph-replayed steps: inter-kernel gaps, what drags the copyBuffer nodes along
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d > $R/gpurun_out/n_trace.log 2>&1
python $R/tools/trace_gaps.py /tmp/kt > $R/gpurun_out/n_gaps.txt 2>&1
cd $R; cat gpurun_out/n_gaps.txt
