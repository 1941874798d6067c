#!/bin/bash
# round-2 GPU call Q2: wgrad on a side stream (parallel graph branch), re-measured with the current kernels
mkdir -p gpurun_out
T="timeout 900"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-deeplab --no-h2d --no-kernel-timing"
run() { env "$@" $T $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])" >> gpurun_out/q2_ab.log; }
run X=default
run CVHIP_ASYNC_WGRAD=2
run X=default
run CVHIP_ASYNC_WGRAD=2
cat gpurun_out/q2_ab.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CVHIP_ASYNC_WGRAD=2 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-deeplab --no-h2d > $R/gpurun_out/q2_trace.log 2>&1
python $R/tools/trace_gaps.py /tmp/kt2 2>&1 | head -8
