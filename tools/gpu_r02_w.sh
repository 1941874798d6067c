#!/bin/bash
# round-2 GPU call W: the whole GPU suite (repeat runs check the noise-floor criteria for flakiness) + smoke
mkdir -p gpurun_out
for i in 1 2; do
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/w_tests_$i.log 2>&1
echo "run $i rc=$?" >> gpurun_out/w_summary.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/w_tests_$i.log >> gpurun_out/w_summary.log
done
cat gpurun_out/w_summary.log
