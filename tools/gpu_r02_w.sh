#!/bin/bash
# round-2 GPU call W: the whole GPU suite + smoke + default bench (final validation)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/w_tests.log 2>&1
echo "pytest rc=$?" > gpurun_out/w_summary.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/w_tests.log >> gpurun_out/w_summary.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/w_summary.log
timeout 900 python bench.py 2>/dev/null | grep "^{" | tail -1 > gpurun_out/w_bench.json
python -c "
import json; d=json.load(open('gpurun_out/w_bench.json')); print('bench', d['value'], d['ms_per_step'], d['with_h2d']['value'], d['config3_deeplabv3plus_r50']['value'], d['roofline']['frac'], d['conv_roofline']['frac'])" >> gpurun_out/w_summary.log
cat gpurun_out/w_summary.log
