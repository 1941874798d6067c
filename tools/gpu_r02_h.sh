#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do timeout 1200 python -m pytest tests/test_gpu_storage_emulator.py tests/test_gpu_modules.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED"; done
