#!/bin/bash
# round-2 GPU call G: storage emulator criteria (after the rounding-flip analysis)
mkdir -p gpurun_out
timeout 900 python tools/emu_compare.py > gpurun_out/g_emu.log 2>&1
timeout 900 python -m pytest tests/test_gpu_storage_emulator.py -q -m gpu -x > gpurun_out/g_test.log 2>&1
tail -30 gpurun_out/g_emu.log; tail -15 gpurun_out/g_test.log
