#!/bin/bash
mkdir -p gpurun_out
T="timeout 900"
$T python tools/emu_compare.py > gpurun_out/f_emu.log 2>&1
$T python -m pytest tests/test_gpu_storage_emulator.py -m gpu -q 2>&1 | tail -15 > gpurun_out/f_t_emu.log
cat gpurun_out/f_emu.log | grep -v Warning | tail -20; tail -5 gpurun_out/f_t_emu.log
