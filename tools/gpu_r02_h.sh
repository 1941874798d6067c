#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/emu_blocks.py > gpurun_out/h_blocks.log 2>&1
grep -v Warning gpurun_out/h_blocks.log | tail -80
