#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_storage_emulator.py -m gpu -q -k yolov7l 2>&1 | grep -E "^E  |assert" | cut -c1-900 | head
