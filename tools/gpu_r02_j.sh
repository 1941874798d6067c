#!/bin/bash
# round-2 GPU call J: what does the A-tile DMA volume cost? (ABL 3 = pixel rows fetched for 2 of 9 taps, results wrong by design)
mkdir -p gpurun_out
T="timeout 600"
export ABL_BATCH=64
( CVHIP_IGEMM_NST2=3 $T python tools/conv_ablate.py
  CVHIP_IGEMM_ABLATE=3 $T python tools/conv_ablate.py
  CVHIP_IGEMM_ABLATE=1 $T python tools/conv_ablate.py
  CVHIP_IGEMM_ABLATE=2 $T python tools/conv_ablate.py
  $T python tools/conv_ablate.py ) > gpurun_out/j_ablate.log 2>&1
grep -v Warn gpurun_out/j_ablate.log | grep "k3 s1"
