timeout 600 python -m pytest tests/test_gpu_wgrad_band.py -x -q 2>&1 | tail -3
for a in 0; do echo "ABL=$a"; CVHIP_WGB_ABL=$a WG_ONLY=k3 timeout 200 python tools/wgrad_bench.py 2>&1 | grep "s1" | head -8; done
