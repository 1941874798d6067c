#!/bin/bash
# Dev tool: isolated sweep of the wgrad launcher's knobs (out-channel tile cap, block target, groups per block) over tools/wgrad_bench.py's shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}
for only in k1 k3; do
  env WG_ONLY=$only timeout 120 python $R/tools/wgrad_bench.py 2>/dev/null | grep -v amdgpu.ids
  for tn in 32 64 128; do
  for blocks in 256 512 768; do
  for groups in 1 2; do
    env WG_ONLY=$only CVHIP_WGRAD_TNMAX=$tn CVHIP_WGRAD_BLOCKS=$blocks CVHIP_WGRAD_GROUPS=$groups timeout 120 python $R/tools/wgrad_bench.py 2>/dev/null | grep -v amdgpu.ids
  done; done; done
done
