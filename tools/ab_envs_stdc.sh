#!/bin/bash
# In-box A/B of environment settings on the STDC1-Seg leg: bash tools/ab_envs_stdc.sh "A=1" "A=2" ... (ROUNDS alternating rounds, default 2)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq ${ROUNDS:-2}); do
  for kv in "$@"; do
    env $kv python - <<PY
import sys, types, torch
sys.path.insert(0, "$R")
import bench
r = bench.stdc_workload(torch.device("cuda:0"), types.SimpleNamespace(no_graph=False), 20, 3)
print("$kv", r["value"], r["ms_per_step"])
PY
  done
done
