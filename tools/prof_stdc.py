"""Dev tool: STDC1-Seg train steps (bench.py's stdc_workload) for `rocprofv3 --kernel-trace --stats`; STEPS (default 40) timed steps so
that the steady state dominates the totals."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
a = types.SimpleNamespace(no_graph=False)
print(bench.stdc_workload(torch.device("cuda:0"), a, int(os.environ.get("STEPS", "40")), 3))
