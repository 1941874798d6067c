"""Dev tool: one graph-replayed train step out of a rocprofv3 --kernel-trace CSV (the kernels between the last two
sgd_ema_kernel launches): span, GPU-idle time, overlap, per-kernel totals, copy nodes.
usage: python tools/trace_gaps.py <dir>"""
import csv, glob, os, sys, collections
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "sgd_ema_kernel" in r[2] or "sgd_nesterov_ema" in r[2]]
print("kernels", len(rows), "step markers", len(marks))
a, b = marks[-2] + 1, marks[-1] + 1
step = rows[a:b]
span = step[-1][1] - step[0][0]
busy_end = step[0][1]
idle = 0
gaps = []
for s, e, k in step[1:]:
    if s > busy_end:
        idle += s - busy_end
        gaps.append(s - busy_end)
    busy_end = max(busy_end, e)
tot = sum(e - s for s, e, _ in step)
print("one replayed step: %d kernels, span %.3f ms, sum of kernel durations %.3f ms, no-kernel-running %.3f ms (%d gaps, median %.2f us, max %.1f us)" %
      (len(step), span / 1e6, tot / 1e6, idle / 1e6, len(gaps), sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0, max(gaps) / 1e3 if gaps else 0))
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in step:
    agg[k][0] += 1
    agg[k][1] += e - s
fam = collections.Counter()
for k, (n, d) in agg.items():
    f = ("bn/act passes" if ("ew_kernel" in k or "colreduce" in k) else "bn finalize (small)" if ("sum_partials" in k or "bn_finalize" in k or "rows_reduce" in k) else
         "igemm + patch" if ("igemm" in k or "conv_patch" in k) else "wgrad" if "wgrad" in k else "1x1 stream" if "conv1x1_stream" in k else "bwd1x1 fused" if "bwd1x1" in k else
         "stem" if "stem" in k else "aten" if "at::native" in k else "copy nodes" if "copyBuffer" in k else "other cvhip")
    fam[f] += d
print("by family (ms):", ", ".join("%s %.2f" % (k, v / 1e6) for k, v in fam.most_common()))
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("  %5d x %8.1f us = %8.3f ms  %s" % (n, d / n / 1e3, d / 1e6, k[:100]))
