"""Dev tool: per-kernel SQ picture of one eager train step from a rocprofv3 --pmc pass (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE): resident waves per SIMD, wait shares, MFMA-pipe busy share.
usage: python tools/pmc_step_summary.py <dir>"""
import csv, glob, os, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
rows = []
for k, c in agg.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0            # summed over 8 XCDs -> kernel cycles
    if gui <= 0:
        continue
    wc = c.get("SQ_WAVE_CYCLES", 0.0) * 4.0               # quad-cycles -> cycles, summed over waves
    rows.append((gui, k, cnt[k], wc / (gui * 1024.0), c.get("SQ_WAIT_ANY", 0) * 4 / max(wc, 1), c.get("SQ_WAIT_INST_ANY", 0) * 4 / max(wc, 1),
                 c.get("SQ_ACTIVE_INST_ANY", 0) * 4 / max(wc, 1), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024.0)))
rows.sort(reverse=True)
print("%-70s %5s %9s %6s %6s %6s %6s %6s" % ("kernel", "n", "Mcyc", "w/SIMD", "wait", "stall", "issue", "mfma"))
for gui, k, n, occ, wa, wi, ac, mf in rows[:40]:
    print("%-70s %5d %9.2f %6.2f %6.2f %6.2f %6.2f %6.2f" % (k[:70], n, gui / 1e6, occ, wa, wi, ac, mf))
