"""Dev tool: per-kernel means of every counter found in rocprofv3 --pmc counter_collection CSVs under the given directories.
usage: python tools/pmc_sq.py <dir> [<dir> ...] [--match substring]"""
import csv, glob, os, sys, collections
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = None
if "--match" in sys.argv:
    match = sys.argv[sys.argv.index("--match") + 1]
    dirs = [d for d in dirs if d != match]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                if match and match not in k:
                    continue
                a = agg[k][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
for k, cs in agg.items():
    print(k[:140])
    for c, (n, v) in sorted(cs.items()):
        print("    %-32s n=%4d  mean %16.1f" % (c, n, v / n))
