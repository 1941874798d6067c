"""Dev tool (VERDICT r03 task 4): what the cache hierarchy does for each kernel of the train step.

    python tools/cache_residency.py <dir of pass A> <dir of pass B> [<kernel-trace csv>]

  pass A: rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum            -> L2 (per-XCD, 4 MiB) hit rate per kernel
  pass B: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
                                                                          -> bytes that LEAVE the L2 towards the fabric
                                                                             (Infinity Cache / HBM): 32 B or 64 B per read request,
                                                                             32 B or 64 B per write request
gfx950 / ROCm 7.2 expose NO Infinity-Cache (MALL) hit counter (rocprofv3 -L: profiles/r04_counters_tcc.txt): what can be measured is
the L2 hit rate and the L2 -> fabric traffic; the MALL's contribution shows only in TIME (same kernel on an operand set that fits the
256 MiB cache vs on rotating sets: tools/patch_bench.py NSETS=1, tools/stream_rotate_probe.py)."""
import collections
import csv
import glob
import os
import sys


def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"]
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                key = (k, row.get("Dispatch_Id"))
                if key not in seen:
                    seen.add(key)
                    cnt[k] += 1
    return agg, cnt


def main():
    A, nA = load(sys.argv[1])
    B, nB = load(sys.argv[2])
    rows = []
    for k in set(A) | set(B):
        a, b = A.get(k, {}), B.get(k, {})
        n = max(nA.get(k, 0), nB.get(k, 0), 1)
        hit, miss = a.get("TCC_HIT_sum", 0.0), a.get("TCC_MISS_sum", 0.0)
        rd, rd32 = b.get("TCC_EA0_RDREQ_sum", 0.0), b.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        wr, wr64 = b.get("TCC_EA0_WRREQ_sum", 0.0), b.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        rbytes = (rd32 * 32 + (rd - rd32) * 64) / max(nB.get(k, 1), 1)
        wbytes = (wr64 * 64 + (wr - wr64) * 32) / max(nB.get(k, 1), 1)
        rows.append((rbytes * nB.get(k, 0) + wbytes * nB.get(k, 0), k, n, hit / max(hit + miss, 1.0), rbytes, wbytes))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("L2 -> fabric bytes of all profiled launches: %.2f GB" % (tot / 1e9))
    print("%-96s %5s %8s %12s %12s" % ("kernel", "n", "L2 hit", "EA read MB", "EA write MB"))
    for _, k, n, hr, rb, wb in rows[:45]:
        print("%-96s %5d %8.3f %12.2f %12.2f" % (k[:96], n, hr, rb / 1e6, wb / 1e6))


if __name__ == "__main__":
    main()
