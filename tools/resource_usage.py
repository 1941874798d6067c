"""Dev tool: per-kernel register / scratch / occupancy table of the built library, from the compiler remarks build.py keeps next to
each object (csrc/_obj/*.usage.txt). A kernel that spills (ScratchSize > 0) is flagged: a runtime flag that keeps extra per-lane
state live across a main loop can push a tuned kernel over its register budget without any functional symptom (round 3: the
bwd1x1 tail sums cost the 128 x 128 configuration 104 spilled VGPRs and 0.44 ms of the YOLOv5-s step).

    python tools/resource_usage.py [--all]        (default: kernels with scratch only)
"""
import glob
import os
import re
import subprocess
import sys

OBJ = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cvpytorch_amd", "csrc", "_obj")


def demangle(names):
    try:
        r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        out = r.stdout.splitlines()
        if len(out) == len(names):
            return out
    except OSError:
        pass
    return names


def parse(path):
    """[(kernel, {field: int})] of one usage file"""
    out, cur = [], None
    for ln in open(path):
        m = re.search(r"remark: \s*([A-Za-z /\[\]]+?):\s*(\S+)", ln)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = {}
            out.append((val, cur))
        elif cur is not None:
            try:
                cur[key] = int(val)
            except ValueError:
                pass
    return out


def table():
    rows = []
    for f in sorted(glob.glob(os.path.join(OBJ, "*.usage.txt"))):
        for k, d in parse(f):
            rows.append((os.path.basename(f).replace(".usage.txt", ""), k, d))
    return rows


if __name__ == "__main__":
    rows = table()
    names = demangle([k for _, k, _ in rows])
    show_all = "--all" in sys.argv
    print("%-22s %5s %5s %7s %4s  %s" % ("object", "VGPR", "AGPR", "scratch", "occ", "kernel"))
    for (obj, _, d), nm in zip(rows, names):
        sc = d.get("ScratchSize [bytes/lane]", 0)
        if sc or show_all:
            print("%-22s %5d %5d %7d %4d  %s" % (obj, d.get("VGPRs", -1), d.get("AGPRs", -1), sc, d.get("Occupancy [waves/SIMD]", -1), nm[:150]))
