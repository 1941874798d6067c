"""In-tree build of libcvhip.so (hipcc, gfx950 only). No torch involvement: the library is a plain
C-ABI shared object (include/cvhip.h) loaded through ctypes. The measurement / known-answer kernels of csrc/probes.hip go into a
second library, libcvhip_probes.so (include/cvhip_probes.h), which the product never loads.

    python -m cvpytorch_amd.build [--force]
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libcvhip.so")
PROBES_LIB = os.path.join(HERE, "libcvhip_probes.so")
PROBE_SOURCES = ("probes.hip",)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = [
    "api.hip",
    "conv_igemm.hip",
    "conv_patch.hip",
    "conv_band.hip",
    "conv_wgrad.hip",
    "conv_wgrad_band.hip",
    "bn_act.hip",
    "pool_resize.hip",
    "weights_optim.hip",
    "dwconv.hip",
    "post.hip",
    "loss_kernels.hip", "yolo_loss.hip", "simota_loss.hip", "conv1x1_stream.hip", "conv_stem.hip", "conv1x1_bwd.hip", "comm.hip", "post_batch.hip", "ota_assign.hip", "probes.hip",
]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-munsafe-fp-atomics",
    "-Wall",
    "-Wno-unused-function",
]


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "cvhip.h"))
    hdrs.append(os.path.join(HERE, "..", "include", "cvhip_probes.h"))
    return hdrs


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# sources without 16-bit operands are compiled once; every other source a second time with -DCVHIP_F16 (fp16 storage,
# entry points suffixed _f16: csrc/common.h, csrc/f16_names.h)
SINGLE_PRECISION = ("comm.hip", "post_batch.hip", "probes.hip")


def _compile(job):
    src, f16 = job
    obj = os.path.join(OBJ, src.replace(".hip", ".f16.o" if f16 else ".o"))
    path = os.path.join(CSRC, src)
    if not _newer(obj, [path] + _deps()):
        return obj, None
    # -Rpass-analysis=kernel-resource-usage: per-kernel VGPR / scratch / occupancy remarks, kept next to the object
    # (<obj>.usage.txt; tools/resource_usage.py tabulates them, tests/test_abi_plan.py fails on a kernel that started to spill)
    cmd = [HIPCC] + FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + (["-DCVHIP_F16=1"] if f16 else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    remarks = [ln for ln in r.stderr.splitlines() if "kernel-resource-usage" in ln]
    with open(obj + ".usage.txt", "w") as fh:
        fh.write("\n".join(remarks) + "\n")
    other = "\n".join(ln for ln in r.stderr.splitlines() if "kernel-resource-usage" not in ln and "remark generated" not in ln).strip()
    return obj, other


def _sync_f16_names():
    """Regenerate csrc/f16_names.h and include/cvhip_f16.h (tools/gen_f16_names.py) when an entry point was added or removed:
    the fp16 build renames every typed entry point, a stale list would define the new ones twice."""
    import importlib.util
    path = os.path.join(HERE, "..", "tools", "gen_f16_names.py")
    if not os.path.exists(path):
        return
    spec = importlib.util.spec_from_file_location("gen_f16_names", path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    a, b, _ = gen.render()
    for target, text in ((os.path.join(CSRC, "f16_names.h"), a), (os.path.join(HERE, "..", "include", "cvhip_f16.h"), b)):
        if not os.path.exists(target) or open(target).read() != text:
            open(target, "w").write(text)


def build_lib(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    _sync_f16_names()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = [(s, False) for s in srcs] + [(s, True) for s in srcs if s not in SINGLE_PRECISION]
    if force:
        for s, f16 in jobs:
            o = os.path.join(OBJ, s.replace(".hip", ".f16.o" if f16 else ".o"))
            if os.path.exists(o):
                os.remove(o)
    with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        results = list(ex.map(_compile, jobs))
    objs = [o for (o, _), (s, _f) in zip(results, jobs) if s not in PROBE_SOURCES]
    probe_objs = [o for (o, _), (s, _f) in zip(results, jobs) if s in PROBE_SOURCES]
    for (o, warn), (s, _f) in zip(results, jobs):
        if warn and verbose:
            sys.stderr.write("[cvhip build] %s:\n%s\n" % (s, warn))
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if probe_objs and (force or _newer(PROBES_LIB, probe_objs)):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", PROBES_LIB] + probe_objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link of libcvhip_probes.so failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
