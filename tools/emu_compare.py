"""Dev tool: engine vs storage emulator vs fp32 oracle gradient agreement (prints the distribution; tests/test_gpu_storage_emulator.py asserts it)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_storage_emulator import engine_vs_emulator
for prec in ("bf16", "fp16"):
    r = engine_vs_emulator(prec)
    ee, eo, mo, mm = (np.array(r[k]) for k in ("cos_engine_emulator", "cos_engine_oracle", "cos_emulator_oracle", "cos_emulator_emulator"))
    print(prec, "loss engine %.6f emulator %.6f oracle %.6f" % (r["loss_engine"], r["loss_emulator"], r["loss_oracle"]))
    for name, v in (("engine~emulator", ee), ("engine~oracle", eo), ("emulator~oracle", mo), ("emulator~emulator64", mm)):
        print("  %-18s median %.5f  p10 %.5f  min %.5f" % (name, np.median(v), np.percentile(v, 10), v.min()))
    print("  worst engine~emulator:", [(round(c, 4), n) for c, n in sorted(zip(ee, r["names"]))[:6]])
