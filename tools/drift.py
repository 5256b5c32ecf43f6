#!/usr/bin/env python3
"""Does the SpMV time drift while the GPU stays busy (clock / power / thermal state)?  One operator, back-to-back
launches for ~25 s, the average of every 0.5 s window printed with the SMI clocks when available."""
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

A = dev.DeviceCSR.poisson_big(512, 512, 512)
n = A.shape[0]
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], capture_output=True,
                             text=True, timeout=5).stdout
        d = json.loads(out)
        c = next(iter(d.values()))
        keep = {k: v for k, v in c.items() if any(t in k.lower() for t in ("sclk", "mclk", "power", "temperature (sensor junction"))}
        return keep
    except Exception as e:  # noqa: BLE001
        return {"smi": str(e)[:60]}


t0 = time.time()
print(json.dumps({"t": 0.0, "smi": smi()}), flush=True)
k = 0
while time.time() - t0 < 25.0:
    ms = time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 300)
    k += 1
    rec = {"t": round(time.time() - t0, 2), "ms": round(ms, 4)}
    if k % 8 == 0:
        rec["smi"] = smi()
    print(json.dumps(rec), flush=True)
