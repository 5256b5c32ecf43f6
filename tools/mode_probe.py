#!/usr/bin/env python3
"""Which of the two timing modes does this process land in, and what differs?  Prints the device pointers of
x / y, free memory before / after, and the csr_spmv_w4 time at 512^3 (run-to-run spread)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

L = lib()
f0, t0 = C.c_int64(), C.c_int64()
check(L.psp_mem_info(C.byref(f0), C.byref(t0)))
order = os.environ.get("PROBE_ORDER", "Axy")
objs = {}
n = 512 ** 3
for ch in order:
    if ch == "A":
        objs["A"] = dev.DeviceCSR.poisson(512, 512, 512)
    elif ch == "x":
        objs["x"] = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
    elif ch == "y":
        objs["y"] = dev.DeviceBuffer(n)
    elif ch == "p":  # padding allocation to shift what follows
        objs.setdefault("pad", []).append(dev.DeviceBuffer(int(os.environ.get("PROBE_PAD", "1000003"))))
A, x, y = objs["A"], objs["x"], objs["y"]
f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
time_launches(f, 10)
ts = [time_launches(f, 20) for _ in range(5)]
f1 = C.c_int64()
check(L.psp_mem_info(C.byref(f1), C.byref(t0)))
print(json.dumps({"ms": min(ts), "ms_all": [round(t, 4) for t in ts], "x": hex(x.ptr), "y": hex(y.ptr),
                  "free_before_GB": f0.value / 1e9, "used_GB": (f0.value - f1.value) / 1e9, "order": order}))
