#!/usr/bin/env python3
"""Does the SpMV time depend on WHERE the operator / vectors were allocated?  Several copies of the
same 512^3 operator are created in one process (earlier ones kept alive, so every copy has other
addresses) and timed in turn; then the same copies again with fresh vectors."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

N = 512
n = N ** 3
xh = np.random.default_rng(0).standard_normal(n)
mats, res = [], []
x = dev.DeviceBuffer.from_host(xh)
y = dev.DeviceBuffer(n)
for k in range(6):
    A = dev.DeviceCSR.poisson_big(N, N, N)  # w4 layout only: 7.8 GB per copy
    mats.append(A)
    time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 3)
    t = min(time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 10) for _ in range(3))
    res.append(t)
    print(json.dumps({"copy": k, "ms": t}), flush=True)
print("second pass, same copies, same vectors:", flush=True)
for k, A in enumerate(mats):
    t = min(time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 10) for _ in range(3))
    print(json.dumps({"copy": k, "ms": t}), flush=True)
print("fresh vectors:", flush=True)
for j in range(3):
    x2 = dev.DeviceBuffer.from_host(xh)
    y2 = dev.DeviceBuffer(n)
    ts = [min(time_launches(lambda: A.matvec_dev(x2.ptr, y2.ptr), 10) for _ in range(2)) for A in mats[:3]]
    print(json.dumps({"vectors": j, "ms": ts}), flush=True)
