#!/usr/bin/env python3
"""Time to first result on the irregular stand-in (config C5 shape): handle creation (upload), first product
(builds the kernel tables; RCM renumbering where it applies), second product, first / second MINRES solve."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from pysparse_amd.tools.standins import fem_sss_arrays  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shuffle", type=int, default=512)
a = ap.parse_args()
L = lib()
check(L.psp_set_device(0))
xw = dev.DeviceBuffer(16)  # context creation outside the timings
n, ind, col, val, diag = fem_sss_arrays(68, 68, 67, a.shuffle, 0)
res = {"n": n, "nnz_lower": len(col), "shuffle": a.shuffle}
t = time.perf_counter()
S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
check(L.psp_synchronize())
res["create_s"] = time.perf_counter() - t
x = np.random.default_rng(1).standard_normal(n)
y = np.empty(n)
for k in ("first_matvec_s", "second_matvec_s"):
    t = time.perf_counter()
    S.matvec(x, y)
    res[k] = time.perf_counter() - t
res["kernel"] = S.kernel_info()[0]
b = np.zeros(n)
b[0] = 1.0
K = dev.DeviceJacobi(S)
for k in ("first_minres_s", "second_minres_s"):
    xh = np.zeros(n)
    t = time.perf_counter()
    r = dev.minres(S, b, xh, 1e-10, 2000, K)
    res[k] = time.perf_counter() - t
res["minres"] = list(r)
print(json.dumps(res), flush=True)
